"""Kernel-level GPU tests: the MFMA tile engines behind every entry point, called through the C ABI."""
import ctypes

import numpy as np
import pytest
import torch

from polygon_amd import _lib

pytestmark = pytest.mark.gpu

DT = {"f32": (torch.float32, _lib.PG_F32, 2e-5), "bf16": (torch.bfloat16, _lib.PG_BF16, 1e-2)}


def _gemm_nt(dt, A, B, bias=None, accumulate=None, alpha=1.0, want_t=False):
    tdt, code, _ = DT[dt]
    M, K = A.shape
    N = B.shape[0]
    C32 = torch.zeros(M, N, device="cuda") if accumulate is None else accumulate.clone()
    Ct = torch.zeros(M, N, device="cuda", dtype=tdt) if want_t else None
    L = _lib.lib()
    st = L.pg_gemm_nt(code, _lib.ptr(A), ctypes.c_long(A.stride(0)), _lib.ptr(B), ctypes.c_long(B.stride(0)), M, N, K,
                      _lib.ptr(bias), _lib.ptr(C32), _lib.ptr(Ct), ctypes.c_long(N), int(accumulate is not None),
                      ctypes.c_float(alpha), _lib.stream_ptr())
    _lib.check(st, "pg_gemm_nt")
    torch.cuda.synchronize()
    return C32, Ct


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(128, 128, 64), (1000, 1536, 512), (77, 51, 512), (4096, 512, 1536), (33, 200, 24)])
def test_gemm_nt(dt, shape):
    tdt, code, tol = DT[dt]
    M, N, K = shape
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g).to("cuda", tdt)
    B = torch.randn(N, K, generator=g).to("cuda", tdt)
    bias = torch.randn(N, generator=g).cuda()
    ref = A.double() @ B.double().T
    C32, Ct = _gemm_nt(dt, A, B, want_t=True)
    scale = ref.abs().max().item()
    assert (C32.double() - ref).abs().max().item() <= tol * scale
    assert (Ct.double() - ref).abs().max().item() <= max(tol, 8e-3 if dt == "bf16" else tol) * scale
    prev = torch.randn(M, N, generator=g).cuda()
    C32, _ = _gemm_nt(dt, A, B, bias=bias, accumulate=prev, alpha=0.5)
    ref2 = 0.5 * ref + bias.double() + prev.double()
    assert (C32.double() - ref2).abs().max().item() <= tol * scale


def test_gemm_nt_asymmetric_identity():
    # A = I with an asymmetric B catches a transposed accumulator map (guide section 3)
    A = torch.eye(64, device="cuda")
    B = (torch.arange(64 * 64, device="cuda", dtype=torch.float32).reshape(64, 64) % 251)
    C32, _ = _gemm_nt("f32", A, B)
    assert torch.equal(C32, B.T.contiguous())
    C32, _ = _gemm_nt("bf16", A.bfloat16(), B.bfloat16())
    assert torch.equal(C32, B.bfloat16().float().T.contiguous())


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(128, 128, 64), (1536, 512, 5000), (51, 512, 3001), (64, 768, 777), (256, 256, 31)])
def test_gemm_tn(dt, shape):
    tdt, code, tol = DT[dt]
    M, N, R = shape
    g = torch.Generator(device="cpu").manual_seed(M + N + R)
    lda = (M + 7) // 8 * 8
    Afull = torch.randn(R, lda, generator=g).to("cuda", tdt)
    B = torch.randn(R, N, generator=g).to("cuda", tdt)
    C = torch.ones(M, N, device="cuda")
    L = _lib.lib()
    st = L.pg_gemm_tn(code, _lib.ptr(Afull), ctypes.c_long(lda), _lib.ptr(B), ctypes.c_long(N), M, N, R, _lib.ptr(C),
                      ctypes.c_long(N), ctypes.c_float(2.0), _lib.stream_ptr())
    _lib.check(st, "pg_gemm_tn")
    torch.cuda.synchronize()
    ref = 1.0 + 2.0 * (Afull[:, :M].double().T @ B.double())
    scale = ref.abs().max().item()
    assert (C.double() - ref).abs().max().item() <= tol * scale


SLAB_BYTES = 776 * 128 * 128 * 4


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("colsums", [True, False])
@pytest.mark.parametrize("shape", [(1536, 512, 1), (1536, 512, 31), (1536, 512, 33), (1536, 512, 5000), (1536, 512, 50021),
                                   (512, 512, 777), (768, 256, 3000), (1536, 512, 6144)])
def test_gemm_tn_slabs_and_colsums(dt, shape, colsums):
    """The forms the training step uses: two-stage split-K through a slab buffer (large outputs; in bf16 with M, N
    multiples of 256, >= 8 tiles and no column sums the 256 x 256 LDS-DMA kernel), with and without the fused column
    sums, against fp64 matmul of the same (rounded) operands.  Row counts around the 32-row tile and a ragged 50 021
    exercise the zero-page rows."""
    tdt, code, tol = DT[dt]
    M, N, R = shape
    g = torch.Generator(device="cpu").manual_seed(M + N + R)
    A = torch.randn(R, M, generator=g).to("cuda", tdt)
    B = torch.randn(R, N, generator=g).to("cuda", tdt)
    C = torch.ones(M, N, device="cuda")
    ca, cb = torch.full((M,), 3.0, device="cuda"), torch.full((N,), -1.0, device="cuda")
    slab = torch.empty(SLAB_BYTES // 4, device="cuda")
    slab.fill_(float("nan"))   # every slab word the reduction reads must have been written
    L = _lib.lib()
    for rep in range(2):       # twice: the second call adds on top (C and the sums are accumulated, the slabs reused)
        st = L.pg_gemm_tn_ws(code, _lib.ptr(A), ctypes.c_long(M), _lib.ptr(B), ctypes.c_long(N), M, N, R, _lib.ptr(C),
                             ctypes.c_long(N), ctypes.c_float(0.5), _lib.ptr(ca) if colsums else None,
                             _lib.ptr(cb) if colsums else None, _lib.ptr(slab), ctypes.c_size_t(SLAB_BYTES), _lib.stream_ptr())
        _lib.check(st, "pg_gemm_tn_ws")
    torch.cuda.synchronize()
    ref = 1.0 + 2 * 0.5 * (A.double().T @ B.double())
    scale = ref.abs().max().item()
    assert (C.double() - ref).abs().max().item() <= tol * scale
    if colsums:
        ra, rb = 3.0 + A.double().sum(0), -1.0 + B.double().sum(0)
        assert (ca.double() - ra).abs().max().item() <= tol * max(1.0, ra.abs().max().item())
        assert (cb.double() - rb).abs().max().item() <= tol * max(1.0, rb.abs().max().item())


def test_gemm_tn256_integer_exact():
    """Small-integer data (exact in bf16) through the 256 x 256 kernel: every element of the swizzled LDS image, the
    transposed fragment reads and the slab reduction, bit for bit."""
    R, M, N = 200, 1536, 512
    A = ((torch.arange(R * M).reshape(R, M) * 7) % 13 - 6).float()
    B = ((torch.arange(R * N).reshape(R, N) * 5) % 11 - 5).float()
    ref = A.double().T @ B.double()
    C = torch.zeros(M, N, device="cuda")
    slab = torch.empty(SLAB_BYTES // 4, device="cuda")
    Ad, Bd = A.to("cuda", torch.bfloat16), B.to("cuda", torch.bfloat16)
    st = _lib.lib().pg_gemm_tn_ws(_lib.PG_BF16, _lib.ptr(Ad), ctypes.c_long(M), _lib.ptr(Bd), ctypes.c_long(N), M, N, R,
                                  _lib.ptr(C), ctypes.c_long(N), ctypes.c_float(1.0), None, None,
                                  _lib.ptr(slab), ctypes.c_size_t(SLAB_BYTES), _lib.stream_ptr())
    _lib.check(st)
    torch.cuda.synchronize()
    assert torch.equal(C.cpu().double(), ref)


def test_gemm_tn_integer_exact():
    # small-integer data: exact in bf16 and f32, asymmetric -> checks the transposed LDS reads element by element
    R, M, N = 96, 128, 128
    A = ((torch.arange(R * M).reshape(R, M) * 7) % 13 - 6).float()
    B = ((torch.arange(R * N).reshape(R, N) * 5) % 11 - 5).float()
    ref = A.double().T @ B.double()
    for dt in ("f32", "bf16"):
        tdt, code, _ = DT[dt]
        C = torch.zeros(M, N, device="cuda")
        Ad, Bd = A.to("cuda", tdt), B.to("cuda", tdt)
        st = _lib.lib().pg_gemm_tn(code, _lib.ptr(Ad), ctypes.c_long(M), _lib.ptr(Bd),
                                   ctypes.c_long(N), M, N, R, _lib.ptr(C), ctypes.c_long(N), ctypes.c_float(1.0),
                                   _lib.stream_ptr())
        _lib.check(st)
        torch.cuda.synchronize()
        assert torch.equal(C.cpu().double(), ref), dt


def _scatter_ref(tok, G, Vp):
    dT = torch.zeros(Vp, G.shape[1], dtype=torch.float64)
    ok = (tok >= 0) & (tok < Vp)
    dT.index_add_(0, tok[ok].long(), G[ok].double())
    return dT, G[ok].double().sum(0)


@pytest.mark.parametrize("dt", ["f32", "bf16"])
@pytest.mark.parametrize("case", [
    # (rows, columns, row stride, Vp, scratch: 0 = one pass, else bytes -> several accumulating passes, skew)
    (22000, 1536, 1536, 64, 0, True),      # decoder of the bench configuration
    (333, 96, 104, 64, 0, True),           # fewer rows than a chunk for most tokens, padded rows, C not a multiple of 128
    (1, 8, 8, 64, 0, False),               # one row
    (5000, 768, 768, 128, 0, False),       # wide vocabulary: the 512-thread sort
    (9000, 512, 512, 64, 200 << 10, True), # scratch for 3200 rows per pass: three accumulating passes
    (300000, 32, 32, 64, 0, True),         # more rows than one sort launch takes (255 per thread): two passes
])
def test_tok_scatter_sum(dt, case):
    """The embedding-table gradient (pg_scatter.h: counting sort of the rows by token, per-chunk sums in registers, reduce)
    against an fp64 index_add of the same (rounded) rows; tokens outside [0, Vp) are left out; dT is overwritten."""
    tdt, code, tol = DT[dt]
    N, C, ldg, Vp, scratch_bytes, skew = case
    g = torch.Generator(device="cpu").manual_seed(N + C)
    tok = torch.randint(0, Vp, (N,), generator=g, dtype=torch.int32)
    if skew:                                # a few symbols carry most rows, as in SMILES
        r = torch.rand(N, generator=g)
        tok = torch.where(r < 0.45, torch.full_like(tok, 3), torch.where(r < 0.65, torch.full_like(tok, 4), tok))
    if N > 100:
        tok[7] = -1; tok[11] = Vp           # out of range: skipped
    Gfull = torch.randn(N, ldg, generator=g).to(tdt)
    L = _lib.lib()
    L.pg_tok_scatter_scratch_bytes.restype = ctypes.c_size_t
    need = L.pg_tok_scatter_scratch_bytes(ctypes.c_long(N), C, Vp)
    assert need > 0
    nbytes = scratch_bytes if scratch_bytes else need
    assert nbytes <= need
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    tok_d, G_d = tok.cuda(), Gfull.cuda()
    dT = torch.full((Vp, C), float("nan"), device="cuda")
    cs = torch.full((C,), 2.0, device="cuda")
    st = L.pg_tok_scatter_sum(code, _lib.ptr(tok_d), _lib.ptr(G_d), ctypes.c_long(ldg), ctypes.c_long(N), C, Vp, _lib.ptr(dT),
                              _lib.ptr(cs), _lib.ptr(scratch), ctypes.c_size_t(nbytes), _lib.stream_ptr())
    _lib.check(st, "pg_tok_scatter_sum")
    torch.cuda.synchronize()
    ref, ref_cs = _scatter_ref(tok, Gfull[:, :C].float(), Vp)
    scale = max(ref.abs().max().item(), 1.0)
    assert (dT.double().cpu() - ref).abs().max().item() <= 1e-5 * scale      # fp32 sums of exactly representable rows
    assert (cs.double().cpu() - 2.0 - ref_cs).abs().max().item() <= 1e-5 * max(ref_cs.abs().max().item(), 1.0) * 4
    # deterministic: the same call again gives the same bits
    dT2 = torch.empty_like(dT)
    st = L.pg_tok_scatter_sum(code, _lib.ptr(tok_d), _lib.ptr(G_d), ctypes.c_long(ldg), ctypes.c_long(N), C, Vp, _lib.ptr(dT2),
                              None, _lib.ptr(scratch), ctypes.c_size_t(nbytes), _lib.stream_ptr())
    _lib.check(st, "pg_tok_scatter_sum")
    torch.cuda.synchronize()
    assert torch.equal(dT, dT2)


def test_tok_scatter_sum_rejects_bad_arguments():
    L = _lib.lib()
    L.pg_tok_scatter_scratch_bytes.restype = ctypes.c_size_t
    assert L.pg_tok_scatter_scratch_bytes(ctypes.c_long(100), 64, 129) == 0
    tok = torch.zeros(100, dtype=torch.int32, device="cuda")
    G = torch.zeros(100, 64, device="cuda")
    dT = torch.zeros(64, 64, device="cuda")
    small = torch.empty(1024, dtype=torch.uint8, device="cuda")
    st = L.pg_tok_scatter_sum(_lib.PG_F32, _lib.ptr(tok), _lib.ptr(G), ctypes.c_long(64), ctypes.c_long(100), 64, 64, _lib.ptr(dT),
                              None, _lib.ptr(small), ctypes.c_size_t(1024), _lib.stream_ptr())
    assert st == -1 and b"scratch too small" in L.pg_last_error()
    st = L.pg_tok_scatter_sum(_lib.PG_F32, _lib.ptr(tok), _lib.ptr(G), ctypes.c_long(64), ctypes.c_long(100), 62, 64, _lib.ptr(dT),
                              None, _lib.ptr(small), ctypes.c_size_t(1024), _lib.stream_ptr())
    assert st == -1
