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
