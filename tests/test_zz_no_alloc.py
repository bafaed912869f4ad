"""pg_clip_adam_static (the entry point that allocates nothing: include/polygon_hip.h) against pg_clip_adam, through the
C ABI, bit for bit (same kernels, same arithmetic), on two calls in a row, the second with a smaller input -- the
library-owned partial-sum array keeps the first call's contents, so stale partials would show in the second result."""
import ctypes

import pytest
import torch

from polygon_amd import _lib

pytestmark = pytest.mark.gpu


def _step(entry, n, p, g, m, v, step, max_norm, scratch=None):
    L = _lib.lib()
    gnorm = torch.zeros((), device="cuda")
    args = [ctypes.c_long(n), _lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), step, ctypes.c_float(3e-4),
            ctypes.c_float(0.9), ctypes.c_float(0.999), ctypes.c_float(1e-8), ctypes.c_float(max_norm)]
    if entry == "pg_clip_adam":
        args.append(_lib.ptr(scratch))
    args += [_lib.ptr(gnorm), _lib.stream_ptr()]
    _lib.check(getattr(L, entry)(*args), entry)
    return gnorm


def test_clip_adam_static_matches_clip_adam_twice():
    gen = torch.Generator(device="cpu").manual_seed(11)
    results = {}
    for entry in ("pg_clip_adam_static", "pg_clip_adam"):
        out = []
        scratch = torch.full((4096,), 7.0, device="cuda")
        # first call: the full model size with a gradient large enough to be clipped; second: a much smaller buffer
        for n, scale, step in ((4_614_236, 3.0, 1), (70_001, 0.01, 2)):
            gen.manual_seed(n)
            p = torch.randn(n, generator=gen).cuda()
            g = (torch.randn(n, generator=gen) * scale).cuda()
            m = (torch.randn(n, generator=gen) * 0.1).cuda()
            v = torch.rand(n, generator=gen).cuda()
            gn = _step(entry, n, p, g, m, v, step, 50.0, scratch)
            torch.cuda.synchronize()
            out.append((p.cpu(), m.cpu(), v.cpu(), gn.cpu(), g.cpu()))
        results[entry] = out
    for (p1, m1, v1, g1, gr), (p2, m2, v2, g2, _) in zip(results["pg_clip_adam_static"], results["pg_clip_adam"]):
        assert torch.equal(g1, g2)
        assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)
        # and the norm is the gradient's (not a stale or partial sum): fp32 tree sum vs float64
        ref = float(gr.double().pow(2).sum().sqrt())
        assert abs(float(g1) - ref) <= 1e-5 * ref
    # the first call must actually have clipped (norm > 50), the second not
    assert float(results["pg_clip_adam_static"][0][3]) > 50.0 > float(results["pg_clip_adam_static"][1][3])


def test_clip_adam_static_rejects_oversize():
    L = _lib.lib()
    x = torch.zeros(8, device="cuda")
    st = L.pg_clip_adam_static(ctypes.c_long((1 << 28) + 1), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), _lib.ptr(x), 1,
                               ctypes.c_float(1e-3), ctypes.c_float(0.9), ctypes.c_float(0.999), ctypes.c_float(1e-8),
                               ctypes.c_float(0.0), None, _lib.stream_ptr())
    assert st == -1
    assert b"pg_clip_adam_static" in L.pg_last_error()
