"""Shared helpers of the parity tests (test-side only; may import the oracle)."""
import numpy as np
import torch

from oracle import vae_oracle as O

TINY = dict(Hq=32, Dz=16, Hd=64, L=3)
REAL = dict(Hq=256, Dz=128, Hd=512, L=3)


def canon_named(model):
    tops = ("x_emb", "encoder_rnn", "q_mu", "q_logvar", "decoder_rnn", "decoder_lat", "decoder_fc")
    return {n: p for n, p in model.named_parameters() if n.split(".")[0] in tops}


def build_model(dims, seed, compute_dtype="fp32", device="cuda", eos_bias=0.0, **kw):
    from polygon_amd.vae.vae_model import VAE
    P = O.init_params(seed=seed, **dims)
    if eos_bias:
        P["decoder_fc.bias"] = P["decoder_fc.bias"].copy()
        P["decoder_fc.bias"][O.EOS] += eos_bias
    if "Lq" in dims or "bidir" in dims:        # other encoder shapes (vae_model.py:63-70)
        kw = dict(dict(q_n_layers=dims.get("Lq", 1), q_bidir=bool(dims.get("bidir", False))), **kw)
    model = VAE(q_d_h=dims["Hq"], d_z=dims["Dz"], d_d_h=dims["Hd"], d_n_layers=dims["L"], compute_dtype=compute_dtype, **kw)
    named = canon_named(model)
    assert set(named) == set(P)
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(torch.from_numpy(P[k]))
    return model.to(device), {k: v.astype(np.float64) for k, v in P.items()}


def _np(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().float().cpu().numpy()
    return np.asarray(a, dtype=np.float64)


def rel_err(a, b):
    a, b = _np(a), _np(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def dropout_masks(seed, p_drop, n_masks, B, T, H, b_mul=1, b_add=0, dtype=np.float64):
    """The inter-layer dropout masks the HIP kernels apply (pg_dropout_masks_host: the kernels' own hash, evaluated on the
    host) as the list of [B,T,H] float64 arrays oracle.decoder_forward(drop_masks=...) takes."""
    import ctypes
    from polygon_amd import _lib
    out = np.empty((n_masks, B, T, H), dtype=np.float32)
    _lib.check(_lib.lib().pg_dropout_masks_host(ctypes.c_uint64(seed), ctypes.c_float(p_drop), n_masks, B, T, H, b_mul, b_add,
                                                out.ctypes.data_as(ctypes.c_void_p)), "pg_dropout_masks_host")
    return [out[l] if dtype == np.float32 else out[l].astype(dtype) for l in range(n_masks)]   # float32: views, no copies
