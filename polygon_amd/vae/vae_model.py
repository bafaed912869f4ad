"""MI355X-native mirror of the reference SMILES VAE (polygon/vae/vae_model.py).

Same class, constructor kwargs, attributes, methods and 74-key ``state_dict`` as the reference
(vae_model.py:22-358); the modules are kept only as parameter containers.  All arithmetic of
``forward`` / ``forward_encoder`` / ``forward_decoder`` / ``sample`` / ``encode`` / ``decode`` runs in the
hand-written gfx950 kernels of ``libpolygon_hip.so`` (C ABI: include/polygon_hip.h) on the model's
``cuda`` device.  There is NO CPU arithmetic path: calling the model on a CPU device raises.

Extra constructor kwarg (inert in the reference, which stores unknown kwargs verbatim):
``compute_dtype`` = ``"fp32"`` (default; f32 MFMA, matches the reference CPU path to <=1e-3) or ``"bf16"``
(bf16 MFMA operands, f32 accumulation, f32 recurrent state, f32 master weights).
"""
import ctypes

import numpy as np
import torch
import torch.nn as nn

from polygon_amd import _lib
from polygon_amd.utils.moses_utils import OneHotVocab
from polygon_amd.utils.smiles_char_dict import SmilesCharDictionary


def get_vocabulary(data=None, device=None):
    """vae_model.py:9-20."""
    if data:
        vocabulary = OneHotVocab.from_data(data)
    else:
        vocabulary = OneHotVocab(SmilesCharDictionary().idx_char.values())
    if device:
        vocabulary.vectors = vocabulary.vectors.to(device)
    return vocabulary


class _Dims(ctypes.Structure):
    _fields_ = [(k, ctypes.c_int) for k in ("V", "E", "Hq", "Dz", "Hd", "L", "pad", "bos", "eos", "unk", "Lq", "qbidir")]


class _Dp(ctypes.Structure):
    _fields_ = [("kl_inv_batch", ctypes.c_float), ("ce_count", ctypes.c_float), ("b_mul", ctypes.c_int),
                ("b_add", ctypes.c_int)]


class TokenBatch(list):
    """What ``collate`` returns: behaves as the reference's list of 1-D int64 tensors (materialised lazily) and
    carries the packed form the kernels consume: ``tokens`` int32 [B,Tmax] on the device, ``lengths`` int32 on host."""

    def __init__(self, tokens, lengths, strings=None):
        super().__init__()
        self.tokens = tokens
        self.lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        self.strings = strings
        self._items = None
        # data parallelism (set by a sharding collate): this rank's pg_dp of the global batch, the global batch's row and
        # token counts, and the global row index of each local row (None: the batch is not a shard)
        self.dp = None
        self.global_rows = int(self.lengths.shape[0])
        self.global_tokens = int(self.lengths.sum())

    def _mat(self):
        if self._items is None:
            t64 = self.tokens.long()
            self._items = [t64[i, :int(n)] for i, n in enumerate(self.lengths)]
        return self._items

    def __len__(self):
        return int(self.lengths.shape[0])

    def __iter__(self):
        return iter(self._mat())

    def __getitem__(self, i):
        return self._mat()[i]

    @property
    def n_tokens(self):
        return int(self.lengths.sum())

    def shard(self, rank, world):
        """Rows rank, rank+world, ... of the (already length-sorted) batch: each shard stays sorted.  The shard carries
        the pg_dp that normalises its losses by the GLOBAL counts (include/polygon_hip.h)."""
        idx = torch.arange(rank, len(self), world, device=self.tokens.device)
        out = TokenBatch(self.tokens.index_select(0, idx).contiguous(), self.lengths[rank::world])
        out.set_global(rank, world, len(self), self.n_tokens)
        return out

    def micro(self, j, k):
        """Rows j, j+k, ... as a batch of their own that still normalises by THIS batch's global counts and keys its
        dropout masks by the global row (the pg_dp of a shard of a shard): gradients and losses of the k micro-batches
        add up to the whole batch's.  Used to keep batches above the persistent kernels' row limit on those kernels."""
        idx = torch.arange(j, len(self), k, device=self.tokens.device)
        out = TokenBatch(self.tokens.index_select(0, idx).contiguous(), self.lengths[j::k])
        mul, add = (self.dp.b_mul, self.dp.b_add) if self.dp is not None else (1, 0)
        out.global_rows, out.global_tokens = self.global_rows, self.global_tokens
        out.dp = _Dp(kl_inv_batch=1.0 / max(1, self.global_rows), ce_count=float(self.global_tokens - self.global_rows),
                     b_mul=int(k * mul), b_add=int(j * mul + add))
        return out

    def set_global(self, rank, world, global_rows, global_tokens):
        self.global_rows, self.global_tokens = int(global_rows), int(global_tokens)
        # every row contributes len-1 cross-entropy targets (the tokeniser never emits pad)
        self.dp = _Dp(kl_inv_batch=1.0 / max(1, self.global_rows), ce_count=float(self.global_tokens - self.global_rows),
                      b_mul=int(world), b_add=int(rank))


def _as_batch(x, pad, device):
    if isinstance(x, TokenBatch):
        if x.tokens.device != device:
            return TokenBatch(x.tokens.to(device), x.lengths)
        return x
    if not isinstance(x, (list, tuple)):
        x = [x]
    if len(x) == 0:
        raise RuntimeError("empty batch")
    lengths = np.array([int(t.shape[0]) for t in x], dtype=np.int32)
    if (np.diff(lengths) > 0).any():
        # torch.nn.utils.rnn.pack_sequence (vae_model.py:225) raises the same way on unsorted input
        raise RuntimeError("`lengths` array must be sorted in decreasing order; the reference collate sorts the batch "
                           "(vae_model.py:143)")
    tokens = nn.utils.rnn.pad_sequence([t.to(device) for t in x], batch_first=True, padding_value=pad)
    return TokenBatch(tokens.to(torch.int32).contiguous(), lengths)


class _Engine:
    """Per-model glue to libpolygon_hip: flat parameter buffer, shadow weights, workspaces."""

    def __init__(self, model):
        self.model = model
        self.flat = None
        self.shadow = None
        self.shadow_key = None
        self.dims = None
        self.n_params = 0

    # ---- parameters ------------------------------------------------------------------------------
    def params(self):
        return list(self.model.vae.parameters())

    def ensure_flat(self):
        """Makes the parameters (25 for the default shapes) views of one flat fp32 buffer in optimiser order (include/polygon_hip.h)."""
        m = self.model
        ps = self.params()
        dev = ps[0].device
        if dev.type != "cuda":
            raise _lib.PolygonHipError(
                "polygon_amd.VAE computes only on an MI355X (`cuda`) device; the model is on %s. Move it with "
                "model.to('cuda') -- there is no CPU path in this package." % dev)
        ok = self.flat is not None and self.flat.device == dev
        if ok:
            off = 0
            base = self.flat.data_ptr()
            for p in ps:
                if p.data_ptr() != base + 4 * off or p.dtype != torch.float32:
                    ok = False
                    break
                off += p.numel()
        if not ok:
            for p in ps:
                if p.dtype != torch.float32:
                    raise _lib.PolygonHipError("parameters must stay float32 master weights (choose bf16 arithmetic with "
                                               "VAE(compute_dtype='bf16'))")
            total = sum(p.numel() for p in ps)
            flat = torch.empty(total, device=dev, dtype=torch.float32)
            off = 0
            with torch.no_grad():
                for p in ps:
                    n = p.numel()
                    flat[off:off + n].copy_(p.data.reshape(-1))
                    p.data = flat[off:off + n].view(p.shape)
                    off += n
            self.flat = flat
            self.n_params = total
            self.shadow_key = None
            d = _Dims(V=len(m.vocabulary), E=m.x_emb.weight.shape[1], Hq=m.q_d_h, Dz=m.d_z, Hd=m.d_d_h, L=m.d_n_layers,
                      pad=m.pad, bos=m.bos, eos=m.eos, unk=m.unk, Lq=int(m.q_n_layers), qbidir=int(bool(m.q_bidir)))
            self.dims = d
            L = _lib.lib()
            L.pg_param_count.restype = ctypes.c_long
            n_lib = L.pg_param_count(ctypes.byref(d))
            if n_lib != total:
                raise _lib.PolygonHipError("parameter layout mismatch: model has %d values, library expects %d" % (total, n_lib))
        return self.flat

    def dtype_code(self):
        return _lib.PG_BF16 if self.model.hip_dtype == torch.bfloat16 else _lib.PG_F32

    @property
    def generic_encoder(self):
        """True for the encoder shapes served by pg_encoder_gen_fwd / _bwd (q_n_layers > 1 or q_bidir, vae_model.py:63-70);
        the reference's default shape has its own kernels (pg_encoder_fwd / _bwd)."""
        m = self.model
        return bool(m.q_bidir) or int(m.q_n_layers) != 1

    @property
    def n_enc(self):
        """Number of leading parameters the encoder's BPTT writes: x_emb, the encoder GRU's, q_mu, q_logvar."""
        m = self.model
        return 1 + 4 * int(m.q_n_layers) * (2 if m.q_bidir else 1) + 4

    @property
    def vpad(self):
        """PG_VPAD(V) of include/polygon_hip.h: width of the logits / probability rows the library writes."""
        return 64 if self.dims.V <= 64 else 128

    def q_dropout(self):
        """nn.GRU's inter-layer dropout of the encoder (vae_model.py:69): train mode and more than one layer only."""
        m = self.model
        return float(m.q_dropout) if (m.training and int(m.q_n_layers) > 1) else 0.0

    def invalidate(self):
        self.shadow_key = None

    def ensure_shadow(self):
        flat = self.ensure_flat()
        key = (flat.data_ptr(), self.dtype_code(), tuple(p._version for p in self.params()))
        if key != self.shadow_key:
            L = _lib.lib()
            L.pg_shadow_bytes.restype = ctypes.c_size_t
            nbytes = L.pg_shadow_bytes(ctypes.byref(self.dims), self.dtype_code())
            if nbytes == 0:
                raise _lib.PolygonHipError("unsupported model dims: " + L.pg_last_error().decode())
            if self.shadow is None or self.shadow.numel() != nbytes or self.shadow.device != flat.device:
                self.shadow = torch.empty(nbytes, dtype=torch.uint8, device=flat.device)
            _lib.check(L.pg_prepare_weights(ctypes.byref(self.dims), self.dtype_code(), _lib.ptr(flat), _lib.ptr(self.shadow),
                                            _lib.stream_ptr()), "pg_prepare_weights")
            self.shadow_key = key
        return self.shadow

    def train_ws(self, B, n_tokens):
        L = _lib.lib()
        L.pg_train_workspace_bytes.restype = ctypes.c_size_t
        nbytes = L.pg_train_workspace_bytes(ctypes.byref(self.dims), self.dtype_code(), int(B), ctypes.c_long(int(n_tokens)))
        if nbytes == 0:
            raise _lib.PolygonHipError("pg_train_workspace_bytes: " + L.pg_last_error().decode())
        return torch.empty(nbytes, dtype=torch.uint8, device=self.flat.device)

    def enc_ws(self, B, n_tokens, keep=False):
        """Workspace the encoder calls take: the training workspace for the default shape, the generic encoder's own one
        otherwise.  keep=True reuses one buffer between calls (the fused step)."""
        if not self.generic_encoder:
            return None
        L = _lib.lib()
        L.pg_encoder_gen_workspace_bytes.restype = ctypes.c_size_t
        need = L.pg_encoder_gen_workspace_bytes(ctypes.byref(self.dims), self.dtype_code(), int(B), ctypes.c_long(int(n_tokens)))
        if need == 0:
            raise _lib.PolygonHipError("pg_encoder_gen_workspace_bytes: " + L.pg_last_error().decode())
        dev = self.flat.device
        if not keep:
            return torch.empty(need, dtype=torch.uint8, device=dev)
        ws = getattr(self, "_ews", None)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self._ews = torch.empty(int(need * 1.25), dtype=torch.uint8, device=dev)
        return ws

    # ---- raw calls ---------------------------------------------------------------------------------
    def encoder_fwd(self, ws, batch, eps, dp, save, z, mu, logvar, kl, p_drop=0.0, seed=0):
        """ws: the training workspace (default encoder shape) or enc_ws(...) (any other shape; p_drop / seed = q_dropout)."""
        L = _lib.lib()
        B, Tmax = batch.tokens.shape
        if self.generic_encoder:
            _lib.check(L.pg_encoder_gen_fwd(ctypes.byref(self.dims), self.dtype_code(), _lib.ptr(self.flat), _lib.ptr(ws),
                                            ctypes.c_size_t(ws.numel()), B, Tmax, _lib.ptr(batch.tokens),
                                            batch.lengths.ctypes.data_as(ctypes.c_void_p), _lib.ptr(eps),
                                            ctypes.byref(dp) if dp is not None else None, ctypes.c_float(p_drop),
                                            ctypes.c_uint64(seed), int(save), _lib.ptr(z), _lib.ptr(mu), _lib.ptr(logvar),
                                            _lib.ptr(kl), _lib.stream_ptr()), "pg_encoder_gen_fwd")
            return
        _lib.check(L.pg_encoder_fwd(ctypes.byref(self.dims), self.dtype_code(), _lib.ptr(self.flat), _lib.ptr(self.shadow),
                                    _lib.ptr(ws), ctypes.c_size_t(ws.numel()), B, Tmax, _lib.ptr(batch.tokens),
                                    batch.lengths.ctypes.data_as(ctypes.c_void_p), _lib.ptr(eps),
                                    ctypes.byref(dp) if dp is not None else None, int(save), _lib.ptr(z), _lib.ptr(mu),
                                    _lib.ptr(logvar), _lib.ptr(kl), _lib.stream_ptr()), "pg_encoder_fwd")

    def decoder_fwd(self, ws, batch, z, dp, p_drop, seed, all_steps, save, recon):
        L = _lib.lib()
        B, Tmax = batch.tokens.shape
        logits = ctypes.c_void_p(0)
        _lib.check(L.pg_decoder_fwd(ctypes.byref(self.dims), self.dtype_code(), _lib.ptr(self.flat), _lib.ptr(self.shadow),
                                    _lib.ptr(ws), ctypes.c_size_t(ws.numel()), B, Tmax, _lib.ptr(batch.tokens),
                                    batch.lengths.ctypes.data_as(ctypes.c_void_p), _lib.ptr(z),
                                    ctypes.byref(dp) if dp is not None else None, ctypes.c_float(p_drop),
                                    ctypes.c_uint64(seed), int(all_steps), int(save), _lib.ptr(recon), ctypes.byref(logits),
                                    _lib.stream_ptr()), "pg_decoder_fwd")
        return logits.value

    def decoder_bwd(self, ws, batch, dp, p_drop, seed, all_steps, g_host, g_dev, grads, dz):
        L = _lib.lib()
        B, Tmax = batch.tokens.shape
        _lib.check(L.pg_decoder_bwd(ctypes.byref(self.dims), self.dtype_code(), _lib.ptr(self.flat), _lib.ptr(self.shadow),
                                    _lib.ptr(ws), ctypes.c_size_t(ws.numel()), B, Tmax,
                                    batch.lengths.ctypes.data_as(ctypes.c_void_p),
                                    ctypes.byref(dp) if dp is not None else None, ctypes.c_float(p_drop),
                                    ctypes.c_uint64(seed), int(all_steps), ctypes.c_float(g_host), _lib.ptr(g_dev),
                                    _lib.ptr(grads), _lib.ptr(dz), _lib.stream_ptr()), "pg_decoder_bwd")

    def encoder_bwd(self, ws, batch, dp, dz, g_host, g_dev, grads, p_drop=0.0, seed=0):
        L = _lib.lib()
        B, Tmax = batch.tokens.shape
        if self.generic_encoder:
            _lib.check(L.pg_encoder_gen_bwd(ctypes.byref(self.dims), self.dtype_code(), _lib.ptr(self.flat), _lib.ptr(ws),
                                            ctypes.c_size_t(ws.numel()), B, Tmax, batch.lengths.ctypes.data_as(ctypes.c_void_p),
                                            ctypes.byref(dp) if dp is not None else None, ctypes.c_float(p_drop),
                                            ctypes.c_uint64(seed), _lib.ptr(dz), ctypes.c_float(g_host), _lib.ptr(g_dev),
                                            _lib.ptr(grads), _lib.stream_ptr()), "pg_encoder_gen_bwd")
            return
        _lib.check(L.pg_encoder_bwd(ctypes.byref(self.dims), self.dtype_code(), _lib.ptr(self.flat), _lib.ptr(self.shadow),
                                    _lib.ptr(ws), ctypes.c_size_t(ws.numel()), B, Tmax,
                                    batch.lengths.ctypes.data_as(ctypes.c_void_p),
                                    ctypes.byref(dp) if dp is not None else None, _lib.ptr(dz), ctypes.c_float(g_host),
                                    _lib.ptr(g_dev), _lib.ptr(grads), _lib.stream_ptr()), "pg_encoder_bwd")

    # ---- fused training step (no autograd): the data-parallel hot path ------------------------------------
    def decoder_param_start(self):
        """Index in the flat buffers where the decoder-only parameters (decoder_lat .. decoder_fc) start."""
        ps = self.params()
        return sum(p.numel() for p in ps[:self.n_enc])

    def train_step(self, batch, kl_weight, grads, eps=None, dp=None, p_drop=0.0, seed=0, comm=None, q_drop=0.0, q_seed=0):
        """forward + hand-written BPTT of  kl_weight * kl + recon  into the flat ``grads`` buffer (zeroed here).
        ``comm`` (optional polygon_amd.parallel.GradComm) all-reduces the decoder bucket while the encoder BPTT
        runs, then the encoder bucket.  p_drop / seed: the decoder's inter-layer dropout (d_dropout); q_drop / q_seed: the
        encoder's (q_dropout, only for encoders of more than one layer).  Returns device scalars (kl, recon)."""
        L = _lib.lib()
        # deferral on for the whole step: the decoder's shadow weights are rebuilt on the library's side stream beside the
        # encoder forward, its large weight-gradient GEMMs run there beside the dz chain and the encoder BPTT
        old = L.pg_defer_weight_grads(1)
        try:
            k = self.micro_batches(len(batch))
            if k == 1 or dp is not None:
                return self._train_step(batch, kl_weight, grads, eps, dp, p_drop, seed, comm, q_drop, q_seed)
            # More rows than the one-launch recurrences take (PERSIST_MAX_ROWS): k interleaved micro-batches, each sorted and
            # length-balanced like a data-parallel shard, accumulate into the same gradient buffer; the losses add up.
            # The all-reduce (comm) rides on the last one, over the accumulated sums.
            if eps is None:
                eps = torch.randn(len(batch), self.dims.Dz, device=self.flat.device)
            kl = recon = None
            for j in range(k):
                a, b = self._train_step(batch.micro(j, k), kl_weight, grads, eps[j::k].contiguous(), None, p_drop, seed,
                                        comm if j == k - 1 else None, q_drop, q_seed, zero=(j == 0))
                kl, recon = (a.clone(), b.clone()) if j == 0 else (kl + a, recon + b)
            return kl, recon
        finally:
            L.pg_defer_weight_grads(old)

    PERSIST_MAX_ROWS = 1024     # dec_persist_supported (pg_persist_dec.hip): rows per launch of the persistent decoder kernels

    def micro_batches(self, B):
        """Number of micro-batches the fused step splits B rows into: 1 unless the bf16 persistent decoder kernels cover
        this model and B exceeds their row limit (fp32 and other shapes run the per-step kernels, which take any B)."""
        m = self.model
        if self.dtype_code() != _lib.PG_BF16 or int(m.d_d_h) != 512 or not 1 <= int(m.d_n_layers) <= 4:
            return 1
        return max(1, -(-int(B) // self.PERSIST_MAX_ROWS))

    def _train_step(self, batch, kl_weight, grads, eps, dp, p_drop, seed, comm, q_drop=0.0, q_seed=0, zero=True):
        self.ensure_shadow()
        dev = self.flat.device
        B = len(batch)
        L = _lib.lib()
        L.pg_train_workspace_bytes.restype = ctypes.c_size_t
        need = L.pg_train_workspace_bytes(ctypes.byref(self.dims), self.dtype_code(), int(B), ctypes.c_long(batch.n_tokens))
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self._ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=dev)
        sm = getattr(self, "_small", None)
        if sm is None or sm[0].shape[0] < B or sm[0].device != dev:
            sm = self._small = (torch.empty(max(B, 64), self.dims.Dz, device=dev), torch.empty(max(B, 64), self.dims.Dz, device=dev),
                                torch.empty(2, device=dev))
        z, dz, scal = sm
        if dp is None:
            dp = getattr(batch, "dp", None)        # a shard made by the sharding collate carries its own pg_dp
        if eps is None:
            eps = torch.randn(B, self.dims.Dz, device=dev)
        ews = self.enc_ws(B, batch.n_tokens, keep=True)
        if ews is None:
            ews = ws
        if zero:
            # FusedClipAdam.step (pg_clip_adam_zero) leaves its buffer zeroed on the stream it ran on: nothing to fill then
            clean = getattr(self, "clean_grads", None)
            if clean != (grads.data_ptr(), torch.cuda.current_stream(dev).cuda_stream):
                grads.zero_()
        self.clean_grads = None        # dirtied from here on
        self.encoder_fwd(ews, batch, eps, dp, 1, z, None, None, scal[0:1], q_drop, q_seed)
        self.decoder_fwd(ws, batch, z, dp, p_drop, seed, 0, 1, scal[1:2])
        # the decoder's large weight-gradient GEMMs run on the library's side stream, beside the dz chain and the encoder
        # BPTT (joined at the end of pg_encoder_bwd).  Under data parallelism the decoder bucket's all-reduce waits for
        # them on the communication stream (GradComm.start_decoder_bucket), not on this one.
        self.decoder_bwd(ws, batch, dp, p_drop, seed, 0, 1.0, None, grads, dz)
        if comm is not None:
            comm.start_decoder_bucket(grads, self.decoder_param_start())
        self.encoder_bwd(ews, batch, dp, dz, float(kl_weight), None, grads, q_drop, q_seed)
        if comm is not None:
            comm.finish(grads, self.decoder_param_start())
        return scal[0], scal[1]

    def eval_losses(self, batch, eps=None, dp=None):
        """(kl, recon) of a batch, forward only (no activations saved, no dropout): the validation pass of the fused
        trainer.  With ``dp`` the values are this rank's share of the global-batch losses (sum over ranks = global)."""
        k = self.micro_batches(len(batch)) if dp is None else 1
        if k > 1:     # as train_step: keep every launch within the persistent kernels' row limit
            self.ensure_flat()
            if eps is None:
                eps = torch.randn(len(batch), self.dims.Dz, device=self.flat.device)
            parts = [self.eval_losses(batch.micro(j, k), eps[j::k].contiguous()) for j in range(k)]
            return sum(p[0] for p in parts), sum(p[1] for p in parts)
        self.ensure_shadow()
        dev = self.flat.device
        B = len(batch)
        L = _lib.lib()
        L.pg_train_workspace_bytes.restype = ctypes.c_size_t
        need = L.pg_train_workspace_bytes(ctypes.byref(self.dims), self.dtype_code(), int(B), ctypes.c_long(batch.n_tokens))
        ws = getattr(self, "_ws", None)
        if ws is None or ws.numel() < need or ws.device != dev:
            ws = self._ws = torch.empty(int(need * 1.25), dtype=torch.uint8, device=dev)
        z = torch.empty(B, self.dims.Dz, device=dev)
        scal = torch.empty(2, device=dev)
        if dp is None:
            dp = getattr(batch, "dp", None)
        if eps is None:
            eps = torch.randn(B, self.dims.Dz, device=dev)      # the reference draws eps in eval mode too (vae_model.py:233)
        ews = self.enc_ws(B, batch.n_tokens, keep=True)
        self.encoder_fwd(ws if ews is None else ews, batch, eps, dp, 0, z, None, None, scal[0:1])
        self.decoder_fwd(ws, batch, z, dp, 0.0, 0, 0, 0, scal[1:2])
        return scal[0], scal[1]

    def grad_views(self, flat_grad):
        out, off = [], 0
        for p in self.params():
            n = p.numel()
            out.append(flat_grad[off:off + n].view(p.shape))
            off += n
        return out


class _EncoderFn(torch.autograd.Function):
    """forward_encoder with hand-written BPTT (pg_encoder_fwd / pg_encoder_bwd)."""

    @staticmethod
    def forward(ctx, eng, batch, eps, need_grad, p_drop, seed, *params):
        dev = eng.flat.device
        B = len(batch)
        Dz = eng.dims.Dz
        ws = eng.enc_ws(B, batch.n_tokens)
        if ws is None:
            ws = eng.train_ws(B, batch.n_tokens)
        z = torch.empty(B, Dz, device=dev)
        mu = torch.empty(B, Dz, device=dev)
        logvar = torch.empty(B, Dz, device=dev)
        kl = torch.empty((), device=dev)
        eng.encoder_fwd(ws, batch, eps, None, need_grad, z, mu, logvar, kl, p_drop, seed)
        ctx.eng, ctx.batch, ctx.ws, ctx.cfg = eng, batch, ws, (p_drop, seed)
        ctx.mark_non_differentiable(mu, logvar)
        return z, kl, mu, logvar

    @staticmethod
    def backward(ctx, gz, gkl, gmu, glv):
        eng = ctx.eng
        grads = torch.zeros(eng.n_params, device=eng.flat.device)
        gkl_dev = gkl.contiguous() if gkl is not None else None
        eng.ensure_shadow()
        eng.encoder_bwd(ctx.ws, ctx.batch, None, gz.contiguous() if gz is not None else None,
                        1.0 if gkl is not None else 0.0, gkl_dev, grads, *ctx.cfg)
        gv = eng.grad_views(grads)
        # decoder-only parameters (decoder_lat .. decoder_fc) receive nothing from the encoder
        n_enc = eng.n_enc
        return (None,) * 6 + tuple(gv[:n_enc]) + (None,) * (len(gv) - n_enc)


class _DecoderFn(torch.autograd.Function):
    """forward_decoder with hand-written BPTT (pg_decoder_fwd / pg_decoder_bwd)."""

    @staticmethod
    def forward(ctx, eng, batch, p_drop, seed, all_steps, need_grad, want_logits, z, *params):
        dev = eng.flat.device
        B = len(batch)
        ws = eng.train_ws(B, batch.n_tokens)
        recon = torch.empty((), device=dev)
        zc = z.detach().to(torch.float32).contiguous()
        logits_ptr = eng.decoder_fwd(ws, batch, zc, None, p_drop, seed, all_steps, need_grad, recon)
        ctx.eng, ctx.batch, ctx.ws = eng, batch, ws
        ctx.cfg = (p_drop, seed, all_steps)
        if want_logits:
            # packed [rows, Vp] fp32 logits live inside the workspace: expose them as a tensor view of it
            n = batch.n_tokens if all_steps else batch.n_tokens - B
            off = logits_ptr - ws.data_ptr()
            vp = eng.vpad
            logits = ws[off:off + n * vp * 4].view(torch.float32).view(n, vp).clone()
        else:
            logits = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(logits)
        return recon, logits

    @staticmethod
    def backward(ctx, grecon, glogits):
        eng = ctx.eng
        p_drop, seed, all_steps = ctx.cfg
        dev = eng.flat.device
        grads = torch.zeros(eng.n_params, device=dev)
        dz = torch.empty(len(ctx.batch), eng.dims.Dz, device=dev)
        eng.ensure_shadow()
        eng.decoder_bwd(ctx.ws, ctx.batch, None, p_drop, seed, all_steps, 1.0, grecon.contiguous(), grads, dz)
        gv = eng.grad_views(grads)
        # encoder-only parameters (encoder_rnn, q_mu, q_logvar = indices 1..n_enc-1) receive nothing from the decoder
        pg = [gv[0]] + [None] * (eng.n_enc - 1) + gv[eng.n_enc:]
        return (None, None, None, None, None, None, None, dz) + tuple(pg)


class VAE(nn.Module):
    def __init__(self, **kwargs):
        super().__init__()
        # defaults (vae_model.py:27-38)
        self.q_cell = "gru"
        self.q_bidir = False
        self.q_d_h = 256
        self.q_n_layers = 1
        self.q_dropout = 0.5
        self.d_cell = "gru"
        self.d_n_layers = 3
        self.d_dropout = 0
        self.d_z = 128
        self.d_d_h = 512
        self.freeze_embeddings = False
        self.vocabulary = None
        self.compute_dtype = "fp32"
        self.__dict__.update(kwargs)

        if self.vocabulary is None:
            self.vocabulary = get_vocabulary()
        for ss in ('bos', 'eos', 'unk', 'pad'):
            setattr(self, ss, getattr(self.vocabulary, ss))

        n_vocab, d_emb = len(self.vocabulary), self.vocabulary.vectors.size(1)
        self.x_emb = nn.Embedding(n_vocab, d_emb, self.pad)
        self.x_emb.weight.data.copy_(self.vocabulary.vectors)
        if self.freeze_embeddings:
            self.x_emb.weight.requires_grad = False

        if self.q_cell != 'gru':
            raise ValueError("Invalid q_cell type, should be one of the ('gru',)")
        self.encoder_rnn = nn.GRU(d_emb, self.q_d_h, num_layers=self.q_n_layers, batch_first=True,
                                  dropout=self.q_dropout if self.q_n_layers > 1 else 0, bidirectional=self.q_bidir)
        q_d_last = self.q_d_h * (2 if self.q_bidir else 1)
        self.q_mu = nn.Linear(q_d_last, self.d_z)
        self.q_logvar = nn.Linear(q_d_last, self.d_z)

        if self.d_cell != 'gru':
            raise ValueError("Invalid d_cell type, should be one of the ('gru',)")
        self.decoder_rnn = nn.GRU(d_emb + self.d_z, self.d_d_h, num_layers=self.d_n_layers, batch_first=True,
                                  dropout=float(self.d_dropout) if self.d_n_layers > 1 else 0)
        self.decoder_lat = nn.Linear(self.d_z, self.d_d_h)
        self.decoder_fc = nn.Linear(self.d_d_h, n_vocab)

        self.encoder = nn.ModuleList([self.encoder_rnn, self.q_mu, self.q_logvar])
        self.decoder = nn.ModuleList([self.decoder_lat, self.decoder_rnn, self.decoder_fc])
        self.vae = nn.ModuleList([self.x_emb, self.encoder, self.decoder])

        if not 1 <= int(self.q_n_layers) <= 4:
            raise ValueError("q_n_layers must be in [1, 4] (PG_MAX_ENC_LAYERS, include/polygon_hip.h), got %r" % (self.q_n_layers,))
        cd = self.compute_dtype
        if cd in ("fp32", "float32", torch.float32):
            self.__dict__['hip_dtype'] = torch.float32
        elif cd in ("bf16", "bfloat16", torch.bfloat16):
            self.__dict__['hip_dtype'] = torch.bfloat16
        else:
            raise ValueError("compute_dtype must be 'fp32' or 'bf16', got %r" % (cd,))
        self.__dict__['_engine'] = _Engine(self)
        self.__dict__['_eps_override'] = None

    # ---- reference helpers ---------------------------------------------------------------------------
    @property
    def device(self):
        return next(self.parameters()).device

    def string2tensor(self, string, device='model'):
        ids = self.vocabulary.string2ids(string, add_bos=True, add_eos=True)
        return torch.tensor(ids, dtype=torch.long, device=self.device if device == 'model' else device)

    def tensor2string(self, tensor):
        return self.vocabulary.ids2string(tensor.tolist(), rem_bos=True, rem_eos=True)

    def get_collate_device(self):
        return self.device

    def get_collate_fn(self, shard=None):
        """vae_model.py:139-149: sorts the batch by length (descending, in place, stable) and tokenises it.  The
        result is a TokenBatch: a list of int64 tensors like the reference's, uploaded as ONE int32 matrix.
        ``shard=(rank, world)`` (data parallelism, no counterpart in the reference): ``data`` is the GLOBAL batch; it is
        sorted on lengths only, this rank keeps rows rank, rank+world, ... (each shard stays sorted and length-balanced,
        SURVEY.md 8e) and tokenises and uploads just those - no rank pays for rows it discards."""
        def collate(data):
            data.sort(key=len, reverse=True)
            n_global = len(data)
            if shard is not None and shard[1] > 1:
                tok_global = sum(map(len, data)) + 2 * n_global
                data = data[shard[0]::shard[1]]
            tokens, lengths = self.vocabulary.encode_batch(data)
            dev = self.device
            t = torch.from_numpy(tokens)
            if dev.type == "cuda":
                t = t.pin_memory().to(dev, non_blocking=True)
            batch = TokenBatch(t, lengths, strings=data)
            if shard is not None and shard[1] > 1:
                batch.set_global(shard[0], shard[1], n_global, tok_global)
            return batch
        return collate

    # ---- hot path --------------------------------------------------------------------------------------
    def _prep(self, x):
        eng = self._engine
        eng.ensure_shadow()
        return eng, _as_batch(x, self.pad, eng.flat.device)

    def forward(self, x):
        """vae_model.py:151-166 -> (kl_loss, recon_loss), attached to autograd."""
        z, kl_loss = self.forward_encoder(x)
        recon_loss = self.forward_decoder(x, z)
        return kl_loss, recon_loss

    def forward_encoder(self, x, return_mu=False, eps=None):
        """vae_model.py:216-239.  ``eps`` (optional [B, d_z]) replaces the reference's ``torch.randn_like(mu)``."""
        eng, batch = self._prep(x)
        B = len(batch)
        if eps is None:
            eps = self._eps_override
        if eps is None:
            eps = torch.randn(B, self.d_z, device=eng.flat.device)
        eps = eps.to(eng.flat.device, torch.float32).contiguous()
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in eng.params())
        q_drop = eng.q_dropout()
        q_seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if q_drop > 0 else 0
        z, kl, mu, logvar = _EncoderFn.apply(eng, batch, eps, need_grad, q_drop, q_seed, *eng.params())
        if return_mu:
            return z, kl, mu
        return z, kl

    def forward_decoder(self, x, z, return_y=False):
        """vae_model.py:241-275."""
        eng, batch = self._prep(x)
        p_drop = float(self.d_dropout) if (self.training and self.d_n_layers > 1) else 0.0
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if p_drop > 0 else 0
        need_grad = torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in eng.params()))
        z = z.to(eng.flat.device)
        recon, logits = _DecoderFn.apply(eng, batch, p_drop, seed, bool(return_y), need_grad, bool(return_y), z,
                                         *eng.params())
        if not return_y:
            return recon
        return recon, self._unpack_logits(batch, logits)

    def _unpack_logits(self, batch, logits):
        """Padded y [B, Tmax, V] from the packed logits (positions past a row's length hold decoder_fc.bias, as
        pad_packed_sequence's zeros give in the reference)."""
        B, Tmax = batch.tokens.shape
        V = len(self.vocabulary)
        lens = batch.lengths.astype(np.int64)
        t_idx, b_idx = _packed_index(lens)
        y = self.decoder_fc.bias.detach().expand(B, Tmax, V).clone()
        y[torch.from_numpy(b_idx).to(y.device), torch.from_numpy(t_idx).to(y.device)] = logits[:, :V]
        return y

    def encode(self, x):
        """vae_model.py:168-181 -> mu."""
        if not isinstance(x, list):
            x = [x]
        z, kl_loss, mu = self.forward_encoder(x, return_mu=True)
        return mu

    def decode(self, z, x=None):
        """vae_model.py:183-200."""
        if x is not None:
            rl, y = self.forward_decoder(x, z, return_y=True)
            xr = y.argmax(2)
            xrt = []
            for i in xr:
                try:
                    q = i[:(i == self.eos).nonzero()[0]]
                    xrt.append(q)
                except IndexError:
                    xrt.append(i)
            smiles = [self.tensor2string(i_x) for i_x in xrt]
        else:
            smiles = self.sample(z.shape[0], z=z, multinomial=True)
        return smiles

    @torch.no_grad()
    def encode_smiles(self, smiles_list, batch_size=1024):
        """mu [n, d_z] of MANY SMILES in INPUT order: the batched form of ``encode(collate([s]))`` that the reference's
        LatentDistance scorer calls once per molecule (polygon/utils/custom_scoring_fcn.py:165-169).
        Strings are length-sorted per chunk as the kernels require and the rows are put back in the caller's order."""
        n = len(smiles_list)
        out = torch.empty(n, self.q_mu.out_features, device=self.device)
        order = sorted(range(n), key=lambda i: len(smiles_list[i]), reverse=True)    # stable, like the reference collate
        collate = self.get_collate_fn()
        for a in range(0, n, batch_size):
            idx = order[a:a + batch_size]
            batch = collate([smiles_list[i] for i in idx])                            # already sorted: order is kept
            mu = VAE.encode(self, batch)            # the device-side encode, whatever a subclass hands its callers
            out[torch.tensor(idx, device=out.device)] = mu
        return out

    def get_latent_space_from_smiles(self, smiles):
        x = self.string2tensor(smiles)
        z, kl_loss = self.forward_encoder([x])
        return z[0]

    def sample_z_prior(self, n_batch):
        return torch.randn(n_batch, self.q_mu.out_features, device=self.x_emb.weight.device)

    def perturb_z(self, z, noise_norm, constant_norm=False):
        """vae_model.py:287-298 (host numpy arithmetic, as the reference)."""
        if noise_norm > 0.0:
            noise_vec = np.random.normal(0, 1, size=z.shape)
            noise_vec = noise_vec / np.linalg.norm(noise_vec)
            if constant_norm:
                return z + (noise_norm * noise_vec)      # tensor + float64 ndarray, exactly as the reference
            noise_amp = np.random.uniform(0, noise_norm, size=(z.shape[0], 1))
            return z + torch.tensor(noise_amp * noise_vec, dtype=z.dtype)
        return z

    def sample_tokens(self, n_batch, max_len=100, z=None, temp=1.0, multinomial=True, uniforms=None, seed=None,
                      row_offset=0, forced=None, return_probs=False):
        """Token matrix of ``sample``: (tokens uint8 [n, max_len], end_pads int32 [n]) on the device.
        ``uniforms`` [max_len-1, n] pins the inverse-CDF draws (parity mode); otherwise Philox(seed, row, step).
        Parity hooks (pg_sample_forced): ``forced`` uint8 [n, max_len] teacher-forces the token fed to every step;
        ``return_probs`` adds the per-step probabilities float [max_len-1, n, 64 or 128 (V > 64)] as a third result."""
        eng = self._engine
        with torch.no_grad():
            eng.ensure_shadow()
            dev = eng.flat.device
            if z is None:
                z = self.sample_z_prior(n_batch)
            z = z.to(dev, torch.float32).contiguous()
            if z.shape[0] != n_batch:
                raise RuntimeError("n_batch (%d) must equal z.shape[0] (%d)" % (n_batch, z.shape[0]))
            L = _lib.lib()
            L.pg_sample_workspace_bytes.restype = ctypes.c_size_t
            CH = 65536                 # rows per library call: bounds the workspace (~22 KB per row) whatever n_batch is
            rows = min(n_batch, CH)
            nbytes = L.pg_sample_workspace_bytes(ctypes.byref(eng.dims), eng.dtype_code(), int(rows))
            if nbytes == 0:
                raise _lib.PolygonHipError("pg_sample_workspace_bytes: " + L.pg_last_error().decode())
            ws = getattr(eng, "_sample_ws", None)      # kept between calls (generate samples 8192 rows every epoch)
            if ws is None or ws.numel() < nbytes or ws.device != dev:
                ws = eng._sample_ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            tokens = torch.empty(n_batch, max_len, dtype=torch.uint8, device=dev)
            end_pads = torch.empty(n_batch, dtype=torch.int32, device=dev)
            if uniforms is not None:
                uniforms = uniforms.to(dev, torch.float32).contiguous()
                if tuple(uniforms.shape) != (max_len - 1, n_batch):
                    raise RuntimeError("uniforms must have shape [max_len-1, n_batch]")
            if seed is None:
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            hooks = forced is not None or return_probs
            if hooks and n_batch > CH:
                raise RuntimeError("the parity hooks (forced / return_probs) take at most %d rows" % CH)
            probs = torch.zeros(max_len - 1, n_batch, eng.vpad, device=dev) if return_probs else None
            if forced is not None:
                forced = forced.to(dev, torch.uint8).contiguous()
                if tuple(forced.shape) != (n_batch, max_len):
                    raise RuntimeError("forced must have shape [n_batch, max_len]")
            for r0 in range(0, n_batch, CH):
                m = min(CH, n_batch - r0)
                u = None if uniforms is None else (uniforms if m == n_batch else uniforms[:, r0:r0 + m].contiguous())
                args = [ctypes.byref(eng.dims), eng.dtype_code(), _lib.ptr(eng.flat), _lib.ptr(eng.shadow), _lib.ptr(ws),
                        ctypes.c_size_t(ws.numel()), int(m), int(max_len), _lib.ptr(z[r0:r0 + m]), _lib.ptr(u),
                        ctypes.c_uint64(seed), ctypes.c_long(int(row_offset) + r0), ctypes.c_float(float(temp)),
                        int(bool(multinomial)), _lib.ptr(tokens[r0:r0 + m]), _lib.ptr(end_pads[r0:r0 + m])]
                if hooks:
                    _lib.check(L.pg_sample_forced(*args, _lib.ptr(forced), _lib.ptr(probs), _lib.stream_ptr()), "pg_sample_forced")
                else:
                    _lib.check(L.pg_sample(*args, _lib.stream_ptr()), "pg_sample")
            return (tokens, end_pads, probs) if return_probs else (tokens, end_pads)

    def sample(self, n_batch, max_len=100, z=None, temp=1.0, multinomial=True):
        """vae_model.py:300-358 -> list of strings (no early exit; specials mid-stream rendered literally)."""
        tokens, end_pads = self.sample_tokens(n_batch, max_len=max_len, z=z, temp=temp, multinomial=multinomial)
        return self.vocabulary.decode_batch(tokens.cpu().numpy(), end_pads.cpu().numpy())

    def sample_sharded(self, n_batch, max_len=100, temp=1.0, multinomial=True, seed=None, group=None):
        """``sample`` over the ranks of a torch.distributed job (no counterpart in the reference): the rows
        are split contiguously, rank r decodes rows [r*q, (r+1)*q) with no collective during the decode, and ONE final
        all-gather of the uint8 token matrix [n, max_len] + end positions hands every rank the whole list.  The prior
        draws z and the sampler's Philox stream are keyed by (seed, global row), so the strings do not depend on the
        number of ranks.  ``seed`` None: rank 0 draws one and broadcasts it.  One process: same result, no collective."""
        import torch.distributed as dist
        from polygon_amd.parallel import GradComm, world_info
        rank, world = world_info()
        dev = self.device
        if seed is None:
            t = torch.randint(0, 2 ** 62, (1,))
            if world > 1:
                comm = GradComm(group)
                t = t.to(dev) if not comm.staged else t
                comm.broadcast(t, 0)
            seed = int(t.item())
        z_all = torch.randn(n_batch, self.q_mu.out_features, device=dev,
                            generator=torch.Generator(device=dev).manual_seed(seed % (2 ** 63 - 1)))
        q = (n_batch + world - 1) // world
        r0, r1 = min(rank * q, n_batch), min((rank + 1) * q, n_batch)
        tok = torch.full((q, max_len), self.pad, dtype=torch.uint8, device=dev)
        ep = torch.zeros(q, dtype=torch.int32, device=dev)
        if r1 > r0:
            t, e = self.sample_tokens(r1 - r0, max_len=max_len, z=z_all[r0:r1], temp=temp, multinomial=multinomial,
                                      seed=seed, row_offset=r0)
            tok[:r1 - r0], ep[:r1 - r0] = t, e
        if world > 1:
            comm = GradComm(group)
            if comm.staged:      # gloo: through the host
                tok_h, ep_h = tok.cpu(), ep.cpu()
                toks = [torch.empty_like(tok_h) for _ in range(world)]
                eps_ = [torch.empty_like(ep_h) for _ in range(world)]
                dist.all_gather(toks, tok_h, group=group)
                dist.all_gather(eps_, ep_h, group=group)
                tok_all, ep_all = torch.cat(toks), torch.cat(eps_)
            else:                # RCCL: one all-gather per tensor on the device, then one D2H copy
                tok_all = torch.empty(world * q, max_len, dtype=torch.uint8, device=dev)
                ep_all = torch.empty(world * q, dtype=torch.int32, device=dev)
                dist.all_gather_into_tensor(tok_all, tok, group=group)
                dist.all_gather_into_tensor(ep_all, ep, group=group)
            tok, ep = tok_all, ep_all
        return self.vocabulary.decode_batch(tok[:n_batch].cpu().numpy(), ep[:n_batch].cpu().numpy())


def _packed_index(lens):
    """(t, b) of every packed row of a length-sorted batch running all its steps."""
    T = int(lens.max())
    t_idx, b_idx = [], []
    for t in range(T):
        bt = int((lens > t).sum())
        t_idx.append(np.full(bt, t, dtype=np.int64))
        b_idx.append(np.arange(bt, dtype=np.int64))
    return np.concatenate(t_idx), np.concatenate(b_idx)
