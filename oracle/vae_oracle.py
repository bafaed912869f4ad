"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the POLYGON SMILES-VAE hot path.

A numpy restatement, written from the equations, of what the reference computes with
stock torch modules.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module; the product path (``polygon_amd``) never does.

Parity pinning: the reference ships no tests and no golden vectors, and its
arithmetic lives in torch (``torch>=1.4.0``, container torch 2.10.0+rocm7.0).  The oracle is
therefore pinned by fixtures generated in the build container from the *real* reference,
imported through ``oracle/ref_loader.py`` by ``oracle/make_golden.py`` and committed under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks every function here against them.

Each function cites the reference lines it restates (paths relative to /root/reference).
Parameter names are the reference's 25 canonical state_dict keys, e.g.
``x_emb.weight``, ``encoder_rnn.weight_ih_l0``, ``decoder_rnn.weight_hh_l2``, ``decoder_fc.bias``.
"""
import math

import numpy as np

# --------------------------------------------------------------------------------------
# vocabulary / tokeniser  (polygon/utils/smiles_char_dict.py:18-27, moses_utils.py:17-95)
# --------------------------------------------------------------------------------------
# The 47 single characters of SmilesCharDictionary.char_idx (values of idx_char), which the
# VAE passes to OneHotVocab (vae_model.py:16-17).  CharVocab sorts them by code point and
# appends the four specials (moses_utils.py:38).
_CHARS = [' ', 'Q', '\n', '#', '%', '(', ')', '+', '-', '.', '0', '1', '2', '3', '4', '5', '6', '7',
          '8', '9', '=', 'A', 'B', 'C', 'F', 'H', 'I', 'N', 'O', 'P', 'S', 'X', 'Y', 'Z', '[', ']',
          'b', 'c', 'n', 'o', 'p', 's', '@', 'R', '/', '\\', 'E']
BOS_S, EOS_S, PAD_S, UNK_S = '<bos>', '<eos>', '<pad>', '<unk>'
ALL_SYMS = sorted(_CHARS) + [BOS_S, EOS_S, PAD_S, UNK_S]
C2I = {c: i for i, c in enumerate(ALL_SYMS)}
I2C = {i: c for i, c in enumerate(ALL_SYMS)}
N_VOCAB = len(ALL_SYMS)
BOS, EOS, PAD, UNK = C2I[BOS_S], C2I[EOS_S], C2I[PAD_S], C2I[UNK_S]


class vocabulary:
    """Context manager: restates the tokeniser over the characters of a data-derived vocabulary
    (OneHotVocab.from_data, moses_utils.py:25-42: sorted characters, then the four specials) for its duration.
    Every function below looks the table up at call time."""

    def __init__(self, chars):
        self.syms = sorted(set(chars) - {BOS_S, EOS_S, PAD_S, UNK_S}) + [BOS_S, EOS_S, PAD_S, UNK_S]

    def __enter__(self):
        g = globals()
        self.saved = {k: g[k] for k in ('ALL_SYMS', 'C2I', 'I2C', 'N_VOCAB', 'BOS', 'EOS', 'PAD', 'UNK')}
        c2i = {c: i for i, c in enumerate(self.syms)}
        g.update(ALL_SYMS=self.syms, C2I=c2i, I2C={i: c for i, c in enumerate(self.syms)}, N_VOCAB=len(self.syms),
                 BOS=c2i[BOS_S], EOS=c2i[EOS_S], PAD=c2i[PAD_S], UNK=c2i[UNK_S])
        return self

    def __exit__(self, *a):
        globals().update(self.saved)


def string2ids(string, add_bos=False, add_eos=False):
    """moses_utils.py:63-83 -- per CHARACTER lookup, unknown -> unk."""
    ids = [C2I.get(c, UNK) for c in string]
    if add_bos:
        ids = [BOS] + ids
    if add_eos:
        ids = ids + [EOS]
    return ids


def ids2string(ids, rem_bos=True, rem_eos=True):
    """moses_utils.py:85-95 -- strips ids[0]==bos and ids[-1]==eos, specials rendered literally."""
    ids = list(ids)
    if len(ids) == 0:
        return ''
    if rem_bos and ids[0] == BOS:
        ids = ids[1:]
    if rem_eos and ids[-1] == EOS:      # IndexError on [bos] alone, as the reference
        ids = ids[:-1]
    return ''.join(I2C.get(int(i), UNK_S) for i in ids)


def collate(strings):
    """vae_model.py:142-147 -- stable sort by length descending, then bos + ids + eos."""
    data = sorted(list(strings), key=len, reverse=True)
    return [np.asarray(string2ids(s, True, True), dtype=np.int64) for s in data]


# --------------------------------------------------------------------------------------
# schedulers (polygon/vae/vae_misc.py:6-61)
# --------------------------------------------------------------------------------------
def kl_weight(epoch, n_epoch, kl_start=0, kl_w_start=0.0, kl_w_end=1.0):
    """vae_misc.py:18-22."""
    inc = (kl_w_end - kl_w_start) / (n_epoch - kl_start)
    k = (epoch - kl_start) if epoch >= kl_start else 0
    return kl_w_start + k * inc


def sgdr_lr_table(base_lr, n_steps, lr_n_period=10, lr_n_mult=1, lr_end=0.0003):
    """vae_misc.py:25-61: lr after the ctor's implicit step and after each explicit step()."""
    out = []
    cur, t_end = 0, lr_n_period
    for _ in range(n_steps + 1):
        cur += 1
        out.append(lr_end + (base_lr - lr_end) * (1 + math.cos(math.pi * cur / t_end)) / 2)
        if cur == t_end:
            cur = 0
            t_end = lr_n_mult * t_end
    return out


# --------------------------------------------------------------------------------------
# model arithmetic
# --------------------------------------------------------------------------------------
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_cell(gi, gh, h):
    """torch.nn.GRU cell, gate order r|z|n: returns h', (r, z, n, hn)."""
    H = h.shape[-1]
    r = _sigmoid(gi[..., :H] + gh[..., :H])
    z = _sigmoid(gi[..., H:2 * H] + gh[..., H:2 * H])
    hn = gh[..., 2 * H:]
    n = np.tanh(gi[..., 2 * H:] + r * hn)
    return (1.0 - z) * n + z * h, (r, z, n, hn)


def _pad(seqs, pad=None):
    pad = PAD if pad is None else pad      # looked up at call time: tests restate the oracle over other vocabularies
    B = len(seqs)
    lens = np.array([len(s) for s in seqs], dtype=np.int64)
    T = int(lens.max())
    x = np.full((B, T), pad, dtype=np.int64)
    for i, s in enumerate(seqs):
        x[i, :len(s)] = s
    return x, lens


def _gru_forward(x_in, lens, h0, w_ih, w_hh, b_ih, b_hh, keep=True, reverse=False):
    """Runs one GRU layer over padded [B,T,I] inputs with packed semantics: row b only steps
    while t < lens[b]; outputs past the length are zero (pad_packed_sequence).  reverse=True is the
    second direction of nn.GRU(bidirectional=True) on a PackedSequence: t runs T-1 .. 0, so row b
    starts from h0 at its own last token (each sequence reversed within its length); cache[t] and
    out[:, t] stay indexed by t."""
    B, T, _ = x_in.shape
    H = w_hh.shape[1]
    h = h0.copy()
    out = np.zeros((B, T, H), dtype=x_in.dtype)
    cache = [None] * T if keep else []
    for t in (range(T - 1, -1, -1) if reverse else range(T)):
        act = lens > t
        gi = x_in[:, t] @ w_ih.T + b_ih
        gh = h @ w_hh.T + b_hh
        hp, (r, z, n, hn) = gru_cell(gi, gh, h)
        hprev = h
        h = np.where(act[:, None], hp, h)
        out[:, t] = np.where(act[:, None], hp, 0.0)
        if keep:
            cache[t] = (act, hprev, r, z, n, hn)
    return out, h, cache


def _gru_backward(d_out, d_hT, x_in, cache, w_ih, w_hh, reverse=False):
    """BPTT for _gru_forward (reverse: of its reverse direction, so t runs 0 .. T-1).  d_out [B,T,H]
    grads of outputs, d_hT [B,H] grad of final state.  Returns dx_in, dh0, dw_ih, dw_hh, db_ih, db_hh."""
    B, T, _ = x_in.shape
    H = w_hh.shape[1]
    dw_ih = np.zeros_like(w_ih); dw_hh = np.zeros_like(w_hh)
    db_ih = np.zeros(3 * H, dtype=w_ih.dtype); db_hh = np.zeros(3 * H, dtype=w_ih.dtype)
    dx = np.zeros_like(x_in)
    dh = d_hT.copy()
    for t in (range(T) if reverse else range(T - 1, -1, -1)):
        act, hprev, r, z, n, hn = cache[t]
        m = act[:, None]
        dht = np.where(m, dh + d_out[:, t], 0.0)     # grad wrt h_t for stepping rows
        dn = dht * (1.0 - z)
        dz = dht * (hprev - n)
        dnp = dn * (1.0 - n * n)
        dzp = dz * z * (1.0 - z)
        drp = dnp * hn * r * (1.0 - r)
        dgi = np.concatenate([drp, dzp, dnp], axis=1)
        dgh = np.concatenate([drp, dzp, dnp * r], axis=1)
        dw_ih += dgi.T @ x_in[:, t]
        dw_hh += dgh.T @ hprev
        db_ih += dgi.sum(0)
        db_hh += dgh.sum(0)
        dx[:, t] = dgi @ w_ih
        dh = np.where(m, dht * z + dgh @ w_hh, dh)   # rows not stepping pass their grad through
    return dx, dh, dw_ih, dw_hh, db_ih, db_hh


def encoder_shape(P):
    """(q_n_layers, q_bidir) read off the parameter names (vae_model.py:63-70)."""
    Lq = 0
    while 'encoder_rnn.weight_ih_l%d' % Lq in P:
        Lq += 1
    return Lq, 'encoder_rnn.weight_ih_l0_reverse' in P


def encoder_forward(P, seqs, eps, keep=True, drop_masks=None):
    """vae_model.py:216-239 for any encoder shape of vae_model.py:63-70.  seqs: list of int arrays (bos..eos);
    eps [B,Dz] replaces randn_like.  drop_masks: optional list (len Lq-1) of [B,T,Hq*dirs] multiplicative masks
    (already scaled) on the outputs of layers 0..Lq-2 (nn.GRU's inter-layer dropout, q_dropout).  The heads read the
    last layer's final states, forward then reverse (vae_model.py:228-229).
    Returns dict(z, kl, mu, logvar, h) (+ caches)."""
    x, lens = _pad(seqs)
    emb = P['x_emb.weight'][x]
    Lq, bidir = encoder_shape(P)
    Hq = P['encoder_rnn.weight_hh_l0'].shape[1]
    h0 = np.zeros((len(seqs), Hq), dtype=emb.dtype)
    layers = []
    inp = emb
    for l in range(Lq):
        outs, fins, caches = [], [], []
        for sfx in (('', '_reverse') if bidir else ('',)):
            nm = 'encoder_rnn.%%s_l%d%s' % (l, sfx)
            out, hT, cache = _gru_forward(inp, lens, h0, P[nm % 'weight_ih'], P[nm % 'weight_hh'], P[nm % 'bias_ih'],
                                          P[nm % 'bias_hh'], keep, reverse=bool(sfx))
            outs.append(out); fins.append(hT); caches.append(cache)
        layers.append((inp, caches))
        inp = np.concatenate(outs, axis=-1)
        if drop_masks is not None and l < Lq - 1:
            inp = inp * drop_masks[l]
    h = np.concatenate(fins, axis=-1)
    mu = h @ P['q_mu.weight'].T + P['q_mu.bias']
    logvar = h @ P['q_logvar.weight'].T + P['q_logvar.bias']
    z = mu + np.exp(logvar / 2) * eps
    kl = 0.5 * (np.exp(logvar) + mu ** 2 - 1 - logvar).sum(1).mean()
    return dict(z=z, kl=kl, mu=mu, logvar=logvar, h=h, x=x, lens=lens, emb=emb, cache=layers[0][1][0], eps=eps,
                layers=layers, drop_masks=drop_masks)


def n_decoder_layers(P):
    L = 0
    while 'decoder_rnn.weight_ih_l%d' % L in P:
        L += 1
    return L


def decoder_forward(P, seqs, z, drop_masks=None, keep=True):
    """vae_model.py:241-275.  drop_masks: optional list (len L-1) of [B,T,Hd] multiplicative masks
    (already scaled by 1/(1-p)) applied to the outputs of layers 0..L-2 (torch GRU inter-layer
    dropout, vae_model.py:90).  Returns dict(recon, y [B,T,V], ...)."""
    x, lens = _pad(seqs)
    B, T = x.shape
    L = n_decoder_layers(P)
    emb = P['x_emb.weight'][x]
    x_in = np.concatenate([emb, np.repeat(z[:, None, :], T, axis=1)], axis=-1)
    h0 = z @ P['decoder_lat.weight'].T + P['decoder_lat.bias']
    layers = []
    inp = x_in
    for l in range(L):
        out, _, cache = _gru_forward(inp, lens, h0, P['decoder_rnn.weight_ih_l%d' % l],
                                     P['decoder_rnn.weight_hh_l%d' % l], P['decoder_rnn.bias_ih_l%d' % l],
                                     P['decoder_rnn.bias_hh_l%d' % l], keep)
        layers.append((inp, cache, out))
        inp = out
        if drop_masks is not None and l < L - 1:
            inp = out * drop_masks[l]
    y = inp @ P['decoder_fc.weight'].T + P['decoder_fc.bias']
    logits = y[:, :-1].reshape(-1, y.shape[-1])
    tgt = x[:, 1:].reshape(-1)
    w = (tgt != PAD)
    mx = logits.max(1, keepdims=True)
    lse = mx[:, 0] + np.log(np.exp(logits - mx).sum(1))
    nll = lse - logits[np.arange(len(tgt)), tgt]
    count = int(w.sum())
    recon = (nll * w).sum() / count
    return dict(recon=recon, y=y, x=x, lens=lens, emb=emb, h0=h0, layers=layers, top=inp, z=z,
                count=count, drop_masks=drop_masks)


def vae_forward(P, seqs, eps, drop_masks=None, enc_drop_masks=None):
    """vae_model.py:151-166 -> (kl, recon) plus caches."""
    enc = encoder_forward(P, seqs, eps, drop_masks=enc_drop_masks)
    dec = decoder_forward(P, seqs, enc['z'], drop_masks)
    return enc, dec


def vae_backward(P, enc, dec, g_kl=1.0, g_recon=1.0):
    """Gradients of g_kl*kl + g_recon*recon wrt the 25 parameters (what loss.backward() yields at
    vae_trainer.py:134 with g_kl = kl_weight, g_recon = 1).  Returns (grads dict, dz)."""
    G = {k: np.zeros_like(v) for k, v in P.items()}
    x, lens = dec['x'], dec['lens']
    B, T = x.shape
    L = n_decoder_layers(P)
    V = P['decoder_fc.weight'].shape[0]
    # ---- cross entropy (vae_model.py:268-272)
    y = dec['y']
    logits = y[:, :-1]
    tgt = x[:, 1:]
    w = (tgt != PAD)
    mx = logits.max(-1, keepdims=True)
    p = np.exp(logits - mx)
    p /= p.sum(-1, keepdims=True)
    onehot = np.zeros_like(p)
    bi, ti = np.nonzero(w)
    onehot[bi, ti, tgt[bi, ti]] = 1.0
    dlog = (p - onehot) * w[..., None] * (g_recon / dec['count'])
    dy = np.zeros_like(y)
    dy[:, :-1] = dlog
    top = dec['top']
    G['decoder_fc.weight'] = dy.reshape(-1, V).T @ top.reshape(B * T, -1)
    G['decoder_fc.bias'] = dy.reshape(-1, V).sum(0)
    d_out = dy @ P['decoder_fc.weight']
    # ---- decoder GRU stack
    dh0 = np.zeros_like(dec['h0'])
    for l in range(L - 1, -1, -1):
        inp, cache, out = dec['layers'][l]
        if dec['drop_masks'] is not None and l < L - 1:
            d_out = d_out * dec['drop_masks'][l]
        dx, dh0_l, dwi, dwh, dbi, dbh = _gru_backward(d_out, np.zeros_like(dh0), inp, cache,
                                                      P['decoder_rnn.weight_ih_l%d' % l],
                                                      P['decoder_rnn.weight_hh_l%d' % l])
        G['decoder_rnn.weight_ih_l%d' % l] = dwi
        G['decoder_rnn.weight_hh_l%d' % l] = dwh
        G['decoder_rnn.bias_ih_l%d' % l] = dbi
        G['decoder_rnn.bias_hh_l%d' % l] = dbh
        dh0 += dh0_l
        d_out = dx
    E = P['x_emb.weight'].shape[1]
    d_emb_dec = d_out[..., :E]
    dz = d_out[..., E:].sum(1)
    G['decoder_lat.weight'] = dh0.T @ dec['z']
    G['decoder_lat.bias'] = dh0.sum(0)
    dz = dz + dh0 @ P['decoder_lat.weight']
    demb = np.zeros_like(P['x_emb.weight'])
    np.add.at(demb, x.reshape(-1), d_emb_dec.reshape(B * T, E))
    # ---- reparameterisation + KL (vae_model.py:232-236)
    mu, logvar, eps = enc['mu'], enc['logvar'], enc['eps']
    dmu = dz + g_kl * mu / B
    dlv = dz * eps * 0.5 * np.exp(logvar / 2) + g_kl * 0.5 * (np.exp(logvar) - 1.0) / B
    h = enc['h']
    G['q_mu.weight'] = dmu.T @ h
    G['q_mu.bias'] = dmu.sum(0)
    G['q_logvar.weight'] = dlv.T @ h
    G['q_logvar.bias'] = dlv.sum(0)
    dh = dmu @ P['q_mu.weight'] + dlv @ P['q_logvar.weight']
    # ---- encoder GRU stack: the final-state gradient enters the last layer, forward half | reverse half
    Te = enc['x'].shape[1]
    Lq, bidir = encoder_shape(P)
    Hq = P['encoder_rnn.weight_hh_l0'].shape[1]
    sfxs = ('', '_reverse') if bidir else ('',)
    d_outs = np.zeros((B, Te, Hq * len(sfxs)), dtype=h.dtype)
    for l in range(Lq - 1, -1, -1):
        inp, caches = enc['layers'][l]
        if enc['drop_masks'] is not None and l < Lq - 1:
            d_outs = d_outs * enc['drop_masks'][l]
        dxe = np.zeros_like(inp)
        for k, sfx in enumerate(sfxs):
            nm = 'encoder_rnn.%%s_l%d%s' % (l, sfx)
            d_fin = dh[:, k * Hq:(k + 1) * Hq] if l == Lq - 1 else np.zeros((B, Hq), dtype=h.dtype)
            dx, _, dwi, dwh, dbi, dbh = _gru_backward(d_outs[..., k * Hq:(k + 1) * Hq], d_fin, inp, caches[k],
                                                      P[nm % 'weight_ih'], P[nm % 'weight_hh'], reverse=bool(sfx))
            G[nm % 'weight_ih'] = dwi
            G[nm % 'weight_hh'] = dwh
            G[nm % 'bias_ih'] = dbi
            G[nm % 'bias_hh'] = dbh
            dxe = dxe + dx
        d_outs = dxe
    np.add.at(demb, enc['x'].reshape(-1), dxe.reshape(B * Te, -1))
    demb[PAD] = 0.0          # nn.Embedding(padding_idx=pad), vae_model.py:54
    G['x_emb.weight'] = demb
    return G, dz


PARAM_ORDER_HEAD = ['x_emb.weight', 'encoder_rnn.weight_ih_l0', 'encoder_rnn.weight_hh_l0',
                    'encoder_rnn.bias_ih_l0', 'encoder_rnn.bias_hh_l0', 'q_mu.weight', 'q_mu.bias',
                    'q_logvar.weight', 'q_logvar.bias', 'decoder_lat.weight', 'decoder_lat.bias']   # the default shape


def param_order(P):
    """Order of model.vae.parameters() (vae_model.py:101-115) = optimiser order (vae_trainer.py:99-100); nn.GRU lists
    weight_ih, weight_hh, bias_ih, bias_hh per layer, forward then reverse."""
    Lq, bidir = encoder_shape(P)
    names = ['x_emb.weight']
    for l in range(Lq):
        for sfx in (('', '_reverse') if bidir else ('',)):
            names += ['encoder_rnn.%s_l%d%s' % (k, l, sfx) for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    names += ['q_mu.weight', 'q_mu.bias', 'q_logvar.weight', 'q_logvar.bias', 'decoder_lat.weight', 'decoder_lat.bias']
    for l in range(n_decoder_layers(P)):
        names += ['decoder_rnn.%s_l%d' % (k, l) for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    names += ['decoder_fc.weight', 'decoder_fc.bias']
    return names


def clip_grad_norm(G, max_norm=50.0):
    """torch.nn.utils.clip_grad_norm_ (vae_trainer.py:136): global L2 norm, coef clamped to 1."""
    total = math.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in G.values()))
    coef = min(max_norm / (total + 1e-6), 1.0)
    return {k: g * coef for k, g in G.items()}, total


def adam_step(P, G, state, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (vae_trainer.py:217-218): state = dict(step, m, v)."""
    state['step'] += 1
    t = state['step']
    bc1 = 1.0 - beta1 ** t
    bc2 = 1.0 - beta2 ** t
    for k in P:
        m = state['m'][k] = beta1 * state['m'][k] + (1 - beta1) * G[k]
        v = state['v'][k] = beta2 * state['v'][k] + (1 - beta2) * G[k] * G[k]
        denom = np.sqrt(v) / math.sqrt(bc2) + eps
        P[k] = P[k] - (lr / bc1) * m / denom
    return P


def adam_state(P):
    return dict(step=0, m={k: np.zeros_like(v) for k, v in P.items()},
                v={k: np.zeros_like(v) for k, v in P.items()})


# --------------------------------------------------------------------------------------
# sampler (vae_model.py:300-358) with the pinned inverse-CDF draw (SURVEY.md section 7 "hard parts")
# --------------------------------------------------------------------------------------
def inverse_cdf_draw(probs, u):
    """Pinned draw: float32 running sum in vocab order, first k with cdf_k > u, clamped to V-1.
    probs [n,V] float32-representable, u [n] in [0,1)."""
    p32 = probs.astype(np.float32)
    cdf = np.cumsum(p32, axis=1, dtype=np.float32)
    k = (cdf > u.astype(np.float32)[:, None]).argmax(1)
    none = ~(cdf > u.astype(np.float32)[:, None]).any(1)
    k[none] = probs.shape[1] - 1
    return k.astype(np.int64), cdf


def sample_tokens(P, z, uniforms=None, max_len=100, temp=1.0, multinomial=True, return_probs=False, forced=None,
                  return_fed=False):
    """Token matrix of VAE.sample: returns (x [n,max_len] int64, end_pads [n]).  uniforms
    [max_len-1, n] feeds inverse_cdf_draw at each step (multinomial=True); argmax otherwise.
    No early exit: all max_len-1 steps run for every row (vae_model.py:329).
    Teacher forcing for step-by-step parity checks (no counterpart in the reference): forced [n,max_len] replaces the
    token FED to the next step (the drawn one is still what x / end_pads record); return_fed adds the matrix of tokens
    actually fed (column i = the token drawn at step i, which the reference keeps feeding after eos, vae_model.py:330)."""
    n = z.shape[0]
    L = n_decoder_layers(P)
    h = [z @ P['decoder_lat.weight'].T + P['decoder_lat.bias'] for _ in range(L)]
    w = np.full(n, BOS, dtype=np.int64)
    x = np.full((n, max_len), PAD, dtype=np.int64)
    x[:, 0] = BOS
    end_pads = np.full(n, max_len, dtype=np.int64)
    eos_mask = np.zeros(n, dtype=bool)
    fed = np.full((n, max_len), BOS, dtype=np.int64)
    probs_log = []
    for i in range(1, max_len):
        inp = np.concatenate([P['x_emb.weight'][w], z], axis=1)
        for l in range(L):
            gi = inp @ P['decoder_rnn.weight_ih_l%d' % l].T + P['decoder_rnn.bias_ih_l%d' % l]
            gh = h[l] @ P['decoder_rnn.weight_hh_l%d' % l].T + P['decoder_rnn.bias_hh_l%d' % l]
            h[l], _ = gru_cell(gi, gh, h[l])
            inp = h[l]
        y = (inp @ P['decoder_fc.weight'].T + P['decoder_fc.bias']) / temp
        y = y - y.max(1, keepdims=True)
        p = np.exp(y)
        p /= p.sum(1, keepdims=True)
        if return_probs:
            probs_log.append(p.copy())
        if multinomial:
            w, _ = inverse_cdf_draw(p, uniforms[i - 1])
        else:
            w = p.argmax(1).astype(np.int64)
        x[~eos_mask, i] = w[~eos_mask]
        i_eos = (~eos_mask) & (w == EOS)
        end_pads[i_eos] = i + 1
        eos_mask |= i_eos
        fed[:, i] = w
        if forced is not None:
            w = np.asarray(forced[:, i], dtype=np.int64)
    out = (x, end_pads)
    if return_probs:
        out += (probs_log,)
    if return_fed:
        out += (fed,)
    return out


def tokens_to_strings(x, end_pads):
    """vae_model.py:354-358."""
    return [ids2string(x[i, :int(end_pads[i])].tolist()) for i in range(x.shape[0])]


def decode_teacher_forced(P, seqs, z):
    """VAE.decode(z, x) (vae_model.py:183-197): argmax of teacher-forced logits, cut at first eos."""
    dec = decoder_forward(P, seqs, z, keep=False)
    xr = dec['y'].argmax(2)
    out = []
    for row in xr:
        pos = np.nonzero(row == EOS)[0]
        out.append(ids2string(row[:pos[0]].tolist() if len(pos) else row.tolist()))
    return out


# --------------------------------------------------------------------------------------
# synthetic GuacaMol-shaped data
# --------------------------------------------------------------------------------------
def synthetic_smiles(n, seed=20250310):
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.normal(47, 12, size=n)), 10, 98).astype(np.int64)
    chars = [c for c in sorted(_CHARS) if c not in (' ', '\n', 'Q')]
    pr = {'C': .30, 'c': .22, '(': .07, ')': .07, 'O': .06, 'N': .05, '1': .05, '=': .04, '2': .03, 'n': .03}
    rest = (1.0 - sum(pr.values())) / (len(chars) - len(pr))
    p = np.array([pr.get(c, rest) for c in chars])
    p /= p.sum()
    idx = rng.choice(len(chars), size=int(lens.sum()), p=p)
    arr = np.array(chars)[idx]
    out, o = [], 0
    for L in lens:
        out.append(''.join(arr[o:o + L]))
        o += L
    return out


def init_params(seed=0, V=N_VOCAB, Hq=256, Dz=128, Hd=512, L=3, dtype=np.float32, Lq=1, bidir=False):
    """Random parameters with torch's default init ranges (uniform +-1/sqrt(H)); x_emb = eye.  Lq / bidir: the
    encoder shape (q_n_layers, q_bidir); the default shape draws exactly what it always drew."""
    rng = np.random.default_rng(seed)

    def U(shape, k):
        return rng.uniform(-k, k, size=shape).astype(dtype)
    P = {'x_emb.weight': np.eye(V, dtype=dtype)}
    k = 1 / math.sqrt(Hq)
    nd = 2 if bidir else 1
    for l in range(Lq):
        for sfx in (('', '_reverse') if bidir else ('',)):
            nm = 'encoder_rnn.%%s_l%d%s' % (l, sfx)
            P[nm % 'weight_ih'] = U((3 * Hq, V if l == 0 else Hq * nd), k); P[nm % 'weight_hh'] = U((3 * Hq, Hq), k)
            P[nm % 'bias_ih'] = U((3 * Hq,), k); P[nm % 'bias_hh'] = U((3 * Hq,), k)
    for nm in ('q_mu', 'q_logvar'):
        P[nm + '.weight'] = U((Dz, Hq * nd), k); P[nm + '.bias'] = U((Dz,), k)
    k = 1 / math.sqrt(Dz)
    P['decoder_lat.weight'] = U((Hd, Dz), k); P['decoder_lat.bias'] = U((Hd,), k)
    k = 1 / math.sqrt(Hd)
    for l in range(L):
        I = V + Dz if l == 0 else Hd
        P['decoder_rnn.weight_ih_l%d' % l] = U((3 * Hd, I), k); P['decoder_rnn.weight_hh_l%d' % l] = U((3 * Hd, Hd), k)
        P['decoder_rnn.bias_ih_l%d' % l] = U((3 * Hd,), k); P['decoder_rnn.bias_hh_l%d' % l] = U((3 * Hd,), k)
    P['decoder_fc.weight'] = U((V, Hd), k); P['decoder_fc.bias'] = U((V,), k)
    return P
