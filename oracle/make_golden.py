"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz|json from the REAL reference.

Run in the build container only (needs /root/reference):  python oracle/make_golden.py
The reference is imported through oracle/ref_loader.py (RDKit stub); its
`torch.randn_like` (eps, vae_model.py:233) and `torch.multinomial` (vae_model.py:337) are
monkey-patched so that the random inputs are the ones stored in the fixture.  Only inputs and
outputs (data) are written; no reference source text is stored.
Weights are NOT stored: they are regenerated from oracle.vae_oracle.init_params(seed, dims).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import vae_oracle as O          # noqa: E402
from oracle.ref_loader import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)

REF2CANON = {'x_emb': 'x_emb', 'encoder_rnn': 'encoder_rnn', 'q_mu': 'q_mu', 'q_logvar': 'q_logvar',
             'decoder_rnn': 'decoder_rnn', 'decoder_lat': 'decoder_lat', 'decoder_fc': 'decoder_fc'}


def canon_named(model):
    """The 25 unique tensors under their top-level names."""
    out = {}
    for name, p in model.named_parameters():
        if name.split('.')[0] in REF2CANON:
            out[name] = p
    return out


def load_params(model, P):
    named = canon_named(model)
    assert set(named) == set(P), (sorted(named), sorted(P))
    with torch.no_grad():
        for k, p in named.items():
            p.copy_(torch.from_numpy(np.asarray(P[k], dtype=np.float32)))


def build(ns, dims, seed, vocab=None):
    if vocab is not None:     # a data-derived vocabulary (vae_model.py:9-20, OneHotVocab.from_data): V = E = len(vocab)
        P = O.init_params(seed=seed, V=len(vocab), **dims)
        model = ns.VAE(vocabulary=vocab, q_d_h=dims['Hq'], d_z=dims['Dz'], d_d_h=dims['Hd'], d_n_layers=dims['L'])
        load_params(model, P)
        return model, P
    P = O.init_params(seed=seed, **dims)
    # q_dropout = 0 for the encoder shapes with more than one layer: torch's own dropout stream cannot be pinned from
    # outside (the dropout path is pinned through pg_dropout_masks_host instead, as the decoder's)
    model = ns.VAE(q_d_h=dims['Hq'], d_z=dims['Dz'], d_d_h=dims['Hd'], d_n_layers=dims['L'],
                   q_n_layers=dims.get('Lq', 1), q_bidir=bool(dims.get('bidir', False)), q_dropout=0.0)
    load_params(model, P)
    return model, P


class patched_eps:
    def __init__(self, eps):
        self.eps = torch.from_numpy(eps)

    def __enter__(self):
        self.orig = torch.randn_like
        torch.randn_like = lambda t, **kw: self.eps.to(t.dtype)
        return self

    def __exit__(self, *a):
        torch.randn_like = self.orig


def vocab_fixture(ns):
    model = ns.VAE()
    voc = model.vocabulary
    cases = ["CCl", "c1ccccc1Br", "C[nH]1cccc1", "N[C@@H](C)C(=O)O", "C/C=C\\C", "", "CC(=O)Oc1ccccc1C(=O)O",
             "[Na+].[Cl-]", "C#N", "Q Q\n", "<pad>", "C%10CCCCC%10", "lrie", "O=S(=O)(N)c1ccc(cc1)Se",
             "FC(F)(F)c1ccccc1", "C1CC1", "XYZARE", "caffeine", "C[C@H](N)C(O)=O", "é", "CNC(=O)c1ccccc1I",
             "B(O)O", "P(=O)(O)O", "s1cccc1", "o1cccc1", "p1ccccc1", "b1ccccc1", "[13CH4]", "C=C-C#C", "0123456789"]
    rec = {'all_syms': [voc.i2c[i] for i in range(len(voc))],
           'bos': voc.bos, 'eos': voc.eos, 'pad': voc.pad, 'unk': voc.unk,
           'string2ids': [[s, voc.string2ids(s, add_bos=True, add_eos=True)] for s in cases]}
    id_cases = [[47, 24, 24, 48], [47, 24, 49, 24, 48], [24, 24], [47], [48], [47, 48], [], [47, 50, 24, 47, 48, 48],
                [49, 49], [24, 48, 24], [47, 0, 1, 2, 48]]
    def _i2s(ids):
        try:
            return voc.ids2string(ids)
        except IndexError:          # reference raises on [bos] alone (moses_utils.py:88-90)
            return {'raises': 'IndexError'}
    rec['ids2string'] = [[ids, _i2s(ids)] for ids in id_cases]
    data = ["CCC", "C", "CCCCC", "NN", "OO", "c1ccccc1", "SSS", "F"]
    d2 = list(data)
    tens = model.get_collate_fn()(d2)
    rec['collate_in'] = data
    rec['collate_sorted'] = d2
    rec['collate_ids'] = [t.tolist() for t in tens]
    rec['state_dict_keys'] = list(model.state_dict().keys())
    rec['param_order'] = [n for n, _ in model.vae.named_parameters()]
    rec['param_shapes'] = [list(p.shape) for p in model.vae.parameters()]
    rec['n_params'] = int(sum(p.numel() for p in model.vae.parameters()))
    # schedulers (vae_misc.py)
    kl = ns.vae_misc.KLAnnealer(10, kl_start=2, kl_w_start=0.1, kl_w_end=0.9)
    rec['kl_annealer'] = {'args': [10, 2, 0.1, 0.9], 'values': [kl(i) for i in range(10)]}
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(1))], lr=3e-3)
    sch = ns.vae_misc.CosineAnnealingLRWithRestart(opt, lr_n_period=4, lr_n_restarts=3, lr_n_mult=2, lr_end=3e-4)
    lrs = [opt.param_groups[0]['lr']]
    for _ in range(14):
        sch.step()
        lrs.append(opt.param_groups[0]['lr'])
    rec['sgdr'] = {'base_lr': 3e-3, 'lr_n_period': 4, 'lr_n_mult': 2, 'lr_end': 3e-4, 'lrs': lrs}
    with open(os.path.join(OUT, 'vocab.json'), 'w') as f:
        json.dump(rec, f, indent=0)


def train_fixture(ns, tag, dims, strings, seed, steps=3, kl_w=0.37, full=True, vocab=None, extra=None):
    model, P = build(ns, dims, seed, vocab)
    model.train()
    data = list(strings)
    batch = model.get_collate_fn()(data)
    B = len(batch)
    rng = np.random.default_rng(seed + 100)
    eps_all = rng.standard_normal((steps, B, dims['Dz'])).astype(np.float32)
    out = {'strings': np.array(strings), 'eps': eps_all, 'kl_w': np.float32(kl_w)}
    if vocab is not None:     # the ids the reference's own tokeniser and collate produce, and its special ids
        T = max(len(t) for t in batch)
        ids = np.full((B, T), model.pad, dtype=np.int64)
        for i, t in enumerate(batch):
            ids[i, :len(t)] = t.numpy()
        out.update(ids=ids, lens=np.array([len(t) for t in batch]), specials=np.array([model.bos, model.eos, model.pad, model.unk]),
                   alphabet=np.array([vocab.i2c[i] for i in range(len(vocab))]))
    if extra:
        out.update(extra(model))
    # forward intermediates on step 0
    with patched_eps(eps_all[0]):
        z, kl, mu = model.forward_encoder(batch, return_mu=True)
    recon, y = model.forward_decoder(batch, z, return_y=True)
    out.update(z=z.detach().numpy(), mu=mu.detach().numpy(), kl=kl.item(), recon=recon.item(), y=y.detach().numpy())
    opt = torch.optim.Adam((p for p in model.vae.parameters() if p.requires_grad), lr=3e-4)
    losses, gnorms = [], []
    for s in range(steps):
        with patched_eps(eps_all[s]):
            kl_l, rec_l = model(batch)
        loss = kl_w * kl_l + rec_l
        opt.zero_grad()
        loss.backward()
        if s == 0:
            named = canon_named(model)
            for k, p in named.items():
                g = p.grad.detach().numpy()
                if full:
                    out['grad/' + k] = g.copy()
                else:
                    out['gnorm/' + k] = np.float64(np.linalg.norm(g.astype(np.float64)))
                    out['gslice/' + k] = g.reshape(-1)[:: max(1, g.size // 257)][:257].copy()
        gn = torch.nn.utils.clip_grad_norm_((p for p in model.vae.parameters() if p.requires_grad), 50)
        opt.step()
        losses.append([kl_l.item(), rec_l.item(), loss.item()])
        gnorms.append(float(gn))
    out['losses'] = np.array(losses)
    out['gnorms'] = np.array(gnorms)
    named = canon_named(model)
    for k, p in named.items():
        a = p.detach().numpy()
        out['pfinal/' + k] = a.copy() if full else a.reshape(-1)[:: max(1, a.size // 257)][:257].copy()
    np.savez_compressed(os.path.join(OUT, 'train_%s.npz' % tag), **out)
    print(tag, 'kl/recon', out['kl'], out['recon'], 'gnorms', gnorms)


def sample_fixture(ns, tag, dims, n, seed, max_len=100, temp=1.0, eos_bias=0.0):
    model, P = build(ns, dims, seed)
    if eos_bias:
        with torch.no_grad():
            model.decoder_fc.bias[O.EOS] += eos_bias
    model.eval()
    rng = np.random.default_rng(seed + 200)
    z = rng.standard_normal((n, dims['Dz'])).astype(np.float32)
    u = rng.random((max_len - 1, n)).astype(np.float32)
    step = {'i': 0}
    V = len(model.vocabulary)

    def fake_multinomial(p, num):
        # pinned inverse-CDF draw; uniforms closer than 1e-3 to a cdf edge are moved to the
        # midpoint of the widest bin so that fp32-level differences cannot flip the draw
        i = step['i']
        p32 = p.detach().numpy().astype(np.float32)
        cdf = np.cumsum(p32, axis=1, dtype=np.float32)
        for r in range(p32.shape[0]):
            if np.min(np.abs(cdf[r] - u[i, r])) < 1e-3:
                kbig = int(np.argmax(p32[r]))
                lo = cdf[r, kbig - 1] if kbig > 0 else 0.0
                u[i, r] = np.float32(0.5 * (lo + cdf[r, kbig]))
        k, _ = O.inverse_cdf_draw(p32, u[i])
        step['i'] += 1
        return torch.from_numpy(k)[:, None]

    orig = torch.multinomial
    torch.multinomial = fake_multinomial
    try:
        strings = model.sample(n, max_len=max_len, z=torch.from_numpy(z), temp=temp)
    finally:
        torch.multinomial = orig
    strings_arg = model.sample(n, max_len=max_len, z=torch.from_numpy(z), temp=temp, multinomial=False)
    np.savez_compressed(os.path.join(OUT, 'sample_%s.npz' % tag), z=z, u=u, strings=np.array(strings),
                        strings_argmax=np.array(strings_arg), temp=np.float32(temp),
                        eos_bias=np.float32(eos_bias), max_len=max_len)
    print(tag, 'sample lens', [len(s) for s in strings][:8])


TINY_STRINGS = ["CCl", "c1ccccc1Br", "C[nH]1cccc1", "N[C@@H](C)C(=O)O", "C", "CC(=O)Oc1ccccc1C(=O)O", "C#N", "OO",
                "CCO", "c1ccncc1"]


def wide_vocab_fixture(ns):
    """train_wide.npz: a data-derived vocabulary of 92 characters + 4 specials = 96 symbols (> 64: the second lane's worth
    of the softmax / cross-entropy / draw kernels), tiny hidden sizes; plus argmax and pinned-multinomial sample strings."""
    rng = np.random.default_rng(2025)
    chars = [chr(c) for c in range(33, 127) if chr(c) not in '<>']          # 92 printable characters
    strings = [''.join(rng.choice(chars, size=int(n))) for n in rng.integers(1, 40, size=14)]
    strings[0] = ''.join(chars[:46])
    strings[1] = ''.join(chars[46:])                                         # every character occurs
    vocab = ns.moses_utils.OneHotVocab.from_data(strings)
    assert len(vocab) == 96, len(vocab)
    dims = dict(Hq=32, Dz=16, Hd=64, L=3)

    def sampler(model):
        model.eval()
        r = np.random.default_rng(77)
        n, max_len = 10, 30
        z = r.standard_normal((n, dims['Dz'])).astype(np.float32)
        u = r.random((max_len - 1, n)).astype(np.float32)
        step = {'i': 0}

        def fake_multinomial(p, num):
            i = step['i']
            p32 = p.detach().numpy().astype(np.float32)
            cdf = np.cumsum(p32, axis=1, dtype=np.float32)
            for row in range(p32.shape[0]):
                if np.min(np.abs(cdf[row] - u[i, row])) < 1e-3:
                    kbig = int(np.argmax(p32[row]))
                    lo = cdf[row, kbig - 1] if kbig > 0 else 0.0
                    u[i, row] = np.float32(0.5 * (lo + cdf[row, kbig]))
            k, _ = O.inverse_cdf_draw(p32, u[i])
            step['i'] += 1
            return torch.from_numpy(k)[:, None]
        orig = torch.multinomial
        torch.multinomial = fake_multinomial
        try:
            s_m = model.sample(n, max_len=max_len, z=torch.from_numpy(z))
        finally:
            torch.multinomial = orig
        s_a = model.sample(n, max_len=max_len, z=torch.from_numpy(z), multinomial=False)
        model.train()
        return dict(sample_z=z, sample_u=u, sample_strings=np.array(s_m), sample_strings_argmax=np.array(s_a),
                    sample_max_len=max_len)
    train_fixture(ns, 'wide', dims, strings[2:] + strings[:2], seed=21, steps=3, full=True, vocab=vocab, extra=sampler)


def enc_shape_fixtures(ns, tiny, strings):
    """Other encoder shapes (vae_model.py:63-70: q_n_layers, q_bidir): train_enc_*.npz."""
    for tag, extra in (('enc_bi', dict(Lq=1, bidir=True)), ('enc_l2', dict(Lq=2, bidir=False)),
                       ('enc_bi3', dict(Lq=3, bidir=True))):
        train_fixture(ns, tag, dict(tiny, **extra), strings, seed=11, steps=3, full=True)


def main():
    ns = load_reference()
    if sys.argv[1:] == ['wide']:
        return wide_vocab_fixture(ns)
    if sys.argv[1:] == ['enc']:      # only the encoder-shape fixtures (the others are byte-stable, no need to rewrite them)
        tiny = dict(Hq=32, Dz=16, Hd=64, L=3)
        return enc_shape_fixtures(ns, tiny, TINY_STRINGS)
    vocab_fixture(ns)
    tiny = dict(Hq=32, Dz=16, Hd=64, L=3)
    real = dict(Hq=256, Dz=128, Hd=512, L=3)
    tiny_strings = TINY_STRINGS
    train_fixture(ns, 'tiny', tiny, tiny_strings, seed=1, steps=3, full=True)
    train_fixture(ns, 'real', real, O.synthetic_smiles(24, seed=7), seed=2, steps=3, full=False)
    enc_shape_fixtures(ns, tiny, tiny_strings)
    wide_vocab_fixture(ns)
    sample_fixture(ns, 'tiny', tiny, n=12, seed=3, max_len=24, eos_bias=2.0)
    sample_fixture(ns, 'real', real, n=16, seed=4, max_len=100, eos_bias=3.0)
    sample_fixture(ns, 'real_t07', real, n=8, seed=5, max_len=60, temp=0.7, eos_bias=0.0)


if __name__ == '__main__':
    main()
