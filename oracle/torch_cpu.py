"""TEST INFRASTRUCTURE ONLY -- the CPU baseline in stock torch.nn ops.

The reference computes its hot path with stock torch modules on the CPU (nn.Embedding, nn.GRU over a PackedSequence,
nn.Linear, F.cross_entropy, torch.multinomial, clip_grad_norm_, optim.Adam: /root/reference/polygon/vae/vae_model.py:54-98,
216-275,300-351 and vae_trainer.py:133-138).  This module assembles the same computation from those stock ops, written
from the equations of SURVEY.md 8a, so that bench.py's ``cpu_baseline`` can time what a user of the reference would run
on the host cores of the GPU box (the reference's own source never travels there).  It is checked against the numpy
oracle, which is pinned to the reference's golden vectors (tests/test_oracle_golden.py).

Only tests/ and bench.py's cpu_baseline leg import this.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils import rnn

from oracle import vae_oracle as O


class StockVAE(nn.Module):
    def __init__(self, P, d_dropout=0.0):
        super().__init__()
        V, E = P["x_emb.weight"].shape
        Hq = P["encoder_rnn.weight_hh_l0"].shape[1]
        Dz = P["q_mu.weight"].shape[0]
        Hd = P["decoder_rnn.weight_hh_l0"].shape[1]
        L = O.n_decoder_layers(P)
        self.x_emb = nn.Embedding(V, E, padding_idx=O.PAD)
        self.encoder_rnn = nn.GRU(E, Hq, batch_first=True)
        self.q_mu, self.q_logvar = nn.Linear(Hq, Dz), nn.Linear(Hq, Dz)
        self.decoder_rnn = nn.GRU(E + Dz, Hd, num_layers=L, batch_first=True, dropout=d_dropout if L > 1 else 0.0)
        self.decoder_lat, self.decoder_fc = nn.Linear(Dz, Hd), nn.Linear(Hd, V)
        self.L = L
        with torch.no_grad():
            own = dict(self.named_parameters())
            assert set(own) == set(P), sorted(set(own) ^ set(P))
            for k, v in P.items():
                own[k].copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))

    def losses(self, seqs, eps=None):
        """(kl, recon) of a length-sorted list of int64 id tensors (bos..eos)."""
        lens = [len(s) for s in seqs]
        _, h = self.encoder_rnn(rnn.pack_sequence([self.x_emb(s) for s in seqs]))
        h = h[-1]
        mu, logvar = self.q_mu(h), self.q_logvar(h)
        if eps is None:
            eps = torch.randn_like(mu)
        z = mu + (logvar / 2).exp() * eps
        kl = 0.5 * (logvar.exp() + mu ** 2 - 1 - logvar).sum(1).mean()
        x = rnn.pad_sequence(seqs, batch_first=True, padding_value=O.PAD)
        inp = torch.cat([self.x_emb(x), z.unsqueeze(1).expand(-1, x.size(1), -1)], dim=-1)
        h0 = self.decoder_lat(z).unsqueeze(0).repeat(self.L, 1, 1)
        out, _ = self.decoder_rnn(rnn.pack_padded_sequence(inp, lens, batch_first=True), h0)
        out, _ = rnn.pad_packed_sequence(out, batch_first=True)
        y = self.decoder_fc(out)
        recon = F.cross_entropy(y[:, :-1].reshape(-1, y.size(-1)), x[:, 1:].reshape(-1), ignore_index=O.PAD)
        return kl, recon

    def train_step(self, seqs, kl_w, opt, eps=None, clip=50.0):
        kl, recon = self.losses(seqs, eps)
        opt.zero_grad()
        (kl_w * kl + recon).backward()
        gn = nn.utils.clip_grad_norm_(self.parameters(), clip)
        opt.step()
        return float(kl.detach()), float(recon.detach()), float(gn)

    @torch.no_grad()
    def sample(self, z, max_len=100, temp=1.0):
        """The sampling loop with torch.multinomial (no early exit, eos bookkeeping as the reference)."""
        n = z.shape[0]
        h = self.decoder_lat(z).unsqueeze(0).repeat(self.L, 1, 1)
        w = torch.full((n,), O.BOS, dtype=torch.long)
        x = torch.full((n, max_len), O.PAD, dtype=torch.long)
        x[:, 0] = O.BOS
        end = torch.full((n,), max_len, dtype=torch.long)
        done = torch.zeros(n, dtype=torch.bool)
        for i in range(1, max_len):
            o, h = self.decoder_rnn(torch.cat([self.x_emb(w), z], dim=-1).unsqueeze(1), h)
            w = torch.multinomial(F.softmax(self.decoder_fc(o.squeeze(1)) / temp, dim=-1), 1)[:, 0]
            x[~done, i] = w[~done]
            new = ~done & (w == O.EOS)
            end[new] = i + 1
            done |= new
        return x, end


def to_tensors(seqs):
    return [torch.as_tensor(np.asarray(s), dtype=torch.long) for s in seqs]
