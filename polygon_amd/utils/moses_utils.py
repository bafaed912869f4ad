"""Tokeniser / vocabulary of the reference, bit-exact (polygon/utils/moses_utils.py:11-101,184-202).

Host-side and tiny: ids 0..46 are the alphabet characters in code-point order, then bos=47, eos=48,
pad=49, unk=50.  Lookup is per CHARACTER (``Cl`` -> ``C``, unk).  ``encode_batch`` is the vectorised
form used by the collate path (one byte LUT pass over the whole batch instead of a Python loop per
character); it produces exactly ``string2ids(s, add_bos=True, add_eos=True)`` for every string.
"""
import random

import numpy as np
import torch


def set_torch_seed_to_all_gens(_):
    seed = torch.initial_seed() % (2 ** 32 - 1)
    random.seed(seed)
    np.random.seed(seed)


class SS:
    bos = '<bos>'
    eos = '<eos>'
    pad = '<pad>'
    unk = '<unk>'


class CharVocab:
    @classmethod
    def from_data(cls, data, *args, **kwargs):
        chars = set()
        for string in data:
            chars.update(string)
        return cls(chars, *args, **kwargs)

    def __init__(self, chars, ss=SS):
        specials = [ss.bos, ss.eos, ss.pad, ss.unk]
        if any(s in chars for s in specials):
            raise ValueError('SS in chars')
        syms = sorted(list(chars)) + specials
        self.ss = ss
        self.c2i = {c: i for i, c in enumerate(syms)}
        self.i2c = dict(enumerate(syms))
        self._lut = None

    def __len__(self):
        return len(self.c2i)

    @property
    def bos(self):
        return self.c2i[self.ss.bos]

    @property
    def eos(self):
        return self.c2i[self.ss.eos]

    @property
    def pad(self):
        return self.c2i[self.ss.pad]

    @property
    def unk(self):
        return self.c2i[self.ss.unk]

    def char2id(self, char):
        return self.c2i.get(char, self.unk)

    def id2char(self, id):
        return self.i2c.get(id, self.ss.unk)

    def string2ids(self, string, add_bos=False, add_eos=False):
        ids = [self.c2i.get(c, self.unk) for c in string]
        if add_bos:
            ids.insert(0, self.bos)
        if add_eos:
            ids.append(self.eos)
        return ids

    def ids2string(self, ids, rem_bos=True, rem_eos=True):
        if len(ids) == 0:
            return ''
        if rem_bos and ids[0] == self.bos:
            ids = ids[1:]
        if rem_eos and ids[-1] == self.eos:     # IndexError on a lone bos, exactly as the reference
            ids = ids[:-1]
        return ''.join(self.i2c.get(i, self.ss.unk) for i in ids)

    # ---- vectorised batch forms ------------------------------------------------
    def _byte_lut(self):
        if self._lut is None:
            lut = np.full(256, self.unk, dtype=np.int32)
            wide = False
            for c, i in self.c2i.items():
                if len(c) == 1:
                    if ord(c) < 256:
                        lut[ord(c)] = i
                    else:
                        wide = True
            self._lut = (lut, wide)
        return self._lut

    def encode_batch(self, strings):
        """-> (tokens int32 [B, Tmax] padded with pad, lengths int32 [B]); rows keep the given order."""
        lut, wide = self._byte_lut()
        B = len(strings)
        lens = np.fromiter((len(s) for s in strings), dtype=np.int32, count=B) + 2
        T = int(lens.max()) if B else 0
        out = np.full((B, T), self.pad, dtype=np.int32)
        if B == 0:
            return out, lens
        out[:, 0] = self.bos
        if wide:
            flat = np.fromiter((self.c2i.get(c, self.unk) for s in strings for c in s), dtype=np.int32)
        else:
            # latin-1 keeps one byte per code point < 256; anything wider becomes '?' which is then forced to unk
            raw = np.frombuffer(''.join(strings).encode('latin-1', 'replace'), dtype=np.uint8)
            flat = lut[raw]
            if any(not s.isascii() for s in strings):
                joined = ''.join(strings)
                bad = np.fromiter((ord(c) > 255 for c in joined), dtype=bool, count=len(joined))
                flat = np.where(bad, self.unk, flat)
        rows = np.repeat(np.arange(B), lens - 2)
        starts = np.cumsum(lens - 2) - (lens - 2)
        cols = np.arange(flat.shape[0]) - np.repeat(starts, lens - 2) + 1
        out[rows, cols] = flat
        out[np.arange(B), lens - 1] = self.eos
        return out, lens

    def decode_batch(self, tokens, end_pads):
        """Rows tokens[i, :end_pads[i]] -> strings with ids2string semantics (leading bos / trailing eos stripped)."""
        i2c = self.i2c
        unk = self.ss.unk
        bos, eos = self.bos, self.eos
        out = []
        tl = tokens.tolist()
        for row, e in zip(tl, end_pads.tolist()):
            ids = row[:e]
            if len(ids) == 0:
                out.append('')
                continue
            if ids[0] == bos:
                ids = ids[1:]
            if ids[-1] == eos:
                ids = ids[:-1]
            out.append(''.join([i2c.get(i, unk) for i in ids]))
        return out


class OneHotVocab(CharVocab):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.vectors = torch.eye(len(self.c2i))


class CircularBuffer:
    """Loss averaging buffer; ``mean`` is over the WHOLE zero-initialised array (moses_utils.py:184-202)."""

    def __init__(self, size):
        self.max_size = size
        self.data = np.zeros(self.max_size)
        self.size = 0
        self.pointer = -1

    def add(self, element):
        self.size = min(self.size + 1, self.max_size)
        self.pointer = (self.pointer + 1) % self.max_size
        self.data[self.pointer] = element
        return element

    def last(self):
        assert self.pointer != -1, "Can't get an element from an empty buffer!"
        return self.data[self.pointer]

    def mean(self):
        return self.data.mean()
