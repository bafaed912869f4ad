"""Synthetic GuacaMol-shaped SMILES strings: no dataset ships with the reference and there is no network,
so benchmarks and smoke tests draw lengths ~ round(N(47, 12^2)) clipped to [10, 98] and characters from a SMILES-like
categorical over the reference alphabet.  The arithmetic of the hot path is value-independent; only lengths matter."""
import numpy as np

from polygon_amd.utils.smiles_char_dict import SmilesCharDictionary

_FREQ = {'C': .30, 'c': .22, '(': .07, ')': .07, 'O': .06, 'N': .05, '1': .05, '=': .04, '2': .03, 'n': .03}


def synthetic_smiles(n, seed=20250310):
    rng = np.random.default_rng(seed)
    lens = np.clip(np.rint(rng.normal(47, 12, size=n)), 10, 98).astype(np.int64)
    chars = [c for c in sorted(SmilesCharDictionary().char_idx) if c not in (' ', '\n', 'Q')]
    rest = (1.0 - sum(_FREQ.values())) / (len(chars) - len(_FREQ))
    p = np.array([_FREQ.get(c, rest) for c in chars])
    p /= p.sum()
    arr = np.array(chars)[rng.choice(len(chars), size=int(lens.sum()), p=p)]
    out, o = [], 0
    for L in lens:
        out.append(''.join(arr[o:o + L]))
        o += L
    return out
