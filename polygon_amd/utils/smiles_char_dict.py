"""Fixed SMILES alphabet of the reference (polygon/utils/smiles_char_dict.py:18-30).

Only ``idx_char.values()`` feeds the VAE vocabulary (vae_model.py:16-17); the multi-character
``encode``/``decode`` helpers are never called on the VAE path and are kept for API parity.
"""

_ALPHABET = (
    (' ', 0), ('Q', 1), ('\n', 2), ('Z', 3), ('F', 4), ('I', 5), ('H', 6), ('A', 7), ('c', 8), ('O', 9), ('N', 10),
    ('B', 11), ('P', 12), ('S', 13), ('Y', 14), ('X', 15), ('[', 16), ('n', 17), (']', 18), ('C', 19), ('#', 20),
    ('b', 21), ('%', 22), ('p', 23), (')', 24), ('(', 25), ('+', 26), ('-', 27), ('s', 28), ('o', 29), ('.', 30),
    ('1', 31), ('0', 32), ('3', 33), ('2', 34), ('5', 35), ('4', 36), ('7', 37), ('6', 38), ('9', 39), ('8', 40),
    ('=', 41), ('@', 42), ('R', 43), ('/', 44), ('\\', 45), ('E', 46),
)
_MULTI = (("Br", 'Y'), ("Cl", 'X'), ("Si", 'A'), ('Se', 'Z'), ('@@', 'R'), ('se', 'E'))


class SmilesCharDictionary(object):
    PAD = ' '
    BEGIN = 'Q'
    END = '\n'

    def __init__(self, max_len=120):
        self.char_idx = dict(_ALPHABET)
        self.idx_char = {i: c for c, i in _ALPHABET}
        self.encode_dict = dict(_MULTI)
        self.decode_dict = {tok: sym for sym, tok in _MULTI}

    def encode(self, smiles):
        for sym, tok in _MULTI:
            smiles = smiles.replace(sym, tok)
        return smiles

    def decode(self, smiles):
        for sym, tok in _MULTI:
            smiles = smiles.replace(tok, sym)
        return smiles

    def get_char_num(self):
        return len(self.idx_char)

    @property
    def begin_idx(self):
        return self.char_idx[self.BEGIN]

    @property
    def end_idx(self):
        return self.char_idx[self.END]

    @property
    def pad_idx(self):
        return self.char_idx[self.PAD]
