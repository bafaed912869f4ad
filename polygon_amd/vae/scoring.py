"""Host glue between the reference's scoring stack (polygon/utils/*, RDKit / sklearn, unchanged, out of scope) and the
MI355X VAE.  Nothing here scores a molecule: it only (1) lets the reference's ``build_scoring_function`` load the VAE its
``latent_distance`` rows need, (2) answers the LatentDistance scorers' one-molecule-at-a-time ``raw_score`` calls from one
batched encoder pass (same numbers), (3) spreads ``score_list`` over host threads and over the
ranks of a data-parallel job (the reference creates a joblib pool and never uses it: vae_generator.py:54,66-69,141-144).

Reference call sites: polygon/utils/utils.py:285-300 (``load_model(vae_model, row.model, "cpu")`` + ``LatentDistance``),
polygon/utils/custom_scoring_fcn.py:149-183 (``raw_score``), polygon/vae/vae_generator.py:145-149 (``score_list``).
"""
import numpy as np
import torch

from polygon_amd.parallel import world_info
from polygon_amd.vae.latent_distance import latent_distance_scores
from polygon_amd.vae.vae_model import VAE


class HostScoringVAE(VAE):
    """The class to hand to ``build_scoring_function(..., vae_model=HostScoringVAE)``.

    The reference builds the scorer's VAE with ``model_class().to("cpu")`` and then does host tensor arithmetic on what
    ``encode`` returns (``torch.norm(z_targets - z)``, ``.numpy()``).  polygon_amd has no CPU arithmetic path, so this
    subclass keeps its parameters on the MI355X whatever device it is "moved" to and hands ``encode``'s result back as a
    CPU tensor: the scorer's host code works unchanged, the encoder runs in the HIP kernels."""

    compute_device = "cuda"

    def to(self, *args, **kwargs):
        dev = kwargs.get("device", args[0] if args else None)
        if isinstance(dev, (str, torch.device)) and torch.device(dev).type == "cpu":
            return super().to(self.compute_device)
        return super().to(*args, **kwargs)

    def encode(self, x):
        return super().encode(x).detach().cpu()


def _is_latent_scorer(s):
    return all(hasattr(s, a) for a in ("z_targets", "n_top", "agg", "model", "raw_score"))


class BatchedObjective:
    """Wraps the aggregate objective returned by ``build_scoring_function``: before a ``score_list`` call every
    LatentDistance-like scorer (duck-typed: ``model``, ``z_targets``, ``n_top``, ``agg``) has the raw scores of the WHOLE
    list computed in one batched encoder pass, and its ``raw_score`` answers from that table (falling back to the scorer's
    own code for a string it has not seen).  ``n_jobs`` > 1 scores chunks of the list on host threads - safe because
    the threads then never reach the GPU model - and under torch.distributed each rank scores its share of the list and
    the scores are all-gathered, so that the RDKit / sklearn work of a generation is spread over the ranks' host cores.
    The scores are the reference's own (same scorers, same modifiers, same aggregation)."""

    def __init__(self, objective, n_jobs=1, batch_size=1024, distribute=True):
        self.objective = objective
        self.n_jobs = max(1, int(n_jobs or 1))
        self.batch_size = batch_size
        self.distribute = distribute
        self.latent = [s for s in getattr(objective, "scoring_functions", []) if _is_latent_scorer(s)]
        self._tables = []
        for s in self.latent:
            table = {}
            self._tables.append(table)
            s.raw_score = self._cached(s, table, s.raw_score)

    @staticmethod
    def _cached(scorer, table, original):
        def raw_score(smiles):
            v = table.get(smiles)
            return original(smiles) if v is None else v
        return raw_score

    def prime(self, smiles_list):
        for s, table in zip(self.latent, self._tables):
            table.clear()
            if len(smiles_list) == 0:
                continue
            raw = latent_distance_scores(s.model, list(smiles_list), s.z_targets, n_top=s.n_top, agg=s.agg,
                                         batch_size=self.batch_size)
            table.update(zip(smiles_list, raw))

    def _score_chunked(self, smiles_list):
        if self.n_jobs == 1 or len(smiles_list) < 2 * self.n_jobs:
            return list(self.objective.score_list(smiles_list))
        from joblib import Parallel, delayed
        chunks = [smiles_list[i::self.n_jobs] for i in range(self.n_jobs)]
        parts = Parallel(n_jobs=self.n_jobs, prefer="threads")(delayed(self.objective.score_list)(c) for c in chunks)
        out = [None] * len(smiles_list)
        for i, p in enumerate(parts):
            out[i::self.n_jobs] = list(p)
        return out

    def score_list(self, smiles_list):
        smiles_list = list(smiles_list)
        rank, world = world_info()
        if world == 1 or not self.distribute:
            self.prime(smiles_list)
            return self._score_chunked(smiles_list)
        mine = smiles_list[rank::world]
        self.prime(mine)
        part = [float(v) for v in self._score_chunked(mine)]
        parts = [None] * world
        torch.distributed.all_gather_object(parts, part)
        out = [None] * len(smiles_list)
        for r, p in enumerate(parts):
            out[r::world] = p
        return out

    def score(self, smiles):
        return self.objective.score(smiles)

    def __getattr__(self, name):          # everything else (scoring_functions, weights, ...) is the wrapped objective's
        return getattr(self.objective, name)
