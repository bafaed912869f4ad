"""Generation loop of ``polygon generate`` (reference: polygon/vae/vae_generator.py:21-270) over the HIP sampler/trainer.

Per generation: sample ``mols_to_sample`` molecules from the prior (ONE sharded kernel sequence, 99 decode steps) ->
canonicalise on the host (RDKit, unchanged) -> drop what was seen -> score on the host (unchanged) -> keep the
``keep_top`` best ever -> shuffle, 75/25 split -> fine-tune the VAE on them with a fresh Adam (``VAETrainer.fit``).
"--opti gauss" only selects the score-modifier class inside the host scoring function (SURVEY.md headline facts).
The canonicaliser is injectable so that the loop can be exercised without RDKit.
"""
import logging
import os
import time
from functools import total_ordering
from typing import List, Set

import numpy as np

from polygon_amd.parallel import world_info
from polygon_amd.utils.utils import save_model
from polygon_amd.vae.vae_trainer import VAETrainer

logger = logging.getLogger(__name__)
logger.addHandler(logging.NullHandler())


@total_ordering
class OptResult:
    """(score, smiles) ordering, vae_generator.py:21-29."""

    def __init__(self, smiles: str, score: float) -> None:
        self.smiles = smiles
        self.score = score

    def _key(self):
        return (self.score, self.smiles)

    def __eq__(self, other):
        return self._key() == other._key()

    def __lt__(self, other):
        return self._key() < other._key()


def _default_canonicalizer(smiles, include_stereocenters=True):
    from polygon_amd.utils.utils import canonicalize_list
    return canonicalize_list(smiles, include_stereocenters=include_stereocenters)


class SmilesVaeMoleculeGenerator:
    def __init__(self, model, max_len: int, device, out_dir: str, lr=0.0003, n_jobs=1, model_save=True,
                 canonicalizer=None, batched_scoring=False) -> None:
        self.device = device
        self.model = model
        self.max_len = max_len
        self.output_dir = out_dir
        self.n_jobs = n_jobs
        self.model_save = model_save
        self.lr = lr
        self.canonicalizer = canonicalizer or _default_canonicalizer
        # opt-in: LatentDistance raw scores from one batched encoder pass, score_list spread over
        # n_jobs host threads and over the ranks of a data-parallel job (polygon_amd/vae/scoring.py); same scores
        self.batched_scoring = batched_scoring
        self.timings = []      # (sample_s, score_s, finetune_s) per generation, as the reference logs them (:190-193)
        # the reference's inner trainer: constant KL weight 1, constant lr, no checkpointing (vae_generator.py:56-63)
        self.trainer = VAETrainer(model=self.model, model_save=None, device=self.device, log_dir=self.output_dir,
                                  kl_w_start=1, kl_w_end=1, lr_start=self.lr, lr_end=self.lr)

    def sample(self, n_batch) -> List[str]:
        """vae_generator.py:72-80.  Under torch.distributed the rows are decoded by all ranks (VAE.sample_sharded: no
        collective but one final gather) and every rank gets the whole list."""
        if world_info()[1] > 1:
            return self.model.sample_sharded(n_batch, max_len=self.max_len)
        return self.model.sample(n_batch, max_len=self.max_len)

    def optimise(self, objective, start_population, keep_top, n_epochs, mols_to_sample, optimize_n_epochs,
                 optimize_batch_size, pretrain_n_epochs, save_frequency=10, ind_scorers=None,
                 save_payloads=False) -> List[OptResult]:
        rank, world = world_info()
        if self.batched_scoring or world > 1 or (self.n_jobs or 1) > 1:
            from polygon_amd.vae.scoring import BatchedObjective
            objective = BatchedObjective(objective, n_jobs=self.n_jobs)
        self.pretrain_on_initial_population(objective, start_population, pretrain_epochs=pretrain_n_epochs)
        objective.score_list(start_population)          # the reference scores it and then drops the result (:101-115)
        results: List[OptResult] = []
        seen: Set[str] = set()
        for epoch in range(1, 1 + n_epochs):
            t0 = time.time()
            samples = self.sample(mols_to_sample)
            t1 = time.time()
            canonical = set(self.canonicalizer(samples, include_stereocenters=True))
            payload = sorted(canonical.difference(seen))   # sorted: reproducible between runs
            seen.update(canonical)
            raw_scores = None
            if ind_scorers:
                from polygon.utils.utils import get_raw_scores   # host scoring of the reference, unchanged
                raw_scores = get_raw_scores(payload, ind_scorers, aggregate_scoring_function=objective)
                scores = raw_scores['Aggregate'].tolist()
            else:
                scores = objective.score_list(payload)
            new = sorted((OptResult(smiles=s, score=v) for s, v in zip(payload, scores)), reverse=True)
            t2 = time.time()
            results.extend(new[0:keep_top])
            results.sort(reverse=True)
            subset = [r.smiles for r in results][0:keep_top]
            np.random.shuffle(subset)
            if world > 1:        # one shuffle for the whole job: the ranks fine-tune on the same global batches
                import torch.distributed as dist
                box = [subset]
                dist.broadcast_object_list(box, 0)
                subset = box[0]
            cut = int(3 * len(subset) / 4)
            train_set, test_set = subset[0:cut], subset[cut:]
            opt_batch_size = min(len(train_set), optimize_batch_size)
            if optimize_n_epochs > 0:
                self.trainer.fit(train_set, test_set, n_epoch=optimize_n_epochs, batch_size=opt_batch_size,
                                 save_frequency=None)
            t3 = time.time()
            self.timings.append((t1 - t0, t2 - t1, t3 - t2))
            logger.info(f'Generation {epoch} --- timings: sample: {(t1 - t0):.3f} s, score: {(t2 - t1):.3f} s, '
                        f'finetune: {(t3 - t2):.3f} s')
            logger.info('Top 4:\n' + '\n'.join(f'\t{r.score:.3f}: {r.smiles}' for r in results[:4]))
            if rank != 0:
                continue         # files are written once, by rank 0
            if (self.model_save is not None) and (epoch % save_frequency == 0):
                save_model(self.trainer.model, os.path.join(self.output_dir, f'GDM_{epoch}.pt'))
            if save_payloads:
                if ind_scorers:
                    raw_scores.to_csv(os.path.join(self.output_dir, f'GDM_{epoch}_payload.csv'), index=False,
                                      header=True, sep=',')
                else:
                    with open(os.path.join(self.output_dir, f'GDM_{epoch}_payload.txt'), 'w') as handle:
                        handle.writelines(f'{s}\n' for s in payload)
            with open(os.path.join(self.output_dir, 'GDM_ongoing_top_scoring_molecules.txt'), 'w') as handle:
                handle.writelines(f'{r.smiles}\n' for r in results[0:keep_top])
        return sorted(results, reverse=True)

    def pretrain_on_initial_population(self, scoring_function, start_population, pretrain_epochs) -> List[OptResult]:
        """vae_generator.py:232-270: fine-tune on the (canonicalised) start population; no-op when it is empty."""
        seed: List[OptResult] = []
        n_start = len(start_population)
        training = self.canonicalizer(start_population, include_stereocenters=True)
        if n_start == 0:
            return seed
        logger.info("Pretraining on Starting Population")
        self.trainer.fit(training, batch_size=min(int(len(training)), 32), n_epoch=pretrain_epochs)
        logger.info("Finished Pretraining")
        return seed
