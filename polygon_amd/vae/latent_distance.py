"""Batched raw scores of the reference's LatentDistance scorer (polygon/utils/custom_scoring_fcn.py:149-183).

The reference encodes ONE molecule per ``raw_score`` call on the host; the default scoring definition uses that scorer
twice per sampled molecule (data/scoring_definition.csv:4-5), i.e. 16 384 single-row encoder passes per generation.
``latent_distance_scores`` runs the same arithmetic -- mu of the molecule, L2 distances to the target mus, the n_top
smallest, then mean / max / min -- for a whole list at once through the HIP encoder.  The score modifiers and the scoring
classes themselves stay in the reference's host modules, unchanged (out of scope)."""
import torch


@torch.no_grad()
def encode_targets(model, smiles_targets):
    """z_targets as LatentDistance.__init__ builds them (collate sorts the targets in place, then encode -> mu)."""
    return model.encode(model.get_collate_fn()(smiles_targets))


@torch.no_grad()
def latent_distance_scores(model, smiles_list, z_targets, n_top=None, agg="mean", batch_size=1024):
    """numpy float32 [len(smiles_list)]: what ``LatentDistance.raw_score(s)`` returns for every s, in input order."""
    if agg not in ("mean", "max", "min"):
        raise ValueError("agg must be mean, max or min (custom_scoring_fcn.py:173-178)")
    k = z_targets.shape[0] if n_top is None else int(n_top)
    mu = model.encode_smiles(list(smiles_list), batch_size=batch_size)          # [n, d_z]
    # exact differences as the reference's torch.norm(z_targets - z, dim=1): cdist's default switches to the matmul
    # formulation above 25 rows, which loses ~1e-3 absolute on near-identical vectors - the small distances this scorer
    # rewards and ranks by
    d = torch.cdist(mu, z_targets.to(mu.device), p=2, compute_mode="donot_use_mm_for_euclid_dist")   # [n, n_targets]
    d = d.sort(dim=1)[0][:, :k]
    r = d.mean(1) if agg == "mean" else d.max(1)[0] if agg == "max" else d.min(1)[0]
    return r.float().cpu().numpy()
