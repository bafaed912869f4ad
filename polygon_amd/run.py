"""``polygon train`` / ``polygon generate`` entry points over the MI355X hot path (reference: polygon/run.py).

Only the two sub-commands on the VAE path are mirrored; flags and defaults are the
reference's (run.py:108-407).  Extra flag: ``--compute_dtype {fp32,bf16}``.  ``generate`` needs the reference's host
scoring stack (RDKit + its scoring modules), which stays on the CPU unchanged and is imported from the installed
``polygon`` package.  Under ``torchrun`` (WORLD_SIZE > 1) training is data parallel over RCCL.
"""
import argparse
import logging
import os
import sys
import time

import torch

from polygon_amd import __version__
from polygon_amd.parallel import init_distributed
from polygon_amd.utils.utils import load_model, set_random_seed, torch_device
from polygon_amd.vae import VAE, SmilesVaeMoleculeGenerator, VAETrainer


def _str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ('yes', 'true', 't', 'y', '1'):
        return True
    if v.lower() in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def _global_arguments(p):
    g = p.add_argument_group("Optional Runtime Arguments")
    g.add_argument("--version", action="version", version=__version__)
    g.add_argument("--verbose", action="store_true", default=False)
    g.add_argument("--quiet", action="store_true", default=False)
    g.add_argument("--debug", action="store_true", default=False)
    g.add_argument("--compute_dtype", default="fp32", choices=["fp32", "bf16"],
                   help="arithmetic of the HIP kernels (master weights stay fp32)")


def _trainer_arguments(g, kl_w_start):
    g.add_argument('--n_batch', type=int, default=512)
    g.add_argument('--clip_grad', type=int, default=50)
    g.add_argument('--kl_start', type=int, default=0)
    g.add_argument('--kl_w_start', type=float, default=kl_w_start)
    g.add_argument('--kl_w_end', type=float, default=1.0)
    g.add_argument('--lr_start', type=float, default=3 * 1e-4)
    g.add_argument('--lr_n_period', type=int, default=10)
    g.add_argument('--lr_n_restarts', type=int, default=6)
    g.add_argument('--lr_n_mult', type=int, default=1)
    g.add_argument('--lr_end', type=float, default=3 * 1e-4)
    g.add_argument('--n_last', type=int, default=1000)
    g.add_argument('--n_jobs', type=int, default=1)
    g.add_argument('--n_workers', type=int, default=0)


def train_parser(sub):
    p = sub.add_parser("train", help="Train a VAE SMILES model")
    p.add_argument('--train_data', type=str, required=True)
    p.add_argument("--model_def", type=str, default='vae_model')
    p.add_argument('--validation_data', type=str, default=None)
    p.add_argument('--model_save', type=str, default="model_{}.pt".format(int(time.time())))
    p.add_argument('--save_frequency', type=int, default=20)
    p.add_argument('--log_file', type=str, required=False)
    p.add_argument('--vocab_save', type=str)
    p.add_argument('--vocab_load', type=str)
    t = p.add_argument_group("Optional Training Parameters")
    t.add_argument("--n_epoch", type=int, default=50)
    t.add_argument("--device", type=torch_device, default="cuda:0")
    t.add_argument('--seed', type=int, default=0)      # parsed and never applied, as in the reference (run.py:305-308)
    _trainer_arguments(t, kl_w_start=1.0)
    m = p.add_argument_group('Model Arguments')
    m.add_argument('--q_cell', type=str, default='gru', choices=['gru'])
    m.add_argument('--q_bidir', default=False, action='store_true')
    m.add_argument('--q_d_h', type=int, default=256)
    m.add_argument('--q_n_layers', type=int, default=1)
    m.add_argument('--q_dropout', type=float, default=0.5)
    m.add_argument('--d_cell', type=str, default='gru', choices=['gru'])
    m.add_argument('--d_n_layers', type=int, default=3)
    m.add_argument('--d_dropout', type=float, default=0.2)
    m.add_argument('--d_z', type=int, default=128)
    m.add_argument('--d_d_h', type=int, default=512)
    m.add_argument('--m_dropout', type=float, default=0.2)
    m.add_argument('--n_mid_layers', type=int, default=1)
    m.add_argument('--batchnorm_conv', type=_str2bool, default=True)
    m.add_argument('--batchnorm_mid', type=_str2bool, default=True)
    m.add_argument('--lambda_scale', type=float, default=1)
    m.add_argument('--freeze_embeddings', default=False, action='store_true')
    _global_arguments(p)
    return p


def generate_parser(sub):
    p = sub.add_parser("generate", help="Generate SMILES by reinforcment learning")
    p.add_argument('--model_path', default=None, required=True)
    s = p.add_argument_group("Files Required for Scoring")
    s.add_argument('--encoding', default=None)
    s.add_argument('--cell_line_model', default=None)
    s.add_argument('--fscores', default=None)
    s.add_argument('--opti', default='gauss')
    s.add_argument("--scoring_definition", default=None)
    p.add_argument("--model_def", type=str, default='vae_model')
    p.add_argument('--outF', default='./')
    p.add_argument('--starting_population', default=None)
    r = p.add_argument_group("Optional Runtime Arguments")
    r.add_argument('--max_len', default=100, type=int)
    r.add_argument('--seed', default=None, type=int)
    r.add_argument('--keep_top', default=512, type=int)
    r.add_argument('--n_epochs', default=20, type=int)
    r.add_argument('--mols_to_sample', default=1024, type=int)
    r.add_argument('--optimize_batch_size', default=256, type=int)
    r.add_argument('--optimize_n_epochs', default=2, type=int)
    r.add_argument('--pretrain_n_epochs', default=2, type=int)
    r.add_argument('--benchmark_num_samples', default=4096, type=int)
    r.add_argument('--benchmark_trajectory', action='store_true', default=False)
    r.add_argument("--device", type=torch_device, default="cuda:0")
    r.add_argument('--save_frequency', type=int, default=5)
    r.add_argument('--save_payloads', action='store_true', default=False)
    r.add_argument('--save_individual_scores', action='store_true', default=False)
    r.add_argument('--batched_scoring', action='store_true', default=False,
                   help='(polygon_amd) LatentDistance raw scores from one batched encoder pass per generation and '
                        'score_list over --n_jobs host threads; same scores as the one-molecule-at-a-time loop')
    _trainer_arguments(p.add_argument_group("Optional Training Parameters"), kl_w_start=0)
    _global_arguments(p)
    return p


def get_parser():
    parser = argparse.ArgumentParser(prog="polygon", description="POLYGON SMILES VAE on MI355X")
    sub = parser.add_subparsers(dest="command")
    train_parser(sub)
    generate_parser(sub)
    return parser


def _read_lines(path):
    with open(path, "r") as handle:
        return [line.rstrip() for line in handle.readlines()]


def train_main(args):
    """run.py:576-620."""
    rank, world, local = init_distributed()
    device = torch.device(args.device if world == 1 else "cuda:%d" % local)
    if device.type.startswith("cuda"):
        torch.cuda.set_device(device.index or 0)
    train_data = _read_lines(args.train_data)
    validation_data = _read_lines(args.validation_data) if args.validation_data else None
    if args.vocab_load:
        logging.warning("Current ignoring custom vocabs.")
    kwargs = dict(vars(args))
    kwargs['device'] = str(device)
    model = VAE(**kwargs)
    model.to(device)
    trainer = VAETrainer(model, **kwargs)
    trainer.fit(train_data, val_data=validation_data)
    logging.info("Training Complete")


def generate_main(args):
    """run.py:622-718; scoring comes from the reference's own host modules."""
    rank, world, local = init_distributed()
    device = torch.device(args.device if world == 1 else "cuda:%d" % local)
    args.device = str(device)
    if device.type.startswith("cuda"):
        torch.cuda.set_device(device.index or 0)
    if args.seed:                                     # seed 0 means "no seed", as in the reference (run.py:636)
        set_random_seed(args.seed, args.device)
    try:
        from polygon.utils.utils import build_scoring_function, load_smiles_from_file
    except ImportError as e:
        raise ImportError("`generate` scores molecules with the reference's host stack (polygon.utils + RDKit), which "
                          "stays on the CPU unchanged; install the reference package next to polygon_amd") from e
    scorers = None
    # latent_distance rows make the reference build a VAE "on the cpu" and call encode() one molecule at a time
    # (utils.py:285-300): HostScoringVAE keeps that model on the MI355X and hands CPU tensors back to the host scorer
    from polygon_amd.vae.scoring import HostScoringVAE
    HostScoringVAE.compute_device = str(device)
    built = build_scoring_function(args.scoring_definition, args.encoding, args.cell_line_model, args.fscores,
                                   opti=args.opti, return_individual=bool(args.save_individual_scores),
                                   vae_model=HostScoringVAE)
    if args.save_individual_scores:
        scorers, scoring_function = built
    else:
        scoring_function = built
    os.makedirs(args.outF, exist_ok=True)
    model = load_model(VAE, args.model_path, args.device, model_params=dict(compute_dtype=args.compute_dtype))
    if args.starting_population is None:
        starting_population = []
    elif os.path.exists(args.starting_population):
        starting_population = load_smiles_from_file(args.starting_population)
    else:
        try:
            starting_population = model.sample(int(args.starting_population))
        except ValueError:
            logging.error(f'Did not understand starting population: {args.starting_population}')
            raise ValueError()
    generator = SmilesVaeMoleculeGenerator(model=model, max_len=args.max_len, device=device, out_dir=args.outF,
                                           lr=args.lr_start, n_jobs=args.n_jobs,
                                           batched_scoring=bool(getattr(args, "batched_scoring", False)))
    molecules = generator.optimise(objective=scoring_function, start_population=starting_population,
                                   n_epochs=args.n_epochs, mols_to_sample=args.mols_to_sample, keep_top=args.keep_top,
                                   optimize_batch_size=args.optimize_batch_size,
                                   optimize_n_epochs=args.optimize_n_epochs, pretrain_n_epochs=args.pretrain_n_epochs,
                                   ind_scorers=scorers, save_payloads=args.save_payloads,
                                   save_frequency=args.save_frequency)
    if rank == 0:
        with open(os.path.join(args.outF, "GDM_final_molecules.txt"), 'w') as handle:
            handle.writelines("{}\n".format(m.smiles) for m in molecules)
    logging.getLogger().info("Molecule Generation Complete")


def main(argv=None):
    parser = get_parser()
    args = parser.parse_args(argv)
    if args.command is None:
        parser.print_help()
        sys.exit(0)
    level = logging.WARNING
    if args.quiet:
        level = logging.ERROR
    if args.verbose:
        level = logging.INFO
    if args.debug:
        level = logging.DEBUG
    logging.basicConfig(format='%(asctime)s [%(levelname)-8s] %(message)s', level=level)
    if args.command == "train":
        train_main(args)
    elif args.command == "generate":
        generate_main(args)
    else:
        logging.error("Did not recognize command.")


if __name__ == "__main__":
    main()
