"""polygon_amd -- MI355X-native hot path of POLYGON's SMILES VAE (train + sample).

Drop-in Python API of ``polygon.vae`` (VAE, VAETrainer, SmilesVaeMoleculeGenerator) whose arithmetic runs in
hand-written gfx950 HIP kernels behind the C ABI of ``include/polygon_hip.h``.
"""
__version__ = "0.1.0"
