from polygon_amd.vae.vae_model import VAE, TokenBatch, get_vocabulary  # noqa: F401
from polygon_amd.vae.vae_trainer import VAETrainer, FusedClipAdam  # noqa: F401
from polygon_amd.vae.vae_misc import KLAnnealer, CosineAnnealingLRWithRestart  # noqa: F401
from polygon_amd.vae.vae_generator import SmilesVaeMoleculeGenerator, OptResult  # noqa: F401
