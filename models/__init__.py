"""`models` as the reference's callers import it (demo_sample.py:10, trainer.py:10): re-exports var_amd.models."""
from var_amd.models import VAR, VQVAE, VectorQuantizer2, build_vae_var  # noqa: F401
