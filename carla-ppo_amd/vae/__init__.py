"""Drop-in `vae` package (reference vae/__init__.py:1-2 re-exports the model classes)."""
from .models import ConvVAE, MlpVAE, VAE, bce_loss, bce_loss_v2, mse_loss  # noqa: F401
