"""Drop-in for the reference's `models` package (reference models/__init__.py:5-11):
`from models import ConvAutoencoder`, `from models.video_autoencoder import VideoAutoencoder`."""
from .autoencoder import Autoencoder, ConvAutoencoder, Decoder, Encoder

__all__ = ["ConvAutoencoder", "Encoder", "Decoder", "Autoencoder"]
