"""MI355X-native anomaly-scoring hot path of KuldeepChoksi/video-anomaly-detection.

The directory name is fixed by the build contract and is not a Python identifier: import it with
`importlib.import_module("video-anomaly-detection_amd")`, or use the drop-in `models` / `utils`
packages at the repository root, which re-export the reference's import surface.
"""
from . import hip, synth                                              # noqa: F401
from .autoencoder import Autoencoder, ConvAutoencoder, Decoder, Encoder   # noqa: F401
from .video_autoencoder import (ConvLSTM, ConvLSTMCell, VideoAutoencoder,    # noqa: F401
                                VideoDecoder, VideoEncoder)
from . import losses                                                   # noqa: F401
from .losses import CombinedLoss, SSIMLoss                           # noqa: F401
from . import training                                                 # noqa: F401
from .training import ImageTrainer, VideoTrainer                                    # noqa: F401
from . import scoring                                                 # noqa: F401
