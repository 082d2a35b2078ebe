"""MI355X-native contrastive video-texture hot path — a drop-in for the operator path of the
reference's contrastive_video_textures/ package."""
from . import _lib, ops  # noqa: F401  (C-ABI loader first: it fails loudly when the HIP library is missing)
from . import audio_frontend, logger, resnet3d, slowfast, utils, vggish  # noqa: F401
from . import models, texture  # noqa: F401
from . import validate as _validate_mod, train as _train_mod, dataset, dist, classic  # noqa: F401
from .models import ContrastivePredictionTemporal, ModelBuilder3D, InfoNCECriterion  # noqa: F401
from .vggish import VGGish  # noqa: F401
from .dataset import AudioVideoSegments  # noqa: F401
from .validate import validate  # noqa: F401
from .train import train  # noqa: F401
