"""MI355X-native contrastive video-texture hot path — a drop-in for the operator path of the
reference's contrastive_video_textures/ package."""
import os as _os

# HIP gives a process 4 hardware queues by default, and a stream that shares a queue with another waits for its neighbour's
# kernels.  The training step runs on three streams of its own (main, query encoder, the target encoder's fast pathway); with two more
# streams alive in the process — a collective library's, another engine's — it measured 3.5 % slower (profiles/r05/hw_queues.log:
# 425 -> 411 clips/s), and not with 8 queues.  Read by the HIP runtime when it initialises: set before anything touches the GPU
# (a value already in the environment wins).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib, ops  # noqa: E402,F401  (C-ABI loader first: it fails loudly when the HIP library is missing)
from . import audio_frontend, logger, resnet3d, slowfast, utils, vggish  # noqa: F401
from . import models, texture  # noqa: F401
from . import validate as _validate_mod, train as _train_mod, dataset, dist, classic  # noqa: F401
from .models import ContrastivePredictionTemporal, ModelBuilder3D, InfoNCECriterion  # noqa: F401
from .vggish import VGGish  # noqa: F401
from .dataset import AudioVideoSegments  # noqa: F401
from .validate import validate  # noqa: F401
from .train import train  # noqa: F401
