"""MI355X-native contrastive video-texture hot path (drop-in for the reference's
contrastive_video_textures/ operator path).  See DESIGN.md."""
from . import _lib, ops  # noqa: F401

__all__ = ["_lib", "ops"]
