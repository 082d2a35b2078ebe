"""Import alias: the package directory is `audio-video-textures_amd` (not a Python
identifier), so `import avtex` loads it through importlib and aliases it."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("audio-video-textures_amd")
sys.modules[__name__] = _pkg
