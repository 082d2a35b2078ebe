"""Import alias: the package directory is `audio-video-textures_amd` (not a Python identifier), so
`import avtex` loads it through importlib and aliases it and its submodules (`avtex.ops`, ...)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_PKG = "audio-video-textures_amd"
_pkg = importlib.import_module(_PKG)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_PKG + "."):
        sys.modules["avtex" + _name[len(_PKG):]] = _mod
sys.modules[__name__] = _pkg
