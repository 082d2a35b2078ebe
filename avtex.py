"""Import alias: the package directory is `audio-video-textures_amd` (not a Python identifier), so
`import avtex` loads it through importlib and aliases it and its submodules (`avtex.ops`, ...).

Submodules imported LATER under either name resolve to ONE module object (a meta-path finder maps `avtex.x` to
`audio-video-textures_amd.x`): without it `from avtex.fused_vggish import VGGishMFMA` after the package had imported
`.fused_vggish` itself would create a second copy of the module, and `isinstance` across the two copies fails."""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_PKG = "audio-video-textures_amd"
_ALIAS = "avtex"


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        mod = importlib.import_module(self.real)  # the one module object, under its real name
        self.spec = getattr(mod, "__spec__", None)
        return mod

    def exec_module(self, module):
        # already executed by the import above; importlib's module_from_spec has just overwritten the module's __spec__ (and
        # __loader__) with the alias's no-op ones — put the real ones back so that importlib.reload() and anything that relies on
        # __spec__.name == __name__ keep working
        if self.spec is not None:
            module.__spec__ = self.spec
            module.__loader__ = self.spec.loader


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if fullname == _ALIAS or not fullname.startswith(_ALIAS + "."):
            return None
        real = _PKG + fullname[len(_ALIAS):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_PKG)
for _name, _mod in list(sys.modules.items()):
    if _name.startswith(_PKG + "."):
        sys.modules[_ALIAS + _name[len(_PKG):]] = _mod
sys.modules[__name__] = _pkg
