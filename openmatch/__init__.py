"""`openmatch` — drop-in import name.  Every `openmatch.<sub>` module IS the corresponding
`openmatch_amd.<sub>` module (same object), so code written against thunlp/OpenMatch
(`from openmatch.modeling import DRModel`, `from openmatch.retriever import Retriever`, ...)
runs on the MI355X-native implementation unchanged."""
import importlib
import importlib.abc
import importlib.util
import sys

import openmatch_amd as _impl

__version__ = getattr(_impl, "__version__", "0")


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, module):
        self._module = module

    def create_module(self, spec):
        return self._module

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith("openmatch."):
            return None
        real = "openmatch_amd" + fullname[len("openmatch"):]
        try:
            module = importlib.import_module(real)
        except ModuleNotFoundError as e:
            if e.name == real:
                return None
            raise
        return importlib.util.spec_from_loader(fullname, _AliasLoader(module))


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())
