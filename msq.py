"""Import alias: the package directory is `microscopiq-llm-quantization_amd` (not a valid
Python identifier), so `import msq` loads it under a usable name."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("microscopiq-llm-quantization_amd")
sys.modules[__name__] = _pkg
for _k, _v in list(sys.modules.items()):
    if _k.startswith("microscopiq-llm-quantization_amd."):
        sys.modules["msq." + _k.split(".", 1)[1]] = _v
