"""Import shim: the product package lives in `vk-raytracing-engine_amd/` (a directory name
Python cannot import directly); `import vkrt_amd` loads that package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vk-raytracing-engine_amd")
_spec = importlib.util.spec_from_file_location(
    "vkrt_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vkrt_amd"] = _mod
_spec.loader.exec_module(_mod)
