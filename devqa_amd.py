"""Import alias: the product package lives in ``de-vqa_amd/`` (a name Python
cannot import directly); ``import devqa_amd`` loads it from there."""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "de-vqa_amd")
_spec = importlib.util.spec_from_file_location(
    "devqa_amd", os.path.join(_root, "__init__.py"), submodule_search_locations=[_root])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["devqa_amd"] = _mod
_spec.loader.exec_module(_mod)
