"""Import alias: the product package lives in the directory ``video-llamagen_amd/`` (not a valid
Python identifier), this module makes it importable as ``video_llamagen_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "video-llamagen_amd")
_spec = importlib.util.spec_from_file_location(
    "video_llamagen_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["video_llamagen_amd"] = _mod
_spec.loader.exec_module(_mod)
