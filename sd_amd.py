"""Import alias for the package directory ``stable-diffusion_amd/`` (a dash is not a legal
module name).  ``import sd_amd`` yields that package: ``sd_amd.pipeline.generate`` etc."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stable-diffusion_amd")
_spec = importlib.util.spec_from_file_location(
    "sd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["sd_amd"] = _pkg
_spec.loader.exec_module(_pkg)
