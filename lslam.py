"""Import alias: registers the package directory ``creating-2d-laser-slam-from-scratch_amd``
(not a valid identifier) as the importable package ``lslam_amd``."""
import importlib.util
import pathlib
import sys

_DIR = pathlib.Path(__file__).resolve().parent / "creating-2d-laser-slam-from-scratch_amd"

if "lslam_amd" not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        "lslam_amd", _DIR / "__init__.py", submodule_search_locations=[str(_DIR)]
    )
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules["lslam_amd"] = _mod
    _spec.loader.exec_module(_mod)

lslam_amd = sys.modules["lslam_amd"]
