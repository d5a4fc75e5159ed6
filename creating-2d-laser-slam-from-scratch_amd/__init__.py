"""MI355X-native 2-D laser SLAM front-end hot path (Karto correlative scan matcher + Hector
log-odds grid update) behind the C ABI of include/lslam_gpu.h.

The directory name follows the project naming rule and is not a valid Python identifier; import
it through the repo-root alias module::

    import lslam            # registers this package as ``lslam_amd``
    from lslam_amd import api, synth
"""
from . import build as build  # noqa: F401
from . import synth as synth  # noqa: F401
from . import api as api  # noqa: F401
from . import shard as shard  # noqa: F401

__all__ = ["api", "synth", "build"]
