import os
import pathlib
import sys

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import lslam  # noqa: E402,F401  (registers the package as ``lslam_amd``)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu-marked tests run via gpurun)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle_lib():
    """Builds (if needed) and returns the oracle python bindings."""
    from oracle import pyoracle

    pyoracle.build("restate")
    if (pathlib.Path("/root/reference/lesson6/lib/open_karto").is_dir()):
        pyoracle.build("ref")
        pyoracle.build("ref_gpu")  # no-op unless liblslam_gpu.so has been built
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    from lslam_amd import api

    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="session")
def workload():
    """cfg-3 style workload: 24-scan running window + 12 queries (seeded)."""
    from lslam_amd import synth

    return synth.make_match_workload(n_base=24, n_query=12, seed=4)


@pytest.fixture(scope="session")
def workload_spread():
    """cfg-4 style workload: queries spread around the window, shared grid."""
    from lslam_amd import synth

    return synth.make_match_workload(n_base=24, n_query=40, seed=5, query_spread=2.0)
