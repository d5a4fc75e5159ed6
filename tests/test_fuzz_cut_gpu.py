"""A 30-second cut of tools/fuzz_round5.py under -m gpu, so that the randomised sweeps the driver never
ran -- random windowed log-odds batches vs sequential updateByScan, random closed-loop trajectories with look-ahead vs scan by
scan, every coarse / fine numerator of random batches vs GetResponse (Mapper.cpp:819-856), loop-closure-size lattices -- are part
of the suite: 6 cases of each kind from a fixed seed, and the numerator sweep once more through the step kernel."""
import os
import pathlib
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def _fuzz(kind, n, seed, env_extra=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, str(ROOT / "tools" / "fuzz_round5.py"), kind, str(n), str(seed)], env=env, capture_output=True,
                       text=True, timeout=600)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0, tail
    assert "0 case(s) differ" in p.stdout, tail
    assert p.stdout.count("seed") >= n, tail


@pytest.mark.parametrize("kind,n", [("windows", 12), ("lookahead", 6), ("sums", 6), ("dense", 4)])
def test_fuzz_cut(kind, n):
    _fuzz(kind, n, 6000)


@pytest.mark.parametrize("waves", [3, 4])
def test_fuzz_sums_through_the_step_kernel(waves):
    _fuzz("sums", 4, 6100, {"LSLAM_FUZZ_STEP_KERNEL": str(waves)})
