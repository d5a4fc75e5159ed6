"""`python bench.py --gpus N` really starts N ranks, shards SURVEY §8(e)'s [r*B/W,(r+1)*B/W) partition over them and
gathers byte-identical results.  On the 1-GPU test box the two ranks talk gloo and share GPU 0
(LSLAM_BENCH_BACKEND=gloo); on an 8-GPU node the same command line runs RCCL over xGMI."""
import json
import os
import pathlib
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = pathlib.Path(__file__).resolve().parent.parent


def run_bench(tmp_path, gpus, extra=(), backend=None, env_extra=None, may_fail=False, timeout=600):
    out = tmp_path / f"res_{gpus}.npy"
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LSLAM_BENCH_BACKEND", "LSLAM_BENCH_FORCE_DIST",
              "LSLAM_BENCH_SHARE_GPU"):
        env.pop(k, None)
    if backend:
        env["LSLAM_BENCH_BACKEND"] = backend
    env.update(env_extra or {})
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "1", "--no-cpu",
           "--no-diagnostics", "--batch", "1024", "--dump-results", str(out), *extra]
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    except subprocess.TimeoutExpired as e:
        if may_fail:
            return None, "timeout: " + str(e)[:200]
        raise
    if may_fail and p.returncode != 0:
        return None, p.stderr
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0 only
    return json.loads(lines[0]), np.load(out)


def test_two_ranks_equal_one_rank(tmp_path):
    one, r1 = run_bench(tmp_path, 1)
    two, r2 = run_bench(tmp_path, 2, backend="gloo")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "strong" and two["config"]["scans_per_gpu_per_step"] == 512
    assert one["results_ok"] == two["results_ok"] == 1024
    assert r1.shape == r2.shape == (1024, 112)
    assert r1.tobytes() == r2.tobytes()  # the gathered records of the sharded run are the single-rank ones
    assert two["gather_ms"] is not None


def test_eight_ranks_equal_one_rank(tmp_path):
    """The partition the driver's 8-GPU run uses -- 8 ranks x 128 of 1024 scans here, [r*B/8, (r+1)*B/8) -- on the one GPU
    of the test box (gloo, all ranks on GPU 0): gathered records byte-equal to the single-rank run, one time per rank."""
    one, r1 = run_bench(tmp_path, 1)
    eight, r8 = run_bench(tmp_path, 8, backend="gloo")
    assert eight["n_gpus"] == 8 and eight["config"]["scans_per_gpu_per_step"] == 128 and eight["results_ok"] == 1024
    assert r1.tobytes() == r8.tobytes()
    assert len(eight["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in eight["per_rank_ms_per_step"])


def test_broadcast_grid_equals_rebuild(tmp_path):
    """--broadcast-grid: rank 0 rasterises the window, shard.broadcast_grid replicates the 4 MB grid and every other
    rank installs it through lslam_matcher_set_grid_u8_dev (no AddScans there).  Gathered records byte-equal to the
    default path, where every rank rebuilds the grid itself."""
    one, r1 = run_bench(tmp_path, 1)
    two, r2 = run_bench(tmp_path, 2, extra=("--broadcast-grid",), backend="gloo")
    assert two["n_gpus"] == 2 and two["results_ok"] == 1024
    assert r1.tobytes() == r2.tobytes()
    assert len(two["per_rank_ms_per_step"]) == 2


def test_weak_scaling_mode_and_gpu_count_check(tmp_path):
    two, r2 = run_bench(tmp_path, 2, extra=("--scaling", "weak"), backend="gloo")
    assert two["scaling"] == "weak" and two["config"]["scans_per_step"] == 2048 and r2.shape == (2048, 112)
    # asking for more GPUs than the box has must fail loudly, not wrap local_rank around
    import torch

    n = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LSLAM_BENCH_BACKEND")}
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(n + 1), "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "visible" in (p.stderr + p.stdout)


def test_rccl_backend_on_the_one_gpu_box(tmp_path):
    """The torch `nccl` (= RCCL) branch of bench.py, executed here before an 8-GPU node executes it: a process group of ONE
    rank (LSLAM_BENCH_FORCE_DIST=1) runs init_process_group(backend="nccl", device_id=...), the barrier, the max-over-ranks
    all_reduce, the device-tensor all_gather of the records and the --broadcast-grid broadcast + set_grid_dev install on
    real RCCL.  Records byte-equal to the plain single-process run."""
    one, r1 = run_bench(tmp_path, 1)
    assert one["backend"] is None and one["rccl_ranks"] is None and len(one["devices"]) == 1
    nc, rn = run_bench(tmp_path, 1, extra=("--broadcast-grid",), backend="nccl", env_extra={"LSLAM_BENCH_FORCE_DIST": "1"})
    assert nc["backend"] == "nccl" and nc["rccl_ranks"] == 1 and nc["n_gpus"] == 1
    assert nc["gather_ms"] is not None and len(nc["per_rank_ms_per_step"]) == 1
    assert len(nc["devices"]) == 1 and "cuda:0" in nc["devices"][0]
    assert nc["pipelined_records_identical"] is True and nc["pipeline_depth"] == 1 and nc["pipelined_leg"]["depth"] == 2
    assert r1.tobytes() == rn.tobytes()


def test_rccl_two_ranks_sharing_the_gpu(tmp_path):
    """Two RCCL ranks on device 0 (LSLAM_BENCH_SHARE_GPU=1).  RCCL is entitled to refuse a communicator with a duplicate
    GPU; when it does the test is skipped with its message, when it does not the sharded records must equal the
    single-rank ones."""
    two, r2 = run_bench(tmp_path, 2, backend="nccl", env_extra={"LSLAM_BENCH_SHARE_GPU": "1"}, may_fail=True, timeout=180)
    if two is None:
        lines = [l for l in (r2 or "").splitlines() if any(w in l for w in ("NCCL", "RCCL", "nccl", "Duplicate", "duplicate", "Error", "timeout"))]
        pytest.skip("RCCL does not form a communicator of two ranks on one device here: " + (lines[0][:300] if lines else "no message"))
    one, r1 = run_bench(tmp_path, 1)
    assert two["backend"] == "nccl" and two["rccl_ranks"] == 2 and two["results_ok"] == 1024
    assert r1.tobytes() == r2.tobytes()


def test_pipelined_timed_region_equals_plain(tmp_path):
    """--pipeline-depth 1 / 2 / 4: the timed region's records are the same bytes."""
    base, r1 = run_bench(tmp_path, 1, extra=("--pipeline-depth", "1"))
    for d in (2, 4):
        j, r = run_bench(tmp_path, 1, extra=("--pipeline-depth", str(d), "--steps", "9"))
        assert j["pipeline_depth"] == d and j["pipelined_records_identical"] is True
        assert j["plain"]["ms_per_step"] > 0 and j["roofline"]["leg"].startswith("plain steps")
        assert r.tobytes() == r1.tobytes()


def test_n1_line_through_torchrun_equals_plain_invocation(tmp_path):
    """The driver's SCALE protocol starts every N -- N = 1 included -- through `python -m torch.distributed.run`; its BENCH
    protocol runs `python bench.py`.  The two N = 1 lines must describe the same measurement: same metric / config / leg,
    same records, `value` within the run-to-run noise of a 40-step region, and roofline + cpu_baseline present in both."""
    common = ("--steps", "40", "--warmup", "5", "--batch", "4096", "--no-secondary", "--no-diagnostics", "--sustained-s", "0",
              "--cpu-sample", "64", "--cpu-cores", "1")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LSLAM_BENCH_BACKEND", "LSLAM_BENCH_FORCE_DIST",
              "LSLAM_BENCH_SHARE_GPU"):
        env.pop(k, None)

    def run(cmd, tag):
        out = tmp_path / f"{tag}.npy"
        p = subprocess.run([*cmd, str(ROOT / "bench.py"), "--gpus", "1", *common, "--dump-results", str(out)], env=env,
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, p.stdout
        return json.loads(lines[0]), np.load(out)

    plain, rp = run([sys.executable], "plain")
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    tr, rt = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                  "--master-port", str(port)], "torchrun")
    assert rp.tobytes() == rt.tobytes()
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "scaling", "dtype", "pipeline_depth", "results_ok"):
        assert plain[k] == tr[k], k
    assert plain["config"] == tr["config"]
    assert tr["backend"] is None and plain["backend"] is None  # WORLD_SIZE 1: no process group either way (same barriers, same clock)
    for line in (plain, tr):
        assert line["roofline"]["kernel"] == "resp_rows_coarse" and line["roofline"]["valu_mix_frac"] is not None
        assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["max_pose_err_vs_gpu"] < 1e-9
    assert abs(tr["value"] / plain["value"] - 1.0) < 0.08, (tr["value"], plain["value"])
