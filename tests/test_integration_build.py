"""CPU-side check of the drop-in sources under integration/: they compile against the reference's OWN headers
(/root/reference: open_karto's Mapper.h, lesson4's hector_mapping headers), link against liblslam_gpu.so, and the
link-time substitution really happened -- the one strong definition of karto::ScanMatcher::MatchScan in
oracle/_ref_gpu/libkarto_ref_gpu.so is the one from integration/karto_scan_matcher_gpu.cpp (it calls the C ABI), while
the reference's own definition is still the one inside oracle/_ref/libkarto_ref.so.  No GPU needed; skipped where the
reference is absent (the GPU box: the prebuilt libraries travel, tests/test_ref_drives_gpu.py uses them there)."""
import os
import pathlib
import subprocess

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent
REF = pathlib.Path("/root/reference/lesson6/lib/open_karto/src/Mapper.cpp")
SYM = "_ZN5karto11ScanMatcher9MatchScanEPNS_18LocalizedRangeScanERKSt6vectorIS2_SaIS2_EERNS_5Pose2ERNS_7Matrix3Ebb"


@pytest.fixture(scope="module")
def built():
    if not REF.exists():
        pytest.skip("/root/reference absent")
    from lslam_amd import build

    build.build_library()
    subprocess.run(["make", "-s", "-C", str(ROOT / "oracle"), "ref", "ref_gpu"], check=True)
    return ROOT / "oracle"


def _nm(path, *flags):
    return subprocess.run(["nm", *flags, str(path)], check=True, capture_output=True, text=True).stdout


def test_link_time_substitution_of_match_scan(built):
    weak_obj = _nm(built / "_ref_gpu" / "Mapper_weak.o")
    ours = _nm(built / "_ref_gpu" / "karto_scan_matcher_gpu.o")
    assert any(l.endswith(" W " + SYM) or f" W {SYM}" in l for l in weak_obj.splitlines())  # the reference's definition, weakened
    assert f" T {SYM}" in ours                                                                 # the strong one
    assert " U lslam_matcher_match_scan" in ours                                               # ... which goes through the C ABI
    lib = _nm(built / "_ref_gpu" / "libkarto_ref_gpu.so", "-D")
    assert f" T {SYM}" in lib and " U lslam_matcher_match_scan" in lib
    # Mapper::Process and the graph are the reference's compiled code in both libraries
    for name in ("_ZN5karto6Mapper7ProcessEPNS_18LocalizedRangeScanE", "_ZN5karto11MapperGraph12TryCloseLoopEPNS_18LocalizedRangeScanERKNS_4NameE"):
        assert name in lib and name in _nm(built / "_ref" / "libkarto_ref.so", "-D")
    assert "lslam_matcher_match_scan" not in _nm(built / "_ref" / "libkarto_ref.so", "-D")      # the pure reference never sees the GPU
    assert " U lslam_occgrid_create_from_scans" in lib  # seam B2: integration/karto_occupancy_grid_gpu.hpp behind the driver


def test_hector_map_rep_is_the_interface(built):
    lib = _nm(built / "_ref_gpu" / "libhector_ref_gpu.so", "-D", "-C")
    assert "lslam::HectorMapRepGpu" in lib and "lslam_map_match_data" in lib and "lslam_map_update_by_scan" in lib
    assert "typeinfo for hectorslam::MapRepresentationInterface" in lib
    assert "lslam_map_create" not in _nm(built / "_ref" / "libhector_ref.so", "-D")


def test_destructor_is_substituted_too_and_the_twins_load(built):
    """~ScanMatcher (Mapper.cpp:119-124) is weakened in the copy of Mapper.o and defined by the integration file (it
    releases the device matcher: Mapper::Reset deletes and re-creates its matchers); the scan-cache entry points are what
    MatchScan goes through; and both GPU-driven twins resolve every symbol at load time (a C-linkage slip in the driver
    once surfaced only on the GPU box)."""
    import ctypes

    weak_obj = _nm(built / "_ref_gpu" / "Mapper_weak.o")
    ours = _nm(built / "_ref_gpu" / "karto_scan_matcher_gpu.o")
    for d in ("_ZN5karto11ScanMatcherD0Ev", "_ZN5karto11ScanMatcherD1Ev", "_ZN5karto11ScanMatcherD2Ev"):
        assert f" W {d}" in weak_obj and f" T {d}" in ours, d
    for sym in ("lslam_matcher_match_scan_cached", "lslam_scan_cache_create", "lslam_scan_cache_put", "lslam_scan_cache_forget",
                "lslam_matcher_destroy"):
        assert f" U {sym}" in ours, sym
    for name in ("libkarto_ref_gpu.so", "libhector_ref_gpu.so"):
        ctypes.CDLL(str(built / "_ref_gpu" / name), mode=os.RTLD_NOW)


def test_predicted_pose_equals_compute_weighted_mean(built):
    """integration/karto_scan_matcher_gpu.cpp prepares the scan Mapper::Process has just matched at the pose AddEdges will
    give it: its restatement of the single-mean case must equal the reference's own MapperGraph::ComputeWeightedMean BIT
    for BIT (the scan cache compares poses bitwise) -- and that pose is, as a rule, NOT the raw mean."""
    import ctypes

    import numpy as np

    from oracle import pyoracle as po

    L = ctypes.CDLL(str(built / "_ref_gpu" / "libkarto_ref_gpu.so"))
    L.kref_create.restype = ctypes.c_void_p
    L.kref_create.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.kref_weighted_mean_check.argtypes = [ctypes.c_void_p] * 5
    L.kref_destroy.argtypes = [ctypes.c_void_p]
    from lslam_amd import synth

    cfg, laser = po.default_cfg(), po.laser_struct(synth.Laser())
    h = L.kref_create(ctypes.byref(cfg), ctypes.byref(laser))
    assert h
    rng = np.random.default_rng(3)
    moved = 0
    for _ in range(2000):
        mean = np.array([rng.uniform(-50, 50), rng.uniform(-50, 50), rng.uniform(-3.14, 3.14)])
        a, b, c = rng.uniform(1e-4, 2e-3), rng.uniform(1e-4, 2e-3), rng.uniform(-5e-5, 5e-5)
        cov = np.array([[a, c, 0.0], [c, b, 0.0], [0.0, 0.0, rng.uniform(1e-6, 1e-4)]])  # what CorrelateScan fills (Mapper.cpp:535-692)
        ref, pred = np.zeros(3), np.zeros(3)
        L.kref_weighted_mean_check(h, mean.ctypes.data, cov.ctypes.data, ref.ctypes.data, pred.ctypes.data)
        assert ref.tobytes() == pred.tobytes()
        assert np.abs(ref - mean).max() < 1e-9
        moved += ref.tobytes() != mean.tobytes()
    assert moved > 200  # the weighted mean of ONE mean is that mean only up to rounding: why the raw mean is not the guess
    L.kref_destroy(h)
