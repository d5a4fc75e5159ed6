"""ctypes host binding of liblslam_gpu.so (include/lslam_gpu.h).

Python is only the harness language of this repo (tests, bench); the classes mirror the
reference's operator interface for the hot path so the parity tests read like calls into the
reference:

* ``ScanMatcher``  <-> karto::ScanMatcher      (Mapper.h:1127-1279): Create / MatchScan /
  CorrelateScan-level batched search / GetCorrelationGrid
* ``OccGridMap``   <-> hectorslam::OccGridMapBase + MapRepMultiMap (H/map/OccGridMapBase.h,
  H/slam_main/MapRepMultiMap.h): updateByScan / updateByScanJustOnce / getGridMap

There is NO CPU fallback here: if the shared library or a GPU is missing, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import math
import weakref

import numpy as np

from . import build as _build

LSLAM_OK = 0
ERRORS = {
    -1: "LSLAM_ERR_INVALID_ARGUMENT",
    -2: "LSLAM_ERR_NO_DEVICE",
    -3: "LSLAM_ERR_INDEX_OUT_OF_RANGE",
    -4: "LSLAM_ERR_PROBABILITY_SEARCH",
    -5: "LSLAM_ERR_NO_BEST_POSE",
    -6: "LSLAM_ERR_HIP",
    -7: "LSLAM_ERR_SMEAR_DEVIATION",
    -8: "LSLAM_ERR_UNSUPPORTED",
}


class LslamError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERRORS.get(code, code)}: {msg}")
        self.code = code


class MatcherConfig(C.Structure):
    """lslam_matcher_config (ScanMatcher::Create args + the Mapper parameters the matcher reads)."""

    _fields_ = [
        ("search_size", C.c_double),
        ("resolution", C.c_double),
        ("smear_deviation", C.c_double),
        ("range_threshold", C.c_double),
        ("coarse_search_angle_offset", C.c_double),
        ("coarse_angle_resolution", C.c_double),
        ("fine_search_angle_offset", C.c_double),
        ("distance_variance_penalty", C.c_double),
        ("angle_variance_penalty", C.c_double),
        ("minimum_distance_penalty", C.c_double),
        ("minimum_angle_penalty", C.c_double),
        ("use_response_expansion", C.c_int32),
        ("reserved", C.c_int32),
    ]


class LaserParams(C.Structure):
    """lslam_laser (karto::LaserRangeFinder parameters)."""

    _fields_ = [
        ("minimum_angle", C.c_double),
        ("maximum_angle", C.c_double),
        ("angular_resolution", C.c_double),
        ("minimum_range", C.c_double),
        ("maximum_range", C.c_double),
        ("range_threshold", C.c_double),
        ("offset_x", C.c_double),
        ("offset_y", C.c_double),
        ("offset_heading", C.c_double),
    ]


class MatchResult(C.Structure):
    _fields_ = [
        ("pose", C.c_double * 3),
        ("response", C.c_double),
        ("covariance", C.c_double * 9),
        ("status", C.c_int32),
        ("flags", C.c_int32),
    ]


RESULT_DTYPE = np.dtype(
    [("pose", "f8", 3), ("response", "f8"), ("covariance", "f8", (3, 3)), ("status", "i4"), ("flags", "i4")]
)
assert RESULT_DTYPE.itemsize == 112 == C.sizeof(MatchResult)


class FrontEndConfig(C.Structure):
    """lslam_frontend_config: the Mapper parameters Mapper::Process / MapperGraph read (Mapper.cpp:1457-1604)."""

    _fields_ = [
        ("scan_buffer_size", C.c_int32),
        ("use_scan_barycenter", C.c_int32),
        ("do_loop_closing", C.c_int32),
        ("loop_match_minimum_chain_size", C.c_int32),
        ("scan_buffer_maximum_scan_distance", C.c_double),
        ("minimum_travel_distance", C.c_double),
        ("minimum_travel_heading", C.c_double),
        ("minimum_time_interval", C.c_double),
        ("link_match_minimum_response_fine", C.c_double),
        ("link_scan_maximum_distance", C.c_double),
        ("loop_search_maximum_distance", C.c_double),
        ("loop_match_maximum_variance_coarse", C.c_double),
        ("loop_match_minimum_response_coarse", C.c_double),
        ("loop_match_minimum_response_fine", C.c_double),
        ("loop_search_space_dimension", C.c_double),
        ("loop_search_space_resolution", C.c_double),
        ("loop_search_space_smear_deviation", C.c_double),
    ]


def frontend_config(**kw) -> FrontEndConfig:
    """Library defaults (lslam_frontend_config_defaults) with keyword overrides."""
    c = FrontEndConfig()
    lib().lslam_frontend_config_defaults(C.byref(c))
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


class HectorScan(C.Structure):
    """lslam_hector_scan: LaserScan header + the node's filters + the base_link -> laser transform."""

    _fields_ = [(k, C.c_float) for k in (
        "angle_min", "angle_increment", "range_min", "range_max", "range_cutoff", "sqr_laser_min_dist",
        "sqr_laser_max_dist", "use_max_scan_range", "laser_z_min", "laser_z_max", "laser_x", "laser_y", "laser_z",
        "laser_yaw")]


def hector_scan(laser, min_dist=0.4, max_dist=30.0, use_max=20.0, cutoff=30.0, z_min=-1.0, z_max=1.0,
                laser_pose=(0.0, 0.0, 0.0, 0.0)) -> HectorScan:
    """The hector_slam node's parameters (hector_slam.cc:40-75, 193) for a synth.Laser."""
    return HectorScan(laser.angle_min, laser.angle_increment, laser.range_min, laser.range_max, cutoff,
                      min_dist * min_dist, max_dist * max_dist, use_max, z_min, z_max, *laser_pose)


class DeskewParams(C.Structure):
    _fields_ = [("angle_min", C.c_float), ("angle_increment", C.c_float), ("range_min", C.c_float), ("range_max", C.c_float),
                ("scan_time_start", C.c_double), ("time_increment", C.c_double), ("use_imu", C.c_int32),
                ("use_odom", C.c_int32), ("start_odom_time", C.c_double), ("end_odom_time", C.c_double),
                ("odom_incre_x", C.c_float), ("odom_incre_y", C.c_float), ("odom_incre_z", C.c_float), ("pad", C.c_float)]


def deskew_scan(ctx, ranges_f32, params: DeskewParams, imu_time=None, imu_rot=None):
    """lesson5 CorrectLaserScan on the device -> (xyz [n,3] float32, valid [n] bool).  imu_rot: [n_imu, 3]."""
    r = np.ascontiguousarray(ranges_f32, dtype=np.float32)
    n_imu = 0 if imu_time is None else len(imu_time)
    it = np.ascontiguousarray(imu_time if n_imu else [0.0], dtype=np.float64)
    ir = np.ascontiguousarray(np.asarray(imu_rot if n_imu else [[0.0, 0.0, 0.0]], dtype=np.float64).T)
    xyz = np.zeros((len(r), 3), np.float32)
    valid = np.zeros(len(r), np.uint8)
    ctx.check(ctx.L.lslam_deskew_scan(ctx.h, r.ctypes.data, len(r), C.byref(params), it.ctypes.data, ir[0].ctypes.data,
                                      ir[1].ctypes.data, ir[2].ctypes.data, max(n_imu, 1) if params.use_imu else 0,
                                      xyz.ctypes.data, valid.ctypes.data))
    return xyz, valid.astype(bool)


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("total_ms", C.c_double)]


def baseline_config(**kw) -> MatcherConfig:
    """BASELINE.json cfg 3/4: 0.05 m cells, +-0.5 m / +-20 deg window, 2005x2005 grid."""
    d = dict(
        search_size=1.0,
        resolution=0.05,
        smear_deviation=0.03,
        range_threshold=49.5,
        coarse_search_angle_offset=0.349,
        coarse_angle_resolution=0.0349,
        fine_search_angle_offset=0.00349,
        distance_variance_penalty=0.3 * 0.3,
        angle_variance_penalty=math.radians(20.0) ** 2,
        minimum_distance_penalty=0.5,
        minimum_angle_penalty=0.9,
        use_response_expansion=0,
        reserved=0,
    )
    d.update(kw)
    return MatcherConfig(**d)


def laser_params(laser, range_threshold: float = 49.5, offset=(0.0, 0.0, 0.0)) -> LaserParams:
    """From a synth.Laser (LaserScan header fields, karto_slam.cc:384-395)."""
    return LaserParams(laser.angle_min, laser.angle_max, laser.angle_increment, laser.range_min,
                       laser.range_max, range_threshold, *offset)


_LIB = None


def lib() -> C.CDLL:
    """Load (building if needed) liblslam_gpu.so and declare the prototypes."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # LSLAM_GPU_LIB: developer override naming a prebuilt variant of the library (kernel A/B runs,
    # tools/ab_variants.py); it is still the HIP library -- there is no CPU fallback behind it
    path = os.environ.get("LSLAM_GPU_LIB") or _build.build_library()
    L = C.CDLL(str(path))
    vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
    L.lslam_abi_version.restype = i32
    L.lslam_create.argtypes = [i32, C.POINTER(vp)]
    L.lslam_destroy.argtypes = [vp]
    L.lslam_last_error.restype = C.c_char_p
    L.lslam_last_error.argtypes = [vp]
    L.lslam_synchronize.argtypes = [vp]
    L.lslam_stream.restype = vp
    L.lslam_stream.argtypes = [vp]
    L.lslam_dev_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.lslam_dev_free.argtypes = [vp, vp]
    L.lslam_dev_upload.argtypes = [vp, vp, vp, C.c_size_t]
    L.lslam_dev_download.argtypes = [vp, vp, vp, C.c_size_t]
    L.lslam_profile_enable.argtypes = [vp, i32]
    L.lslam_profile_only.argtypes = [vp, C.c_char_p]
    L.lslam_profile_reset.argtypes = [vp]
    L.lslam_profile_read.argtypes = [vp, vp, i32]
    L.lslam_matcher_config_defaults.argtypes = [C.POINTER(MatcherConfig)]
    L.lslam_matcher_config_defaults.restype = None
    L.lslam_matcher_create.argtypes = [vp, C.POINTER(MatcherConfig), C.POINTER(LaserParams), C.POINTER(vp)]
    L.lslam_matcher_destroy.argtypes = [vp]
    L.lslam_matcher_destroy.restype = None
    L.lslam_matcher_num_beams.argtypes = [vp]
    L.lslam_matcher_grid_info.argtypes = [vp, vp, vp]
    L.lslam_matcher_get_grid_u8.argtypes = [vp, vp]
    L.lslam_matcher_get_kernel_u8.argtypes = [vp, vp]
    L.lslam_matcher_set_grid_u8.argtypes = [vp, vp, vp]
    L.lslam_matcher_set_grid_u8_dev.argtypes = [vp, vp, vp]
    L.lslam_matcher_grid_dev_ptr.restype = vp
    L.lslam_matcher_grid_dev_ptr.argtypes = [vp]
    L.lslam_sensor_pose_from_robot.argtypes = [C.POINTER(LaserParams), vp, vp]
    L.lslam_sensor_pose_from_robot.restype = None
    L.lslam_robot_pose_from_sensor.argtypes = [C.POINTER(LaserParams), vp, vp]
    L.lslam_robot_pose_from_sensor.restype = None
    L.lslam_matcher_set_base_scans.argtypes = [vp, i32, vp, i32, vp, vp]
    L.lslam_matcher_match_scan.argtypes = [vp, i32, vp, i32, vp, vp, vp, i32, i32, vp]
    L.lslam_matcher_match_batch.argtypes = [vp, i32, vp, i32, vp, i32, i32, vp]
    L.lslam_matcher_match_batch_dev_f32.argtypes = [vp, i32, vp, i32, vp, i32, i32, vp]
    L.lslam_matcher_match_batch_dev_f64.argtypes = [vp, i32, vp, i32, vp, i32, i32, vp]
    L.lslam_matcher_debug_lookup_table.argtypes = [vp, vp, vp, dbl, dbl, dbl, vp, C.POINTER(i32)]
    L.lslam_matcher_debug_coarse_sums.argtypes = [vp, vp, vp, vp, C.POINTER(i32), C.POINTER(i32),
                                                  C.POINTER(i32), i32]
    L.lslam_matcher_debug_coarse_sums_batch.argtypes = [vp, i32, vp, i32, vp, vp]
    L.lslam_matcher_debug_fine_sums_batch.argtypes = [vp, i32, vp, i32, vp, vp, vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.lslam_matcher_debug_valid_mask.argtypes = [vp, vp, vp, vp, vp]
    L.lslam_frontend_create.argtypes = [vp, i32, dbl, dbl, dbl, C.POINTER(vp)]
    L.lslam_frontend_destroy.argtypes = [vp]
    L.lslam_frontend_destroy.restype = None
    L.lslam_frontend_reset.argtypes = [vp]
    L.lslam_frontend_process.argtypes = [vp, vp, i32, vp, C.POINTER(i32), vp, vp, C.POINTER(dbl)]
    L.lslam_frontend_running_scans.argtypes = [vp]
    L.lslam_frontend_config_defaults.argtypes = [C.POINTER(FrontEndConfig)]
    L.lslam_frontend_config_defaults.restype = None
    L.lslam_frontend_create_ex.argtypes = [vp, C.POINTER(FrontEndConfig), C.POINTER(vp)]
    L.lslam_frontend_process_stamped.argtypes = [vp, vp, i32, vp, dbl, C.POINTER(i32), vp, vp, C.POINTER(dbl)]
    L.lslam_frontend_num_scans.argtypes = [vp]
    L.lslam_frontend_scan_pose.argtypes = [vp, i32, vp]
    L.lslam_frontend_stats.argtypes = [vp, vp]
    L.lslam_frontend_process_many.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, vp]
    L.lslam_frontend_lookahead_stats.argtypes = [vp, vp]
    L.lslam_frontend_spec_chain_stats.argtypes = [vp, vp]
    L.lslam_debug_frontend_anchor_row.argtypes = [vp, C.c_int, vp]
    L.lslam_occgrid_create_from_scans.argtypes = [vp, C.POINTER(LaserParams), i32, vp, i32, vp, dbl, C.POINTER(vp)]
    L.lslam_occgrid_destroy.argtypes = [vp]
    L.lslam_occgrid_destroy.restype = None
    L.lslam_occgrid_info.argtypes = [vp, vp, vp, C.POINTER(dbl)]
    L.lslam_occgrid_read_u8.argtypes = [vp, vp]
    L.lslam_occgrid_read_ros_i8.argtypes = [vp, vp]
    L.lslam_occgrid_scan_bounds.argtypes = [vp, C.POINTER(LaserParams), i32, vp, i32, vp, vp]
    L.lslam_occgrid_create_partial.argtypes = [vp, C.POINTER(LaserParams), i32, vp, i32, vp, dbl, vp, C.POINTER(vp)]
    L.lslam_occgrid_counter_words.argtypes = [vp, C.POINTER(C.c_size_t)]
    L.lslam_occgrid_export_counters.argtypes = [vp, vp, i32]
    L.lslam_occgrid_import_counters.argtypes = [vp, vp, i32, i32]
    L.lslam_occgrid_counters_dev_ptr.restype = vp
    L.lslam_occgrid_counters_dev_ptr.argtypes = [vp]
    L.lslam_occgrid_create_sharded.argtypes = [vp, C.POINTER(LaserParams), i32, vp, i32, vp, dbl, vp, C.POINTER(vp)]
    L.lslam_pool_occgrid_from_scans.argtypes = [vp, C.POINTER(LaserParams), i32, vp, i32, vp, dbl, C.POINTER(vp)]
    L.lslam_map_create.argtypes = [vp, i32, i32, C.c_float, C.c_float, C.c_float, i32, C.POINTER(vp)]
    L.lslam_map_destroy.argtypes = [vp]
    L.lslam_map_destroy.restype = None
    L.lslam_map_reset.argtypes = [vp]
    L.lslam_map_set_update_factor_free.argtypes = [vp, C.c_float]
    L.lslam_map_set_update_factor_occupied.argtypes = [vp, C.c_float]
    L.lslam_map_levels.argtypes = [vp]
    L.lslam_map_size.argtypes = [vp, i32, C.POINTER(i32), C.POINTER(i32)]
    L.lslam_map_scale_to_map.restype = C.c_float
    L.lslam_map_scale_to_map.argtypes = [vp, i32]
    L.lslam_map_update_by_scan.argtypes = [vp, vp, i32, vp, vp]
    L.lslam_map_update_by_scan_dev.argtypes = [vp, vp, i32, vp, vp]
    L.lslam_map_update_just_once.argtypes = [vp, vp, i32, vp, C.c_float, C.c_float, dbl]
    L.lslam_map_match_data.argtypes = [vp, vp, i32, vp, vp, vp, vp]
    L.lslam_map_cached_points.argtypes = [vp]
    L.lslam_map_set_option.argtypes = [vp, i32, i32]
    L.lslam_map_batch_stats.argtypes = [vp, vp]
    L.lslam_map_read_logodds.argtypes = [vp, i32, vp]
    L.lslam_map_read_occupancy_i8.argtypes = [vp, i32, vp]
    L.lslam_map_flush.argtypes = [vp]
    L.lslam_map_cells_dev_ptr.restype = vp
    L.lslam_map_cells_dev_ptr.argtypes = [vp, i32]
    L.lslam_map_set_scan.argtypes = [vp, vp, i32, C.POINTER(HectorScan), C.POINTER(i32)]
    L.lslam_map_read_container.argtypes = [vp, vp, i32, vp]
    L.lslam_map_match_container.argtypes = [vp, vp, vp, vp]
    L.lslam_map_update_by_container.argtypes = [vp, vp]
    L.lslam_map_update_batch.argtypes = [vp, i32, vp, vp, vp, vp]
    L.lslam_map_update_batch_dev.argtypes = [vp, i32, vp, vp, vp, vp]
    L.lslam_pool_create.argtypes = [i32, C.POINTER(MatcherConfig), C.POINTER(LaserParams), C.POINTER(vp)]
    L.lslam_pool_create_on.argtypes = [vp, i32, C.POINTER(MatcherConfig), C.POINTER(LaserParams), C.POINTER(vp)]
    L.lslam_pool_destroy.argtypes = [vp]
    L.lslam_pool_destroy.restype = None
    L.lslam_pool_devices.argtypes = [vp]
    L.lslam_pool_last_error.argtypes = [vp]
    L.lslam_pool_last_error.restype = C.c_char_p
    L.lslam_pool_set_base_scans.argtypes = [vp, i32, vp, i32, vp, vp, i32]
    L.lslam_pool_match_batch.argtypes = [vp, i32, vp, i32, vp, i32, i32, vp]
    L.lslam_deskew_scan.argtypes = [vp, vp, i32, C.POINTER(DeskewParams), vp, vp, vp, vp, i32, vp, vp]
    L.lslam_clock_sample.argtypes = [vp, vp]
    L.lslam_matcher_get_option.argtypes = [vp, i32]
    L.lslam_matcher_set_option.argtypes = [vp, i32, i32]
    L.lslam_matcher_flush.argtypes = [vp]
    L.lslam_matcher_pipelined_steps.argtypes = [vp]
    L.lslam_matcher_pipelined_steps.restype = C.c_int64
    L.lslam_matcher_step_kernel_launches.argtypes = [vp]
    L.lslam_matcher_step_kernel_launches.restype = C.c_int64
    L.lslam_matcher_lone_kernel_launches.argtypes = [vp]
    L.lslam_matcher_lone_kernel_launches.restype = C.c_int64
    i64 = C.c_int64
    L.lslam_scan_cache_create.argtypes = [vp, C.POINTER(LaserParams), C.POINTER(vp)]
    L.lslam_scan_cache_destroy.argtypes = [vp]
    L.lslam_scan_cache_destroy.restype = None
    L.lslam_scan_cache_put.argtypes = [vp, i64, vp]
    L.lslam_scan_cache_contains.argtypes = [vp, i64]
    L.lslam_scan_cache_prepare.argtypes = [vp, i64, vp]
    L.lslam_scan_cache_forget.argtypes = [vp, i64]
    L.lslam_scan_cache_size.argtypes = [vp]
    L.lslam_scan_cache_counters.argtypes = [vp, vp]
    L.lslam_matcher_match_scan_cached.argtypes = [vp, vp, i32, vp, vp, i64, vp, vp, i32, vp]
    L.lslam_matcher_read_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.lslam_matcher_read_beam_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    _LIB = L
    return L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a: np.ndarray) -> int:
    """Address of a contiguous array's first byte.  `a.ctypes.data` builds a helper object per access (~1 us, four of them per
    per-scan call were a twentieth of a streamed scan); the buffer protocol gives the same address in a third of the time."""
    try:
        return C.addressof(C.c_char.from_buffer(a))
    except (TypeError, ValueError, BufferError):  # read-only or empty arrays
        return a.ctypes.data


class Context:
    """lslam_context: one GPU, one HIP stream."""

    def __init__(self, device: int = 0):
        self.L = lib()
        h = C.c_void_p()
        rc = self.L.lslam_create(device, C.byref(h))
        if rc != LSLAM_OK:
            raise LslamError(rc, self.L.lslam_last_error(None).decode())
        self.h = h
        self.device = device
        self._children = weakref.WeakSet()  # handles that must be released before the context

    def _adopt(self, child):
        self._children.add(child)

    def check(self, rc: int):
        if rc != LSLAM_OK:
            raise LslamError(rc, self.L.lslam_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            for child in list(self._children):  # matcher / map / grid handles hold a pointer to the context
                child.close()
            self.L.lslam_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the order of finalisers is arbitrary (a context may already be gone under a handle
        # that points into it) and the process is exiting anyway: release explicitly, or not at all
        try:
            if sys.is_finalizing():
                return
            self.close()
        except Exception:  # incl. module globals already torn down (sys is None) late in shutdown
            pass

    def synchronize(self):
        self.check(self.L.lslam_synchronize(self.h))

    def clock_sample(self) -> np.ndarray:
        """[256][3] uint64: {CU id + 1, shader-clock ticks, 100 MHz ticks} from 256 single-wave blocks (lslam_clock_sample)."""
        out = np.zeros(768, dtype=np.uint64)
        self.check(self.L.lslam_clock_sample(self.h, out.ctypes.data))
        return out.reshape(256, 3)

    @staticmethod
    def clock_ghz(before: np.ndarray, after: np.ndarray):
        """Average shader clock [GHz] between two clock_sample() results: per CU that reported in both, (t1 - t0) / ((r1 - r0) /
        100 MHz); the median over those CUs (None when no CU is in both samples)."""
        b = {int(k): (int(t), int(r)) for k, t, r in before if k}
        ratios = []
        for k, t, r in after:
            if int(k) in b:
                t0, r0 = b[int(k)]
                if int(r) > r0 and int(t) > t0:
                    ratios.append((int(t) - t0) / ((int(r) - r0) / 1e8))
        return float(np.median(ratios)) / 1e9 if ratios else None

    @property
    def stream(self) -> int:
        return self.L.lslam_stream(self.h)

    # raw HBM helpers (bench/tests); torch tensors' data_ptr() work just as well
    def alloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        self.check(self.L.lslam_dev_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr: int):
        self.check(self.L.lslam_dev_free(self.h, ptr))

    def upload(self, ptr: int, arr: np.ndarray):
        a = np.ascontiguousarray(arr)
        self.check(self.L.lslam_dev_upload(self.h, ptr, a.ctypes.data, a.nbytes))

    def download(self, ptr: int, arr: np.ndarray):
        assert arr.flags["C_CONTIGUOUS"]
        self.check(self.L.lslam_dev_download(self.h, arr.ctypes.data, ptr, arr.nbytes))

    def profile(self, on: bool):
        self.check(self.L.lslam_profile_enable(self.h, int(on)))

    def profile_only(self, kernel_name: str | None):
        """Time only the kernels launched under this name (None = all): fewer events in the stream."""
        self.check(self.L.lslam_profile_only(self.h, kernel_name.encode() if kernel_name else None))

    def profile_reset(self):
        self.check(self.L.lslam_profile_reset(self.h))

    def profile_read(self) -> dict:
        buf = (KernelTime * 64)()
        n = self.L.lslam_profile_read(self.h, buf, 64)
        return {buf[i].name.decode(): (int(buf[i].launches), float(buf[i].total_ms)) for i in range(n)}


class ScanMatcher:
    """karto::ScanMatcher on the GPU.  Poses are SENSOR poses (x, y, heading)."""

    def __init__(self, ctx: Context, cfg: MatcherConfig, laser: LaserParams):
        self.ctx, self.L, self.cfg, self.laser = ctx, ctx.L, cfg, laser
        h = C.c_void_p()
        ctx.check(self.L.lslam_matcher_create(ctx.h, C.byref(cfg), C.byref(laser), C.byref(h)))
        self.h = h
        self._frontends = weakref.WeakSet()
        ctx._adopt(self)

    # reference spelling
    @classmethod
    def Create(cls, ctx, cfg, laser):
        """ScanMatcher::Create: returns None where the reference returns NULL."""
        try:
            return cls(ctx, cfg, laser)
        except LslamError as e:
            if e.code == -1:
                return None
            raise

    def close(self):
        if getattr(self, "h", None):
            for fe in list(self._frontends):  # a front-end borrows its matcher
                fe.close()
            self.L.lslam_matcher_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the order of finalisers is arbitrary (a context may already be gone under a handle
        # that points into it) and the process is exiting anyway: release explicitly, or not at all
        try:
            if sys.is_finalizing():
                return
            self.close()
        except Exception:  # incl. module globals already torn down (sys is None) late in shutdown
            pass

    @property
    def num_beams(self) -> int:
        return self.L.lslam_matcher_num_beams(self.h)

    def grid_info(self) -> dict:
        i = np.zeros(8, dtype=np.int32)
        off = np.zeros(2)
        self.ctx.check(self.L.lslam_matcher_grid_info(self.h, i.ctypes.data, off.ctypes.data))
        keys = ("width", "height", "stride", "roi_x", "roi_y", "roi_w", "roi_h", "kernel_size")
        d = {k: int(v) for k, v in zip(keys, i)}
        d["offset"] = off
        return d

    def GetCorrelationGrid(self) -> np.ndarray:
        gi = self.grid_info()
        out = np.zeros((gi["height"], gi["stride"]), dtype=np.uint8)
        self.ctx.check(self.L.lslam_matcher_get_grid_u8(self.h, out.ctypes.data))
        return out

    def kernel(self) -> np.ndarray:
        k = self.grid_info()["kernel_size"]
        out = np.zeros((k, k), dtype=np.uint8)
        self.ctx.check(self.L.lslam_matcher_get_kernel_u8(self.h, out.ctypes.data))
        return out

    def set_grid(self, grid: np.ndarray, offset):
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        gi = self.grid_info()
        assert g.size == gi["height"] * gi["stride"]
        o = _f64(offset)
        self.ctx.check(self.L.lslam_matcher_set_grid_u8(self.h, g.ctypes.data, o.ctypes.data))

    def set_grid_dev(self, ptr: int, offset):
        o = _f64(offset)
        self.ctx.check(self.L.lslam_matcher_set_grid_u8_dev(self.h, ptr, o.ctypes.data))

    def set_option(self, name: str, value: int):
        opt = {"row_occupancy": 1, "collect_stats": 2, "lds_staged": 3, "pipeline_depth": 4, "step_kernel": 5,
               "step_min_scans": 6, "rows_waves": 7, "check_output_reuse": 8, "lone_kernel": 9}[name]
        self.ctx.check(self.L.lslam_matcher_set_option(self.h, opt, int(value)))

    @property
    def step_kernel_launches(self) -> int:
        """Batched matches that went out as ONE launch (set_option('step_kernel', 3 or 4)) so far."""
        return int(self.L.lslam_matcher_step_kernel_launches(self.h))

    @property
    def lone_kernel_launches(self) -> int:
        """Single-scan matches that went out as ONE launch (set_option('lone_kernel', 4 / 8 / 16)) so far."""
        return int(self.L.lslam_matcher_lone_kernel_launches(self.h))

    def flush(self):
        """Order the context stream behind every pipelined step in flight (set_option('pipeline_depth', D > 1))."""
        self.ctx.check(self.L.lslam_matcher_flush(self.h))

    @property
    def pipelined_steps(self) -> int:
        return int(self.L.lslam_matcher_pipelined_steps(self.h))

    def read_stats(self) -> dict:
        out = (C.c_uint64 * 4)()
        self.ctx.check(self.L.lslam_matcher_read_stats(self.h, out))
        b = (C.c_uint64 * 4)()
        self.ctx.check(self.L.lslam_matcher_read_beam_stats(self.h, b))
        return {"rows_in_range": int(out[0]), "rows_live": int(out[1]), "beam_angles": int(out[2]),
                "beam_angles_queued": int(out[3]), "beams_readable": int(b[0]), "beams_live_in_some_angle": int(b[1]),
                "lds_drains_staged": int(b[2]), "lds_drains_global": int(b[3])}

    @property
    def grid_dev_ptr(self) -> int:
        """Device address of the raw grid bytes.  After WRITING through it call set_grid_dev(ptr, offset) with the same
        pointer: the parity planes, row-occupancy bitmaps and tiled copies derived from the grid are refreshed only
        when the matcher is told that the grid changed."""
        return self.L.lslam_matcher_grid_dev_ptr(self.h)

    def sensor_pose_from_robot(self, robot):
        r, out = _f64(robot), np.zeros(3)
        self.L.lslam_sensor_pose_from_robot(C.byref(self.laser), r.ctypes.data, out.ctypes.data)
        return out

    def robot_pose_from_sensor(self, sensor):
        s, out = _f64(sensor), np.zeros(3)
        self.L.lslam_robot_pose_from_sensor(C.byref(self.laser), s.ctypes.data, out.ctypes.data)
        return out

    def AddScans(self, base_ranges, base_sensor_poses, center_pose):
        """MatchScan steps 1-4 + AddScans: rebuild the grid around center_pose."""
        r, p, c = _f64(base_ranges), _f64(base_sensor_poses), _f64(center_pose)
        r = r.reshape(-1, r.shape[-1]) if r.size else r.reshape(0, max(self.num_beams, 1))
        self.ctx.check(self.L.lslam_matcher_set_base_scans(self.h, r.shape[0], r.ctypes.data, r.shape[1],
                                                           p.ctypes.data, c.ctypes.data))

    def MatchScan(self, query_ranges, query_sensor_pose, base_ranges, base_sensor_poses,
                  doPenalize: bool = True, doRefineMatch: bool = True):
        """ScanMatcher::MatchScan -> (response, mean pose, covariance 3x3)."""
        q, qp = _f64(query_ranges), _f64(query_sensor_pose)
        r, p = _f64(base_ranges), _f64(base_sensor_poses)
        r = r.reshape(-1, r.shape[-1]) if r.size else r.reshape(0, max(self.num_beams, 1))
        res = np.zeros(1, dtype=RESULT_DTYPE)
        self.ctx.check(self.L.lslam_matcher_match_scan(self.h, r.shape[0], _ptr(r), r.shape[1], _ptr(p),
                                                       _ptr(q), _ptr(qp), int(doPenalize),
                                                       int(doRefineMatch), _ptr(res)))
        if res["status"][0] != 0:
            raise LslamError(int(res["status"][0]), "the reference would have thrown here")
        return float(res["response"][0]), res["pose"][0].copy(), res["covariance"][0].copy()

    def match_batch(self, ranges, sensor_poses, doPenalize: bool = True, doRefineMatch: bool = True) -> np.ndarray:
        """Coarse+fine search of S independent scans against the CURRENT grid (host arrays)."""
        r, p = _f64(ranges), _f64(sensor_poses)
        r = r.reshape(-1, r.shape[-1])
        res = np.zeros(r.shape[0], dtype=RESULT_DTYPE)
        self.ctx.check(self.L.lslam_matcher_match_batch(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data,
                                                        int(doPenalize), int(doRefineMatch), res.ctypes.data))
        return res

    def match_batch_dev(self, n_scans: int, ranges_ptr: int, stride: int, poses_ptr: int, out_ptr: int,
                        dtype="f32", doPenalize: bool = True, doRefineMatch: bool = True):
        """Same with float32/float64 ranges, poses and results resident in HBM; asynchronous.  With
        set_option('pipeline_depth', D > 1) consecutive calls are pipelined steps: up to D in flight at once, so give D
        consecutive calls their own result buffers and read them after ctx.synchronize()."""
        fn = self.L.lslam_matcher_match_batch_dev_f32 if dtype == "f32" else self.L.lslam_matcher_match_batch_dev_f64
        self.ctx.check(fn(self.h, n_scans, ranges_ptr, stride, poses_ptr, int(doPenalize), int(doRefineMatch), out_ptr))

    # inspection hooks (parity tests)
    def lookup_table(self, ranges, sensor_pose, angle_center, angle_offset, angle_resolution) -> np.ndarray:
        r, p = _f64(ranges), _f64(sensor_pose)
        na = C.c_int()
        self.ctx.check(self.L.lslam_matcher_debug_lookup_table(self.h, r.ctypes.data, p.ctypes.data, angle_center,
                                                               angle_offset, angle_resolution, None, C.byref(na)))
        out = np.zeros((na.value, self.num_beams), dtype=np.int32)
        self.ctx.check(self.L.lslam_matcher_debug_lookup_table(self.h, r.ctypes.data, p.ctypes.data, angle_center,
                                                               angle_offset, angle_resolution, out.ctypes.data,
                                                               C.byref(na)))
        return out

    def coarse_sums(self, ranges, sensor_pose, force_generic: bool = False) -> np.ndarray:
        r, p = _f64(ranges), _f64(sensor_pose)
        nx, ny, na = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.L.lslam_matcher_debug_coarse_sums(self.h, r.ctypes.data, p.ctypes.data, None,
                                                              C.byref(nx), C.byref(ny), C.byref(na), 0))
        out = np.zeros((ny.value, nx.value, na.value), dtype=np.int32)
        self.ctx.check(self.L.lslam_matcher_debug_coarse_sums(self.h, r.ctypes.data, p.ctypes.data, out.ctypes.data,
                                                              C.byref(nx), C.byref(ny), C.byref(na),
                                                              int(force_generic)))
        return out

    def coarse_sums_batch(self, ranges, sensor_poses) -> np.ndarray:
        """Coarse-pass response numerators of EVERY scan of a batch, [n_scans, ny, nx, na] int32, through the launches a
        coarse-only match of that batch takes (lslam_matcher_debug_coarse_sums_batch)."""
        r, p = _f64(ranges), _f64(sensor_poses)
        n = r.shape[0]
        nx, ny, na = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.L.lslam_matcher_debug_coarse_sums(self.h, r.ctypes.data, p.ctypes.data, None,
                                                              C.byref(nx), C.byref(ny), C.byref(na), 0))
        out = np.zeros((n, ny.value, nx.value, na.value), dtype=np.int32)
        self.ctx.check(self.L.lslam_matcher_debug_coarse_sums_batch(self.h, n, r.ctypes.data, r.shape[1], p.ctypes.data,
                                                                    out.ctypes.data))
        return out

    def fine_sums_batch(self, ranges, sensor_poses):
        """Fine-pass numerators of every scan of a batch and the centre each fine pass was searched around:
        ([n_scans, ny, nx, na] int32, [n_scans, 3]) (lslam_matcher_debug_fine_sums_batch)."""
        r, p = _f64(ranges), _f64(sensor_poses)
        n = r.shape[0]
        nx, ny, na = C.c_int(), C.c_int(), C.c_int()
        self.ctx.check(self.L.lslam_matcher_debug_fine_sums_batch(self.h, n, r.ctypes.data, r.shape[1], p.ctypes.data, None, None,
                                                                  C.byref(nx), C.byref(ny), C.byref(na)))
        out = np.zeros((n, ny.value, nx.value, na.value), dtype=np.int32)
        centers = np.zeros((n, 3))
        self.ctx.check(self.L.lslam_matcher_debug_fine_sums_batch(self.h, n, r.ctypes.data, r.shape[1], p.ctypes.data,
                                                                  centers.ctypes.data, out.ctypes.data, C.byref(nx), C.byref(ny),
                                                                  C.byref(na)))
        return out, centers

    def valid_mask(self, ranges, sensor_pose, viewpoint) -> np.ndarray:
        r, p, v = _f64(ranges), _f64(sensor_pose), _f64(viewpoint)
        out = np.zeros(self.num_beams, dtype=np.uint8)
        self.ctx.check(self.L.lslam_matcher_debug_valid_mask(self.h, r.ctypes.data, p.ctypes.data, v.ctypes.data,
                                                             out.ctypes.data))
        return out


class ScanCache:
    """lslam_scan_cache: the device-side scan cache behind seam B1 (readings, world points and FindValidPoints anchors of
    every scan the caller has named, resident in HBM); MatchScan then names its base scans by id."""

    PENALIZE, REFINE, QUERY_TAKES_RESULT_POSE = 1, 2, 4

    def __init__(self, ctx: Context, laser: LaserParams):
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        ctx.check(self.L.lslam_scan_cache_create(ctx.h, C.byref(laser), C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            self.L.lslam_scan_cache_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            if sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass

    def put(self, scan_id: int, ranges):
        r = _f64(ranges)
        self.ctx.check(self.L.lslam_scan_cache_put(self.h, int(scan_id), r.ctypes.data))

    def prepare(self, scan_id: int, sensor_pose):
        """World points + anchors of a cached scan at this pose, enqueued (asynchronous)."""
        p = _f64(sensor_pose)
        self.ctx.check(self.L.lslam_scan_cache_prepare(self.h, int(scan_id), p.ctypes.data))

    def __contains__(self, scan_id: int) -> bool:
        return bool(self.L.lslam_scan_cache_contains(self.h, int(scan_id)))

    def forget(self, scan_id: int = -1):
        self.ctx.check(self.L.lslam_scan_cache_forget(self.h, int(scan_id)))

    def __len__(self) -> int:
        return int(self.L.lslam_scan_cache_size(self.h))

    def counters(self) -> dict:
        out = np.zeros(5, dtype=np.int64)
        self.ctx.check(self.L.lslam_scan_cache_counters(self.h, out.ctypes.data))
        return dict(zip(("matches", "uploads", "refreshed", "speculated", "resident_bytes"), (int(v) for v in out)))

    def MatchScan(self, matcher: "ScanMatcher", base_ids, base_sensor_poses, query_sensor_pose, query_id: int = -1,
                  query_ranges=None, doPenalize: bool = True, doRefineMatch: bool = True, takes_result_pose: bool = False):
        """ScanMatcher::MatchScan with the base scans named by id -> (response, mean pose, covariance 3x3)."""
        ids = np.ascontiguousarray(base_ids, dtype=np.int64)
        p, qp = _f64(base_sensor_poses), _f64(query_sensor_pose)
        q = _f64(query_ranges) if query_ranges is not None else None
        flags = (1 if doPenalize else 0) | (2 if doRefineMatch else 0) | (4 if takes_result_pose else 0)
        res = np.zeros(1, dtype=RESULT_DTYPE)
        self.ctx.check(self.L.lslam_matcher_match_scan_cached(matcher.h, self.h, len(ids), _ptr(ids), _ptr(p),
                                                              int(query_id), _ptr(q) if q is not None else None,
                                                              _ptr(qp), flags, _ptr(res)))
        if res["status"][0] != 0:
            raise LslamError(int(res["status"][0]), "the reference would have thrown here")
        return float(res["response"][0]), res["pose"][0].copy(), res["covariance"][0].copy()


class MatcherPool:
    """lslam_pool: the batched many-scan mode over several GPUs of one node from ONE process -- one matcher per device,
    the shared grid copied device-to-device, scans [r*B/W, (r+1)*B/W) on device r, results in scan order."""

    def __init__(self, cfg: MatcherConfig, laser: LaserParams, devices=0):
        self.L = lib()
        h = C.c_void_p()
        if isinstance(devices, int):
            rc = self.L.lslam_pool_create(devices, C.byref(cfg), C.byref(laser), C.byref(h))
        else:
            ids = np.ascontiguousarray(devices, dtype=np.int32)
            rc = self.L.lslam_pool_create_on(ids.ctypes.data, len(ids), C.byref(cfg), C.byref(laser), C.byref(h))
        if rc != LSLAM_OK:
            raise LslamError(rc, self.L.lslam_pool_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.lslam_pool_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the order of finalisers is arbitrary (a context may already be gone under a handle
        # that points into it) and the process is exiting anyway: release explicitly, or not at all
        try:
            if sys.is_finalizing():
                return
            self.close()
        except Exception:  # incl. module globals already torn down (sys is None) late in shutdown
            pass

    def _check(self, rc):
        if rc != LSLAM_OK:
            raise LslamError(rc, self.L.lslam_pool_last_error(self.h).decode())

    @property
    def devices(self) -> int:
        return self.L.lslam_pool_devices(self.h)

    def AddScans(self, base_ranges, base_sensor_poses, center_pose, rebuild_everywhere: bool = False):
        r, p, c = _f64(base_ranges), _f64(base_sensor_poses), _f64(center_pose)
        r = r.reshape(-1, r.shape[-1])
        self._check(self.L.lslam_pool_set_base_scans(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data,
                                                     c.ctypes.data, int(rebuild_everywhere)))

    def CreateOccupancyGrid(self, laser: LaserParams, ranges, sensor_poses, resolution: float):
        """OccupancyGrid::CreateFromScans sharded over the pool's devices in this one process: RCCL (ncclCommInitAll +
        all-reduce) when the devices are distinct, counter addition on the first device otherwise.  Returns
        (cells [h, w] uint8, offset_xy)."""
        r, p = _f64(ranges), _f64(sensor_poses)
        r = r.reshape(-1, r.shape[-1])
        h = C.c_void_p()
        self._check(self.L.lslam_pool_occgrid_from_scans(self.h, C.byref(laser), r.shape[0], r.ctypes.data, r.shape[1],
                                                         p.ctypes.data, float(resolution), C.byref(h)))
        dims, off, res = np.zeros(2, np.int32), np.zeros(2), C.c_double()
        self.L.lslam_occgrid_info(h, dims.ctypes.data, off.ctypes.data, C.byref(res))
        out = np.zeros((int(dims[1]), int(dims[0])), dtype=np.uint8)
        rc = self.L.lslam_occgrid_read_u8(h, out.ctypes.data)
        self.L.lslam_occgrid_destroy(h)
        self._check(rc)
        return out, off

    def match_batch(self, ranges, sensor_poses, doPenalize: bool = True, doRefineMatch: bool = True) -> np.ndarray:
        r, p = _f64(ranges), _f64(sensor_poses)
        r = r.reshape(-1, r.shape[-1])
        out = np.zeros(r.shape[0], dtype=RESULT_DTYPE)
        self._check(self.L.lslam_pool_match_batch(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data,
                                                  int(doPenalize), int(doRefineMatch), out.ctypes.data))
        return out


class FrontEnd:
    """karto::Mapper::Process with every processed scan resident in HBM: running-window match, AddEdges with
    LinkNearChains, TryCloseLoop (config=frontend_config(...)); without `config` the round-1 four-parameter form
    (library defaults for the graph side, loop closing off)."""

    def __init__(self, matcher: ScanMatcher, scan_buffer_size=70, scan_buffer_max_distance=20.0,
                 min_travel_distance=0.2, min_travel_heading=math.radians(10.0), config: FrontEndConfig | None = None):
        self.m, self.ctx, self.L = matcher, matcher.ctx, matcher.L
        h = C.c_void_p()
        if config is not None:
            self.ctx.check(self.L.lslam_frontend_create_ex(matcher.h, C.byref(config), C.byref(h)))
        else:
            self.ctx.check(self.L.lslam_frontend_create(matcher.h, scan_buffer_size, scan_buffer_max_distance,
                                                        min_travel_distance, min_travel_heading, C.byref(h)))
        self.h = h
        matcher._frontends.add(self)

    def close(self):
        if getattr(self, "h", None):
            self.L.lslam_frontend_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the order of finalisers is arbitrary (a context may already be gone under a handle
        # that points into it) and the process is exiting anyway: release explicitly, or not at all
        try:
            if sys.is_finalizing():
                return
            self.close()
        except Exception:  # incl. module globals already torn down (sys is None) late in shutdown
            pass

    def Process(self, ranges, odom_pose, time_s: float = 0.0):
        """-> (processed, corrected robot pose, covariance 3x3, response)"""
        r, o = _f64(ranges), _f64(odom_pose)
        ok, resp = C.c_int(), C.c_double()
        pose, cov = np.zeros(3), np.zeros(9)
        self.ctx.check(self.L.lslam_frontend_process_stamped(self.h, _ptr(r), r.shape[0], _ptr(o), time_s,
                                                             C.byref(ok), _ptr(pose), _ptr(cov), C.byref(resp)))
        return bool(ok.value), pose, cov.reshape(3, 3), resp.value

    def ProcessMany(self, ranges, odom_poses, times_s=None):
        """Mapper::Process over scans already at hand, with one scan of look-ahead (the loop search of scan t and the
        running-window match of scan t + 1 share the device).  -> (processed [n] bool, corrected robot poses [n, 3],
        covariances [n, 3, 3], responses [n]); identical to n Process calls."""
        r = _f64(ranges)
        r = r.reshape(-1, r.shape[-1])
        o = _f64(odom_poses).reshape(-1, 3)
        n = r.shape[0]
        assert o.shape[0] == n
        t = None if times_s is None else _f64(times_s)
        ok = np.zeros(n, dtype=np.int32)
        pose, cov, resp = np.zeros((n, 3)), np.zeros((n, 9)), np.zeros(n)
        self.ctx.check(self.L.lslam_frontend_process_many(self.h, n, r.ctypes.data, r.shape[1], o.ctypes.data,
                                                          None if t is None else t.ctypes.data, ok.ctypes.data, pose.ctypes.data,
                                                          cov.ctypes.data, resp.ctypes.data))
        return ok.astype(bool), pose, cov.reshape(n, 3, 3), resp

    def anchor_row(self, scan_id: int) -> np.ndarray:
        """FindValidPoints' anchors of a resident scan at its current pose (beam indices in order)."""
        out = np.zeros(self.m.num_beams + 1, dtype=np.int32)
        self.ctx.check(self.L.lslam_debug_frontend_anchor_row(self.h, int(scan_id), out.ctypes.data))
        return out[1:1 + out[0]].copy()

    def spec_chain_stats(self) -> dict:
        """Speculative anchor chains: world-point refreshes handed one / of those, the ones that recomputed the chain."""
        out = np.zeros(2, dtype=np.int64)
        self.ctx.check(self.L.lslam_frontend_spec_chain_stats(self.h, out.ctypes.data))
        return {"handed": int(out[0]), "recomputed": int(out[1])}

    def lookahead_stats(self) -> dict:
        out = np.zeros(3, dtype=np.int64)
        self.ctx.check(self.L.lslam_frontend_lookahead_stats(self.h, out.ctypes.data))
        return {"started": int(out[0]), "accepted": int(out[1]), "discarded": int(out[2])}

    def num_scans(self) -> int:
        return self.L.lslam_frontend_num_scans(self.h)

    def scan_pose(self, scan_id: int) -> np.ndarray:
        out = np.zeros(3)
        self.ctx.check(self.L.lslam_frontend_scan_pose(self.h, scan_id, out.ctypes.data))
        return out

    def stats(self) -> dict:
        out = np.zeros(6, dtype=np.int64)
        self.ctx.check(self.L.lslam_frontend_stats(self.h, out.ctypes.data))
        return dict(zip(("scans", "edges", "chain_matches", "loop_coarse_matches", "loop_fine_matches", "loops_closed"),
                        (int(v) for v in out)))

    def running_scans(self) -> int:
        return self.L.lslam_frontend_running_scans(self.h)

    def reset(self):
        self.ctx.check(self.L.lslam_frontend_reset(self.h))


class OccupancyGrid:
    """karto::OccupancyGrid (hit/pass counters) built on the GPU from scans at sensor poses."""

    def __init__(self, ctx: Context, h):
        self.ctx, self.L, self.h = ctx, ctx.L, h
        ctx._adopt(self)

    @classmethod
    def CreateFromScans(cls, ctx: Context, laser: LaserParams, ranges, sensor_poses, resolution: float):
        """OccupancyGrid::CreateFromScans; None where the reference returns NULL (no scans)."""
        r, p = _f64(ranges), _f64(sensor_poses)
        r = r.reshape(-1, r.shape[-1]) if r.size else r.reshape(0, 1)
        if r.shape[0] == 0:
            return None
        h = C.c_void_p()
        ctx.check(ctx.L.lslam_occgrid_create_from_scans(ctx.h, C.byref(laser), r.shape[0], r.ctypes.data, r.shape[1],
                                                        p.ctypes.data, resolution, C.byref(h)))
        return cls(ctx, h)

    # ---- sharded build: boxes merge by min/max, counters by integer addition, both exact ----
    @staticmethod
    def scan_bounds(ctx: Context, laser: LaserParams, ranges, sensor_poses) -> np.ndarray:
        """Box (minx, miny, maxx, maxy) ComputeDimensions derives from THESE scans; no scans -> the identity box."""
        r, p = _f64(ranges), _f64(sensor_poses)
        r = r.reshape(-1, r.shape[-1]) if r.size else r.reshape(0, 1)
        box = np.zeros(4)
        ctx.check(ctx.L.lslam_occgrid_scan_bounds(ctx.h, C.byref(laser), r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data,
                                                  box.ctypes.data))
        return box

    @classmethod
    def CreatePartial(cls, ctx: Context, laser: LaserParams, ranges, sensor_poses, resolution: float, box):
        """Counters of these scans (possibly none) on the grid of `box`, the merged box of all shards."""
        r, p, b = _f64(ranges), _f64(sensor_poses), _f64(box)
        r = r.reshape(-1, r.shape[-1]) if r.size else r.reshape(0, 1)
        h = C.c_void_p()
        ctx.check(ctx.L.lslam_occgrid_create_partial(ctx.h, C.byref(laser), r.shape[0], r.ctypes.data, r.shape[1],
                                                     p.ctypes.data, resolution, b.ctypes.data, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def CreateSharded(cls, ctx: Context, laser: LaserParams, ranges, sensor_poses, resolution: float, nccl_comm: int):
        """This rank's scans in, the grid of ALL ranks' scans out: box all-reduce + ONE counter all-reduce, RCCL called by
        the library itself on the context stream (lslam_occgrid_create_sharded).  nccl_comm: an ncclComm_t handle."""
        r, p = _f64(ranges), _f64(sensor_poses)
        r = r.reshape(-1, r.shape[-1]) if r.size else r.reshape(0, 1)
        h = C.c_void_p()
        ctx.check(ctx.L.lslam_occgrid_create_sharded(ctx.h, C.byref(laser), r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data,
                                                     float(resolution), C.c_void_p(nccl_comm), C.byref(h)))
        return cls(ctx, h)

    def counter_words(self) -> int:
        n = C.c_size_t()
        self.ctx.check(self.L.lslam_occgrid_counter_words(self.h, C.byref(n)))
        return int(n.value)

    def export_counters(self) -> np.ndarray:
        """uint32 [2, stride*height]: pass plane, hit plane."""
        out = np.zeros(self.counter_words(), dtype=np.uint32)
        self.ctx.check(self.L.lslam_occgrid_export_counters(self.h, out.ctypes.data, 0))
        return out.reshape(2, -1)

    def import_counters(self, counters, accumulate: bool = False):
        c = np.ascontiguousarray(counters, dtype=np.uint32).reshape(-1)
        if c.size != self.counter_words():
            raise ValueError(f"expected {self.counter_words()} counter words, got {c.size}")
        self.ctx.check(self.L.lslam_occgrid_import_counters(self.h, c.ctypes.data, 0, int(accumulate)))

    def export_counters_dev(self, dev_ptr: int):
        self.ctx.check(self.L.lslam_occgrid_export_counters(self.h, C.c_void_p(dev_ptr), 1))

    def import_counters_dev(self, dev_ptr: int, accumulate: bool = False):
        self.ctx.check(self.L.lslam_occgrid_import_counters(self.h, C.c_void_p(dev_ptr), 1, int(accumulate)))

    def close(self):
        if getattr(self, "h", None):
            self.L.lslam_occgrid_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the order of finalisers is arbitrary (a context may already be gone under a handle
        # that points into it) and the process is exiting anyway: release explicitly, or not at all
        try:
            if sys.is_finalizing():
                return
            self.close()
        except Exception:  # incl. module globals already torn down (sys is None) late in shutdown
            pass

    def info(self):
        d, off, res = np.zeros(2, dtype=np.int32), np.zeros(2), C.c_double()
        self.ctx.check(self.L.lslam_occgrid_info(self.h, d.ctypes.data, off.ctypes.data, C.byref(res)))
        return int(d[0]), int(d[1]), off, res.value

    def data(self) -> np.ndarray:
        w, h, _, _ = self.info()
        out = np.zeros((h, w), dtype=np.uint8)
        self.ctx.check(self.L.lslam_occgrid_read_u8(self.h, out.ctypes.data))
        return out

    def ros_data(self) -> np.ndarray:
        w, h, _, _ = self.info()
        out = np.zeros((h, w), dtype=np.int8)
        self.ctx.check(self.L.lslam_occgrid_read_ros_i8(self.h, out.ctypes.data))
        return out


class OccGridMap:
    """hectorslam log-odds occupancy grid pyramid on the GPU (OccGridMapBase / MapRepMultiMap)."""

    def __init__(self, ctx: Context, size_x: int, size_y: int, cell_length: float, offset=(0.0, 0.0),
                 levels: int = 1):
        self.ctx, self.L = ctx, ctx.L
        h = C.c_void_p()
        ctx.check(self.L.lslam_map_create(ctx.h, size_x, size_y, cell_length, offset[0], offset[1], levels, C.byref(h)))
        self.h = h
        ctx._adopt(self)

    def close(self):
        if getattr(self, "h", None):
            self.L.lslam_map_destroy(self.h)
            self.h = None

    def __del__(self):
        # at interpreter shutdown the order of finalisers is arbitrary (a context may already be gone under a handle
        # that points into it) and the process is exiting anyway: release explicitly, or not at all
        try:
            if sys.is_finalizing():
                return
            self.close()
        except Exception:  # incl. module globals already torn down (sys is None) late in shutdown
            pass

    def reset(self):
        self.ctx.check(self.L.lslam_map_reset(self.h))

    def setUpdateFreeFactor(self, p: float):
        self.ctx.check(self.L.lslam_map_set_update_factor_free(self.h, p))

    def setUpdateOccupiedFactor(self, p: float):
        self.ctx.check(self.L.lslam_map_set_update_factor_occupied(self.h, p))

    @property
    def levels(self) -> int:
        return self.L.lslam_map_levels(self.h)

    def size(self, level: int = 0):
        sx, sy = C.c_int(), C.c_int()
        self.ctx.check(self.L.lslam_map_size(self.h, level, C.byref(sx), C.byref(sy)))
        return sx.value, sy.value

    def getScaleToMap(self, level: int = 0) -> float:
        return self.L.lslam_map_scale_to_map(self.h, level)

    def updateByScan(self, points_xy, origo_xy, robot_pose_world):
        """MapRepMultiMap::updateByScan.  Level 0 takes this container; the levels above 0 take the container cached
        by the LAST matchData call (the reference's dataContainers: empty before the first matchData, stale if a
        different scan was matched).  At most 65536 points (LslamError beyond; the reference has no limit); NaN / Inf
        / out-of-int-range points are dropped like on the reference's x86 build.  Asynchronous: enqueued when this
        returns, readers are ordered after it."""
        p = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        o = np.ascontiguousarray(origo_xy, dtype=np.float32)
        w = np.ascontiguousarray(robot_pose_world, dtype=np.float32)
        self.ctx.check(self.L.lslam_map_update_by_scan(self.h, _ptr(p), p.shape[0], _ptr(o), _ptr(w)))

    def updateByScan_dev(self, points_ptr: int, n: int, origo_xy, robot_pose_world):
        o = np.ascontiguousarray(origo_xy, dtype=np.float32)
        w = np.ascontiguousarray(robot_pose_world, dtype=np.float32)
        self.ctx.check(self.L.lslam_map_update_by_scan_dev(self.h, points_ptr, n, o.ctypes.data, w.ctypes.data))

    def setScan(self, ranges_f32, scan: HectorScan) -> int:
        """LaserScan -> DataContainer on the device (hector_slam.cc:193, 320-362); returns the container size."""
        r = np.ascontiguousarray(ranges_f32, dtype=np.float32)
        n = C.c_int32()
        self.ctx.check(self.L.lslam_map_set_scan(self.h, r.ctypes.data, r.shape[0], C.byref(scan), C.byref(n)))
        return n.value

    def container(self):
        """-> (points [n,2] float32, origo [2]) of the resident container."""
        origo = np.zeros(2, np.float32)
        n = self.L.lslam_map_read_container(self.h, None, 0, origo.ctypes.data)
        pts = np.zeros((max(n, 0), 2), np.float32)
        if n > 0:
            self.L.lslam_map_read_container(self.h, pts.ctypes.data, n, origo.ctypes.data)
        return pts, origo

    def matchContainer(self, begin_estimate_world):
        b = np.ascontiguousarray(begin_estimate_world, dtype=np.float32)
        pose, cov = np.zeros(3, dtype=np.float32), np.zeros(9, dtype=np.float32)
        self.ctx.check(self.L.lslam_map_match_container(self.h, b.ctypes.data, pose.ctypes.data, cov.ctypes.data))
        return pose, cov.reshape(3, 3)

    def updateByContainer(self, robot_pose_world):
        w = np.ascontiguousarray(robot_pose_world, dtype=np.float32)
        self.ctx.check(self.L.lslam_map_update_by_container(self.h, w.ctypes.data))

    def updateByScans(self, points_list, origos_xy, robot_poses_world):
        """Batched update: exactly len(points_list) successive updateByScan calls (every level fed the same scan),
        marked in parallel and applied in one pass over the map."""
        counts = np.array([len(p) for p in points_list], dtype=np.int32)
        pts = (np.concatenate([np.asarray(p, dtype=np.float32).reshape(-1, 2) for p in points_list])
               if len(points_list) else np.zeros((0, 2), np.float32))
        pts = np.ascontiguousarray(pts, dtype=np.float32)
        o = np.ascontiguousarray(np.broadcast_to(np.asarray(origos_xy, dtype=np.float32), (len(counts), 2)))
        w = np.ascontiguousarray(robot_poses_world, dtype=np.float32).reshape(len(counts), 3)
        self.ctx.check(self.L.lslam_map_update_batch(self.h, len(counts), pts.ctypes.data, counts.ctypes.data,
                                                     o.ctypes.data, w.ctypes.data))

    def updateByScans_dev(self, points_ptr: int, counts, origos_xy, robot_poses_world):
        counts = np.ascontiguousarray(counts, dtype=np.int32)
        o = np.ascontiguousarray(np.broadcast_to(np.asarray(origos_xy, dtype=np.float32), (len(counts), 2)))
        w = np.ascontiguousarray(robot_poses_world, dtype=np.float32).reshape(len(counts), 3)
        self.ctx.check(self.L.lslam_map_update_batch_dev(self.h, len(counts), points_ptr, counts.ctypes.data,
                                                         o.ctypes.data, w.ctypes.data))

    def updateByScanJustOnce(self, points_xy_m, origo_xy=(0.0, 0.0), begin=(800.0, 800.0), metres_per_cell=0.05):
        p = np.ascontiguousarray(points_xy_m, dtype=np.float32).reshape(-1, 2)
        o = np.ascontiguousarray(origo_xy, dtype=np.float32)
        self.ctx.check(self.L.lslam_map_update_just_once(self.h, p.ctypes.data, p.shape[0], o.ctypes.data,
                                                         begin[0], begin[1], metres_per_cell))

    def matchData(self, begin_estimate_world, points_xy, origo_xy=(0.0, 0.0)):
        """MapRepMultiMap::matchData -> (pose[3] float32, covMatrix 3x3 float32).  Like the reference
        (MapRepMultiMap.h:161) the container is cached: the NEXT updateByScan feeds the levels above 0
        from it, whatever container that call is handed."""
        p = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        o = np.ascontiguousarray(origo_xy, dtype=np.float32)
        b = np.ascontiguousarray(begin_estimate_world, dtype=np.float32)
        pose, cov = np.zeros(3, dtype=np.float32), np.zeros(9, dtype=np.float32)
        self.ctx.check(self.L.lslam_map_match_data(self.h, p.ctypes.data, p.shape[0], o.ctypes.data, b.ctypes.data,
                                                   pose.ctypes.data, cov.ctypes.data))
        return pose, cov.reshape(3, 3)

    def set_option(self, name: str, value: int):
        """'ordered_sums': matchData adds its sums in point order (bit-equal to the CPU restatement) instead of in parallel."""
        self.ctx.check(self.L.lslam_map_set_option(self.h, {"ordered_sums": 1, "batch_scratch_mb": 2, "batch_radius_cells": 3}[name],
                                                   int(value)))

    def batch_stats(self) -> dict:
        """Scratch of the batched update: bytes held, rounds of the last batch, cells outside their window (must be 0), budget."""
        out = (C.c_int64 * 4)()
        self.ctx.check(self.L.lslam_map_batch_stats(self.h, out))
        return {"scratch_bytes": int(out[0]), "rounds": int(out[1]), "window_misses": int(out[2]), "budget_bytes": int(out[3])}

    def cached_points(self) -> int:
        return self.L.lslam_map_cached_points(self.h)

    def logodds(self, level: int = 0) -> np.ndarray:
        sx, sy = self.size(level)
        out = np.zeros((sy, sx), dtype=np.float32)
        self.ctx.check(self.L.lslam_map_read_logodds(self.h, level, out.ctypes.data))
        return out

    def occupancy_i8(self, level: int = 0) -> np.ndarray:
        sx, sy = self.size(level)
        out = np.zeros((sy, sx), dtype=np.int8)
        self.ctx.check(self.L.lslam_map_read_occupancy_i8(self.h, level, out.ctypes.data))
        return out

    def cells_dev_ptr(self, level: int = 0) -> int:
        return self.L.lslam_map_cells_dev_ptr(self.h, level)
