"""TEST INFRASTRUCTURE ONLY: ctypes bindings of the CPU oracles.

* ``RefKarto``   -> oracle/_ref/libkarto_ref.so : the reference's own open_karto, compiled from
  /root/reference by oracle/Makefile (exists only where that build ran; travels to the GPU box).
* ``PortKarto``  -> oracle/libkarto_oracle.so   : plain-C restatement (oracle/karto_oracle.c).
* ``PortHector`` -> oracle/libhector_oracle.so  : plain-C restatement (oracle/hector_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import subprocess

import numpy as np

HERE = pathlib.Path(__file__).resolve().parent


class KrefCfg(C.Structure):
    _fields_ = [
        ("search_size", C.c_double),
        ("resolution", C.c_double),
        ("smear_deviation", C.c_double),
        ("coarse_angle_offset", C.c_double),
        ("coarse_angle_resolution", C.c_double),
        ("fine_angle_offset", C.c_double),
        ("distance_variance_penalty", C.c_double),
        ("angle_variance_penalty", C.c_double),
        ("minimum_distance_penalty", C.c_double),
        ("minimum_angle_penalty", C.c_double),
        ("use_response_expansion", C.c_int),
        ("scan_buffer_size", C.c_int),
        ("scan_buffer_max_scan_distance", C.c_double),
        ("minimum_travel_distance", C.c_double),
        ("minimum_travel_heading", C.c_double),
        ("do_loop_closing", C.c_int),
        ("loop_match_minimum_chain_size", C.c_int),
        ("link_match_minimum_response_fine", C.c_double),
        ("link_scan_maximum_distance", C.c_double),
        ("loop_search_maximum_distance", C.c_double),
        ("loop_match_maximum_variance_coarse", C.c_double),
        ("loop_match_minimum_response_coarse", C.c_double),
        ("loop_match_minimum_response_fine", C.c_double),
        ("loop_search_space_dimension", C.c_double),
        ("loop_search_space_resolution", C.c_double),
        ("loop_search_space_smear_deviation", C.c_double),
        ("minimum_time_interval", C.c_double),
        ("use_scan_barycenter", C.c_int),
        ("reserved", C.c_int),
    ]


class KrefLaser(C.Structure):
    _fields_ = [
        ("min_angle", C.c_double),
        ("max_angle", C.c_double),
        ("angular_resolution", C.c_double),
        ("min_range", C.c_double),
        ("max_range", C.c_double),
        ("range_threshold", C.c_double),
        ("offset_x", C.c_double),
        ("offset_y", C.c_double),
        ("offset_heading", C.c_double),
    ]


def default_cfg(**kw) -> KrefCfg:
    """BASELINE cfg 3/4 parameters (SURVEY.md §8 sizes table, column 1)."""
    d = dict(
        search_size=1.0,
        resolution=0.05,
        smear_deviation=0.03,
        coarse_angle_offset=0.349,
        coarse_angle_resolution=0.0349,
        fine_angle_offset=0.00349,
        distance_variance_penalty=0.3 * 0.3,  # library default Mapper.cpp:1608 (variance)
        angle_variance_penalty=np.deg2rad(20.0) ** 2,  # Mapper.cpp:1614
        minimum_distance_penalty=0.5,
        minimum_angle_penalty=0.9,
        use_response_expansion=0,
        scan_buffer_size=70,
        scan_buffer_max_scan_distance=20.0,
        minimum_travel_distance=0.2,
        minimum_travel_heading=np.deg2rad(10.0),
        # pose-graph side of Mapper::Process: library defaults (Mapper.cpp:1516-1604); loop closing is OFF unless asked
        # for (LinkNearChains is unconditional in AddEdges either way)
        do_loop_closing=0,
        loop_match_minimum_chain_size=10,
        link_match_minimum_response_fine=0.8,
        link_scan_maximum_distance=10.0,
        loop_search_maximum_distance=4.0,
        loop_match_maximum_variance_coarse=0.4 * 0.4,
        loop_match_minimum_response_coarse=0.8,
        loop_match_minimum_response_fine=0.8,
        loop_search_space_dimension=8.0,
        loop_search_space_resolution=0.05,
        loop_search_space_smear_deviation=0.03,
        minimum_time_interval=3600.0,
        use_scan_barycenter=1,
        reserved=0,
    )
    d.update(kw)
    return KrefCfg(**d)


def laser_struct(laser, range_threshold=49.5, offset=(0.0, 0.0, 0.0)) -> KrefLaser:
    return KrefLaser(
        laser.angle_min,
        laser.angle_max,
        laser.angle_increment,
        laser.range_min,
        laser.range_max,
        range_threshold,
        *offset,
    )


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def build(target: str = "all") -> None:
    """make -C oracle <target>; quiet no-op when everything is fresh."""
    subprocess.run(["make", "-s", "-C", str(HERE), target], check=True)


def have_ref() -> bool:
    return (HERE / "_ref" / "libkarto_ref.so").exists()


class RefKarto:
    """The reference's own karto::ScanMatcher / karto::Mapper behind oracle/ref_driver.cpp."""

    def __init__(self, cfg: KrefCfg, laser: KrefLaser, gpu: bool = False):
        """gpu=True loads oracle/_ref_gpu/libkarto_ref_gpu.so instead: the SAME reference objects (Karto.o, Mapper.o,
        this driver), with karto::ScanMatcher::MatchScan substituted at link time by integration/
        karto_scan_matcher_gpu.cpp -- the reference's Mapper / MapperGraph orchestrating, the HIP kernels matching."""
        path = HERE / "_ref_gpu" / "libkarto_ref_gpu.so" if gpu else HERE / "_ref" / "libkarto_ref.so"
        if not path.exists():
            raise FileNotFoundError(f"{path} (build it here with `make -C oracle {'ref_gpu' if gpu else 'ref'}`)")
        L = C.CDLL(str(path))
        self.gpu = gpu
        if gpu:
            L.lslam_karto_gpu_match_calls.restype = C.c_longlong
            L.lslam_karto_gpu_stats.argtypes = [C.c_void_p]
            L.lslam_karto_gpu_stats.restype = None
        L.kref_reset.argtypes = [C.c_void_p]
        L.kref_mapper_grid_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_create.restype = C.c_void_p
        L.kref_create.argtypes = [C.POINTER(KrefCfg), C.POINTER(KrefLaser)]
        L.kref_destroy.argtypes = [C.c_void_p]
        L.kref_num_beams.argtypes = [C.c_void_p]
        L.kref_match.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                 C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_match_timed.restype = C.c_double
        L.kref_match_timed.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.kref_grid_info.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_grid_copy.argtypes = [C.c_void_p, C.c_void_p]
        L.kref_kernel_copy.argtypes = [C.c_void_p, C.c_void_p]
        L.kref_table_dims.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_table_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_probs_copy.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_get_response.restype = C.c_double
        L.kref_get_response.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.kref_point_readings.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.kref_find_valid_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_process.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_running_scans.argtypes = [C.c_void_p]
        L.kref_graph_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.kref_graph_stats.restype = None
        L.kref_scan_pose.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.kref_occupancy_grid.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_last_error.restype = C.c_char_p
        L.kref_last_error.argtypes = [C.c_void_p]
        L.kref_set_base_scans.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.kref_match_fixed_grid.restype = C.c_double
        L.kref_match_fixed_grid.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_occgrid_from_scans.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
        L.kref_round.restype = C.c_double
        L.kref_round.argtypes = [C.c_double]
        self.L = L
        self.h = L.kref_create(C.byref(cfg), C.byref(laser))
        if not self.h:
            raise RuntimeError("kref_create failed (ScanMatcher::Create returned NULL or threw)")

    def close(self):
        if self.h:
            self.L.kref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gpu_match_calls(self) -> int:
        """MatchScan calls that ran on the device so far (process-wide; gpu=True only)."""
        return int(self.L.lslam_karto_gpu_match_calls()) if self.gpu else 0

    def gpu_stats(self) -> dict:
        """integration/karto_scan_matcher_gpu.cpp's counters (process-wide; gpu=True only)."""
        out = (C.c_longlong * 8)()
        if self.gpu:
            self.L.lslam_karto_gpu_stats(out)
        return dict(zip(("match_calls", "cached_calls", "matchers_alive", "resident_scans", "scans_uploaded", "refreshes",
                         "ns_in_match_scan", "ns_in_device_call"), (int(v) for v in out)))

    def reset(self):
        """Mapper::Reset (Mapper.cpp:1980-1992)."""
        if self.L.kref_reset(self.h) != 0:
            raise RuntimeError(self.L.kref_last_error(self.h).decode())

    def mapper_grid(self):
        """(bytes [h, stride], offset) of the Mapper's sequential matcher's correlation grid, or (None, None)."""
        info = (C.c_int * 8)()
        off = (C.c_double * 2)()
        self.L.kref_grid_info(self.h, info, off)  # same geometry as the stand-alone matcher (same parameters)
        g = np.zeros((info[1], info[2]), dtype=np.uint8)
        o = np.zeros(2)
        if not self.L.kref_mapper_grid_copy(self.h, g.ctypes.data, o.ctypes.data):
            return None, None
        return g, o

    @property
    def num_beams(self) -> int:
        return self.L.kref_num_beams(self.h)

    def match(self, base_ranges, base_poses, q_ranges, q_pose, do_penalize=True, do_refine=True):
        br = np.ascontiguousarray(base_ranges, dtype=np.float64)
        bp = np.ascontiguousarray(base_poses, dtype=np.float64)
        qr = np.ascontiguousarray(q_ranges, dtype=np.float64)
        qp = np.ascontiguousarray(q_pose, dtype=np.float64)
        pose = np.zeros(3)
        cov = np.zeros(9)
        resp = C.c_double()
        rc = self.L.kref_match(self.h, br.shape[0], br.ctypes.data, bp.ctypes.data, qr.shape[-1],
                               qr.ctypes.data, qp.ctypes.data, int(do_penalize), int(do_refine),
                               pose.ctypes.data, cov.ctypes.data, C.byref(resp))
        if rc != 0:
            raise RuntimeError(self.L.kref_last_error(self.h).decode())
        return pose, cov.reshape(3, 3), resp.value

    def match_timed(self, base_ranges, base_poses, q_ranges, q_poses):
        br = np.ascontiguousarray(base_ranges, dtype=np.float64)
        bp = np.ascontiguousarray(base_poses, dtype=np.float64)
        qr = np.ascontiguousarray(q_ranges, dtype=np.float64)
        qp = np.ascontiguousarray(q_poses, dtype=np.float64)
        n = qr.shape[0]
        poses = np.zeros((n, 3))
        resp = np.zeros(n)
        sec = self.L.kref_match_timed(self.h, br.shape[0], br.ctypes.data, bp.ctypes.data,
                                      qr.shape[1], qr.ctypes.data, qp.ctypes.data, n,
                                      poses.ctypes.data, resp.ctypes.data)
        return sec, poses, resp

    def set_base_scans(self, base_ranges, base_poses, center_pose):
        """shared-grid mode: the reference's AddScans around an explicit centre pose."""
        br = np.ascontiguousarray(base_ranges, dtype=np.float64)
        bp = np.ascontiguousarray(base_poses, dtype=np.float64)
        c = np.ascontiguousarray(center_pose, dtype=np.float64)
        rc = self.L.kref_set_base_scans(self.h, br.shape[0], br.ctypes.data, bp.ctypes.data, br.shape[1], c.ctypes.data)
        if rc != 0:
            raise RuntimeError(self.L.kref_last_error(self.h).decode())

    def match_fixed_grid(self, q_ranges, q_poses, do_penalize=True, do_refine=True):
        """coarse+fine CorrelateScan of n scans vs the current grid -> (sec/scan, poses, covs, resp)."""
        qr = np.ascontiguousarray(q_ranges, dtype=np.float64).reshape(-1, np.shape(q_ranges)[-1])
        qp = np.ascontiguousarray(q_poses, dtype=np.float64).reshape(-1, 3)
        n = qr.shape[0]
        poses, covs, resp = np.zeros((n, 3)), np.zeros((n, 3, 3)), np.zeros(n)
        sec = self.L.kref_match_fixed_grid(self.h, qr.shape[1], qr.ctypes.data, qp.ctypes.data, n, int(do_penalize),
                                           int(do_refine), poses.ctypes.data, covs.ctypes.data, resp.ctypes.data)
        if sec < 0:
            raise RuntimeError("reference threw inside CorrelateScan")
        return sec, poses, covs, resp

    def grid_info(self):
        i = np.zeros(8, dtype=np.int32)
        off = np.zeros(2)
        self.L.kref_grid_info(self.h, i.ctypes.data, off.ctypes.data)
        keys = ("width", "height", "stride", "roi_x", "roi_y", "roi_w", "roi_h", "kernel_size")
        d = {k: int(v) for k, v in zip(keys, i)}
        d["offset"] = off
        return d

    def grid(self) -> np.ndarray:
        gi = self.grid_info()
        out = np.zeros((gi["height"], gi["stride"]), dtype=np.uint8)
        self.L.kref_grid_copy(self.h, out.ctypes.data)
        return out

    def kernel(self) -> np.ndarray:
        k = self.grid_info()["kernel_size"]
        out = np.zeros((k, k), dtype=np.uint8)
        self.L.kref_kernel_copy(self.h, out.ctypes.data)
        return out

    def tables(self):
        na, npnt = C.c_int(), C.c_int()
        self.L.kref_table_dims(self.h, C.byref(na), C.byref(npnt))
        out = np.zeros((na.value, npnt.value), dtype=np.int32)
        ang = np.zeros(na.value)
        self.L.kref_table_copy(self.h, out.ctypes.data, ang.ctypes.data)
        return out, ang

    def probs(self) -> np.ndarray:
        d = np.zeros(3, dtype=np.int32)
        self.L.kref_probs_copy(self.h, d.ctypes.data, None)
        out = np.zeros((d[1], d[2]))
        self.L.kref_probs_copy(self.h, d.ctypes.data, out.ctypes.data)
        return out[:, : d[0]]

    def get_response(self, angle_index: int, grid_index: int) -> float:
        return self.L.kref_get_response(self.h, angle_index, grid_index)

    def point_readings(self, ranges, pose, filtered=False) -> np.ndarray:
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(pose, dtype=np.float64)
        out = np.zeros((r.shape[0], 2))
        n = self.L.kref_point_readings(self.h, r.shape[0], r.ctypes.data, p.ctypes.data,
                                       int(filtered), out.ctypes.data)
        return out[:n]

    def find_valid_points(self, ranges, pose, viewpoint) -> np.ndarray:
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(pose, dtype=np.float64)
        v = np.ascontiguousarray(viewpoint, dtype=np.float64)
        out = np.zeros((r.shape[0], 2))
        n = self.L.kref_find_valid_points(self.h, r.shape[0], r.ctypes.data, p.ctypes.data,
                                          v.ctypes.data, out.ctypes.data)
        return out[:n]

    def process(self, ranges, odom_pose):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(odom_pose, dtype=np.float64)
        out = np.zeros(3)
        rc = self.L.kref_process(self.h, r.shape[0], r.ctypes.data, p.ctypes.data, out.ctypes.data, None)
        if rc < 0:
            raise RuntimeError(self.L.kref_last_error(self.h).decode())
        return bool(rc), out

    def running_scans(self) -> int:
        return self.L.kref_running_scans(self.h)

    def graph_stats(self):
        """(vertices, edges) of the reference's pose graph."""
        out = np.zeros(2, dtype=np.int32)
        self.L.kref_graph_stats(self.h, out.ctypes.data)
        return int(out[0]), int(out[1])

    def scan_pose(self, scan_id: int) -> np.ndarray:
        out = np.zeros(3)
        if self.L.kref_scan_pose(self.h, scan_id, out.ctypes.data) != 0:
            raise IndexError(scan_id)
        return out

    def occupancy_grid(self, resolution: float):
        d = np.zeros(2, dtype=np.int32)
        off = np.zeros(2)
        if self.L.kref_occupancy_grid(self.h, resolution, d.ctypes.data, off.ctypes.data, None) != 0:
            return None, None
        out = np.zeros((d[1], d[0]), dtype=np.uint8)
        self.L.kref_occupancy_grid(self.h, resolution, d.ctypes.data, off.ctypes.data, out.ctypes.data)
        return out, off

    def occgrid_from_scans(self, ranges, poses, resolution):
        """OccupancyGrid::CreateFromScans of explicit scans at ROBOT poses -> (grid[h,w] u8, offset)."""
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(poses, dtype=np.float64)
        d, off = np.zeros(2, dtype=np.int32), np.zeros(2)
        rc = self.L.kref_occgrid_from_scans(self.h, r.shape[0], r.ctypes.data, p.ctypes.data, r.shape[1], resolution,
                                            d.ctypes.data, off.ctypes.data, None)
        if rc != 0:
            return None, None
        out = np.zeros((d[1], d[0]), dtype=np.uint8)
        self.L.kref_occgrid_from_scans(self.h, r.shape[0], r.ctypes.data, p.ctypes.data, r.shape[1], resolution,
                                       d.ctypes.data, off.ctypes.data, out.ctypes.data)
        return out, off

    def round(self, v: float) -> float:
        return self.L.kref_round(v)


class KorConfig(C.Structure):
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


def kor_config_from(cfg: KrefCfg, range_threshold: float) -> KorConfig:
    return KorConfig(cfg.search_size, cfg.resolution, cfg.smear_deviation, range_threshold,
                     cfg.coarse_angle_offset, cfg.coarse_angle_resolution, cfg.fine_angle_offset,
                     cfg.distance_variance_penalty, cfg.angle_variance_penalty,
                     cfg.minimum_distance_penalty, cfg.minimum_angle_penalty,
                     cfg.use_response_expansion, 0)


class PortKarto:
    """oracle/karto_oracle.c (plain-C restatement) -- sensor poses in, sensor poses out."""

    def __init__(self, cfg: KrefCfg, laser: KrefLaser):
        path = HERE / "libkarto_oracle.so"
        if not path.exists():
            build("libkarto_oracle.so")
        L = C.CDLL(str(path))
        vp = C.c_void_p
        L.kor_create.restype = vp
        L.kor_create.argtypes = [C.POINTER(KorConfig), C.POINTER(KrefLaser)]
        L.kor_destroy.argtypes = [vp]
        L.kor_num_beams.argtypes = [vp]
        L.kor_grid_info.argtypes = [vp, vp, vp]
        L.kor_grid_data.restype = vp
        L.kor_grid_data.argtypes = [vp]
        L.kor_kernel_data.restype = vp
        L.kor_kernel_data.argtypes = [vp]
        L.kor_sensor_pose_from_robot.argtypes = [vp, vp, vp]
        L.kor_robot_pose_from_sensor.argtypes = [vp, vp, vp]
        L.kor_point_readings.argtypes = [vp, vp, vp, vp]
        L.kor_find_valid_points.argtypes = [vp, C.c_int, vp, vp]
        L.kor_set_base_scans.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
        L.kor_set_grid.argtypes = [vp, vp, vp]
        L.kor_compute_offsets.argtypes = [vp, vp, vp, C.c_double, C.c_double, C.c_double, vp]
        L.kor_response_sum.restype = C.c_int64
        L.kor_response_sum.argtypes = [vp, vp, C.c_int32]
        L.kor_correlate_scan.restype = C.c_double
        L.kor_correlate_scan.argtypes = [vp, vp, vp, vp] + [C.c_double] * 6 + [C.c_int, C.c_int, vp, vp, vp, vp]
        L.kor_match.restype = C.c_double
        L.kor_match.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
        L.kor_match_scan.restype = C.c_double
        L.kor_match_scan.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
        L.kor_probs.argtypes = [vp, vp]
        L.kor_occgrid_from_scans.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_double, vp, vp, vp]
        L.kor_occgrid_bounds.restype = None
        L.kor_occgrid_bounds.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp]
        L.kor_occgrid_partial.argtypes = [vp, C.c_int, vp, C.c_int, vp, C.c_double, vp, vp, vp]
        L.kor_occgrid_update.restype = None
        L.kor_occgrid_update.argtypes = [vp, vp, vp]
        L.kor_frontend_create.restype = vp
        L.kor_frontend_create.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double]
        L.kor_frontend_destroy.argtypes = [vp]
        L.kor_frontend_process.argtypes = [vp, vp, vp, vp, vp, vp]
        L.kor_frontend_running_scans.argtypes = [vp]
        self.L = L
        self.cfg = cfg
        kc = kor_config_from(cfg, laser.range_threshold)
        self.h = L.kor_create(C.byref(kc), C.byref(laser))
        if not self.h:
            raise ValueError("kor_create: invalid parameters (ScanMatcher::Create -> NULL)")
        self.fe = None

    def close(self):
        if getattr(self, "fe", None):
            self.L.kor_frontend_destroy(self.fe)
            self.fe = None
        if getattr(self, "h", None):
            self.L.kor_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def num_beams(self) -> int:
        return self.L.kor_num_beams(self.h)

    def grid_info(self):
        i = np.zeros(8, dtype=np.int32)
        off = np.zeros(2)
        self.L.kor_grid_info(self.h, i.ctypes.data, off.ctypes.data)
        keys = ("width", "height", "stride", "roi_x", "roi_y", "roi_w", "roi_h", "kernel_size")
        d = {k: int(v) for k, v in zip(keys, i)}
        d["offset"] = off
        return d

    def grid(self) -> np.ndarray:
        gi = self.grid_info()
        n = gi["height"] * gi["stride"]
        buf = (C.c_uint8 * n).from_address(self.L.kor_grid_data(self.h))
        return np.frombuffer(buf, dtype=np.uint8).reshape(gi["height"], gi["stride"]).copy()

    def kernel(self) -> np.ndarray:
        k = self.grid_info()["kernel_size"]
        buf = (C.c_uint8 * (k * k)).from_address(self.L.kor_kernel_data(self.h))
        return np.frombuffer(buf, dtype=np.uint8).reshape(k, k).copy()

    def sensor_pose_from_robot(self, robot):
        r = np.ascontiguousarray(robot, dtype=np.float64)
        out = np.zeros(3)
        self.L.kor_sensor_pose_from_robot(self.h, r.ctypes.data, out.ctypes.data)
        return out

    def robot_pose_from_sensor(self, sensor):
        s = np.ascontiguousarray(sensor, dtype=np.float64)
        out = np.zeros(3)
        self.L.kor_robot_pose_from_sensor(self.h, s.ctypes.data, out.ctypes.data)
        return out

    def point_readings(self, ranges, sensor_pose) -> np.ndarray:
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(sensor_pose, dtype=np.float64)
        out = np.zeros((self.num_beams, 2))
        self.L.kor_point_readings(self.h, r.ctypes.data, p.ctypes.data, out.ctypes.data)
        return out

    def find_valid_points(self, pts, viewpoint) -> np.ndarray:
        p = np.ascontiguousarray(pts, dtype=np.float64)
        v = np.ascontiguousarray(viewpoint, dtype=np.float64)
        out = np.zeros_like(p)
        n = self.L.kor_find_valid_points(p.ctypes.data, p.shape[0], v.ctypes.data, out.ctypes.data)
        return out[:n]

    def set_base_scans(self, ranges, sensor_poses, center_pose):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(sensor_poses, dtype=np.float64)
        c = np.ascontiguousarray(center_pose, dtype=np.float64)
        self.L.kor_set_base_scans(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data, c.ctypes.data)

    def set_grid(self, grid, offset):
        g = np.ascontiguousarray(grid, dtype=np.uint8)
        o = np.ascontiguousarray(offset, dtype=np.float64)
        self.L.kor_set_grid(self.h, g.ctypes.data, o.ctypes.data)

    def compute_offsets(self, ranges, sensor_pose, angle_center, angle_offset, angle_resolution):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(sensor_pose, dtype=np.float64)
        na = self.L.kor_compute_offsets(self.h, r.ctypes.data, p.ctypes.data, angle_center,
                                        angle_offset, angle_resolution, None)
        out = np.zeros((na, self.num_beams), dtype=np.int32)
        self.L.kor_compute_offsets(self.h, r.ctypes.data, p.ctypes.data, angle_center, angle_offset,
                                   angle_resolution, out.ctypes.data)
        return out

    def response_sum(self, table_row, pos) -> int:
        t = np.ascontiguousarray(table_row, dtype=np.int32)
        return self.L.kor_response_sum(self.h, t.ctypes.data, int(pos))

    def correlate_scan(self, ranges, sensor_pose, center, off, res, ang_off, ang_res, do_penalize,
                       doing_fine, want_sums=False):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(sensor_pose, dtype=np.float64)
        c = np.ascontiguousarray(center, dtype=np.float64)
        mean, cov, st = np.zeros(3), np.zeros(9), C.c_int()
        sums = None
        if want_sums:
            nx = int(np.floor(off * 2.0 / res + 0.5) + 1)
            na = int(np.floor(ang_off * 2.0 / ang_res + 0.5) + 1)
            sums = np.zeros((nx, nx, na), dtype=np.int32)
        resp = self.L.kor_correlate_scan(self.h, r.ctypes.data, p.ctypes.data, c.ctypes.data, off, off,
                                         res, res, ang_off, ang_res, int(do_penalize), int(doing_fine),
                                         mean.ctypes.data, cov.ctypes.data,
                                         sums.ctypes.data if sums is not None else None, C.byref(st))
        return resp, mean, cov.reshape(3, 3), st.value, sums

    def match(self, ranges, sensor_pose, do_penalize=True, do_refine=True):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(sensor_pose, dtype=np.float64)
        mean, cov, st = np.zeros(3), np.zeros(9), C.c_int()
        resp = self.L.kor_match(self.h, r.ctypes.data, p.ctypes.data, int(do_penalize), int(do_refine),
                                mean.ctypes.data, cov.ctypes.data, C.byref(st))
        if st.value:
            raise RuntimeError(f"kor_match status {st.value}")
        return mean, cov.reshape(3, 3), resp

    def match_scan(self, base_ranges, base_poses, q_ranges, q_pose, do_penalize=True, do_refine=True):
        br = np.ascontiguousarray(base_ranges, dtype=np.float64)
        bp = np.ascontiguousarray(base_poses, dtype=np.float64)
        qr = np.ascontiguousarray(q_ranges, dtype=np.float64)
        qp = np.ascontiguousarray(q_pose, dtype=np.float64)
        mean, cov, st = np.zeros(3), np.zeros(9), C.c_int()
        resp = self.L.kor_match_scan(self.h, br.shape[0], br.ctypes.data, br.shape[1] if br.ndim == 2 else 0,
                                     bp.ctypes.data, qr.ctypes.data, qp.ctypes.data, int(do_penalize),
                                     int(do_refine), mean.ctypes.data, cov.ctypes.data, C.byref(st))
        if st.value:
            raise RuntimeError(f"kor_match_scan status {st.value}")
        return mean, cov.reshape(3, 3), resp

    def probs(self) -> np.ndarray:
        side = int(np.floor(self.cfg.search_size / self.cfg.resolution + 0.5) + 1)
        out = np.zeros((side, side))
        self.L.kor_probs(self.h, out.ctypes.data)
        return out

    def occgrid_from_scans(self, ranges, sensor_poses, resolution):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(sensor_poses, dtype=np.float64)
        d, off = np.zeros(2, dtype=np.int32), np.zeros(2)
        rc = self.L.kor_occgrid_from_scans(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data, resolution,
                                           d.ctypes.data, off.ctypes.data, None)
        if rc != 0:
            return None, None
        out = np.zeros((d[1], d[0]), dtype=np.uint8)
        self.L.kor_occgrid_from_scans(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data, resolution,
                                      d.ctypes.data, off.ctypes.data, out.ctypes.data)
        return out, off

    # the same build in shardable pieces (stand-in for the device path in the CPU multi-rank tests)
    def occgrid_bounds(self, ranges, sensor_poses):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        r = r.reshape(-1, r.shape[-1])
        p = np.ascontiguousarray(sensor_poses, dtype=np.float64)
        box = np.zeros(4)
        self.L.kor_occgrid_bounds(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data, box.ctypes.data)
        return box

    def occgrid_partial(self, ranges, sensor_poses, resolution, box):
        """-> (dims (w, h, stride), counters uint32 [2, h*stride]: pass plane, hit plane)"""
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        r = r.reshape(-1, r.shape[-1])
        p = np.ascontiguousarray(sensor_poses, dtype=np.float64)
        b = np.ascontiguousarray(box, dtype=np.float64)
        d = np.zeros(3, dtype=np.int32)
        self.L.kor_occgrid_partial(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data, resolution,
                                   b.ctypes.data, d.ctypes.data, None)
        cnt = np.zeros((2, max(int(d[1]), 0) * int(d[2])), dtype=np.uint32)
        self.L.kor_occgrid_partial(self.h, r.shape[0], r.ctypes.data, r.shape[1], p.ctypes.data, resolution,
                                   b.ctypes.data, d.ctypes.data, cnt.ctypes.data)
        return d, cnt

    def occgrid_update(self, dims, counters):
        d = np.ascontiguousarray(dims, dtype=np.int32)
        c = np.ascontiguousarray(counters, dtype=np.uint32)
        out = np.zeros((max(int(d[1]), 0), max(int(d[0]), 0)), dtype=np.uint8)
        self.L.kor_occgrid_update(d.ctypes.data, c.ctypes.data, out.ctypes.data)
        return out

    # streaming front-end
    def frontend(self):
        c = self.cfg
        self.fe = self.L.kor_frontend_create(self.h, c.scan_buffer_size, c.scan_buffer_max_scan_distance,
                                             c.minimum_travel_distance, c.minimum_travel_heading)

    def process(self, ranges, odom_pose):
        r = np.ascontiguousarray(ranges, dtype=np.float64)
        p = np.ascontiguousarray(odom_pose, dtype=np.float64)
        out, cov, resp = np.zeros(3), np.zeros(9), C.c_double()
        rc = self.L.kor_frontend_process(self.fe, r.ctypes.data, p.ctypes.data, out.ctypes.data,
                                         cov.ctypes.data, C.byref(resp))
        if rc < 0:
            raise RuntimeError(f"kor_frontend_process status {rc}")
        return bool(rc), out, cov.reshape(3, 3), resp.value

    def running_scans(self) -> int:
        return self.L.kor_frontend_running_scans(self.fe)


class PortHector:
    """oracle/hector_oracle.c: one level of the reference's log-odds map."""

    _L = None

    @classmethod
    def lib(cls):
        if cls._L is None:
            path = HERE / "libhector_oracle.so"
            if not path.exists():
                build("libhector_oracle.so")
            L = C.CDLL(str(path))
            vp = C.c_void_p
            L.hor_create.restype = vp
            L.hor_create.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.c_float]
            L.hor_destroy.argtypes = [vp]
            L.hor_reset.argtypes = [vp]
            L.hor_set_update_free_factor.argtypes = [vp, C.c_float]
            L.hor_set_update_occupied_factor.argtypes = [vp, C.c_float]
            L.hor_scale_to_map.restype = C.c_float
            L.hor_scale_to_map.argtypes = [vp]
            L.hor_update_by_scan.argtypes = [vp, vp, C.c_int, vp, vp]
            L.hor_update_just_once.argtypes = [vp, vp, C.c_int, vp, C.c_float, C.c_float, C.c_double]
            L.hor_read_logodds.argtypes = [vp, vp]
            L.hor_read_update_index.argtypes = [vp, vp]
            L.hor_read_occupancy_i8.argtypes = [vp, vp]
            L.hor_last_cell_visits.restype = C.c_int64
            L.hor_last_cell_visits.argtypes = [vp]
            L.hor_match_data.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp]
            L.hor_level_factor.restype = C.c_float
            L.hor_level_factor.argtypes = [C.c_int]
            L.hor_match_level.argtypes = [vp, vp, C.c_int, C.c_float, vp, C.c_int, vp, vp]
            L.hor_hessian_derivs.argtypes = [vp, vp, C.c_int, vp, vp, vp]
            cls._L = L
        return cls._L

    def __init__(self, size_x, size_y, cell_length, offset=(0.0, 0.0)):
        self.L = self.lib()
        self.sx, self.sy = size_x, size_y
        self.h = self.L.hor_create(size_x, size_y, cell_length, offset[0], offset[1])

    def close(self):
        if getattr(self, "h", None):
            self.L.hor_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.L.hor_reset(self.h)

    def setUpdateFreeFactor(self, p):
        self.L.hor_set_update_free_factor(self.h, p)

    def setUpdateOccupiedFactor(self, p):
        self.L.hor_set_update_occupied_factor(self.h, p)

    def getScaleToMap(self):
        return self.L.hor_scale_to_map(self.h)

    def updateByScan(self, points_xy, origo_xy, pose_world):
        p = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        o = np.ascontiguousarray(origo_xy, dtype=np.float32)
        w = np.ascontiguousarray(pose_world, dtype=np.float32)
        self.L.hor_update_by_scan(self.h, p.ctypes.data, p.shape[0], o.ctypes.data, w.ctypes.data)

    def updateByScanJustOnce(self, points_xy_m, origo_xy=(0.0, 0.0), begin=(800.0, 800.0), metres_per_cell=0.05):
        p = np.ascontiguousarray(points_xy_m, dtype=np.float32).reshape(-1, 2)
        o = np.ascontiguousarray(origo_xy, dtype=np.float32)
        self.L.hor_update_just_once(self.h, p.ctypes.data, p.shape[0], o.ctypes.data, begin[0], begin[1],
                                    metres_per_cell)

    def logodds(self):
        out = np.zeros((self.sy, self.sx), dtype=np.float32)
        self.L.hor_read_logodds(self.h, out.ctypes.data)
        return out

    def occupancy_i8(self):
        out = np.zeros((self.sy, self.sx), dtype=np.int8)
        self.L.hor_read_occupancy_i8(self.h, out.ctypes.data)
        return out

    def last_cell_visits(self) -> int:
        return self.L.hor_last_cell_visits(self.h)

    @classmethod
    def match_data(cls, levels, points_xy, begin_world):
        """MapRepMultiMap::matchData over a list of PortHector levels -> (pose[3], H[3,3])."""
        L = cls.lib()
        arr = (C.c_void_p * len(levels))(*[lv.h for lv in levels])
        p = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(begin_world, dtype=np.float32)
        pose, cov = np.zeros(3, dtype=np.float32), np.zeros(9, dtype=np.float32)
        L.hor_match_data(arr, len(levels), p.ctypes.data, p.shape[0], b.ctypes.data, pose.ctypes.data, cov.ctypes.data)
        return pose, cov.reshape(3, 3)

    @classmethod
    def level_factor(cls, level: int) -> float:
        return cls.lib().hor_level_factor(level)

    def update_index(self):
        out = np.zeros((self.sy, self.sx), dtype=np.int32)
        self.L.hor_read_update_index(self.h, out.ctypes.data)
        return out

    def match_level(self, points_xy, begin_world, max_iterations, factor=1.0):
        p = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(begin_world, dtype=np.float32)
        pose, cov = np.zeros(3, dtype=np.float32), np.zeros(9, dtype=np.float32)
        self.L.hor_match_level(self.h, p.ctypes.data, p.shape[0], factor, b.ctypes.data, max_iterations,
                               pose.ctypes.data, cov.ctypes.data)
        return pose, cov.reshape(3, 3)

    def hessian_derivs(self, points_xy, pose_map):
        p = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        q = np.ascontiguousarray(pose_map, dtype=np.float32)
        H, d = np.zeros(9, np.float32), np.zeros(3, np.float32)
        self.L.hor_hessian_derivs(self.h, p.ctypes.data, p.shape[0], q.ctypes.data, H.ctypes.data, d.ctypes.data)
        return H.reshape(3, 3), d


class PortHectorRep:
    """MapRepMultiMap (H/slam_main/MapRepMultiMap.h) over PortHector levels: constructor geometry (:57-93),
    matchData coarse-to-fine with the per-level containers it caches (:144-167), updateByScan feeding the levels
    above 0 from those cached containers (:174-191) -- also when they are stale or still empty."""

    def __init__(self, map_resolution, size_x, size_y, levels, start=(0.5, 0.5)):
        f32 = np.float32
        res = f32(map_resolution)
        off = (res * f32(size_x) * f32(start[0]), res * f32(size_y) * f32(start[1]))
        self.maps = []
        sx, sy = size_x, size_y
        for _ in range(levels):
            self.maps.append(PortHector(sx, sy, float(res), (float(off[0]), float(off[1]))))
            sx //= 2
            sy //= 2
            res = f32(res * f32(2.0))
        self.levels = levels
        self.cached = [(np.zeros((0, 2), np.float32), np.zeros(2, np.float32)) for _ in range(levels - 1)]

    def setUpdateFactorFree(self, p):
        for m in self.maps:
            m.setUpdateFreeFactor(p)

    def setUpdateFactorOccupied(self, p):
        for m in self.maps:
            m.setUpdateOccupiedFactor(p)

    def reset(self):
        for m in self.maps:
            m.reset()

    def matchData(self, points_xy, begin_world, origo_xy=(0.0, 0.0)):
        p = np.ascontiguousarray(points_xy, dtype=np.float32).reshape(-1, 2)
        o = np.ascontiguousarray(origo_xy, dtype=np.float32)
        for lv in range(1, self.levels):  # setFrom: origo * factor, points * factor
            f = np.float32(PortHector.level_factor(lv))
            self.cached[lv - 1] = (p * f, o * f)
        return PortHector.match_data(self.maps, p, begin_world)

    def updateByScan(self, points_xy, origo_xy, pose_world):
        self.maps[0].updateByScan(points_xy, origo_xy, pose_world)
        for lv in range(1, self.levels):
            pts, org = self.cached[lv - 1]
            self.maps[lv].updateByScan(pts, org, pose_world)

    def cached_points(self, level):
        return len(self.cached[level - 1][0])

    def logodds(self, level=0):
        return self.maps[level].logodds()

    def occupancy_i8(self, level=0):
        return self.maps[level].occupancy_i8()


def have_ref_hector() -> bool:
    return (HERE / "_ref" / "libhector_ref.so").exists()


def have_ref_gpu() -> bool:
    """the reference's orchestrators linked against the HIP path (make -C oracle ref_gpu)"""
    return (HERE / "_ref_gpu" / "libkarto_ref_gpu.so").exists() and (HERE / "_ref_gpu" / "libhector_ref_gpu.so").exists()


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a.reshape(shape) if shape is not None else a


class _HrefLib:
    """oracle/_ref/libhector_ref.so: lesson4's UNMODIFIED hector_mapping headers behind
    oracle/hector_ref_driver.cpp (compiled against oracle/shim/Eigen; exists only where
    `make -C oracle ref_hector` ran with /root/reference present -- travels to the GPU box)."""

    _L = None
    _Lgpu = None

    @classmethod
    def lib(cls, gpu: bool = False):
        if gpu:  # the -DHREF_GPU twin: HectorSlamProcessor's mapRep is integration/hector_map_rep_gpu.hpp
            if cls._Lgpu is None:
                path = HERE / "_ref_gpu" / "libhector_ref_gpu.so"
                if not path.exists():
                    raise FileNotFoundError(f"{path} (build it here with `make -C oracle ref_gpu`)")
                cls._Lgpu = cls._load(path)
            return cls._Lgpu
        if cls._L is None:
            path = HERE / "_ref" / "libhector_ref.so"
            if not path.exists():
                build("ref_hector")
            cls._L = cls._load(path)
        return cls._L

    @staticmethod
    def _load(path):
        if True:
            L = C.CDLL(str(path))
            vp, i, f = C.c_void_p, C.c_int, C.c_float
            sig = {
                "href_map_create": (vp, [i, i, f, f, f]),
                "href_map_destroy": (None, [vp]),
                "href_map_reset": (None, [vp]),
                "href_map_set_update_free_factor": (None, [vp, f]),
                "href_map_set_update_occupied_factor": (None, [vp, f]),
                "href_map_scale_to_map": (f, [vp]),
                "href_map_obstacle_threshold": (f, [vp]),
                "href_map_update_index": (i, [vp]),
                "href_map_update_by_scan": (None, [vp, vp, i, vp, vp]),
                "href_map_update_just_once": (None, [vp, vp, i, vp]),
                "href_map_read_logodds": (None, [vp, vp]),
                "href_map_read_update_index": (None, [vp, vp]),
                "href_map_read_occupancy_i8": (None, [vp, vp]),
                "href_map_match_data": (None, [vp, vp, i, vp, vp, i, vp, vp]),
                "href_map_hessian_derivs": (None, [vp, vp, i, vp, vp, vp]),
                "href_map_world_to_map_pose": (None, [vp, vp, vp]),
                "href_map_map_to_world_pose": (None, [vp, vp, vp]),
                "href_rep_create": (vp, [f, i, i, C.c_uint, f, f]),
                "href_rep_destroy": (None, [vp]),
                "href_rep_reset": (None, [vp]),
                "href_rep_levels": (i, [vp]),
                "href_rep_level_info": (None, [vp, i, vp, vp, vp]),
                "href_rep_set_update_factor_free": (None, [vp, f]),
                "href_rep_set_update_factor_occupied": (None, [vp, f]),
                "href_rep_scale_to_map": (f, [vp]),
                "href_rep_match_data": (None, [vp, vp, i, vp, vp, vp, vp]),
                "href_rep_update_by_scan": (None, [vp, vp, i, vp, vp]),
                "href_rep_on_map_updated": (None, [vp]),
                "href_rep_cached_points": (i, [vp, i]),
                "href_rep_read_logodds": (None, [vp, i, vp]),
                "href_rep_read_update_index": (None, [vp, i, vp]),
                "href_rep_read_occupancy_i8": (None, [vp, i, vp]),
                "href_proc_create": (vp, [f, i, i, f, f, i]),
                "href_proc_destroy": (None, [vp]),
                "href_proc_set_factors": (None, [vp, f, f]),
                "href_proc_set_update_thresholds": (None, [vp, f, f]),
                "href_proc_update": (i, [vp, vp, i, vp, vp, i]),
                "href_proc_last_pose": (None, [vp, vp, vp]),
                "href_proc_levels": (i, [vp]),
                "href_proc_level_dims": (None, [vp, i, vp]),
                "href_proc_read_logodds": (None, [vp, i, vp]),
                "href_proc_read_occupancy_i8": (None, [vp, i, vp]),
                "href_pose_difference_larger_than": (i, [vp, vp, f, f]),
                "href_normalize_angle": (f, [f]),
                "href_sizeof_cell": (i, []),
                "href_is_gpu": (i, []),
            }
            for name, (res, args) in sig.items():
                fn = getattr(L, name)
                fn.restype, fn.argtypes = res, args
            return L


class RefHector:
    """One hectorslam::GridMap level of the reference (same surface as PortHector)."""

    def __init__(self, size_x, size_y, cell_length, offset=(0.0, 0.0)):
        self.L = _HrefLib.lib()
        self.sx, self.sy = size_x, size_y
        self.h = self.L.href_map_create(size_x, size_y, cell_length, offset[0], offset[1])

    def close(self):
        if getattr(self, "h", None):
            self.L.href_map_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        self.L.href_map_reset(self.h)

    def setUpdateFreeFactor(self, p):
        self.L.href_map_set_update_free_factor(self.h, p)

    def setUpdateOccupiedFactor(self, p):
        self.L.href_map_set_update_occupied_factor(self.h, p)

    def getScaleToMap(self):
        return self.L.href_map_scale_to_map(self.h)

    def updateByScan(self, points_xy, origo_xy, pose_world):
        p, o, w = _f32(points_xy, (-1, 2)), _f32(origo_xy), _f32(pose_world)
        self.L.href_map_update_by_scan(self.h, p.ctypes.data, p.shape[0], o.ctypes.data, w.ctypes.data)

    def updateByScanJustOnce(self, points_xy_m, origo_xy=(0.0, 0.0)):
        """The literal demo variant: begin (800,800) and 0.05 m cells are hard-coded by the reference."""
        p, o = _f32(points_xy_m, (-1, 2)), _f32(origo_xy)
        self.L.href_map_update_just_once(self.h, p.ctypes.data, p.shape[0], o.ctypes.data)

    def logodds(self):
        out = np.zeros((self.sy, self.sx), dtype=np.float32)
        self.L.href_map_read_logodds(self.h, out.ctypes.data)
        return out

    def update_index(self):
        out = np.zeros((self.sy, self.sx), dtype=np.int32)
        self.L.href_map_read_update_index(self.h, out.ctypes.data)
        return out

    def occupancy_i8(self):
        out = np.zeros((self.sy, self.sx), dtype=np.int8)
        self.L.href_map_read_occupancy_i8(self.h, out.ctypes.data)
        return out

    def match_level(self, points_xy, begin_world, max_iterations, origo_xy=(0.0, 0.0)):
        p, o, b = _f32(points_xy, (-1, 2)), _f32(origo_xy), _f32(begin_world)
        pose, cov = np.zeros(3, np.float32), np.zeros(9, np.float32)
        self.L.href_map_match_data(self.h, p.ctypes.data, p.shape[0], o.ctypes.data, b.ctypes.data, max_iterations,
                                   pose.ctypes.data, cov.ctypes.data)
        return pose, cov.reshape(3, 3)

    def hessian_derivs(self, points_xy, pose_map):
        p, q = _f32(points_xy, (-1, 2)), _f32(pose_map)
        H, d = np.zeros(9, np.float32), np.zeros(3, np.float32)
        self.L.href_map_hessian_derivs(self.h, p.ctypes.data, p.shape[0], q.ctypes.data, H.ctypes.data, d.ctypes.data)
        return H.reshape(3, 3), d

    def world_to_map_pose(self, w):
        out, w32 = np.zeros(3, np.float32), _f32(w)
        self.L.href_map_world_to_map_pose(self.h, w32.ctypes.data, out.ctypes.data)
        return out

    def map_to_world_pose(self, p):
        out, p32 = np.zeros(3, np.float32), _f32(p)
        self.L.href_map_map_to_world_pose(self.h, p32.ctypes.data, out.ctypes.data)
        return out


class RefHectorRep:
    """hectorslam::MapRepMultiMap of the reference (H/slam_main/MapRepMultiMap.h)."""

    def __init__(self, map_resolution, size_x, size_y, levels, start=(0.5, 0.5)):
        self.L = _HrefLib.lib()
        self.h = self.L.href_rep_create(map_resolution, size_x, size_y, levels, start[0], start[1])
        self.levels = self.L.href_rep_levels(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.L.href_rep_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def level_info(self, level):
        dims, cell, off = np.zeros(2, np.int32), C.c_float(), np.zeros(2, np.float32)
        self.L.href_rep_level_info(self.h, level, dims.ctypes.data, C.addressof(cell), off.ctypes.data)
        return int(dims[0]), int(dims[1]), float(cell.value), (float(off[0]), float(off[1]))

    def setUpdateFactorFree(self, p):
        self.L.href_rep_set_update_factor_free(self.h, p)

    def setUpdateFactorOccupied(self, p):
        self.L.href_rep_set_update_factor_occupied(self.h, p)

    def getScaleToMap(self):
        return self.L.href_rep_scale_to_map(self.h)

    def reset(self):
        self.L.href_rep_reset(self.h)

    def matchData(self, points_xy, begin_world, origo_xy=(0.0, 0.0)):
        p, o, b = _f32(points_xy, (-1, 2)), _f32(origo_xy), _f32(begin_world)
        pose, cov = np.zeros(3, np.float32), np.zeros(9, np.float32)
        self.L.href_rep_match_data(self.h, p.ctypes.data, p.shape[0], o.ctypes.data, b.ctypes.data, pose.ctypes.data,
                                   cov.ctypes.data)
        return pose, cov.reshape(3, 3)

    def updateByScan(self, points_xy, origo_xy, pose_world):
        p, o, w = _f32(points_xy, (-1, 2)), _f32(origo_xy), _f32(pose_world)
        self.L.href_rep_update_by_scan(self.h, p.ctypes.data, p.shape[0], o.ctypes.data, w.ctypes.data)

    def onMapUpdated(self):
        self.L.href_rep_on_map_updated(self.h)

    def cached_points(self, level):
        return self.L.href_rep_cached_points(self.h, level)

    def logodds(self, level=0):
        sx, sy, _, _ = self.level_info(level)
        out = np.zeros((sy, sx), dtype=np.float32)
        self.L.href_rep_read_logodds(self.h, level, out.ctypes.data)
        return out

    def occupancy_i8(self, level=0):
        sx, sy, _, _ = self.level_info(level)
        out = np.zeros((sy, sx), dtype=np.int8)
        self.L.href_rep_read_occupancy_i8(self.h, level, out.ctypes.data)
        return out


class RefHectorProcessor:
    """hectorslam::HectorSlamProcessor of the reference (H/slam_main/HectorSlamProcessor.h)."""

    def __init__(self, map_resolution, size_x, size_y, start=(0.5, 0.5), levels=3, p_free=None, p_occ=None, gpu=False):
        self.L = _HrefLib.lib(gpu)
        assert bool(self.L.href_is_gpu()) == bool(gpu)
        self.h = self.L.href_proc_create(map_resolution, size_x, size_y, start[0], start[1], levels)
        if not self.h:
            raise RuntimeError("href_proc_create failed")
        if p_free is not None:
            self.L.href_proc_set_factors(self.h, p_free, p_occ)

    def close(self):
        if getattr(self, "h", None):
            self.L.href_proc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def update(self, points_xy, pose_hint, origo_xy=(0.0, 0.0), map_without_matching=False) -> bool:
        p, o, b = _f32(points_xy, (-1, 2)), _f32(origo_xy), _f32(pose_hint)
        return bool(self.L.href_proc_update(self.h, p.ctypes.data, p.shape[0], o.ctypes.data, b.ctypes.data,
                                            int(map_without_matching)))

    def last_pose(self):
        pose, cov = np.zeros(3, np.float32), np.zeros(9, np.float32)
        self.L.href_proc_last_pose(self.h, pose.ctypes.data, cov.ctypes.data)
        return pose, cov.reshape(3, 3)

    def logodds(self, level=0):
        d = np.zeros(2, np.int32)
        self.L.href_proc_level_dims(self.h, level, d.ctypes.data)
        out = np.zeros((int(d[1]), int(d[0])), dtype=np.float32)
        self.L.href_proc_read_logodds(self.h, level, out.ctypes.data)
        return out


def href_pose_difference_larger_than(a, b, dist, ang) -> bool:
    L = _HrefLib.lib()
    a32, b32 = _f32(a), _f32(b)  # keep the temporaries alive across the call
    return bool(L.href_pose_difference_larger_than(a32.ctypes.data, b32.ctypes.data, dist, ang))


def have_ref_lesson5() -> bool:
    return (HERE / "_ref" / "liblesson5_ref.so").exists()


class RefLesson5:
    """The reference's own lesson5 LidarUndistortion (lesson5/src/lidar_undistortion.cc compiled unmodified behind
    oracle/lesson5_ref_driver.cpp, against the ROS / tf / PCL / Eigen stand-ins under oracle/shim): IMU and odometry
    messages in, LaserScans in, the de-skewed point cloud of the PREVIOUS scan out -- plus the state CorrectLaserScan read
    (the integrated gyro samples and the odometry increment of PruneImuDeque / PruneOdomDeque), which is what
    lslam_deskew_scan takes as its inputs."""

    CAP = 2000

    def __init__(self, use_imu=True, use_odom=True):
        path = HERE / "_ref" / "liblesson5_ref.so"
        if not path.exists():
            raise FileNotFoundError(f"{path} (build it here with `make -C oracle ref_lesson5`)")
        L = C.CDLL(str(path))
        L.l5_create.restype = C.c_void_p
        L.l5_create.argtypes = [C.c_int, C.c_int]
        L.l5_destroy.argtypes = [C.c_void_p]
        L.l5_add_imu.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.l5_add_odom.argtypes = [C.c_void_p] + [C.c_double] * 8
        L.l5_scan.argtypes = [C.c_void_p, C.c_double] + [C.c_float] * 5 + [C.c_void_p, C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 2
        self.L = L
        self.h = L.l5_create(int(use_imu), int(use_odom))

    def close(self):
        if getattr(self, "h", None):
            self.L.l5_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_imu(self, stamp, w):
        self.L.l5_add_imu(self.h, float(stamp), float(w[0]), float(w[1]), float(w[2]))

    def add_odom(self, stamp, xyz, quat_xyzw):
        self.L.l5_add_odom(self.h, float(stamp), *[float(v) for v in xyz], *[float(v) for v in quat_xyzw])

    def scan(self, stamp, angle_min, angle_increment, time_increment, range_min, range_max, ranges):
        """-> None while the first scan is queued / the node waits for data, else a dict for the scan that WAS corrected."""
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        n = len(r)
        xyz, valid = np.zeros((n, 3), np.float32), np.zeros(n, np.uint8)
        state, imu = np.zeros(8), np.zeros((4, self.CAP))
        hdr, cr = np.zeros(4, np.float32), np.zeros(n, np.float32)
        rc = self.L.l5_scan(self.h, float(stamp), float(angle_min), float(angle_increment), float(time_increment), float(range_min),
                            float(range_max), r.ctypes.data, n, xyz.ctypes.data, valid.ctypes.data, state.ctypes.data,
                            imu.ctypes.data, self.CAP, hdr.ctypes.data, cr.ctypes.data)
        if rc != 1:
            return None
        k = int(state[7]) + 1
        return {"xyz": xyz, "valid": valid.astype(bool), "ranges": cr, "scan_time_start": float(state[0]), "time_increment": float(state[1]),
                "start_odom_time": float(state[2]), "end_odom_time": float(state[3]), "odom_incre": state[4:7].astype(np.float32),
                "imu_time": imu[0, :k].copy(), "imu_rot": imu[1:4, :k].T.copy(),
                "angle_min": float(hdr[0]), "angle_increment": float(hdr[1]), "range_min": float(hdr[2]), "range_max": float(hdr[3])}
