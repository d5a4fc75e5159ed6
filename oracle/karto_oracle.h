/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Karto correlative scan matcher.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / reported baseline.  The product (liblslam_gpu.so) never links,
 * includes or calls anything under oracle/.
 *
 * Parity status: PINNED.  The reference ships no tests or golden vectors, so this
 * restatement is pinned against the reference's own code executed here: oracle/_ref
 * (libkarto_ref.so = /root/reference/lesson6/lib/open_karto compiled unmodified) on seeded
 * inputs (tests/test_oracle_vs_ref.py) and through the committed fixtures under tests/golden/
 * produced from it by tests/golden/make_golden.py.
 *
 * All file:line citations are relative to /root/reference/lesson6/lib/open_karto/
 * (Mapper.cpp = src/Mapper.cpp, Mapper.h / Karto.h / Math.h = include/open_karto/...).
 */
#ifndef KARTO_ORACLE_H
#define KARTO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kor_config {
  double search_size;                 /* CorrelationSearchSpaceDimension  Mapper.cpp:1572-1576 */
  double resolution;                  /* CorrelationSearchSpaceResolution Mapper.cpp:1577-1580 */
  double smear_deviation;             /* ...SmearDeviation                Mapper.cpp:1582-1586 */
  double range_threshold;             /* rangeThreshold arg of ScanMatcher::Create Mapper.cpp:126 */
  double coarse_search_angle_offset;  /* Mapper.cpp:1621-1624 */
  double coarse_angle_resolution;     /* Mapper.cpp:1626-1629 */
  double fine_search_angle_offset;    /* Mapper.cpp:1616-1619 (used as the fine angular STEP, :280) */
  double distance_variance_penalty;   /* a variance: the node's setter squares (Mapper.cpp:1919-1922) */
  double angle_variance_penalty;      /* a variance (Mapper.cpp:1924-1927) */
  double minimum_distance_penalty;    /* Mapper.cpp:1637-1641 */
  double minimum_angle_penalty;       /* Mapper.cpp:1631-1635 */
  int32_t use_response_expansion;     /* Mapper.cpp:1643-1647 */
  int32_t reserved;
} kor_config;

typedef struct kor_laser {
  double minimum_angle, maximum_angle, angular_resolution; /* Karto.h:4130-4135 */
  double minimum_range, maximum_range, range_threshold;    /* Karto.h:4127-4137 */
  double offset_x, offset_y, offset_heading;               /* Sensor OffsetPose (karto_slam.cc:387-389) */
} kor_laser;

typedef struct kor_matcher kor_matcher;

/* ScanMatcher::Create (Mapper.cpp:126-172): NULL on invalid parameters */
kor_matcher* kor_create(const kor_config* cfg, const kor_laser* laser);
void kor_destroy(kor_matcher* m);

/* LaserRangeFinder::Update (Karto.h:4152-4161): round((max-min)/res), the +1 is commented out */
int kor_num_beams(const kor_matcher* m);
/* out[8] = width,height,stride,roi_x,roi_y,roi_w,roi_h,kernel_size ; offset[2] = grid offset */
void kor_grid_info(const kor_matcher* m, int32_t* out, double* offset);
const uint8_t* kor_grid_data(const kor_matcher* m);
const uint8_t* kor_kernel_data(const kor_matcher* m);

/* LocalizedRangeScan::GetSensorAt / SetSensorPose (Karto.h:5280-5313) */
void kor_sensor_pose_from_robot(const kor_matcher* m, const double robot[3], double sensor[3]);
void kor_robot_pose_from_sensor(const kor_matcher* m, const double sensor[3], double robot[3]);

/* LocalizedRangeScan::Update, unfiltered list (Karto.h:5362-5428): out_xy[2*num_beams] */
void kor_point_readings(const kor_matcher* m, const double* ranges, const double sensor_pose[3],
                        double* out_xy);
/* ScanMatcher::FindValidPoints (Mapper.cpp:756-811): returns count */
int kor_find_valid_points(const double* pts_xy, int n, const double viewpoint[2], double* out_xy);

/* MatchScan steps 1-4 + AddScans (Mapper.cpp:212-225, 699-748): recentre on center_pose, clear,
 * rasterise + smear every base scan (viewpoint = center_pose position) */
void kor_set_base_scans(kor_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                        const double* sensor_poses, const double center_pose[3]);
void kor_set_grid(kor_matcher* m, const uint8_t* grid, const double offset[2]);

/* GridIndexLookup::ComputeOffsets (Karto.h:6409-6501): writes n_angles*num_beams int32, returns
 * n_angles; out may be NULL to size */
int kor_compute_offsets(kor_matcher* m, const double* ranges, const double sensor_pose[3],
                        double angle_center, double angle_offset, double angle_resolution,
                        int32_t* out);
/* ScanMatcher::GetResponse numerator (Mapper.cpp:819-856) as an exact integer */
int64_t kor_response_sum(const kor_matcher* m, const int32_t* table_row, int32_t grid_position_index);

/* ScanMatcher::CorrelateScan (Mapper.cpp:309-523) against the current grid.
 * resp_sums (optional): integer response numerators, order y,x,angle (nY*nX*nA) */
double kor_correlate_scan(kor_matcher* m, const double* ranges, const double sensor_pose[3],
                          const double search_center[3], double off_x, double off_y, double res_x,
                          double res_y, double angle_offset, double angle_resolution,
                          int do_penalize, int doing_fine, double mean[3], double cov[9],
                          int32_t* resp_sums, int* status);

/* MatchScan body after AddScans (Mapper.cpp:227-290) against the CURRENT grid */
double kor_match(kor_matcher* m, const double* ranges, const double sensor_pose[3], int do_penalize,
                 int do_refine, double mean[3], double cov[9], int* status);
/* Full ScanMatcher::MatchScan (Mapper.cpp:184-291) */
double kor_match_scan(kor_matcher* m, int n_base, const double* base_ranges, int ranges_stride,
                      const double* base_sensor_poses, const double* q_ranges,
                      const double q_sensor_pose[3], int do_penalize, int do_refine, double mean[3],
                      double cov[9], int* status);

/* last coarse pass' search-space probability grid (Mapper.cpp:431-451): side*side doubles */
int kor_probs(const kor_matcher* m, double* out);

/* karto::OccupancyGrid::CreateFromScans (Karto.h:5659-5990): hit/pass-counter occupancy grid of
 * scans at SENSOR poses (lesson6's published map, karto_slam.cc:507-581).  Two-call protocol:
 * out == NULL fills dims/offset only.  Cell values: 0 unknown, 100 occupied, 255 free.
 * returns 0, or -1 when there are no scans (the reference returns NULL). */
int kor_occgrid_from_scans(const kor_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                           const double* sensor_poses, double resolution, int32_t dims[2],
                           double offset_xy[2], uint8_t* out);
/* The same build in shardable pieces (the counters are plain sums and the box a min/max, so disjoint scan subsets
 * combine exactly): box = minx, miny, maxx, maxy of ComputeDimensions over the given scans; the hit/pass counters
 * of the given scans on the grid of a given box (dims = w, h, stride; counters = pass plane then hit plane,
 * stride*h words each; NULL = dims only); Update() over (summed) counters. */
void kor_occgrid_bounds(const kor_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                        const double* sensor_poses, double box[4]);
int kor_occgrid_partial(const kor_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                        const double* sensor_poses, double resolution, const double box[4], int32_t dims[3],
                        uint32_t* counters);
void kor_occgrid_update(const int32_t dims[3], const uint32_t* counters, uint8_t* out);

/* ---- streaming front-end: the pose-relevant part of Mapper::Process (Mapper.cpp:1999-2079) ----
 * lastTransform propagation (:2021-2025), HasMovedEnough (:2087-2120, time test omitted: the
 * harness has no clock), MatchScan vs the running window (:2040), SetSensorPose (:2044),
 * AddRunningScan (Mapper.h:1365-1386).  Graph edges / loop closure (back-end) are out of scope. */
typedef struct kor_frontend kor_frontend;
kor_frontend* kor_frontend_create(kor_matcher* m, int scan_buffer_size,
                                  double scan_buffer_max_distance, double min_travel_distance,
                                  double min_travel_heading);
void kor_frontend_destroy(kor_frontend* f);
/* returns 1 processed / 0 rejected; out_pose = corrected robot pose */
int kor_frontend_process(kor_frontend* f, const double* ranges, const double odom_pose[3],
                         double out_pose[3], double out_cov[9], double* out_response);
int kor_frontend_running_scans(const kor_frontend* f);

#ifdef __cplusplus
}
#endif
#endif
