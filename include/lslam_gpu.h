/*
 * lslam_gpu.h -- C ABI of the MI355X-native 2-D laser SLAM front-end hot path
 * (Karto-style correlative scan matcher + Hector-style Bresenham log-odds grid update).
 *
 * This is the drop-in boundary.  The reference (xiangli0608/Creating-2D-laser-slam-from-scratch)
 * has no FFI for this path -- its seams are C++ classes -- so every entry point below names the
 * reference interface it stands in for (file:line relative to the reference repo; short names:
 *   Mapper.h/Karto.h = lesson6/lib/open_karto/include/open_karto/..., Mapper.cpp = .../src/Mapper.cpp,
 *   H/ = lesson4/include/lesson4/hector_mapping/).
 * INTEGRATION.md shows the adapter a maintainer of the reference would add on top of it.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.  Opaque handles.
 *  - Every call returns LSLAM_OK (0) or a negative lslam_status; lslam_last_error() gives text.
 *    No exception ever crosses the boundary (the reference throws karto::Exception /
 *    std::runtime_error, Mapper.cpp:444-447,484-487, Karto.h:4488-4499; those surface as
 *    per-scan `status` in lslam_match_result or as the call's return code).
 *  - Poses are (x [m], y [m], heading [rad]).  Matcher poses are SENSOR poses
 *    (LocalizedRangeScan::GetSensorPose, Karto.h:5280); lslam_sensor_pose_from_robot /
 *    lslam_robot_pose_from_sensor convert (Karto.h:5289-5313).
 *  - "host" entry points take host pointers and are synchronous.  "_dev" entry points take
 *    device (HBM) pointers, only enqueue work on the context's HIP stream and return; use
 *    lslam_synchronize() or your own event on lslam_stream().
 *  - A context is bound to one GPU and is not thread-safe, like the reference's matcher
 *    (one grid + one lookup per ScanMatcher instance, Mapper.h:1273-1278).
 *  - There is NO CPU fallback: without a usable gfx950 device lslam_create fails.
 */
#ifndef LSLAM_GPU_H
#define LSLAM_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSLAM_ABI_VERSION 5

typedef enum lslam_status {
  LSLAM_OK = 0,
  LSLAM_ERR_INVALID_ARGUMENT = -1, /* ScanMatcher::Create returning NULL (Mapper.cpp:130-145) */
  LSLAM_ERR_NO_DEVICE = -2,
  LSLAM_ERR_INDEX_OUT_OF_RANGE = -3,  /* karto::Exception from Grid::GridIndex (Karto.h:4488-4499) */
  LSLAM_ERR_PROBABILITY_SEARCH = -4,  /* "Index out of range in probability search" (Mapper.cpp:444-447) */
  LSLAM_ERR_NO_BEST_POSE = -5,        /* "Unable to find best position" (Mapper.cpp:484-487) */
  LSLAM_ERR_HIP = -6,
  LSLAM_ERR_SMEAR_DEVIATION = -7,     /* CalculateKernel range check (Mapper.h:1041-1053) */
  LSLAM_ERR_UNSUPPORTED = -8,
  LSLAM_ERR_NO_DATA = -9              /* a diagnostic was read before anything had been collected for it */
} lslam_status;

typedef struct lslam_context lslam_context;
typedef struct lslam_matcher lslam_matcher;
typedef struct lslam_map lslam_map;

/* ---------------------------------------------------------------------------------------- */
/* context                                                                                  */
/* ---------------------------------------------------------------------------------------- */
int lslam_abi_version(void);
/* device = HIP device ordinal.  Fails (LSLAM_ERR_NO_DEVICE) if no GPU is visible. */
int lslam_create(int device, lslam_context** out);
void lslam_destroy(lslam_context* ctx);
const char* lslam_last_error(const lslam_context* ctx); /* ctx may be NULL: last global error */
int lslam_synchronize(lslam_context* ctx);
void* lslam_stream(lslam_context* ctx); /* the hipStream_t all work of this context runs on */
/* device memory helpers for callers without their own allocator (bench / tests) */
int lslam_dev_alloc(lslam_context* ctx, size_t bytes, void** out);
int lslam_dev_free(lslam_context* ctx, void* p);
int lslam_dev_upload(lslam_context* ctx, void* dst_dev, const void* src_host, size_t bytes);
int lslam_dev_download(lslam_context* ctx, void* dst_host, const void* src_dev, size_t bytes);
/* per-kernel HIP-event timing of subsequent calls (bench.py's roofline numbers).
 * lslam_profile_read returns the number of records written: name (<=31 chars), launches, total ms */
typedef struct lslam_kernel_time {
  char name[32];
  int64_t launches;
  double total_ms;
} lslam_kernel_time;
/* Diagnostics: 256 samples {where + 1, shader-clock ticks (s_memtime), 100 MHz ticks (s_memrealtime)} from 256 single-wave
 * blocks; `where` = XCD << 16 | shader engine / array / CU bits of HW_ID; 0 = that block did not report.  Two calls around a timed
 * region give the clock it ran at -- (t1 - t0) / ((r1 - r0) / 1e8), taken per CU present in both samples, so that nothing depends
 * on whether the counters of different CUs agree.  Synchronises the context stream.  (bench.py prices cycles with it.) */
int lslam_clock_sample(lslam_context* ctx, uint64_t out[768]);
int lslam_profile_enable(lslam_context* ctx, int on);
/* restrict the timing to kernels launched under this name (NULL or "" = every kernel): two events per
 * launch cost ~3 us of stream time, which matters when a batch is five kernels */
int lslam_profile_only(lslam_context* ctx, const char* kernel_name);
int lslam_profile_reset(lslam_context* ctx);
int lslam_profile_read(lslam_context* ctx, lslam_kernel_time* out, int capacity);

/* ---------------------------------------------------------------------------------------- */
/* Karto correlative scan matcher  (replaces karto::ScanMatcher, Mapper.h:1127-1279)        */
/* ---------------------------------------------------------------------------------------- */
typedef struct lslam_matcher_config {
  double search_size;                /* ScanMatcher::Create searchSize   (Mapper.h:1139-1143) */
  double resolution;                 /*                      resolution                        */
  double smear_deviation;            /*                      smearDeviation                    */
  double range_threshold;            /*                      rangeThreshold                    */
  /* the Mapper parameters the matcher reads through friend access (Mapper.cpp:206,238-256,
   * 279-280,405-411).  The two variance penalties are VARIANCES: Mapper::setParam*VariancePenalty
   * squares what the ROS node passes (Mapper.cpp:1919-1927). */
  double coarse_search_angle_offset; /* m_pCoarseSearchAngleOffset */
  double coarse_angle_resolution;    /* m_pCoarseAngleResolution   */
  double fine_search_angle_offset;   /* m_pFineSearchAngleOffset   */
  double distance_variance_penalty;  /* m_pDistanceVariancePenalty */
  double angle_variance_penalty;     /* m_pAngleVariancePenalty    */
  double minimum_distance_penalty;   /* m_pMinimumDistancePenalty  */
  double minimum_angle_penalty;      /* m_pMinimumAnglePenalty     */
  int32_t use_response_expansion;    /* m_pUseResponseExpansion    */
  int32_t reserved;
} lslam_matcher_config;

/* karto::LaserRangeFinder parameters the hot path reads (Karto.h:4127-4137, karto_slam.cc:384-395) */
typedef struct lslam_laser {
  double minimum_angle, maximum_angle, angular_resolution;
  double minimum_range, maximum_range, range_threshold;
  double offset_x, offset_y, offset_heading; /* Sensor::SetOffsetPose (karto_slam.cc:387-389) */
} lslam_laser;

/* Pose2 mean + response + Matrix3 covariance of one match (Mapper.h:1155-1159): 104 B padded to 112 */
typedef struct lslam_match_result {
  double pose[3];       /* rMean: best SENSOR pose */
  double response;      /* return value of MatchScan, in [0,1] */
  double covariance[9]; /* rCovariance, row-major 3x3 */
  int32_t status;       /* LSLAM_OK or the lslam_status the reference would have thrown */
  int32_t flags;        /* bit0: response expansion was used */
} lslam_match_result;

/* default parameters of the library (Mapper.cpp:1572-1647) */
void lslam_matcher_config_defaults(lslam_matcher_config* cfg);

/* ScanMatcher::Create (Mapper.h:1139-1143, Mapper.cpp:126-172): allocates the correlation grid,
 * smear kernel and workspaces in HBM.  Invalid parameters -> LSLAM_ERR_INVALID_ARGUMENT (the
 * reference returns NULL) / LSLAM_ERR_SMEAR_DEVIATION (the reference throws). */
int lslam_matcher_create(lslam_context* ctx, const lslam_matcher_config* cfg,
                         const lslam_laser* laser, lslam_matcher** out);
void lslam_matcher_destroy(lslam_matcher* m); /* ~ScanMatcher (Mapper.cpp:119-124) */

/* LaserRangeFinder::GetNumberOfRangeReadings (Karto.h:4152-4161): round((max-min)/res), no +1 */
int lslam_matcher_num_beams(const lslam_matcher* m);
/* GetCorrelationGrid() geometry (Mapper.h:1226): out[8] = width,height,widthStep,roi_x,roi_y,
 * roi_w,roi_h,kernel_size ; offset_xy = CoordinateConverter offset of the current grid */
int lslam_matcher_grid_info(const lslam_matcher* m, int32_t out[8], double offset_xy[2]);
/* GetCorrelationGrid()->GetDataPointer() contents: height*widthStep bytes */
int lslam_matcher_get_grid_u8(lslam_matcher* m, uint8_t* out_host);
int lslam_matcher_get_kernel_u8(lslam_matcher* m, uint8_t* out_host);
/* shared-grid mode: install a prebuilt correlation grid (same geometry) + its offset */
int lslam_matcher_set_grid_u8(lslam_matcher* m, const uint8_t* grid_host, const double offset_xy[2]);
/* same, from a device buffer (used to replicate one rank's grid after an RCCL broadcast) */
int lslam_matcher_set_grid_u8_dev(lslam_matcher* m, const uint8_t* grid_dev, const double offset_xy[2]);
/* device address of the grid bytes (height*widthStep), e.g. as an RCCL broadcast buffer */
void* lslam_matcher_grid_dev_ptr(lslam_matcher* m);

/* Tuning / diagnostics switches; none of them changes a result.
 *  LSLAM_OPT_ROW_OCCUPANCY (default 1): exact zero-row pruning of the coarse pass.  0 = every in-range lattice row
 *    of every beam is gathered -- the worst case over world sparsity (bench.py reports both step times).
 *  LSLAM_OPT_COLLECT_STATS (default 0): 1 clears the counters and routes coarse passes through an instrumented twin of
 *    the hot kernel (slower: for an untimed diagnostic launch); lslam_matcher_read_stats then returns, summed over
 *    the passes since: [0] lattice rows inside the reference's index range (Mapper.cpp:841-845), [1] rows still live
 *    after pruning, [2] readable beam x angle pairs, [3] those with at least one live row. */
/*  LSLAM_OPT_LDS_STAGED (default 0): 1 routes the coarse pass of chip-filling batches through the LDS-staged variant of
 *    the hot kernel (phase B reads its rows from per-drain patches of the parity planes staged in LDS): an experiment that
 *    was measured and dropped (DESIGN_HISTORY.md B, "LDS-staged experiment"), kept selectable so the measurement can be repeated. */
/*  LSLAM_OPT_PIPELINE_DEPTH (default 1; 1..4): with depth D > 1 consecutive lslam_matcher_match_batch_dev_* calls become
 *    PIPELINED STEPS: they take turns on D internal HIP streams, each with its own set of per-step workspaces, so that
 *    D steps share the chip: their response kernels fill each other's tails and their latency-bound prep / reduce
 *    kernels run side by side instead of one step after the other (what small per-GPU batches of a sharded run need:
 *    512 scans/step 0.118 -> 0.086 ms on one MI355X; DESIGN.md 6 has the timeline).  Same kernels, same arguments, byte-identical
 *    records.  A step is ordered behind everything the CONTEXT stream held when the call was made (inputs, grid
 *    changes) but not behind the steps before it; the context stream falls in behind all steps at the next
 *    lslam_synchronize / lslam_matcher_flush or any matcher entry point that touches the grid.  The caller's side of
 *    the contract: up to D steps are in flight at once, so D consecutive calls must not share an output buffer whose
 *    earlier contents are still wanted, and results are read after lslam_synchronize (or lslam_matcher_flush + work on
 *    the context stream).  lslam_matcher_match_batch (host arrays) splits its batch into up to D sub-batches of >= 256
 *    scans and pipelines those, uploads included; it returns with everything done, as before.  The reference has no
 *    counterpart: karto::ScanMatcher is one grid, one lookup table, one caller (Mapper.h:1273-1278). */
/*  LSLAM_OPT_CHECK_OUTPUT_REUSE (default 0; debug): with 1 a pipelined step whose result buffer overlaps the buffer of a
 *    step that may still be in flight returns LSLAM_ERR_INVALID_ARGUMENT instead of letting two steps write the same
 *    records (the caller's side of the LSLAM_OPT_PIPELINE_DEPTH contract, checked on the host: no device cost). */
/*  LSLAM_OPT_ROWS_WAVES (1, 2, 4 or 8; default 1): waves per block of the coarse response kernel of chip-filling batches.
 *    With W > 1 the W waves of a block take W consecutive candidate angles of ONE scan (k_resp_rows_mw), so they run on one
 *    CU and share its L1 (points, occupancy words, lines of the tiled planes); 1 = one wave per block (k_resp_rows).  Same
 *    numerators either way. */
/*  LSLAM_OPT_STEP_KERNEL (0, 3 or 4): ONE launch per batched match instead of five.  With 3 or 4 a workgroup of that
 *    many wave64s takes one scan through scan_prep -> coarse responses -> coarse reduce -> fine responses -> fine reduce
 *    (k_match_step: the same device functions in the same order, byte-identical records); the int32 response numerators
 *    stay in LDS and the latency-bound phases of one scan run under the gathers of the scans resident beside it.  Applies
 *    to batches of at least LSLAM_OPT_STEP_MIN_SCANS scans (default 64) whose configuration takes the tiled production
 *    kernels: no response expansion, refinement on, coarse lattice rows of 5..16 positions, the 3 x 3 fine lattice;
 *    everything else keeps the five-kernel path (value 0 = always).  lslam_matcher_step_kernel_launches counts the
 *    launches that did go out as one kernel.  The reference has no counterpart (Mapper.cpp:184-291 is one scan at a time). */
/*  LSLAM_OPT_LONE_KERNEL (0, 4, 8 or 16; default 0): the match of ONE scan (lslam_matcher_match_scan, the streaming
 *    front-end, a batch of one) as ONE launch instead of four: k_match_lone keeps the width of the chain -- one (angle, beam
 *    slice) task per wave, blocks of that many wave64s -- and replaces the launches between coarse responses, coarse reduce,
 *    fine responses and fine reduce by counters in device memory (the last block to arrive runs the reduce; the others wait,
 *    bounded, on one word).  Same device functions, same atomically accumulated integers: byte-identical records.  A hand-over
 *    that does not arrive within ~40 ms makes the record's status LSLAM_ERR_HIP; nothing hangs.  Measured: 3 us per scan SLOWER
 *    than the four launches (profiles/r06/experiments/README.md section 6), hence off by default; value + 100 = one task wave
 *    per block (the A/B form).  lslam_matcher_lone_kernel_launches counts the matches that did go out as one kernel.  The
 *    reference has no counterpart. */
enum { LSLAM_OPT_ROW_OCCUPANCY = 1, LSLAM_OPT_COLLECT_STATS = 2, LSLAM_OPT_LDS_STAGED = 3, LSLAM_OPT_PIPELINE_DEPTH = 4,
       LSLAM_OPT_STEP_KERNEL = 5, LSLAM_OPT_STEP_MIN_SCANS = 6, LSLAM_OPT_ROWS_WAVES = 7, LSLAM_OPT_CHECK_OUTPUT_REUSE = 8, LSLAM_OPT_LONE_KERNEL = 9 };
/* current value of an option (negative: error code) */
int lslam_matcher_get_option(const lslam_matcher* m, int option);
/* Order the context stream behind every pipelined step in flight (no host wait).  No-op at depth 1. */
int lslam_matcher_flush(lslam_matcher* m);
/* diagnostics: pipelined steps enqueued so far (0 while LSLAM_OPT_PIPELINE_DEPTH is 1) */
int64_t lslam_matcher_pipelined_steps(const lslam_matcher* m);
/* diagnostics: batched matches that went out as ONE launch (LSLAM_OPT_STEP_KERNEL) so far */
int64_t lslam_matcher_step_kernel_launches(const lslam_matcher* m);
/* diagnostics: single-scan matches that went out as ONE launch (LSLAM_OPT_LONE_KERNEL) so far */
int64_t lslam_matcher_lone_kernel_launches(const lslam_matcher* m);
/* diagnostics: k_match_lone's hand-over words, 16 slots x 8 words {coarse tickets, coarse done, fine ready, fine tickets,
 * fine done, timeouts, 0, 0}; LSLAM_ERR_NO_DATA before the first such launch */
int lslam_debug_lone_sync(lslam_matcher* m, unsigned* out128);
int lslam_matcher_set_option(lslam_matcher* m, int option, int value);
int lslam_matcher_read_stats(lslam_matcher* m, uint64_t out[4]);
/* after an instrumented pass: out[0] = readable (scan, beam) pairs of that batch, out[1] = those with a live lattice row in
 * AT LEAST ONE of the scan's coarse angles (a beam outside out[1] contributes nothing to any candidate of its scan);
 * out[2], out[3] = with LSLAM_OPT_LDS_STAGED: drains of 64 queued beams whose patches fit LDS / that took the global path */
int lslam_matcher_read_beam_stats(lslam_matcher* m, uint64_t out[4]);

/* LocalizedRangeScan::GetSensorAt / SetSensorPose (Karto.h:5280-5313), host-side, double */
void lslam_sensor_pose_from_robot(const lslam_laser* laser, const double robot[3], double sensor[3]);
void lslam_robot_pose_from_sensor(const lslam_laser* laser, const double sensor[3], double robot[3]);

/* MatchScan steps 1-4 + AddScans (Mapper.cpp:212-225,699-748): recentre the grid on
 * center_pose, clear it, rasterise + smear the valid points (FindValidPoints, :756-811, viewpoint =
 * center position) of n_scans base scans.  ranges: n_scans rows of `ranges_stride` doubles
 * (>= num_beams used), sensor_poses: n_scans*3. */
int lslam_matcher_set_base_scans(lslam_matcher* m, int n_scans, const double* ranges,
                                 int ranges_stride, const double* sensor_poses,
                                 const double center_pose[3]);

/* ScanMatcher::MatchScan (Mapper.h:1155-1159, Mapper.cpp:184-291), complete: rebuild the grid
 * from the base scans around the query's sensor pose, coarse + (expansion) + fine search. */
int lslam_matcher_match_scan(lslam_matcher* m, int n_base, const double* base_ranges,
                             int ranges_stride, const double* base_sensor_poses,
                             const double* query_ranges, const double query_sensor_pose[3],
                             int do_penalize, int do_refine, lslam_match_result* out);

/* The search part of MatchScan (Mapper.cpp:227-290 = CorrelateScan coarse/expansion/fine,
 * Mapper.h:1177-1186) for n_scans INDEPENDENT scans against the CURRENT grid (batched
 * many-scan mode; n_scans = 1 is the single-scan case).  Host pointers, synchronous. */
int lslam_matcher_match_batch(lslam_matcher* m, int n_scans, const double* ranges,
                              int ranges_stride, const double* sensor_poses, int do_penalize,
                              int do_refine, lslam_match_result* out);
/* Same with everything resident in HBM: float32 ranges exactly as a sensor_msgs/LaserScan
 * carries them (widened to double on the device like karto_slam.cc:428-433), results written
 * to out_dev.  Asynchronous on the context stream. */
int lslam_matcher_match_batch_dev_f32(lslam_matcher* m, int n_scans, const float* ranges_dev,
                                      int ranges_stride, const double* sensor_poses_dev,
                                      int do_penalize, int do_refine,
                                      lslam_match_result* out_dev);
int lslam_matcher_match_batch_dev_f64(lslam_matcher* m, int n_scans, const double* ranges_dev,
                                      int ranges_stride, const double* sensor_poses_dev,
                                      int do_penalize, int do_refine,
                                      lslam_match_result* out_dev);

/* ---- device-side scan cache behind seam B1 -------------------------------------------------------------------------
 * karto::ScanMatcher::MatchScan (Mapper.h:1155-1159) is handed its base scans as a LocalizedRangeScanVector on every
 * call: the running window (Mapper.cpp:2040), near chains (:942, :1140), loop chains (:991, :1015).  A scan's readings
 * never change and its pose changes about once in its life (when its own match is accepted, Mapper.cpp:2040-2044, or a
 * closed loop re-poses it), so re-sending ~70 x 8.6 KB per call -- what lslam_matcher_match_scan's literal signature
 * costs -- is avoidable: the cache keeps a scan's readings, its world points and the anchor chain of FindValidPoints
 * (Mapper.cpp:756-811) resident in HBM under an id the CALLER chooses (integration/karto_scan_matcher_gpu.cpp numbers the
 * LocalizedRangeScan objects it meets).  One cache serves every matcher of its context that was created for the same
 * laser (the sequential and the loop matcher of one Mapper).  Not thread-safe, like the matcher. */
typedef struct lslam_scan_cache lslam_scan_cache;
int lslam_scan_cache_create(lslam_context* ctx, const lslam_laser* laser, lslam_scan_cache** out);
void lslam_scan_cache_destroy(lslam_scan_cache* cache);
/* readings of scan `scan_id` (>= 0; num_beams doubles, LocalizedRangeScan::GetRangeReadings) into HBM; an id that is
 * already cached is overwritten.  The caller's buffer is free again on return. */
int lslam_scan_cache_put(lslam_scan_cache* cache, int64_t scan_id, const double* ranges);
int lslam_scan_cache_contains(const lslam_scan_cache* cache, int64_t scan_id); /* 1 / 0 */
int lslam_scan_cache_forget(lslam_scan_cache* cache, int64_t scan_id);         /* scan_id < 0: every scan */
int lslam_scan_cache_size(const lslam_scan_cache* cache);
/* World points + FindValidPoints anchors of cached scan `scan_id` at `sensor_pose`, ENQUEUED on the context stream (returns
 * at once): for a caller that knows -- or can guess -- the pose a scan will have when it is next named as a base scan
 * (integration/karto_scan_matcher_gpu.cpp: the pose AddEdges gives the scan Mapper::Process has just matched, Mapper.cpp:
 * 930-973), so that the preparation runs beside the caller's host code instead of in front of its next match.  A guess
 * that turns out wrong costs nothing but the refresh it would have saved. */
int lslam_scan_cache_prepare(lslam_scan_cache* cache, int64_t scan_id, const double sensor_pose[3]);
/* out[5] = matches served, scans uploaded, (scan, pose) refreshes (world points + anchors recomputed because the pose
 * differed bitwise from the cached one), speculative refreshes (below), bytes of HBM held */
int lslam_scan_cache_counters(const lslam_scan_cache* cache, int64_t out[5]);
/* flags of lslam_matcher_match_scan_cached */
enum {
  LSLAM_MATCH_PENALIZE = 1, /* doPenalize */
  LSLAM_MATCH_REFINE = 2,   /* doRefineMatch */
  /* The caller will give the query scan the returned mean as its sensor pose (Mapper::Process: SetSensorPose(bestPose),
   * Mapper.cpp:2040-2044): its world points + anchors at that pose are computed BEHIND the match on the stream, and the
   * call returns as soon as the result record is on the host.  Purely a hint: a pose that turns out different is
   * refreshed again when the scan is next named as a base scan. */
  LSLAM_MATCH_QUERY_TAKES_RESULT_POSE = 4
};
/* ScanMatcher::MatchScan, complete (= lslam_matcher_match_scan, identical results), with the base scans named by id.
 * base_sensor_poses: n_base*3, the scans' CURRENT sensor poses (24 bytes per scan are all that cross the bus).
 * query_id >= 0 and cached: query_ranges may be NULL.  query_id >= 0 and not cached: query_ranges are uploaded INTO the
 * cache under that id (the scan the caller is about to add to its running window).  query_id < 0: anonymous query
 * (TryCloseLoop's temporary scan, Mapper.cpp:1008-1015), nothing is kept.  An unknown base id -> LSLAM_ERR_INVALID_ARGUMENT. */
int lslam_matcher_match_scan_cached(lslam_matcher* m, lslam_scan_cache* cache, int n_base, const int64_t* base_ids,
                                    const double* base_sensor_poses, int64_t query_id, const double* query_ranges,
                                    const double query_sensor_pose[3], int flags, lslam_match_result* out);

/* ---- inspection hooks used by the parity tests (intermediate state of the reference) ---- */
/* GridIndexLookup::ComputeOffsets (Karto.h:6409-6501) for one scan: out = n_angles*num_beams
 * int32 (INT_MAX = INVALID_SCAN); returns n_angles in *n_angles_out */
int lslam_matcher_debug_lookup_table(lslam_matcher* m, const double* ranges,
                                     const double sensor_pose[3], double angle_center,
                                     double angle_offset, double angle_resolution,
                                     int32_t* out_host, int* n_angles_out);
/* integer numerators of GetResponse (Mapper.cpp:819-856) over the COARSE lattice of one scan
 * against the current grid, order y,x,angle; returns counts through nx,ny,na */
int lslam_matcher_debug_coarse_sums(lslam_matcher* m, const double* ranges,
                                    const double sensor_pose[3], int32_t* out_host, int* nx,
                                    int* ny, int* na, int force_generic_kernel);
/* the same for a BATCH: the n_scans scans go through exactly the launches a coarse-only match of that batch takes (so the
 * variant of the response kernel the batch size selects -- beam slices, linear or tiled planes, fp64 or estimate-first table
 * cells -- is the one whose numerators come back); out: n_scans * ny * nx * na int32, each scan in the order y, x, angle */
int lslam_matcher_debug_coarse_sums_batch(lslam_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                                          const double* sensor_poses, int32_t* out);
/* ... and of the FINE pass (Mapper.cpp:276-281: 3 x 3 positions x 11 angles around the coarse pass's mean) of every scan of a
 * batch, through the launches a complete match of that batch takes; centers_out: n_scans * 3, the mean each fine pass was
 * centred on (NaN where the coarse pass failed); out: n_scans * ny * nx * na int32, order y, x, angle.  With out == NULL
 * only the counts are returned. */
int lslam_matcher_debug_fine_sums_batch(lslam_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                                        const double* sensor_poses, double* centers_out, int32_t* out, int* nx, int* ny,
                                        int* na);
/* FindValidPoints mask (Mapper.cpp:756-811) of one scan: out[num_beams] bytes, 1 = kept */
int lslam_matcher_debug_valid_mask(lslam_matcher* m, const double* ranges,
                                   const double sensor_pose[3], const double viewpoint[2],
                                   uint8_t* out_host);

/* ---------------------------------------------------------------------------------------- */
/* Batched many-scan mode across the GPUs of one node, one host process:     */
/* one context + matcher per device, the shared grid replicated device-to-device over xGMI,   */
/* scans [r*B/W, (r+1)*B/W) matched on device r, results in scan order.  No collective: the   */
/* units are independent.  (bench.py is the one-process-per-GPU form with RCCL.)              */
/* ---------------------------------------------------------------------------------------- */
typedef struct lslam_pool lslam_pool;
/* n_devices = 0: every visible GPU; more than are visible -> LSLAM_ERR_INVALID_ARGUMENT */
int lslam_pool_create(int n_devices, const lslam_matcher_config* cfg, const lslam_laser* laser, lslam_pool** out);
/* explicit device ordinals (repeats allowed: several contexts on one GPU, which is how the 1-GPU tests shard) */
int lslam_pool_create_on(const int* devices, int n_devices, const lslam_matcher_config* cfg, const lslam_laser* laser,
                         lslam_pool** out);
void lslam_pool_destroy(lslam_pool* pool);
int lslam_pool_devices(const lslam_pool* pool);
lslam_matcher* lslam_pool_matcher(lslam_pool* pool, int i); /* borrowed */
const char* lslam_pool_last_error(const lslam_pool* pool);
/* lslam_matcher_set_base_scans on the first device, then the 4 MB grid goes to the others by hipMemcpyPeerAsync
 * (replicate_by_rebuild = 0) or every device rasterises the same base scans itself (!= 0) */
int lslam_pool_set_base_scans(lslam_pool* pool, int n_scans, const double* ranges, int ranges_stride,
                              const double* sensor_poses, const double center_pose[3], int replicate_by_rebuild);
/* lslam_matcher_match_batch over the whole pool; out[n_scans] in scan order */
int lslam_pool_match_batch(lslam_pool* pool, int n_scans, const double* ranges, int ranges_stride,
                           const double* sensor_poses, int do_penalize, int do_refine, lslam_match_result* out);

/* ---------------------------------------------------------------------------------------- */
/* Streaming front-end: karto::Mapper::Process (Mapper.cpp:1999-2079) with every processed scan  */
/* resident in HBM (replaces Mapper::Process as karto_slam.cc:444 calls it; the pose graph's    */
/* matches run on the device, its bookkeeping on the host, the optional solver stays outside)   */
/* ---------------------------------------------------------------------------------------- */
typedef struct lslam_frontend lslam_frontend;
/* The Mapper parameters Mapper::Process and MapperGraph read (Mapper.cpp:1457-1604; defaults = the library's).
 * Variances are VARIANCES (the node's setters square their inputs, Mapper.cpp:1919-1927, 1873-1876). */
typedef struct lslam_frontend_config {
  int32_t scan_buffer_size;                  /* ScanBufferSize */
  int32_t use_scan_barycenter;               /* UseScanBarycenter: reference pose of a scan = barycentre of its readings */
  int32_t do_loop_closing;                   /* DoLoopClosing: TryCloseLoop after every processed scan */
  int32_t loop_match_minimum_chain_size;     /* LoopMatchMinimumChainSize (also the minimum near-chain size) */
  double scan_buffer_maximum_scan_distance;  /* ScanBufferMaximumScanDistance */
  double minimum_travel_distance;            /* MinimumTravelDistance */
  double minimum_travel_heading;             /* MinimumTravelHeading [rad] */
  double minimum_time_interval;              /* MinimumTimeInterval [s] */
  double link_match_minimum_response_fine;   /* LinkMatchMinimumResponseFine */
  double link_scan_maximum_distance;         /* LinkScanMaximumDistance */
  double loop_search_maximum_distance;       /* LoopSearchMaximumDistance */
  double loop_match_maximum_variance_coarse; /* LoopMatchMaximumVarianceCoarse */
  double loop_match_minimum_response_coarse; /* LoopMatchMinimumResponseCoarse */
  double loop_match_minimum_response_fine;   /* LoopMatchMinimumResponseFine */
  double loop_search_space_dimension;        /* LoopSearchSpaceDimension: the loop matcher's search_size */
  double loop_search_space_resolution;       /* LoopSearchSpaceResolution */
  double loop_search_space_smear_deviation;  /* LoopSearchSpaceSmearDeviation */
} lslam_frontend_config;
void lslam_frontend_config_defaults(lslam_frontend_config* cfg);
/* Mapper::Process with its pose graph (Mapper.cpp:1999-2079, 862-1390): every processed scan stays resident in HBM;
 * AddEdges' LinkNearChains matches (Mapper.cpp:1124-1149) and TryCloseLoop's coarse (loop matcher, created here from
 * the sequential matcher's parameters + the three loop_search_space_* values, Mapper.cpp:862-871) and fine matches
 * (Mapper.cpp:976-1051) run on the device, the graph walk (breadth-first FindNearLinkedScans, FindNearChains,
 * FindPossibleLoopClosure) on the host.  No ScanSolver: CorrectPoses() is a no-op exactly as in karto::Mapper without
 * an optimizer (the solvers are third-party back-ends).  The matcher must outlive the front-end. */
int lslam_frontend_create_ex(lslam_matcher* m, const lslam_frontend_config* cfg, lslam_frontend** out);
/* round-1 spelling: library defaults for everything not named here, loop closing OFF.
 * scan_buffer_size / scan_buffer_max_distance: ScanBufferSize, ScanBufferMaximumScanDistance
 * (Mapper.cpp:1501-1515); min_travel_*: MinimumTravelDistance / MinimumTravelHeading (:1480-1499). */
int lslam_frontend_create(lslam_matcher* m, int scan_buffer_size, double scan_buffer_max_distance,
                          double min_travel_distance, double min_travel_heading, lslam_frontend** out);
void lslam_frontend_destroy(lslam_frontend* f);
int lslam_frontend_reset(lslam_frontend* f);
/* One LaserScan in (ranges widened to double, odometric ROBOT pose), corrected ROBOT pose out.
 * *processed = 0 when HasMovedEnough rejects the scan (Mapper.cpp:2028-2031): corrected_pose is then the odometric
 * pose carried through the last correction, covariance the identity and response 0.  covariance/response may be NULL.
 * lslam_frontend_process = time 0 (the MinimumTimeInterval test of HasMovedEnough never fires). */
int lslam_frontend_process(lslam_frontend* f, const double* ranges, int n_ranges, const double odom_pose[3],
                           int* processed, double corrected_pose[3], double covariance[9], double* response);
int lslam_frontend_process_stamped(lslam_frontend* f, const double* ranges, int n_ranges, const double odom_pose[3],
                                   double time_s, int* processed, double corrected_pose[3], double covariance[9],
                                   double* response);
/* Mapper::Process for n_scans scans the caller ALREADY HOLDS (offline / batch use: a recorded trajectory), one after the
 * other, with one scan of look-ahead: while the loop search of scan t -- a chain of lone, latency-bound matches on the
 * loop matchers, 98 % of which close nothing -- is in flight, the running-window match of scan t + 1 is enqueued on the
 * sequential matcher; Process(t + 1) finds it done.  A loop that does close re-poses scan t: the look-ahead match is
 * dropped and redone, so scan for scan the poses, the edges and the graph are those of n_scans calls of
 * lslam_frontend_process_stamped.  ranges: n_scans rows of ranges_stride doubles; odom_poses n_scans*3; times_s n_scans or
 * NULL (= 0); processed n_scans; corrected_poses n_scans*3; covariances n_scans*9 or NULL; responses n_scans or NULL.
 * The reference has no counterpart (Mapper::Process takes one scan: Mapper.cpp:1999-2079). */
int lslam_frontend_process_many(lslam_frontend* f, int n_scans, const double* ranges, int ranges_stride,
                                const double* odom_poses, const double* times_s, int32_t* processed,
                                double* corrected_poses, double* covariances, double* responses);
/* diagnostics of the speculative anchor chains (the newest scan's FindValidPoints anchors are worked out beside its own
 * match, on the points at the pose the match starts from, and taken over at the final pose when that is provably the same
 * chain): out[0] = world-point refreshes that were handed one, out[1] = of those, the ones that worked the chain out again. */
int lslam_frontend_spec_chain_stats(lslam_frontend* f, int64_t out[2]);
/* diagnostics: FindValidPoints' anchors (Mapper.cpp:774-787) of resident scan scan_id at its current pose: out[0] = count,
 * out[1..count] = beam indices in order; out holds num_beams + 1 ints */
int lslam_debug_frontend_anchor_row(lslam_frontend* f, int scan_id, int32_t* out);
/* out[3] = look-ahead matches started, accepted, discarded */
int lslam_frontend_lookahead_stats(const lslam_frontend* f, int64_t out[3]);
/* GetAllProcessedScans().size(); corrected ROBOT pose of processed scan `scan_id` as it stands now (a closed loop
 * re-poses the closing scan); out[6] = scans, graph edges, near-chain matches, loop coarse matches, loop fine
 * matches, loops closed */
int lslam_frontend_num_scans(const lslam_frontend* f);
int lslam_frontend_scan_pose(const lslam_frontend* f, int scan_id, double robot_pose[3]);
int lslam_frontend_stats(const lslam_frontend* f, int64_t out[6]);
int lslam_frontend_running_scans(const lslam_frontend* f); /* ScanManager::GetRunningScans().size() */

/* ---------------------------------------------------------------------------------------- */
/* Karto hit/pass-counter occupancy grid (replaces karto::OccupancyGrid::CreateFromScans,     */
/* Karto.h:5659-5673, as SlamKarto::updateMap calls it, karto_slam.cc:507-581)                */
/* ---------------------------------------------------------------------------------------- */
typedef struct lslam_occgrid lslam_occgrid;
/* n_scans scans (rows of `ranges_stride` doubles) at SENSOR poses; the grid is sized by
 * ComputeDimensions (Karto.h:5799-5817).  No scans -> LSLAM_ERR_INVALID_ARGUMENT (reference: NULL). */
int lslam_occgrid_create_from_scans(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges,
                                    int ranges_stride, const double* sensor_poses, double resolution,
                                    lslam_occgrid** out);
void lslam_occgrid_destroy(lslam_occgrid* og);
/* GetWidth/GetHeight, CoordinateConverter offset, resolution */
int lslam_occgrid_info(const lslam_occgrid* og, int32_t dims[2], double offset_xy[2], double* resolution);
/* GetValue(x,y): 0 unknown, 100 occupied, 255 free (GridStates, Karto.h:4193-4198); row-major y*w+x */
int lslam_occgrid_read_u8(lslam_occgrid* og, uint8_t* out_host);
/* nav_msgs/OccupancyGrid data as karto_slam.cc:546-569 fills it: -1 unknown, 100 occupied, 0 free */
int lslam_occgrid_read_ros_i8(lslam_occgrid* og, int8_t* out_host);

/* Sharded build over several GPUs / ranks (SURVEY 8(e) "offline map build from known poses").  The reference's
 * result depends on the scans only through (i) the union of their bounding boxes (ComputeDimensions, Karto.h:
 * 5799-5817: min/max) and (ii) per-cell hit/pass counts (AddScan/RayTrace, Karto.h:5851-5942: integer sums), so
 * disjoint scan subsets combine exactly, in any order:
 *   1. every rank: lslam_occgrid_scan_bounds over ITS scans -> box = {minx, miny, maxx, maxy}
 *      (n_scans == 0 is allowed: the identity box {+big, +big, -big, -big});
 *   2. all-reduce the boxes (min on [0..1], max on [2..3]);
 *   3. every rank: lslam_occgrid_create_partial with the merged box -> the counters of its scans on the common grid;
 *   4. all-reduce(sum) the counter buffers: lslam_occgrid_export_counters -> collective -> lslam_occgrid_import_
 *      counters (accumulate = 0).  The buffer is lslam_occgrid_counter_words 32-bit words: the pass plane
 *      (stride*height, stride = width rounded up to 8) followed by the hit plane, so ONE collective moves both.
 *      on_device = 1: `buf` is a device pointer on the grid's device (RCCL works on it in place), 0: host memory.
 *      accumulate = 1 adds a peer's exported buffer instead (one-process / tree merges).
 *   5. every rank (or only the one that publishes) reads the cells: lslam_occgrid_read_u8 / _read_ros_i8.
 * An empty merged box (no scans anywhere) -> LSLAM_ERR_INVALID_ARGUMENT, the reference's NULL. */
int lslam_occgrid_scan_bounds(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges,
                              int ranges_stride, const double* sensor_poses, double box[4]);
int lslam_occgrid_create_partial(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges,
                                 int ranges_stride, const double* sensor_poses, double resolution, const double box[4],
                                 lslam_occgrid** out);
int lslam_occgrid_counter_words(const lslam_occgrid* og, size_t* words);
/* device address of the counter buffer (pass plane, then hit plane; lslam_occgrid_counter_words uint32 words): a caller
 * with its own RCCL communicator all-reduces it in place */
void* lslam_occgrid_counters_dev_ptr(lslam_occgrid* og);
int lslam_occgrid_export_counters(lslam_occgrid* og, uint32_t* buf, int on_device);
int lslam_occgrid_import_counters(lslam_occgrid* og, const uint32_t* buf, int on_device, int accumulate);

/* Steps 1-4 above as ONE call per rank, with RCCL called directly from this library (librccl is dlopen'ed on first use;
 * LSLAM_ERR_UNSUPPORTED if it cannot be loaded).  nccl_comm: the caller's ncclComm_t for ctx's device (one process per
 * GPU: ncclCommInitRank; one process, one thread per device: ncclCommInitAll).  n_scans = 0 is a valid shard.  Two
 * collectives on the context stream: all-reduce(max) of (-minx, -miny, maxx, maxy), then ONE all-reduce(sum, uint32)
 * over both counter planes in place in HBM (32 MB at 2005^2 cells).  *out holds the grid of ALL ranks' scans. */
int lslam_occgrid_create_sharded(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges,
                                 int ranges_stride, const double* sensor_poses, double resolution, void* nccl_comm,
                                 lslam_occgrid** out);
/* The same over the devices of an lslam_pool in ONE process (scans [r*n/W, (r+1)*n/W) to device r, one host thread per
 * device, communicators from ncclCommInitAll); the grid is returned on the pool's first device.  A pool that names one
 * GPU more than once cannot form an RCCL clique: its partial grids are then merged by counter addition instead. */
int lslam_pool_occgrid_from_scans(lslam_pool* pool, const lslam_laser* laser, int n_scans, const double* ranges,
                                  int ranges_stride, const double* sensor_poses, double resolution, lslam_occgrid** out);

/* ---------------------------------------------------------------------------------------- */
/* Hector log-odds occupancy grid  (replaces hectorslam::OccGridMapBase<LogOddsCell,...>,    */
/* H/map/OccGridMapBase.h, H/map/GridMapLogOdds.h, H/map/GridMapBase.h)                      */
/* ---------------------------------------------------------------------------------------- */
/* GridMapBase ctor (H/map/GridMapBase.h:54-67) x MapRepMultiMap pyramid (H/slam_main/
 * MapRepMultiMap.h:50-93): level i has size>>i cells of cell_length*2^i.
 * top-left offset as in H/map/GridMapBase.h:270-286. */
int lslam_map_create(lslam_context* ctx, int size_x, int size_y, float cell_length,
                     float offset_x, float offset_y, int levels, lslam_map** out);
void lslam_map_destroy(lslam_map* map);
int lslam_map_reset(lslam_map* map);                         /* GridMapBase::reset (:95-110) */
int lslam_map_set_update_factor_free(lslam_map* map, float p);     /* H/map/OccGridMapBase.h:103-106 */
int lslam_map_set_update_factor_occupied(lslam_map* map, float p); /* :108-111 */
int lslam_map_levels(const lslam_map* map);
int lslam_map_size(const lslam_map* map, int level, int* size_x, int* size_y);
float lslam_map_scale_to_map(const lslam_map* map, int level); /* getScaleToMap (:292-295) */
/* MapRepMultiMap::updateByScan (H/slam_main/MapRepMultiMap.h:174-191): OccGridMapBase::updateByScan
 * (H/map/OccGridMapBase.h:118-168) on every pyramid level.  Level 0 takes (points_xy, origo_xy); level
 * i > 0 takes -- exactly like the reference's dataContainers[i-1] -- the container CACHED BY THE LAST
 * lslam_map_match_data CALL, scaled by 1/2^i (DataPointContainer::setFrom, H/scan/DataPointContainer.h:
 * 46-58): empty until the first matchData, stale if the caller updates with a different scan than it
 * matched.  HectorSlamProcessor::update always matches first (HectorSlamProcessor.h:84-110), so in the
 * reference's own flow all levels see the same scan.  A single-level map has no such coupling.
 * points_xy: n points in LEVEL-0 MAP-CELL units, robot frame (hector_slam.cc:320-362);
 * origo_xy: DataContainer origo; pose_world: (x[m], y[m], heading).
 * Points whose end cell is not representable (NaN/Inf or beyond the int32 range) are dropped, as on
 * the reference's x86 build, where the float->int cast yields INT_MIN and the in-map test rejects it.
 * At most 65536 points per container (LSLAM_ERR_UNSUPPORTED beyond; the reference has no limit).
 * ASYNCHRONOUS and pipelined: when the call returns points_xy has been copied and may be reused at once, this scan's
 * marks are enqueued on the context stream and its apply rides in the NEXT update's launch (or in the flush any reader
 * issues: lslam_map_read_*, lslam_map_match_data, lslam_synchronize ... are ordered after the complete update). */
int lslam_map_update_by_scan(lslam_map* map, const float* points_xy, int n,
                             const float origo_xy[2], const float pose_world[3]);
int lslam_map_update_by_scan_dev(lslam_map* map, const float* points_xy_dev, int n,
                                 const float origo_xy[2], const float pose_world[3]);
/* BATCHED update: n_scans containers, each with its own origo and robot pose, applied exactly as n_scans successive
 * [matchData-cached] updateByScan calls in this order would (bit-identical log-odds planes: the float operations of
 * every cell are applied in scan order), but marked in parallel and applied by one pass over the map -- three
 * launches per pyramid level per 64 scans instead of two dependent launches per scan and level.  For callers that know
 * several poses before the map is needed again: an offline map build from known poses, or a front-end whose matcher does
 * not read this map (the Karto front-end of BASELINE config 5).
 * points_xy: the containers back to back (sum of n_points[] points, level-0 map-cell units); n_points[n_scans];
 * origos_xy[n_scans][2]; poses_world[n_scans][3].  Every level is fed the same scan (each scan counts as matched
 * first); afterwards the cached container is the last scan's. */
int lslam_map_update_batch(lslam_map* map, int n_scans, const float* points_xy, const int32_t* n_points,
                           const float* origos_xy, const float* poses_world);
/* same with the points already in HBM (n_points / origos / poses stay host arrays: a few bytes per scan) */
int lslam_map_update_batch_dev(lslam_map* map, int n_scans, const float* points_xy_dev, const int32_t* n_points,
                               const float* origos_xy, const float* poses_world);
/* OccGridMapBase::updateByScanJustOnce (H/map/OccGridMapBase.h:175-217): the lesson4
 * make_hector_map demo variant -- points in METRES, begin cell and 1/0.05 scale hard-coded by
 * the reference (we take them as parameters; pass 800,800,0.05 for the literal behaviour). */
int lslam_map_update_just_once(lslam_map* map, const float* points_xy, int n,
                               const float origo_xy[2], float begin_x, float begin_y,
                               double metres_per_cell);
/* MapRepresentationInterface::matchData (H/slam_main/MapRepresentationInterface.h:58-60) =
 * MapRepMultiMap::matchData (H/slam_main/MapRepMultiMap.h:144-167): coarse-to-fine Gauss-Newton
 * scan-to-map matching on the pyramid.  points_xy / origo_xy = the DataContainer (level-0 map-cell
 * units; origo_xy may be NULL = (0,0); the matcher itself never reads the origo, but the container is
 * cached for the next updateByScan, see above); begin_world = beginEstimateWorld; out_cov = covMatrix
 * (the last Hessian, ScanMatcher.h:82-86).  n = 0 returns begin_world unchanged (ScanMatcher.h:96). */
int lslam_map_match_data(lslam_map* map, const float* points_xy, int n, const float origo_xy[2],
                         const float begin_world[3], float out_pose[3], float out_cov[9]);
/* matchData adds its nine Hessian / gradient sums over the points in PARALLEL (tree sums in fp32, float32 exp / sin / cos):
 * within 1e-5 of the reference's poses on the test sequences, tolerance 1e-4 m / 1e-4 rad.  LSLAM_MAP_OPT_ORDERED_SUMS = 1
 * selects the kernel that adds them in point order like the reference's sequential loop (H/matcher/ScanMatcher.h:94-126)
 * and evaluates the libm calls in double like its host code: bit-equal to the CPU restatement, ~4x slower. */
/* The batched update's scratch (lslam_map_update_batch*): every scan of a batch owns a WINDOW of 8x8-cell tiles -- the
 * square its rays can reach from its begin cell -- in a pool of 64-byte tile slots, not a byte plane of the whole map
 * (a 1081-beam scan with 20 m of range on a 0.025 m map: 2.6 MB, whatever the map's size).
 *  LSLAM_MAP_OPT_BATCH_SCRATCH_MB (default 192): budget of one pyramid level's pool; the scans of a call are applied in
 *    rounds whose windows fit it (one round normally; a single scan whose window is larger gets a round -- and the
 *    memory -- of its own).  Results do not depend on it.
 *  LSLAM_MAP_OPT_BATCH_RADIUS_CELLS (default 0 = none): for lslam_map_update_batch_dev, whose points the host never
 *    sees: an upper bound, in level-0 cells, on |point - origo| of every scan handed in.  Without it such scans get
 *    windows the size of the map (the rounds still keep the scratch inside the budget, there are just more of them).
 *    A bound that is too small loses cells: lslam_map_batch_stats counts them, and must read 0. */
enum { LSLAM_MAP_OPT_ORDERED_SUMS = 1, LSLAM_MAP_OPT_BATCH_SCRATCH_MB = 2, LSLAM_MAP_OPT_BATCH_RADIUS_CELLS = 3 };
int lslam_map_set_option(lslam_map* map, int option, int value);
/* out[0] = bytes of batched-update scratch held over all levels (tile-slot pools + tile flags), out[1] = rounds level 0
 * of the last batch took, out[2] = cells found outside their scan's window so far (synchronises), out[3] = budget, bytes */
int lslam_map_batch_stats(lslam_map* map, int64_t out[4]);
/* Host-only planner (no context, no GPU): the windows and rounds a batched update of n_scans scans would use on a level of
 * sx x sy cells.  begin_cells_xy: the scans' begin cells on that level; reach_cells: per scan the distance of its farthest
 * point from its origo in LEVEL-0 cells (NULL or < 0: unknown = the whole map); level_factor: 1, 0.5, 0.25 ... for levels
 * 0, 1, 2.  windows_out[4k..] = first tile x, first tile y, tiles wide, tiles high (8x8-cell tiles; 0 wide = nothing
 * of the scan reaches the map); base_out[k] = first tile slot inside its round; round_out[k] (may be NULL);
 * *pool_bytes_out = bytes the largest round needs.  Returns the number of rounds (>= 0) or a negative lslam_status. */
int lslam_map_plan_batch_windows(int sx, int sy, int n_scans, const int32_t* begin_cells_xy, const int32_t* n_points,
                                 const double* reach_cells, double level_factor, int64_t budget_bytes,
                                 int32_t* windows_out, uint32_t* base_out, int32_t* round_out, int64_t* pool_bytes_out);
/* LaserScan -> DataContainer ON THE DEVICE: HectorMappingRos::scanCallback's pre-processing (hector_slam.cc:186-205):
 * laser_geometry's projectLaser(scan, cloud, 30.0) and rosPointCloudToDataContainer (hector_slam.cc:320-362).  The
 * container stays resident in HBM; lslam_map_match_container / lslam_map_update_by_container are matchData /
 * updateByScan on it (no per-scan point list crosses PCIe, only the float32 ranges).  The cos/sin table of the beam
 * angles is built on the host once per scan geometry and cached (laser_geometry's co_sine_map_), so the projected
 * float32 points are bit-identical to a host evaluation.  The base_link -> laser transform is planar (yaw +
 * translation); a tilted laser needs the host path. */
typedef struct lslam_hector_scan {
  float angle_min, angle_increment, range_min, range_max; /* sensor_msgs/LaserScan header */
  float range_cutoff;         /* projectLaser's range_cutoff (hector_slam.cc:193 passes 30.0); < 0 = range_max */
  float sqr_laser_min_dist;   /* p_sqr_laser_min_dist_ */
  float sqr_laser_max_dist;   /* p_sqr_laser_max_dist_ */
  float use_max_scan_range;   /* p_use_max_scan_range_ */
  float laser_z_min, laser_z_max; /* p_laser_z_min_value_, p_laser_z_max_value_ */
  float laser_x, laser_y, laser_z, laser_yaw; /* laserTransform_: base_link -> laser */
} lslam_hector_scan;
int lslam_map_set_scan(lslam_map* map, const float* ranges, int n, const lslam_hector_scan* scan, int* n_points);
/* copies up to `capacity` points of the resident container (+ its origo) to the host; returns the container size */
int lslam_map_read_container(lslam_map* map, float* out_xy, int capacity, float origo_xy[2]);
int lslam_map_match_container(lslam_map* map, const float begin_world[3], float out_pose[3], float out_cov[9]);
int lslam_map_update_by_container(lslam_map* map, const float pose_world[3]);
/* size of the cached containers (dataContainers[i].getSize()); 0 before the first matchData */
int lslam_map_cached_points(const lslam_map* map);
/* LogOddsCell::logOddsVal of every cell, row-major y*size_x+x (H/map/GridMapLogOdds.h:85) */
int lslam_map_read_logodds(lslam_map* map, int level, float* out_host);
/* nav_msgs/OccupancyGrid data as hector_slam.cc:287-304 / hector_mapping.cc:186-200 publish
 * it: free(<0) -> 0, occupied(>0) -> 100, else -1 */
int lslam_map_read_occupancy_i8(lslam_map* map, int level, int8_t* out_host);
void* lslam_map_cells_dev_ptr(lslam_map* map, int level); /* float log-odds plane in HBM (flushes, see below) */
/* lslam_map_update_by_scan[_dev] / _by_container are PIPELINED: a call launches [apply of the previous scan | mark of
 * this scan] as ONE kernel and leaves this scan's apply pending (one launch per scan in steady state instead of two
 * dependent ones; bit-identical planes).  Every reader in this library -- lslam_map_read_*, lslam_map_match_*, the batched
 * update, lslam_map_cells_dev_ptr, lslam_synchronize -- enqueues the pending apply first.  A caller that consumes the raw
 * plane from its OWN stream calls lslam_map_flush before recording its event on lslam_stream(). */
int lslam_map_flush(lslam_map* map);

/* ---------------------------------------------------------------------------------------- */
/* lesson5 lidar motion de-skew (LidarUndistortion::CorrectLaserScan, lesson5/src/            */
/* lidar_undistortion.cc:339-447) -- SURVEY 8(f) #4.  Pinned (round 4) against the reference's */
/* own source compiled in place behind ROS / tf / PCL / Eigen stand-ins (oracle/shim, which    */
/* define PCL's getTransformation formula and Eigen 3.3's evaluation orders; tests/            */
/* test_deskew_pin.py): <= 4e-6 m (float32 cos / sin of the Euler angles from the device libm).*/
/* ---------------------------------------------------------------------------------------- */
typedef struct lslam_deskew_params {
  float angle_min, angle_increment, range_min, range_max; /* sensor_msgs/LaserScan header */
  double scan_time_start, time_increment;                 /* header.stamp, time_increment (:154-155) */
  int32_t use_imu, use_odom;
  double start_odom_time, end_odom_time;                  /* :300-301 */
  float odom_incre_x, odom_incre_y, odom_incre_z, pad;    /* :331-334 */
} lslam_deskew_params;
/* imu_time / imu_rot_*: the integrated gyro samples of this scan (imu_time_[0..current_imu_index_], :205-243),
 * n_imu = current_imu_index_ + 1.  out_xyz: n x 3 float32 (zeros for the beams the reference skips), out_valid: n */
int lslam_deskew_scan(lslam_context* ctx, const float* ranges, int n, const lslam_deskew_params* params,
                      const double* imu_time, const double* imu_rot_x, const double* imu_rot_y, const double* imu_rot_z,
                      int n_imu, float* out_xyz, uint8_t* out_valid);

#ifdef __cplusplus
}
#endif
#endif /* LSLAM_GPU_H */
