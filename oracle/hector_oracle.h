/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's Hector log-odds grid update.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Parity status: PINNED against the reference's own code.  lesson4's hector_mapping headers are
 * compiled UNMODIFIED from /root/reference as oracle/_ref/libhector_ref.so (oracle/Makefile target
 * ref_hector, driver oracle/hector_ref_driver.cpp) and tests/test_oracle_vs_ref.py checks this
 * restatement against them bit for bit: updateByScan, updateByScanJustOnce, the 3-level pyramid,
 * getCompleteHessianDerivs, ScanMatcher::matchData, MapRepMultiMap::matchData.  Eigen itself is not
 * in this image (and ROS is not needed for these headers), so they compile against oracle/shim/Eigen;
 * the ONE thing that stays shim-defined is the order in which Eigen evaluates its own 2- and 3-float
 * expressions (affine multiply, 3x3 inverse, 3-term dot product) -- written after Eigen 3.3's sources,
 * see shim/Eigen/Core.  Everything else (control flow, casts, Bresenham, once-per-scan cell semantics,
 * interpolation, Hessian sums, iteration counts, libm calls) is the reference's own compiled code.
 * This file restates H/map/OccGridMapBase.h:118-330, H/map/GridMapLogOdds.h:37-161,
 * H/map/GridMapBase.h:270-286, H/matcher/ScanMatcher.h:60-139, H/map/OccGridMapUtil.h:77-228 in
 * plain float32 C (no FMA: the reference builds lesson4 for baseline x86-64); it is what travels
 * everywhere (the GPU box has no /root/reference) and what the cpu_baseline legs time.
 *
 * Citations: H/ = /root/reference/lesson4/include/lesson4/hector_mapping/.
 */
#ifndef HECTOR_ORACLE_H
#define HECTOR_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct hor_map hor_map;

/* GridMapBase ctor (H/map/GridMapBase.h:54-67): one level */
hor_map* hor_create(int size_x, int size_y, float cell_length, float offset_x, float offset_y);
void hor_destroy(hor_map* m);
void hor_reset(hor_map* m);                           /* clear() (H/map/GridMapBase.h:102-110) */
void hor_set_update_free_factor(hor_map* m, float p); /* H/map/GridMapLogOdds.h:140-143 */
void hor_set_update_occupied_factor(hor_map* m, float p);
float hor_scale_to_map(const hor_map* m);
/* OccGridMapBase::updateByScan (H/map/OccGridMapBase.h:118-168); points in map-cell units */
void hor_update_by_scan(hor_map* m, const float* points_xy, int n, const float origo_xy[2],
                        const float pose_world[3]);
/* OccGridMapBase::updateByScanJustOnce (H/map/OccGridMapBase.h:175-217); points in metres */
void hor_update_just_once(hor_map* m, const float* points_xy, int n, const float origo_xy[2],
                          float begin_x, float begin_y, double metres_per_cell);
void hor_read_logodds(const hor_map* m, float* out);
void hor_read_update_index(const hor_map* m, int32_t* out);
/* publish conversion (hector_slam.cc:287-304): free -> 0, occupied -> 100, else -1 */
void hor_read_occupancy_i8(const hor_map* m, int8_t* out);
/* number of cells traversed (free marks incl. repeats + endpoints) by the last update: the
 * "cell-updates" unit of BASELINE.md §4 */
int64_t hor_last_cell_visits(const hor_map* m);

/* MapRepMultiMap::matchData (H/slam_main/MapRepMultiMap.h:144-167) over `n_levels` maps (level i =
 * cell_length*2^i): coarse-to-fine Gauss-Newton scan-to-map matching (H/matcher/ScanMatcher.h:60-139,
 * H/map/OccGridMapUtil.h:77-228), 3 (+1) iterations per coarse level, 5 (+1) on level 0.
 * points: level-0 map-cell units.  out_cov = the last Hessian H (ScanMatcher.h:82-86).
 * Eigen evaluation orders (as oracle/shim/Eigen states them): 3x3 inverse by cofactors with
 * det = c0*m00 + (c1*m10 + c2*m20); matrix*vector rows as a*x + (b*y + c*z); Affine2f inverse via the
 * 2x2 cofactor inverse and translation = (-Linv) * t. */
void hor_match_data(hor_map* const* levels, int n_levels, const float* points_xy, int n,
                    const float begin_world[3], float out_pose[3], float out_cov[9]);

/* one level of the above (ScanMatcher::matchData, H/matcher/ScanMatcher.h:60-99) and one evaluation of
 * getCompleteHessianDerivs (H/map/OccGridMapUtil.h:77-132) -- intermediate state for the parity tests */
void hor_match_level(const hor_map* m, const float* points_xy, int n, float factor, const float begin_world[3],
                     int max_iterations, float out_pose[3], float out_cov[9]);
void hor_hessian_derivs(const hor_map* m, const float* points_xy, int n, const float pose_map[3], float H[9],
                        float dTr[3]);

/* DataPointContainer::setFrom factor for pyramid level i (H/slam_main/MapRepMultiMap.h:161) */
float hor_level_factor(int level);

#ifdef __cplusplus
}
#endif
#endif
