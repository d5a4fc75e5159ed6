// TEST INFRASTRUCTURE ONLY -- C entry points around the reference's UNMODIFIED lesson4 hector_mapping headers
// (compiled in place from /root/reference, never copied) so that tests can run the reference's own
// OccGridMapBase::updateByScan / updateByScanJustOnce, MapRepMultiMap::matchData / updateByScan and
// HectorSlamProcessor::update.  Eigen is not in this image: the headers compile against oracle/shim/Eigen
// (what that shim defines -- only Eigen's own evaluation order of 2- and 3-float expressions -- is stated
// in shim/Eigen/Core).  Built by `make -C oracle ref_hector` into oracle/_ref/libhector_ref.so.
//
// Only tests/, __graft_entry__.smoke() and the cpu_baseline legs of bench.py / tools/bench_extra.py may load it.
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <vector>

#include <Eigen/Core>
#include <Eigen/Geometry>

// the reference's headers keep their state protected; the driver needs to read it (same device as ref_driver.cpp)
#define protected public
#define private public
#include "lesson4/hector_mapping/slam_main/HectorSlamProcessor.h"
#undef private
#undef protected

// -DHREF_GPU (oracle/Makefile target ref_gpu -> oracle/_ref_gpu/libhector_ref_gpu.so): the SAME reference
// HectorSlamProcessor, but its `mapRep` is the MI355X implementation of MapRepresentationInterface
// (integration/hector_map_rep_gpu.hpp) instead of the reference's MapRepMultiMap -- the reference's orchestrator
// driving the HIP kernels through the C ABI.  Everything else in this file is unchanged.
#ifdef HREF_GPU
#include "hector_map_rep_gpu.hpp"
#endif

using hectorslam::DataContainer;
using hectorslam::GridMap;

namespace {
// the reference prints from its constructors / matcher ("HectorSM map lvl ...", "SearchDir angle change too
// large"); keep the test output clean without touching the headers
struct CoutSilencer {
  std::streambuf* old;
  std::ostringstream sink;
  CoutSilencer() : old(std::cout.rdbuf(sink.rdbuf())) {}
  ~CoutSilencer() { std::cout.rdbuf(old); }
};

void fill(DataContainer& dc, const float* pts, int n, const float origo[2]) {
  dc.clear();
  dc.setOrigo(Eigen::Vector2f(origo[0], origo[1]));
  for (int i = 0; i < n; ++i) dc.add(Eigen::Vector2f(pts[2 * i], pts[2 * i + 1]));
}
void read_logodds(const GridMap& g, float* out) {
  const int n = g.getSizeX() * g.getSizeY();
  for (int i = 0; i < n; ++i) out[i] = g.getCell(i).logOddsVal;
}
void read_index(const GridMap& g, int32_t* out) {
  const int n = g.getSizeX() * g.getSizeY();
  for (int i = 0; i < n; ++i) out[i] = g.getCell(i).updateIndex;
}
// hector_slam.cc:287-304 / hector_mapping.cc:186-200 publish rule, through the reference's own isFree/isOccupied
void read_occupancy(const GridMap& g, int8_t* out) {
  const int n = g.getSizeX() * g.getSizeY();
  std::memset(out, -1, (size_t)n);
  for (int i = 0; i < n; ++i) {
    if (g.isFree(i)) out[i] = 0;
    else if (g.isOccupied(i)) out[i] = 100;
  }
}
}  // namespace

struct href_map {  // one GridMap level + its matcher, as MapRepMultiMap builds them (MapRepMultiMap.h:76-84)
  GridMap* grid;
  hectorslam::OccGridMapUtilConfig<GridMap>* util;
  hectorslam::ScanMatcher<hectorslam::OccGridMapUtilConfig<GridMap>>* matcher;
  DataContainer dc;
};
struct href_rep {
  hectorslam::MapRepMultiMap* rep;
  DataContainer dc;
};
struct href_proc {
  hectorslam::HectorSlamProcessor* proc;
  DataContainer dc;
};

extern "C" {

int href_abi(void) { return 1; }
#ifdef HREF_GPU
int href_is_gpu(void) { return 1; }
#else
int href_is_gpu(void) { return 0; }
#endif
int href_sizeof_cell(void) { return (int)sizeof(LogOddsCell); }

// ---- single GridMap (H/map/GridMap.h:38) --------------------------------------------------------------------
href_map* href_map_create(int size_x, int size_y, float cell_length, float offset_x, float offset_y) {
  href_map* m = new href_map;
  m->grid = new GridMap(cell_length, Eigen::Vector2i(size_x, size_y), Eigen::Vector2f(offset_x, offset_y));
  m->util = new hectorslam::OccGridMapUtilConfig<GridMap>(m->grid);
  m->matcher = new hectorslam::ScanMatcher<hectorslam::OccGridMapUtilConfig<GridMap>>();
  return m;
}
void href_map_destroy(href_map* m) {
  if (!m) return;
  delete m->matcher;
  delete m->util;
  delete m->grid;
  delete m;
}
void href_map_reset(href_map* m) { m->grid->reset(); m->util->resetCachedData(); }
void href_map_set_update_free_factor(href_map* m, float p) { m->grid->setUpdateFreeFactor(p); }
void href_map_set_update_occupied_factor(href_map* m, float p) { m->grid->setUpdateOccupiedFactor(p); }
float href_map_scale_to_map(const href_map* m) { return m->grid->getScaleToMap(); }
float href_map_obstacle_threshold(const href_map* m) { return m->grid->getObstacleThreshold(); }
int href_map_update_index(const href_map* m) { return m->grid->getUpdateIndex(); }
void href_map_update_by_scan(href_map* m, const float* pts, int n, const float origo[2], const float pose[3]) {
  fill(m->dc, pts, n, origo);
  m->grid->updateByScan(m->dc, Eigen::Vector3f(pose[0], pose[1], pose[2]));
  m->util->resetCachedData();  // MapRepMultiMap::onMapUpdated (MapRepMultiMap.h:127-135)
}
// the literal demo variant: pose (800,800,0) and 0.05 m cells are hard-coded by the reference (:182,:202-203)
void href_map_update_just_once(href_map* m, const float* pts_m, int n, const float origo[2]) {
  fill(m->dc, pts_m, n, origo);
  m->grid->updateByScanJustOnce(m->dc, Eigen::Vector3f(0.f, 0.f, 0.f));
  m->util->resetCachedData();
}
void href_map_read_logodds(const href_map* m, float* out) { read_logodds(*m->grid, out); }
void href_map_read_update_index(const href_map* m, int32_t* out) { read_index(*m->grid, out); }
void href_map_read_occupancy_i8(const href_map* m, int8_t* out) { read_occupancy(*m->grid, out); }
// ScanMatcher::matchData on this one level (H/matcher/ScanMatcher.h:60-99); pts in THIS level's cell units
void href_map_match_data(href_map* m, const float* pts, int n, const float origo[2], const float begin_world[3],
                         int max_iterations, float out_pose[3], float out_cov[9]) {
  CoutSilencer quiet;
  fill(m->dc, pts, n, origo);
  Eigen::Matrix3f cov = Eigen::Matrix3f::Zero();
  Eigen::Vector3f r = m->matcher->matchData(Eigen::Vector3f(begin_world[0], begin_world[1], begin_world[2]), *m->util,
                                            m->dc, cov, max_iterations);
  for (int i = 0; i < 3; ++i) out_pose[i] = r[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out_cov[3 * i + j] = cov(i, j);
}
// intermediate state for the parity tests: one getCompleteHessianDerivs evaluation (OccGridMapUtil.h:77-132)
void href_map_hessian_derivs(href_map* m, const float* pts, int n, const float pose_map[3], float H[9], float dTr[3]) {
  const float origo[2] = {0.f, 0.f};
  fill(m->dc, pts, n, origo);
  Eigen::Matrix3f Hm;
  Eigen::Vector3f d;
  m->util->getCompleteHessianDerivs(Eigen::Vector3f(pose_map[0], pose_map[1], pose_map[2]), m->dc, Hm, d);
  for (int i = 0; i < 3; ++i) {
    dTr[i] = d[i];
    for (int j = 0; j < 3; ++j) H[3 * i + j] = Hm(i, j);
  }
}
void href_map_world_to_map_pose(const href_map* m, const float w[3], float out[3]) {
  Eigen::Vector3f r = m->grid->getMapCoordsPose(Eigen::Vector3f(w[0], w[1], w[2]));
  for (int i = 0; i < 3; ++i) out[i] = r[i];
}
void href_map_map_to_world_pose(const href_map* m, const float p[3], float out[3]) {
  Eigen::Vector3f r = m->grid->getWorldCoordsPose(Eigen::Vector3f(p[0], p[1], p[2]));
  for (int i = 0; i < 3; ++i) out[i] = r[i];
}

// ---- MapRepMultiMap (H/slam_main/MapRepMultiMap.h) -----------------------------------------------------------
href_rep* href_rep_create(float map_resolution, int size_x, int size_y, unsigned levels, float start_x, float start_y) {
  CoutSilencer quiet;
  href_rep* r = new href_rep;
  r->rep = new hectorslam::MapRepMultiMap(map_resolution, size_x, size_y, levels, Eigen::Vector2f(start_x, start_y));
  return r;
}
void href_rep_destroy(href_rep* r) {
  if (!r) return;
  delete r->rep;
  delete r;
}
void href_rep_reset(href_rep* r) { r->rep->reset(); }
int href_rep_levels(const href_rep* r) { return r->rep->getMapLevels(); }
void href_rep_level_info(const href_rep* r, int level, int32_t dims[2], float* cell_length, float offset[2]) {
  const GridMap& g = r->rep->getGridMap(level);
  dims[0] = g.getSizeX();
  dims[1] = g.getSizeY();
  *cell_length = g.getCellLength();
  offset[0] = g.getMapDimProperties().getTopLeftOffset()[0];
  offset[1] = g.getMapDimProperties().getTopLeftOffset()[1];
}
void href_rep_set_update_factor_free(href_rep* r, float p) { r->rep->setUpdateFactorFree(p); }
void href_rep_set_update_factor_occupied(href_rep* r, float p) { r->rep->setUpdateFactorOccupied(p); }
float href_rep_scale_to_map(const href_rep* r) { return r->rep->getScaleToMap(); }
void href_rep_match_data(href_rep* r, const float* pts, int n, const float origo[2], const float begin_world[3],
                         float out_pose[3], float out_cov[9]) {
  CoutSilencer quiet;
  fill(r->dc, pts, n, origo);
  Eigen::Matrix3f cov = Eigen::Matrix3f::Zero();
  Eigen::Vector3f p = r->rep->matchData(Eigen::Vector3f(begin_world[0], begin_world[1], begin_world[2]), r->dc, cov);
  for (int i = 0; i < 3; ++i) out_pose[i] = p[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out_cov[3 * i + j] = cov(i, j);
}
// MapRepMultiMap::updateByScan (:174-191): levels > 0 use the containers cached by the LAST matchData call
void href_rep_update_by_scan(href_rep* r, const float* pts, int n, const float origo[2], const float pose[3]) {
  fill(r->dc, pts, n, origo);
  r->rep->updateByScan(r->dc, Eigen::Vector3f(pose[0], pose[1], pose[2]));
}
void href_rep_on_map_updated(href_rep* r) { r->rep->onMapUpdated(); }
int href_rep_cached_points(const href_rep* r, int level) {  // size of dataContainers[level-1]
  if (level <= 0 || level > (int)r->rep->dataContainers.size()) return -1;
  return r->rep->dataContainers[level - 1].getSize();
}
void href_rep_read_logodds(const href_rep* r, int level, float* out) { read_logodds(r->rep->getGridMap(level), out); }
void href_rep_read_update_index(const href_rep* r, int level, int32_t* out) { read_index(r->rep->getGridMap(level), out); }
void href_rep_read_occupancy_i8(const href_rep* r, int level, int8_t* out) { read_occupancy(r->rep->getGridMap(level), out); }

// ---- HectorSlamProcessor (H/slam_main/HectorSlamProcessor.h) -------------------------------------------------
href_proc* href_proc_create(float map_resolution, int size_x, int size_y, float start_x, float start_y, int levels) {
  CoutSilencer quiet;
  href_proc* p = new href_proc;
  p->proc = new hectorslam::HectorSlamProcessor(map_resolution, size_x, size_y, Eigen::Vector2f(start_x, start_y), levels);
#ifdef HREF_GPU
  // swap the map representation behind the reference's processor (HectorSlamProcessor.h:61,141)
  static lslam_context* ctx = nullptr;
  if (!ctx && lslam_create(0, &ctx) != LSLAM_OK) {
    std::cerr << "href_proc_create (gpu): " << lslam_last_error(nullptr) << std::endl;
    delete p->proc;
    delete p;
    return nullptr;
  }
  delete p->proc->mapRep;
  p->proc->mapRep = new lslam::HectorMapRepGpu(ctx, map_resolution, size_x, size_y, (unsigned)levels,
                                               Eigen::Vector2f(start_x, start_y));
  p->proc->reset();
#endif
  return p;
}
void href_proc_destroy(href_proc* p) {
  if (!p) return;
  delete p->proc;
  delete p;
}
void href_proc_set_factors(href_proc* p, float p_free, float p_occ) {
  p->proc->setUpdateFactorFree(p_free);
  p->proc->setUpdateFactorOccupied(p_occ);
}
void href_proc_set_update_thresholds(href_proc* p, float min_dist, float min_angle) {
  p->proc->setMapUpdateMinDistDiff(min_dist);
  p->proc->setMapUpdateMinAngleDiff(min_angle);
}
// HectorSlamProcessor::update (:84-110).  Returns 1 when the map was updated by this scan.
int href_proc_update(href_proc* p, const float* pts, int n, const float origo[2], const float pose_hint[3],
                     int map_without_matching) {
  CoutSilencer quiet;
  fill(p->dc, pts, n, origo);
#ifdef HREF_GPU
  // the device map rep counts its updates: getGridMap(0) would pull the whole level-0 map out of HBM twice per scan just
  // to read one integer (this driver's instrumentation, not something HectorSlamProcessor does)
  lslam::HectorMapRepGpu* rep = static_cast<lslam::HectorMapRepGpu*>(p->proc->mapRep);
  const long before = rep->updateCount();
  p->proc->update(p->dc, Eigen::Vector3f(pose_hint[0], pose_hint[1], pose_hint[2]), map_without_matching != 0);
  return rep->updateCount() != before;
#else
  const int before = p->proc->getGridMap(0).getUpdateIndex();
  p->proc->update(p->dc, Eigen::Vector3f(pose_hint[0], pose_hint[1], pose_hint[2]), map_without_matching != 0);
  return p->proc->getGridMap(0).getUpdateIndex() != before;
#endif
}
void href_proc_last_pose(const href_proc* p, float out_pose[3], float out_cov[9]) {
  const Eigen::Vector3f& q = p->proc->getLastScanMatchPose();
  const Eigen::Matrix3f& c = p->proc->getLastScanMatchCovariance();
  for (int i = 0; i < 3; ++i) out_pose[i] = q[i];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out_cov[3 * i + j] = c(i, j);
}
int href_proc_levels(const href_proc* p) { return p->proc->getMapLevels(); }
void href_proc_level_dims(const href_proc* p, int level, int32_t dims[2]) {
  dims[0] = p->proc->getGridMap(level).getSizeX();
  dims[1] = p->proc->getGridMap(level).getSizeY();
}
void href_proc_read_logodds(const href_proc* p, int level, float* out) { read_logodds(p->proc->getGridMap(level), out); }
void href_proc_read_occupancy_i8(const href_proc* p, int level, int8_t* out) { read_occupancy(p->proc->getGridMap(level), out); }

// util::poseDifferenceLargerThan as THIS toolchain compiles it (H/util/UtilFunctions.h:73-92: the unqualified
// abs(float) there is toolchain-dependent; the tests record which overload was picked)
int href_pose_difference_larger_than(const float a[3], const float b[3], float dist, float ang) {
  return util::poseDifferenceLargerThan(Eigen::Vector3f(a[0], a[1], a[2]), Eigen::Vector3f(b[0], b[1], b[2]), dist, ang);
}
float href_normalize_angle(float a) { return util::normalize_angle(a); }

}  // extern "C"
