// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product path.
//
// oracle/_ref/libkarto_ref.so = the reference's OWN open_karto sources
// (/root/reference/lesson6/lib/open_karto/src/{Karto,Mapper}.cpp, compiled where they lie by
// oracle/Makefile) plus this thin extern "C" driver.  It lets the tests / fixture generator /
// bench cpu_baseline call the unmodified karto::ScanMatcher::MatchScan (Mapper.cpp:184-291),
// karto::Mapper::Process (Mapper.cpp:1999-2079) and dump the intermediate state the parity tests
// compare (correlation grid bytes, lookup tables, search-space probabilities).
//
// The set-up sequence mirrors how the reference's ROS node drives the library
// (lesson6/src/karto_slam.cc:384-444): CreateLaserRangeFinder(Custom) + setters,
// Dataset::Add(laser) (registers it with SensorManager), new LocalizedRangeScan(name, readings),
// SetOdometricPose/SetCorrectedPose, Mapper::Process / ScanMatcher::MatchScan.
//
// No reference source is copied here; private members are reached with the usual
// "#define private public" test trick so that the reference files themselves stay untouched.

#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <limits>
#include <map>
#include <mutex>
#include <set>
#include <shared_mutex>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>
#include <algorithm>
#include <list>
#include <queue>
#include <iomanip>
#include <cassert>
#include <cstdlib>
#include <cstdio>
#include <chrono>
#include <fstream>
#include <limits.h>
#include <math.h>
#include <stddef.h>
#include <string.h>
#include <typeinfo>

#define private public
#define protected public
#include "open_karto/Mapper.h"
#undef private
#undef protected

// -DKREF_GPU (oracle/Makefile target ref_gpu): the published-map build of this driver goes through
// integration/karto_occupancy_grid_gpu.hpp -- the reference's LocalizedRangeScanVector in, a karto::OccupancyGrid* out,
// the cells computed on the GPU -- instead of karto::OccupancyGrid::CreateFromScans (a header-inline static, so a call-site
// change rather than a link-time substitution).  MatchScan is substituted at link time either way (see the Makefile).
#ifdef KREF_GPU
#include "karto_occupancy_grid_gpu.hpp"
static lslam_context* kref_gpu_ctx() {
  static lslam_context* ctx = nullptr;
  if (!ctx && lslam_create(0, &ctx) != LSLAM_OK) throw std::runtime_error(lslam_last_error(nullptr));
  return ctx;
}
#define KREF_CREATE_FROM_SCANS(scans, res) lslam::CreateOccupancyGridFromScans(kref_gpu_ctx(), scans, res)
namespace lslam_karto {  // integration/karto_scan_matcher_gpu.cpp
bool SyncCorrelationGrid(karto::ScanMatcher* pMatcher);
karto::Pose2 PredictPoseAfterAddEdges(const karto::Pose2& rMean, const karto::Matrix3& rCovariance);
}
#else
#define KREF_CREATE_FROM_SCANS(scans, res) karto::OccupancyGrid::CreateFromScans(scans, res)
#endif

using namespace karto;

extern "C" {

struct kref_cfg {
  double search_size;        // CorrelationSearchSpaceDimension
  double resolution;         // CorrelationSearchSpaceResolution
  double smear_deviation;    // CorrelationSearchSpaceSmearDeviation
  double coarse_angle_offset;
  double coarse_angle_resolution;
  double fine_angle_offset;  // FineSearchAngleOffset (used as the fine pass' angular step)
  double distance_variance_penalty;  // already a VARIANCE (the node's setter squares its input)
  double angle_variance_penalty;     // already a VARIANCE
  double minimum_distance_penalty;
  double minimum_angle_penalty;
  int use_response_expansion;
  // front-end (Mapper::Process) parameters
  int scan_buffer_size;
  double scan_buffer_max_scan_distance;
  double minimum_travel_distance;
  double minimum_travel_heading;
  // pose-graph side of Mapper::Process (AddEdges / LinkNearChains / TryCloseLoop); Mapper.cpp:1516-1604
  int do_loop_closing;
  int loop_match_minimum_chain_size;
  double link_match_minimum_response_fine;
  double link_scan_maximum_distance;
  double loop_search_maximum_distance;
  double loop_match_maximum_variance_coarse;  // a VARIANCE (the node's setter squares its input)
  double loop_match_minimum_response_coarse;
  double loop_match_minimum_response_fine;
  double loop_search_space_dimension;
  double loop_search_space_resolution;
  double loop_search_space_smear_deviation;
  double minimum_time_interval;
  int use_scan_barycenter;
  int reserved;
};

struct kref_laser {
  double min_angle, max_angle, angular_resolution;
  double min_range, max_range, range_threshold;
  double offset_x, offset_y, offset_heading;
};

struct KRef {
  Mapper* mapper = nullptr;
  Dataset* dataset = nullptr;
  LaserRangeFinder* laser = nullptr;
  ScanMatcher* matcher = nullptr;  // stand-alone matcher for kref_match
  std::string name;
  std::vector<LocalizedRangeScan*> owned;  // scans created by kref_match (freed per call)
  std::string err;
};

#ifndef KREF_NAME_PREFIX
#define KREF_NAME_PREFIX "kref_laser_"
#endif
static int g_counter = 0;

void* kref_create(const kref_cfg* c, const kref_laser* l) {
  try {
    KRef* k = new KRef();
    std::ostringstream nm;
    // the GPU-driven twin of this library (-DKREF_NAME_PREFIX, oracle/Makefile target ref_gpu) may live in the same
    // process: SensorManager is a process-wide singleton, so the two register their lasers under different names
    nm << KREF_NAME_PREFIX << (g_counter++);
    k->name = nm.str();
    k->mapper = new Mapper();
    Mapper* m = k->mapper;
    m->m_pCorrelationSearchSpaceDimension->SetValue(c->search_size);
    m->m_pCorrelationSearchSpaceResolution->SetValue(c->resolution);
    m->m_pCorrelationSearchSpaceSmearDeviation->SetValue(c->smear_deviation);
    m->m_pCoarseSearchAngleOffset->SetValue(c->coarse_angle_offset);
    m->m_pCoarseAngleResolution->SetValue(c->coarse_angle_resolution);
    m->m_pFineSearchAngleOffset->SetValue(c->fine_angle_offset);
    m->m_pDistanceVariancePenalty->SetValue(c->distance_variance_penalty);
    m->m_pAngleVariancePenalty->SetValue(c->angle_variance_penalty);
    m->m_pMinimumDistancePenalty->SetValue(c->minimum_distance_penalty);
    m->m_pMinimumAnglePenalty->SetValue(c->minimum_angle_penalty);
    m->m_pUseResponseExpansion->SetValue(c->use_response_expansion != 0);
    m->m_pScanBufferSize->SetValue((kt_int32u)c->scan_buffer_size);
    m->m_pScanBufferMaximumScanDistance->SetValue(c->scan_buffer_max_scan_distance);
    m->m_pMinimumTravelDistance->SetValue(c->minimum_travel_distance);
    m->m_pMinimumTravelHeading->SetValue(c->minimum_travel_heading);
    // no ScanSolver is ever attached (spa / g2o / ceres / gtsam are third-party back-ends, out of scope), so a closed
    // loop re-poses the closing scan and links it (Mapper.cpp:1029-1036) but CorrectPoses() has nothing to run
    m->m_pDoLoopClosing->SetValue(c->do_loop_closing != 0);
    m->m_pLoopMatchMinimumChainSize->SetValue((kt_int32u)c->loop_match_minimum_chain_size);
    m->m_pLinkMatchMinimumResponseFine->SetValue(c->link_match_minimum_response_fine);
    m->m_pLinkScanMaximumDistance->SetValue(c->link_scan_maximum_distance);
    m->m_pLoopSearchMaximumDistance->SetValue(c->loop_search_maximum_distance);
    m->m_pLoopMatchMaximumVarianceCoarse->SetValue(c->loop_match_maximum_variance_coarse);
    m->m_pLoopMatchMinimumResponseCoarse->SetValue(c->loop_match_minimum_response_coarse);
    m->m_pLoopMatchMinimumResponseFine->SetValue(c->loop_match_minimum_response_fine);
    m->m_pLoopSearchSpaceDimension->SetValue(c->loop_search_space_dimension);
    m->m_pLoopSearchSpaceResolution->SetValue(c->loop_search_space_resolution);
    m->m_pLoopSearchSpaceSmearDeviation->SetValue(c->loop_search_space_smear_deviation);
    m->m_pMinimumTimeInterval->SetValue(c->minimum_time_interval);
    m->m_pUseScanBarycenter->SetValue(c->use_scan_barycenter != 0);
    m->m_pUseScanMatching->SetValue(true);

    k->dataset = new Dataset();
    k->laser = LaserRangeFinder::CreateLaserRangeFinder(LaserRangeFinder_Custom, Name(k->name));
    k->laser->SetOffsetPose(Pose2(l->offset_x, l->offset_y, l->offset_heading));
    k->laser->SetMinimumRange(l->min_range);
    k->laser->SetMaximumRange(l->max_range);
    k->laser->SetMinimumAngle(l->min_angle);
    k->laser->SetMaximumAngle(l->max_angle);
    k->laser->SetAngularResolution(l->angular_resolution);
    k->laser->SetRangeThreshold(l->range_threshold);
    k->dataset->Add(k->laser);

    k->matcher = ScanMatcher::Create(m, c->search_size, c->resolution, c->smear_deviation,
                                     l->range_threshold);
    if (!k->matcher) {
      delete k->mapper;
      delete k;
      return nullptr;
    }
    return k;
  } catch (std::exception& e) {
    std::cerr << "kref_create: " << e.what() << std::endl;
    return nullptr;
  } catch (karto::Exception& e) {
    std::cerr << "kref_create: " << e.GetErrorMessage() << std::endl;
    return nullptr;
  }
}

void kref_destroy(void* h) {
  KRef* k = (KRef*)h;
  if (!k) return;
  for (auto* s : k->owned) delete s;
  delete k->matcher;
  delete k->mapper;
  // dataset owns the laser + processed scans
  delete k->dataset;
  delete k;
}

// LaserRangeFinder::GetNumberOfRangeReadings (Karto.h:4152-4161): round((max-min)/res), no +1
int kref_num_beams(void* h) { return (int)((KRef*)h)->laser->GetNumberOfRangeReadings(); }

static LocalizedRangeScan* make_scan(KRef* k, const double* ranges, int n, const double* pose) {
  std::vector<kt_double> r(ranges, ranges + n);
  LocalizedRangeScan* s = new LocalizedRangeScan(Name(k->name), r);
  Pose2 p(pose[0], pose[1], pose[2]);
  s->SetOdometricPose(p);
  s->SetCorrectedPose(p);
  return s;
}

// One ScanMatcher::MatchScan call.  Poses are ROBOT (corrected) poses; the sensor pose is
// derived by the reference itself from the laser offset.  Returns 0, or <0 on exception.
int kref_match(void* h, int n_base, const double* base_ranges, const double* base_poses,
               int n_ranges, const double* q_ranges, const double* q_pose, int do_penalize,
               int do_refine, double* out_pose, double* out_cov, double* out_response) {
  KRef* k = (KRef*)h;
  try {
    for (auto* s : k->owned) delete s;
    k->owned.clear();
    LocalizedRangeScanVector base;
    for (int i = 0; i < n_base; i++) {
      LocalizedRangeScan* s =
          make_scan(k, base_ranges + (size_t)i * n_ranges, n_ranges, base_poses + 3 * i);
      k->owned.push_back(s);
      base.push_back(s);
    }
    LocalizedRangeScan* q = make_scan(k, q_ranges, n_ranges, q_pose);
    k->owned.push_back(q);
    Pose2 mean;
    Matrix3 cov;
    cov.SetToIdentity();  // Mapper::Process passes an identity matrix (Mapper.cpp:2033-2034)
    double resp = k->matcher->MatchScan(q, base, mean, cov, do_penalize != 0, do_refine != 0);
    out_pose[0] = mean.GetX();
    out_pose[1] = mean.GetY();
    out_pose[2] = mean.GetHeading();
    for (int r = 0; r < 3; r++)
      for (int c = 0; c < 3; c++) out_cov[3 * r + c] = cov(r, c);
    *out_response = resp;
    return 0;
  } catch (std::exception& e) {
    k->err = e.what();
    return -1;
  } catch (karto::Exception& e) {
    k->err = e.GetErrorMessage();
    return -2;
  }
}

// Same, repeated `reps` times on the same inputs; returns seconds per MatchScan (steady_clock).
double kref_match_timed(void* h, int n_base, const double* base_ranges, const double* base_poses,
                        int n_ranges, const double* q_ranges_all, const double* q_poses_all,
                        int n_queries, double* out_poses, double* out_resp) {
  KRef* k = (KRef*)h;
  for (auto* s : k->owned) delete s;
  k->owned.clear();
  LocalizedRangeScanVector base;
  for (int i = 0; i < n_base; i++) {
    LocalizedRangeScan* s =
        make_scan(k, base_ranges + (size_t)i * n_ranges, n_ranges, base_poses + 3 * i);
    k->owned.push_back(s);
    base.push_back(s);
  }
  std::vector<LocalizedRangeScan*> qs;
  for (int i = 0; i < n_queries; i++) {
    LocalizedRangeScan* q =
        make_scan(k, q_ranges_all + (size_t)i * n_ranges, n_ranges, q_poses_all + 3 * i);
    k->owned.push_back(q);
    qs.push_back(q);
  }
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n_queries; i++) {
    Pose2 mean;
    Matrix3 cov;
    cov.SetToIdentity();
    double r = k->matcher->MatchScan(qs[i], base, mean, cov, true, true);
    if (out_poses) {
      out_poses[3 * i] = mean.GetX();
      out_poses[3 * i + 1] = mean.GetY();
      out_poses[3 * i + 2] = mean.GetHeading();
    }
    if (out_resp) out_resp[i] = r;
  }
  auto t1 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(t1 - t0).count() / std::max(1, n_queries);
}

// geometry of the stand-alone matcher's correlation grid (Mapper.h:1016-1027, Karto.h:4438-4471)
void kref_grid_info(void* h, int* out /* w,h,stride,roi_x,roi_y,roi_w,roi_h,kernel_size */,
                    double* out_offset /* x,y */) {
  KRef* k = (KRef*)h;
  CorrelationGrid* g = k->matcher->GetCorrelationGrid();
  out[0] = g->GetWidth();
  out[1] = g->GetHeight();
  out[2] = g->GetWidthStep();
  out[3] = g->GetROI().GetX();
  out[4] = g->GetROI().GetY();
  out[5] = g->GetROI().GetWidth();
  out[6] = g->GetROI().GetHeight();
  out[7] = g->m_KernelSize;
  out_offset[0] = g->GetCoordinateConverter()->GetOffset().GetX();
  out_offset[1] = g->GetCoordinateConverter()->GetOffset().GetY();
}

void kref_grid_copy(void* h, uint8_t* out) {
  KRef* k = (KRef*)h;
  CorrelationGrid* g = k->matcher->GetCorrelationGrid();
  memcpy(out, g->GetDataPointer(), (size_t)g->GetDataSize());
}

void kref_kernel_copy(void* h, uint8_t* out) {
  KRef* k = (KRef*)h;
  CorrelationGrid* g = k->matcher->GetCorrelationGrid();
  memcpy(out, g->m_pKernel, (size_t)g->m_KernelSize * g->m_KernelSize);
}

// lookup tables left behind by the LAST CorrelateScan (Karto.h:6409-6501): n_angles arrays
int kref_table_dims(void* h, int* n_angles, int* n_points) {
  KRef* k = (KRef*)h;
  GridIndexLookup<kt_int8u>* lk = k->matcher->m_pGridLookup;
  *n_angles = (int)lk->m_Size;
  *n_points = lk->m_Size ? (int)lk->m_ppLookupArray[0]->GetSize() : 0;
  return 0;
}

void kref_table_copy(void* h, int32_t* out, double* out_angles) {
  KRef* k = (KRef*)h;
  GridIndexLookup<kt_int8u>* lk = k->matcher->m_pGridLookup;
  for (kt_int32u a = 0; a < lk->m_Size; a++) {
    const LookupArray* arr = lk->m_ppLookupArray[a];
    memcpy(out + (size_t)a * arr->GetSize(), arr->GetArrayPointer(), sizeof(int32_t) * arr->GetSize());
    if (out_angles) out_angles[a] = lk->m_Angles[a];
  }
}

// search-space probability grid of the last COARSE pass (Mapper.cpp:431-451)
void kref_probs_copy(void* h, int* dims /* w,h,stride */, double* out) {
  KRef* k = (KRef*)h;
  Grid<kt_double>* p = k->matcher->m_pSearchSpaceProbs;
  dims[0] = p->GetWidth();
  dims[1] = p->GetHeight();
  dims[2] = p->GetWidthStep();
  if (out) memcpy(out, p->GetDataPointer(), sizeof(double) * (size_t)p->GetDataSize());
}

// Direct call of the private GetResponse (Mapper.cpp:819-856) on the current grid + tables
double kref_get_response(void* h, int angle_index, int grid_position_index) {
  KRef* k = (KRef*)h;
  return k->matcher->GetResponse((kt_int32u)angle_index, grid_position_index);
}

// World points of one scan at a robot pose, as LocalizedRangeScan::Update computes them
// (Karto.h:5362-5428).  Returns count written (unfiltered list).
int kref_point_readings(void* h, int n_ranges, const double* ranges, const double* pose,
                        int want_filtered, double* out_xy) {
  KRef* k = (KRef*)h;
  LocalizedRangeScan* s = make_scan(k, ranges, n_ranges, pose);
  const PointVectorDouble& pts = s->GetPointReadings(want_filtered != 0);
  int n = (int)pts.size();
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = pts[i].GetX();
    out_xy[2 * i + 1] = pts[i].GetY();
  }
  delete s;
  return n;
}

// FindValidPoints (Mapper.cpp:756-811) for one scan
int kref_find_valid_points(void* h, int n_ranges, const double* ranges, const double* pose,
                           const double* viewpoint, double* out_xy) {
  KRef* k = (KRef*)h;
  LocalizedRangeScan* s = make_scan(k, ranges, n_ranges, pose);
  PointVectorDouble v = k->matcher->FindValidPoints(s, Vector2<kt_double>(viewpoint[0], viewpoint[1]));
  int n = (int)v.size();
  for (int i = 0; i < n; i++) {
    out_xy[2 * i] = v[i].GetX();
    out_xy[2 * i + 1] = v[i].GetY();
  }
  delete s;
  return n;
}

// Streaming front-end: karto::Mapper::Process on one scan (Mapper.cpp:1999-2079); the scan is
// handed to the Dataset when processed (karto_slam.cc:475).  Returns 1 if processed, 0 if
// rejected by HasMovedEnough, <0 on exception.  out_pose = corrected ROBOT pose.
int kref_process(void* h, int n_ranges, const double* ranges, const double* odom_pose,
                 double* out_pose, double* out_cov_unused) {
  KRef* k = (KRef*)h;
  (void)out_cov_unused;
  try {
    LocalizedRangeScan* s = make_scan(k, ranges, n_ranges, odom_pose);
    bool ok = k->mapper->Process(s);
    Pose2 p = s->GetCorrectedPose();
    out_pose[0] = p.GetX();
    out_pose[1] = p.GetY();
    out_pose[2] = p.GetHeading();
    if (ok) {
      k->dataset->Add(s);
      return 1;
    }
    delete s;
    return 0;
  } catch (std::exception& e) {
    k->err = e.what();
    return -1;
  } catch (karto::Exception& e) {
    k->err = e.GetErrorMessage();
    return -2;
  }
}

// Mapper::Reset (Mapper.cpp:1980-1992): deletes the sequential matcher, the graph (with its loop matcher) and the scan
// manager; the next Process() re-initialises.  Scans already handed to the Dataset stay there (the Mapper never owned them).
int kref_reset(void* h) {
  KRef* k = (KRef*)h;
  try {
    k->mapper->Reset();
    return 0;
  } catch (std::exception& e) {
    k->err = e.what();
    return -1;
  }
}

// GetCorrelationGrid() (Mapper.h:1226) of the Mapper's SEQUENTIAL matcher after the last Process(): the bytes and the grid
// offset.  In the GPU-driven twin the host grid is only refreshed on request (integration/karto_scan_matcher_gpu.cpp:
// lslam_karto::SyncCorrelationGrid); returns 0 when there is no matcher (or, GPU twin, nothing has been matched yet).
int kref_mapper_grid_copy(void* h, uint8_t* out, double* offset_xy) {
  KRef* k = (KRef*)h;
  ScanMatcher* sm = k->mapper->m_pSequentialScanMatcher;
  if (!sm) return 0;
#ifdef KREF_GPU
  if (!lslam_karto::SyncCorrelationGrid(sm)) return 0;
#endif
  CorrelationGrid* g = sm->GetCorrelationGrid();
  memcpy(out, g->GetDataPointer(), (size_t)g->GetDataSize());
  const Vector2<kt_double> off = g->GetCoordinateConverter()->GetOffset();
  offset_xy[0] = off.GetX();
  offset_xy[1] = off.GetY();
  return 1;
}

#ifdef KREF_GPU
// The integration file's statement of "the pose AddEdges gives a scan whose only mean is its sequential match" beside the
// reference's own MapperGraph::ComputeWeightedMean (Mapper.cpp:1288-1330, private: this driver sees it through its
// `#define private public`) on the same (mean, covariance): host arithmetic only, no GPU involved.
int kref_weighted_mean_check(void* h, const double* mean, const double* cov9, double* out_ref, double* out_pred) {
  KRef* k = (KRef*)h;
  MapperGraph* g = k->mapper->GetGraph();
  MapperGraph* tmp = nullptr;
  if (!g) g = tmp = new MapperGraph(k->mapper, k->laser->GetRangeThreshold());
  Matrix3 c;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c(i, j) = cov9[3 * i + j];
  Pose2Vector means;
  means.push_back(Pose2(mean[0], mean[1], mean[2]));
  std::vector<Matrix3> covs;
  covs.push_back(c);
  const Pose2 a = g->ComputeWeightedMean(means, covs);
  const Pose2 b = lslam_karto::PredictPoseAfterAddEdges(means[0], c);
  out_ref[0] = a.GetX(); out_ref[1] = a.GetY(); out_ref[2] = a.GetHeading();
  out_pred[0] = b.GetX(); out_pred[1] = b.GetY(); out_pred[2] = b.GetHeading();
  delete tmp;
  return 0;
}
#endif

// pose-graph statistics after the scans processed so far: out[0] = vertices, out[1] = edges (Graph::GetEdges)
void kref_graph_stats(void* h, int out[2]) {
  KRef* k = (KRef*)h;
  out[0] = out[1] = 0;
  MapperGraph* g = k->mapper->GetGraph();
  if (!g) return;
  out[1] = (int)g->GetEdges().size();
  const Graph<LocalizedRangeScan>::VertexMap& vm = g->GetVertices();
  for (Graph<LocalizedRangeScan>::VertexMap::const_iterator it = vm.begin(); it != vm.end(); ++it) out[0] += (int)it->second.size();
}
// corrected ROBOT pose of processed scan `id` as it stands NOW (a closed loop re-poses the closing scan)
int kref_scan_pose(void* h, int id, double out_pose[3]) {
  KRef* k = (KRef*)h;
  if (!k->mapper->m_pMapperSensorManager) return -1;
  LocalizedRangeScanVector& v = k->mapper->m_pMapperSensorManager->GetScans(Name(k->name));
  if (id < 0 || id >= (int)v.size()) return -1;
  Pose2 p = v[id]->GetCorrectedPose();
  out_pose[0] = p.GetX(); out_pose[1] = p.GetY(); out_pose[2] = p.GetHeading();
  return 0;
}

// number of scans currently in the running-scan window (Mapper.h:1365-1386)
int kref_running_scans(void* h) {
  KRef* k = (KRef*)h;
  if (!k->mapper->m_pMapperSensorManager) return 0;
  return (int)k->mapper->m_pMapperSensorManager->GetRunningScans(Name(k->name)).size();
}

// Karto hit/pass occupancy grid of all processed scans (Karto.h:5659-5673), as
// karto_slam.cc:507-581 publishes it.  Two-call protocol: out==NULL -> dims only.
int kref_occupancy_grid(void* h, double resolution, int* dims /* w,h */, double* offset_xy,
                        uint8_t* out) {
  KRef* k = (KRef*)h;
  OccupancyGrid* g = nullptr;
  try {
    g = KREF_CREATE_FROM_SCANS(k->mapper->GetAllProcessedScans(), resolution);
  } catch (std::exception& e) {
    k->err = e.what();
    return -2;
  }
  if (!g) return -1;
  dims[0] = g->GetWidth();
  dims[1] = g->GetHeight();
  offset_xy[0] = g->GetCoordinateConverter()->GetOffset().GetX();
  offset_xy[1] = g->GetCoordinateConverter()->GetOffset().GetY();
  if (out) {
    for (int y = 0; y < g->GetHeight(); y++)
      for (int x = 0; x < g->GetWidth(); x++)
        out[(size_t)y * g->GetWidth() + x] = g->GetValue(Vector2<kt_int32s>(x, y));
  }
  delete g;
  return 0;
}

// ---- shared-grid ("batched many-scan") mode on the reference's own functions -------------------
// MatchScan steps 1-4 + AddScans (Mapper.cpp:212-225) around an explicit centre pose: the private
// AddScans is the reference's; only the centre is ours.
int kref_set_base_scans(void* h, int n_base, const double* base_ranges, const double* base_poses,
                        int n_ranges, const double* center_pose) {
  KRef* k = (KRef*)h;
  try {
    for (auto* s : k->owned) delete s;
    k->owned.clear();
    LocalizedRangeScanVector base;
    for (int i = 0; i < n_base; i++) {
      LocalizedRangeScan* s = make_scan(k, base_ranges + (size_t)i * n_ranges, n_ranges, base_poses + 3 * i);
      k->owned.push_back(s);
      base.push_back(s);
    }
    CorrelationGrid* g = k->matcher->m_pCorrelationGrid;
    Rectangle2<kt_int32s> roi = g->GetROI();
    Vector2<kt_double> offset;
    offset.SetX(center_pose[0] - (0.5 * (roi.GetWidth() - 1) * g->GetResolution()));
    offset.SetY(center_pose[1] - (0.5 * (roi.GetHeight() - 1) * g->GetResolution()));
    g->GetCoordinateConverter()->SetOffset(offset);
    k->matcher->AddScans(base, Vector2<kt_double>(center_pose[0], center_pose[1]));
    return 0;
  } catch (std::exception& e) {
    k->err = e.what();
    return -1;
  } catch (karto::Exception& e) {
    k->err = e.GetErrorMessage();
    return -2;
  }
}

// The search part of MatchScan (Mapper.cpp:227-282) against the CURRENT grid, spelled with the
// reference's public CorrelateScan (Mapper.h:1177-1186).  Returns seconds per scan over the n
// queries (steady_clock around the loop only).
double kref_match_fixed_grid(void* h, int n_ranges, const double* q_ranges_all, const double* q_poses_all,
                             int n_queries, int do_penalize, int do_refine, double* out_poses,
                             double* out_covs, double* out_resp) {
  KRef* k = (KRef*)h;
  ScanMatcher* sm = k->matcher;
  Mapper* mp = k->mapper;
  std::vector<LocalizedRangeScan*> qs;
  for (int i = 0; i < n_queries; i++)
    qs.push_back(make_scan(k, q_ranges_all + (size_t)i * n_ranges, n_ranges, q_poses_all + 3 * i));
  CorrelationGrid* g = sm->m_pCorrelationGrid;
  double total = 0.0;
  try {
    for (int i = 0; i < n_queries; i++) {
      qs[i]->GetPointReadings();  // LocalizedRangeScan::Update outside the timed region
    }
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n_queries; i++) {
      LocalizedRangeScan* pScan = qs[i];
      Pose2 scanPose = pScan->GetSensorPose();
      Pose2 mean;
      Matrix3 cov;
      cov.SetToIdentity();
      Vector2<kt_double> searchDimensions(sm->m_pSearchSpaceProbs->GetWidth(), sm->m_pSearchSpaceProbs->GetHeight());
      Vector2<kt_double> coarseSearchOffset(0.5 * (searchDimensions.GetX() - 1) * g->GetResolution(),
                                            0.5 * (searchDimensions.GetY() - 1) * g->GetResolution());
      Vector2<kt_double> coarseSearchResolution(2 * g->GetResolution(), 2 * g->GetResolution());
      kt_double best = sm->CorrelateScan(pScan, scanPose, coarseSearchOffset, coarseSearchResolution,
                                         mp->m_pCoarseSearchAngleOffset->GetValue(),
                                         mp->m_pCoarseAngleResolution->GetValue(), do_penalize != 0, mean, cov, false);
      // MatchScan's response expansion (Mapper.cpp:242-267), when the configuration has it on: a zero best response
      // widens the angular window by 20 degrees, up to three times -- the reference's own CorrelateScan each time
      if (mp->m_pUseResponseExpansion->GetValue() && math::DoubleEqual(best, 0.0)) {
        kt_double wider = mp->m_pCoarseSearchAngleOffset->GetValue();
        for (int e = 0; e < 3 && math::DoubleEqual(best, 0.0); e++) {
          wider += math::DegreesToRadians(20);
          best = sm->CorrelateScan(pScan, scanPose, coarseSearchOffset, coarseSearchResolution, wider,
                                   mp->m_pCoarseAngleResolution->GetValue(), do_penalize != 0, mean, cov, false);
        }
      }
      if (do_refine) {
        Vector2<kt_double> fineSearchOffset(coarseSearchResolution * 0.5);
        Vector2<kt_double> fineSearchResolution(g->GetResolution(), g->GetResolution());
        best = sm->CorrelateScan(pScan, mean, fineSearchOffset, fineSearchResolution,
                                 0.5 * mp->m_pCoarseAngleResolution->GetValue(),
                                 mp->m_pFineSearchAngleOffset->GetValue(), do_penalize != 0, mean, cov, true);
      }
      if (out_poses) { out_poses[3 * i] = mean.GetX(); out_poses[3 * i + 1] = mean.GetY(); out_poses[3 * i + 2] = mean.GetHeading(); }
      if (out_covs) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) out_covs[9 * i + 3 * r + c] = cov(r, c);
      if (out_resp) out_resp[i] = best;
    }
    auto t1 = std::chrono::steady_clock::now();
    total = std::chrono::duration<double>(t1 - t0).count();
  } catch (...) {
    total = -1.0;
  }
  for (auto* q : qs) delete q;
  return total / std::max(1, n_queries);
}

// OccupancyGrid::CreateFromScans (Karto.h:5659-5673) on explicit scans at explicit ROBOT poses:
// the published map of lesson6 (karto_slam.cc:507-581).  Two-call protocol: out == NULL -> dims only.
int kref_occgrid_from_scans(void* h, int n_scans, const double* ranges, const double* poses, int n_ranges,
                            double resolution, int* dims /* w,h */, double* offset_xy, uint8_t* out) {
  KRef* k = (KRef*)h;
  try {
    LocalizedRangeScanVector scans;
    for (int i = 0; i < n_scans; i++) scans.push_back(make_scan(k, ranges + (size_t)i * n_ranges, n_ranges, poses + 3 * i));
    OccupancyGrid* g = KREF_CREATE_FROM_SCANS(scans, resolution);
    int rc = 0;
    if (!g) {
      rc = -1;
    } else {
      dims[0] = g->GetWidth();
      dims[1] = g->GetHeight();
      offset_xy[0] = g->GetCoordinateConverter()->GetOffset().GetX();
      offset_xy[1] = g->GetCoordinateConverter()->GetOffset().GetY();
      if (out)
        for (int y = 0; y < g->GetHeight(); y++)
          for (int x = 0; x < g->GetWidth(); x++) out[(size_t)y * g->GetWidth() + x] = g->GetValue(Vector2<kt_int32s>(x, y));
      delete g;
    }
    for (auto* s : scans) delete s;
    return rc;
  } catch (std::exception& e) {
    k->err = e.what();
    return -2;
  } catch (karto::Exception& e) {
    k->err = e.GetErrorMessage();
    return -3;
  }
}

const char* kref_last_error(void* h) { return ((KRef*)h)->err.c_str(); }

int kref_sizeof_pose2() { return (int)sizeof(Pose2); }
int kref_sizeof_matrix3() { return (int)sizeof(Matrix3); }
double kref_round(double v) { return math::Round(v); }

}  // extern "C"
