// karto_scan_matcher_gpu.cpp -- the reference's OWN karto::Mapper driving the MI355X scan matcher.
//
// This translation unit DEFINES karto::ScanMatcher::MatchScan (Mapper.h:1155-1159; the reference's body is
// Mapper.cpp:184-291) and karto::ScanMatcher::~ScanMatcher (Mapper.cpp:119-124) on top of the C ABI of
// include/lslam_gpu.h.  Linked into open_karto in place of the reference's definitions, every caller of MatchScan inside
// the unmodified library -- Mapper::Process (Mapper.cpp:2040), MapperGraph::AddEdges (:942), LinkNearChains (:1140),
// TryCloseLoop (:991 coarse on the loop matcher, :1015 fine) -- runs its correlative search on the GPU while the pose
// graph, the scan manager and the dataset stay the reference's own code.  Nothing else of ScanMatcher is replaced: Create
// (Mapper.cpp:126-172) still allocates the host-side CorrelationGrid / search-space grid, which this file only reads for
// their GEOMETRY; the destructor additionally releases the device matcher that mirrors the instance (Mapper::Reset,
// Mapper.cpp:1980-1992, deletes and re-creates its matchers).
//
// How a maintainer of the reference uses it:
//   * either delete the bodies of ScanMatcher::MatchScan and ScanMatcher::~ScanMatcher from Mapper.cpp and add this file
//     to the library's sources,
//   * or, without touching any reference file, mark the reference's definitions weak in the compiled object
//     (objcopy --weaken-symbol=<mangled name> Mapper.o) and link this file's strong definitions next to it: that is what
//     oracle/Makefile's `ref_gpu` target does for the parity tests (link-time substitution).
//
// MatchScan is a member of ScanMatcher, and ScanMatcher is a friend of Mapper (Mapper.h:1745), so the nine Mapper
// parameters are read exactly where the reference reads them (Mapper.cpp:206,238-256,279-280,405-411); no access
// hack is needed.  The karto::LaserRangeFinder -> lslam_laser conversion the C ABI asks for is laser_from() below
// (what karto_slam.cc:384-395 configures per sensor).
//
// Device-side scan cache (round 4).  The reference hands MatchScan its base scans as a LocalizedRangeScanVector on every
// call; forwarding that literally re-uploads ~70 x 8.6 KB per call.  Scans the Mapper manages (unique id >= 0, owned by the
// caller's Dataset for the Mapper's lifetime, Mapper.h:1308-1320) are instead kept resident in an lslam_scan_cache --
// readings, world points and FindValidPoints anchors -- under a number this file assigns per LocalizedRangeScan object, so
// a call sends the new scan's readings once and 24 bytes of pose per base scan (lslam_matcher_match_scan_cached; identical
// results by construction, tests/test_scan_cache_gpu.py).  A hit needs the same object, the same unique id and the same
// readings array; temporaries (TryCloseLoop's stack scan, Mapper.cpp:1004-1015) and unmanaged scans travel uncached.
// LSLAM_KARTO_NO_CACHE=1 selects the literal forwarding of round 3.
//
// GetCorrelationGrid() (Mapper.h:1226) is an inline accessor of the HOST grid, which no reference code reads after Create
// (only MatchScan / AddScans write it, and those run on the device now): it stays zero-filled.  A caller that wants to
// look at the correlation grid of the last match calls lslam_karto::SyncCorrelationGrid(matcher) first.
//
// Threading: the reference's matcher is single-threaded and not re-entrant.  All state of this file is
// behind one mutex, so two Mappers on two threads are serialised rather than corrupted.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "open_karto/Mapper.h"

#include "karto_occupancy_grid_gpu.hpp"  // lslam::LaserFrom (karto::LaserRangeFinder -> lslam_laser); includes lslam_gpu.h

namespace lslam_karto {

static std::mutex& mutex() {
  static std::mutex m;
  return m;
}

// one HIP context per process, on device $LSLAM_DEVICE (default 0)
static lslam_context* context() {
  static lslam_context* ctx = nullptr;
  if (!ctx) {
    const char* e = std::getenv("LSLAM_DEVICE");
    int rc = lslam_create(e ? std::atoi(e) : 0, &ctx);
    if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_create: ") + lslam_last_error(nullptr));
  }
  return ctx;
}

static bool cache_enabled() {
  static const bool on = [] {
    const char* e = std::getenv("LSLAM_KARTO_NO_CACHE");
    return !(e && *e && *e != '0');
  }();
  return on;
}

static lslam_laser laser_from(karto::LaserRangeFinder* lrf) { return lslam::LaserFrom(lrf); }

struct GpuMatcher {
  lslam_matcher* h = nullptr;
  lslam_matcher_config cfg;
  lslam_laser laser;
  int grid_w = 0, grid_h = 0;  // geometry of the karto::CorrelationGrid it mirrors
  std::vector<double> ranges, poses;
  std::vector<int64_t> ids;
  long long calls = 0, cached_calls = 0;
  long long ns_total = 0, ns_device = 0;  // wall time inside MatchScan / inside the lslam call of it (host profile)
  bool sequential = false;                // this instance is its Mapper's m_pSequentialScanMatcher (noted by MatchScan)
};

// keyed by the karto::ScanMatcher instance (a library that adopts this file would hold the handle as a member)
static std::map<const karto::ScanMatcher*, GpuMatcher>& registry() {
  static std::map<const karto::ScanMatcher*, GpuMatcher> r;
  return r;
}

// ---- resident scans: one cache per laser geometry, shared by the sequential and the loop matcher -------------------------
struct ResidentScan {
  int64_t id;
  kt_int32s unique_id;        // -1: provisional (cached as the query of Mapper::Process before AddScan numbered it)
  const kt_double* readings;  // LocalizedRangeScan::GetRangeReadings() at upload time
  uint64_t checksum;          // of the uploaded readings; re-checked when a provisional entry is adopted and on every
                              // kFullCheckEvery-th hit
  uint64_t fingerprint;       // of kFingerprintSamples readings spread over the scan; re-checked on EVERY hit
  uint64_t last_use;          // Residents::clock at the last hit (capacity bound: least recently used goes first)
  uint32_t hits = 0;          // hits so far (every kFullCheckEvery-th one re-reads the whole scan)
};
struct Residents {
  lslam_laser laser;
  lslam_scan_cache* cache = nullptr;
  std::unordered_map<const karto::LocalizedRangeScan*, ResidentScan> scans;
  int64_t next_id = 0;
  uint64_t clock = 0;
};
// HBM held per resident scan: readings + world points + anchors = 28 B per beam (30.3 KB at 1081 beams).  Without a bound
// the cache grows with the Mapper's Dataset -- the same lifetime the reference gives its scans on the host.
// $LSLAM_KARTO_CACHE_MAX_SCANS (default 0 = unbounded) evicts the least recently used entries beyond it; an evicted
// scan is simply uploaded again when a later MatchScan names it.
static size_t cache_capacity() {
  static const size_t cap = [] {
    const char* e = std::getenv("LSLAM_KARTO_CACHE_MAX_SCANS");
    return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)0;
  }();
  return cap;
}
static std::vector<Residents>& residents() {
  static std::vector<Residents> r;
  return r;
}

static bool same_laser(const lslam_laser& a, const lslam_laser& b) { return std::memcmp(&a, &b, sizeof a) == 0; }
static bool same_cfg(const lslam_matcher_config& a, const lslam_matcher_config& b) {
  return std::memcmp(&a, &b, sizeof a) == 0;
}

static uint64_t checksum(const kt_double* r, int n) {
  uint64_t h = 1469598103934665603ull;
  for (int i = 0; i < n; i++) {
    uint64_t b;
    std::memcpy(&b, r + i, 8);
    h = (h ^ b) * 1099511628211ull;
  }
  return h;
}

// An address, an id and a readings pointer can all come back after a Dataset freed its scans and allocated new ones
// (ADVICE r04): a hit is therefore also held against a fingerprint of the CONTENTS -- 16 readings spread over the scan,
// a few loads per base scan per call where the full checksum would cost more than the device call it guards.
constexpr int kFingerprintSamples = 16;
constexpr uint32_t kFullCheckEvery = 64;
static uint64_t fingerprint(const kt_double* r, int n) {
  uint64_t h = 1469598103934665603ull ^ (uint64_t)n;
  if (n <= 0) return h;
  for (int k = 0; k < kFingerprintSamples; k++) {
    uint64_t b;
    std::memcpy(&b, r + (size_t)((long long)k * (n - 1) / (kFingerprintSamples - 1)), 8);
    h = (h ^ b) * 1099511628211ull;
  }
  return h;
}

static Residents& residents_for(const lslam_laser& laser) {
  for (Residents& r : residents())
    if (same_laser(r.laser, laser)) return r;
  residents().emplace_back();
  Residents& r = residents().back();
  r.laser = laser;
  int rc = lslam_scan_cache_create(context(), &laser, &r.cache);
  if (rc != LSLAM_OK) {
    residents().pop_back();
    throw std::runtime_error(std::string("lslam_scan_cache_create: ") + lslam_last_error(context()));
  }
  return r;
}

// the resident entry of a MANAGED scan (unique id >= 0), or NULL: same object, same readings array, and either the same
// unique id or a provisional entry (cached as the query of Mapper::Process, numbered by AddScan since) with equal contents
static ResidentScan* find_resident(Residents& R, karto::LocalizedRangeScan* s, int nb) {
  const kt_double* rd = s->GetRangeReadings();
  const kt_int32s uid = s->GetUniqueId();
  auto it = R.scans.find(s);
  if (uid < 0 || it == R.scans.end() || it->second.readings != rd) return nullptr;
  if (it->second.fingerprint != fingerprint(rd, nb)) return nullptr;  // same address, other contents: a recycled scan
  // ... and every kFullCheckEvery-th hit re-reads ALL readings (a recycled scan that happens to agree in the 16
  // samples would otherwise be served stale for ever): 8.6 KB hashed once per 64 device calls
  if ((++it->second.hits % kFullCheckEvery) == 0 && it->second.checksum != checksum(rd, nb)) return nullptr;
  if (it->second.unique_id == uid) {
    it->second.last_use = ++R.clock;
    return &it->second;
  }
  if (it->second.unique_id < 0 && it->second.checksum == checksum(rd, nb) && lslam_scan_cache_contains(R.cache, it->second.id)) {
    it->second.unique_id = uid;  // (contains: a match that threw may have left the provisional entry without its readings)
    it->second.last_use = ++R.clock;
    return &it->second;
  }
  return nullptr;
}

// capacity bound: drop the least recently used entries (never one touched at or after `keep_from`, i.e. by this call)
static void evict_beyond_capacity(Residents& R, uint64_t keep_from) {
  const size_t cap = cache_capacity();
  while (cap && R.scans.size() > cap) {
    auto victim = R.scans.end();
    for (auto it = R.scans.begin(); it != R.scans.end(); ++it)
      if (it->second.last_use < keep_from && (victim == R.scans.end() || it->second.last_use < victim->second.last_use)) victim = it;
    if (victim == R.scans.end()) return;  // everything resident is in use by this call
    lslam_scan_cache_forget(R.cache, victim->second.id);
    R.scans.erase(victim);
  }
}

// id of a MANAGED scan in the cache, uploading its readings on a miss
static int64_t resident_id(Residents& R, karto::LocalizedRangeScan* s, int nb) {
  if (ResidentScan* e = find_resident(R, s, nb)) return e->id;
  const kt_double* rd = s->GetRangeReadings();
  auto it = R.scans.find(s);
  ResidentScan e;
  e.id = it != R.scans.end() ? it->second.id : R.next_id++;
  e.unique_id = s->GetUniqueId();
  e.readings = rd;
  e.checksum = checksum(rd, nb);
  e.fingerprint = fingerprint(rd, nb);
  e.last_use = ++R.clock;
  int rc = lslam_scan_cache_put(R.cache, e.id, rd);
  if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_scan_cache_put: ") + lslam_last_error(context()));
  R.scans[s] = e;
  return e.id;
}

static void drop_residents() {
  for (Residents& r : residents())
    if (r.cache) lslam_scan_cache_destroy(r.cache);
  residents().clear();
}

extern "C" void lslam_karto_gpu_stats(long long out[8]);
long long gpu_match_calls() {  // for the tests: how many MatchScan calls really ran on the device
  long long all[8];
  lslam_karto_gpu_stats(all);
  return all[0];
}

void release_all() {
  std::lock_guard<std::mutex> lock(mutex());
  for (auto& kv : registry())
    if (kv.second.h) lslam_matcher_destroy(kv.second.h);
  registry().clear();
  drop_residents();
}

// What Mapper::Process does with the mean of its sequential match: AddEdges (Mapper.cpp:930-973) collects it with the
// means of the near-chain matches and gives the scan ComputeWeightedMean(means, covariances) (Mapper.cpp:1288-1330) as its
// sensor pose -- with ONE mean (no near chain linked: ~90 % of the scans) that is inverse(inverse(C)) * inverse(C) * mean,
// the same pose up to a last-bit rounding, but not the same BITS.  The scan cache matches poses bitwise, so the pose the
// scan is prepared at behind its own match is this function's, not the raw mean.  Same statements, same karto::Matrix3 /
// Pose2 operations, same order as the reference for a single (mean, covariance); a wrong guess (near chains linked, a
// closed loop) only costs the refresh in front of the next match that it was meant to save.
karto::Pose2 PredictPoseAfterAddEdges(const karto::Pose2& rMean, const karto::Matrix3& rCovariance) {
  using namespace karto;
  Matrix3 sumOfInverses;
  Matrix3 inverse = rCovariance.Inverse();
  sumOfInverses += inverse;
  Matrix3 inverseOfSumOfInverses = sumOfInverses.Inverse();
  Pose2 accumulatedPose;
  kt_double thetaX = 0.0;
  kt_double thetaY = 0.0;
  Pose2 pose = rMean;
  kt_double angle = pose.GetHeading();
  thetaX += cos(angle);
  thetaY += sin(angle);
  Matrix3 weight = inverseOfSumOfInverses * inverse;
  accumulatedPose += weight * pose;
  thetaX /= 1;
  thetaY /= 1;
  accumulatedPose.SetHeading(atan2(thetaY, thetaX));
  return accumulatedPose;
}

// The host CorrelationGrid behind GetCorrelationGrid() (Mapper.h:1226) <- the device grid of this matcher's last match
// (bytes + CoordinateConverter offset).  Returns false when the matcher has not matched anything yet.
bool SyncCorrelationGrid(karto::ScanMatcher* pMatcher) {
  std::lock_guard<std::mutex> lock(mutex());
  auto it = registry().find(pMatcher);
  if (it == registry().end() || !it->second.h) return false;
  karto::CorrelationGrid* g = pMatcher->GetCorrelationGrid();
  int32_t info[8];
  double off[2];
  lslam_matcher_grid_info(it->second.h, info, off);
  if (info[0] != g->GetWidth() || info[1] != g->GetHeight() || info[2] != g->GetWidthStep()) return false;
  if (lslam_matcher_get_grid_u8(it->second.h, g->GetDataPointer()) != LSLAM_OK) return false;
  g->GetCoordinateConverter()->SetOffset(karto::Vector2<kt_double>(off[0], off[1]));
  return true;
}

}  // namespace lslam_karto

extern "C" long long lslam_karto_gpu_match_calls(void) { return lslam_karto::gpu_match_calls(); }
extern "C" void lslam_karto_gpu_release(void) { lslam_karto::release_all(); }
// The hook a caller that removes scans from its Dataset while the Mapper lives calls per scan (before freeing it): the
// resident copy (28 B per beam of HBM) is dropped and the address may be reused at once.  Unknown scans are ignored.
extern "C" void lslam_karto_gpu_forget_scan(const karto::LocalizedRangeScan* scan) {
  using namespace lslam_karto;
  std::lock_guard<std::mutex> lock(mutex());
  for (Residents& R : residents()) {
    auto it = R.scans.find(scan);
    if (it == R.scans.end()) continue;
    if (R.cache) lslam_scan_cache_forget(R.cache, it->second.id);
    R.scans.erase(it);
  }
}
// out[8] = device MatchScan calls, of which through the scan cache, matcher instances alive, resident scans,
//          scans uploaded, (scan, pose) refreshes in front of a match, ns spent inside MatchScan, ns of those inside the
//          lslam_matcher_match_scan[_cached] call (upload + kernels + wait).  Counters of matchers already destroyed are
//          kept in a process-wide tally.
static long long g_gone[8] = {0, 0, 0, 0, 0, 0, 0, 0};
extern "C" void lslam_karto_gpu_stats(long long out[8]) {
  using namespace lslam_karto;
  std::lock_guard<std::mutex> lock(mutex());
  for (int i = 0; i < 8; i++) out[i] = g_gone[i];
  for (auto& kv : registry()) {
    out[0] += kv.second.calls;
    out[1] += kv.second.cached_calls;
    out[2] += kv.second.h ? 1 : 0;
    out[6] += kv.second.ns_total;
    out[7] += kv.second.ns_device;
  }
  for (Residents& r : residents()) {
    int64_t c[5];
    if (r.cache && lslam_scan_cache_counters(r.cache, c) == LSLAM_OK) {
      out[3] += lslam_scan_cache_size(r.cache);
      out[4] += c[1];
      out[5] += c[2];
    }
  }
}

namespace karto {

static const kt_double kMaxVariance = 500.0;  // MAX_VARIANCE, file-local in the reference (Mapper.cpp:35)

// Mapper.cpp:119-124 + the device matcher mirroring this instance.  When the SEQUENTIAL matcher of a Mapper goes
// (Mapper::Reset, Mapper.cpp:1982; ~Mapper) the scans it numbered are forgotten too: a re-initialised Mapper numbers
// its scans from 0 again (Mapper.h:1315-1318).
ScanMatcher::~ScanMatcher() {
  {
    using namespace lslam_karto;
    std::lock_guard<std::mutex> lock(lslam_karto::mutex());
    auto it = registry().find(this);
    if (it != registry().end()) {
      // (the role was noted by MatchScan: the destructor does not look at m_pMapper, which a caller may already have deleted)
      if (it->second.h) {
        if (it->second.sequential)
          for (Residents& r : residents())
            if (same_laser(r.laser, it->second.laser)) {
              if (r.cache) lslam_scan_cache_forget(r.cache, -1);
              r.scans.clear();
              r.next_id = 0;
            }
        lslam_matcher_destroy(it->second.h);
      }
      g_gone[0] += it->second.calls;
      g_gone[1] += it->second.cached_calls;
      g_gone[6] += it->second.ns_total;
      g_gone[7] += it->second.ns_device;
      registry().erase(it);
    }
  }
  delete m_pCorrelationGrid;
  delete m_pSearchSpaceProbs;
  delete m_pGridLookup;
}

kt_double ScanMatcher::MatchScan(LocalizedRangeScan* pScan, const LocalizedRangeScanVector& rBaseScans, Pose2& rMean,
                                 Matrix3& rCovariance, kt_bool doPenalize, kt_bool doRefineMatch) {
  using namespace lslam_karto;
  // Mapper.cpp:195-209: a scan without readings cannot be matched
  Pose2 scanPose = pScan->GetSensorPose();
  if (pScan->GetNumberOfRangeReadings() == 0) {
    rMean = scanPose;
    rCovariance(0, 0) = kMaxVariance;
    rCovariance(1, 1) = kMaxVariance;
    rCovariance(2, 2) = 4 * math::Square(m_pMapper->m_pCoarseAngleResolution->GetValue());
    return 0.0;
  }
  std::lock_guard<std::mutex> lock(lslam_karto::mutex());
  const auto t_in = std::chrono::steady_clock::now();

  // ---- which parameter set built this matcher (Mapper.cpp:1964-1968 sequential, :865-867 loop) ----------------------
  const kt_double resolution = m_pCorrelationGrid->GetResolution();
  const kt_int32s side = m_pSearchSpaceProbs->GetWidth();
  lslam_matcher_config cfg;
  lslam_matcher_config_defaults(&cfg);
  struct Set { kt_double dim, res, smear; };
  const Set sets[2] = {{m_pMapper->m_pCorrelationSearchSpaceDimension->GetValue(),
                        m_pMapper->m_pCorrelationSearchSpaceResolution->GetValue(),
                        m_pMapper->m_pCorrelationSearchSpaceSmearDeviation->GetValue()},
                       {m_pMapper->m_pLoopSearchSpaceDimension->GetValue(), m_pMapper->m_pLoopSearchSpaceResolution->GetValue(),
                        m_pMapper->m_pLoopSearchSpaceSmearDeviation->GetValue()}};
  int pick = -1;
  const bool sequential = this == m_pMapper->m_pSequentialScanMatcher;
  const int first = sequential ? 0 : 1;
  for (int t = 0; t < 2 && pick < 0; t++) {
    const Set& s = sets[(first + t) & 1];
    if (s.res == resolution && static_cast<kt_int32s>(math::Round(s.dim / s.res) + 1) == side) pick = (first + t) & 1;
  }
  if (pick < 0) throw std::runtime_error("lslam: ScanMatcher was not created from the Mapper's search-space parameters");
  cfg.search_size = sets[pick].dim;
  cfg.resolution = sets[pick].res;
  cfg.smear_deviation = sets[pick].smear;
  LaserRangeFinder* lrf = pScan->GetLaserRangeFinder();
  const lslam_laser laser = laser_from(lrf);
  cfg.range_threshold = laser.range_threshold;
  cfg.coarse_search_angle_offset = m_pMapper->m_pCoarseSearchAngleOffset->GetValue();
  cfg.coarse_angle_resolution = m_pMapper->m_pCoarseAngleResolution->GetValue();
  cfg.fine_search_angle_offset = m_pMapper->m_pFineSearchAngleOffset->GetValue();
  cfg.distance_variance_penalty = m_pMapper->m_pDistanceVariancePenalty->GetValue();
  cfg.angle_variance_penalty = m_pMapper->m_pAngleVariancePenalty->GetValue();
  cfg.minimum_distance_penalty = m_pMapper->m_pMinimumDistancePenalty->GetValue();
  cfg.minimum_angle_penalty = m_pMapper->m_pMinimumAnglePenalty->GetValue();
  cfg.use_response_expansion = m_pMapper->m_pUseResponseExpansion->GetValue() ? 1 : 0;

  // ---- the device matcher mirroring this instance (re-created when a parameter or the sensor changed) ---------------
  GpuMatcher& g = registry()[this];
  if (!g.h || !same_cfg(g.cfg, cfg) || !same_laser(g.laser, laser) || g.grid_w != m_pCorrelationGrid->GetWidth() ||
      g.grid_h != m_pCorrelationGrid->GetHeight()) {
    if (g.h) lslam_matcher_destroy(g.h);
    g.h = nullptr;
    int rc = lslam_matcher_create(context(), &cfg, &laser, &g.h);
    if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_matcher_create: ") + lslam_last_error(context()));
    int32_t info[8];
    double off[2];
    lslam_matcher_grid_info(g.h, info, off);
    // same correlation grid as ScanMatcher::Create built on the host (Mapper.cpp:147-171): the range threshold the
    // reference passed to Create is the laser's (Mapper.cpp:2011), so the two must agree
    if (info[0] != m_pCorrelationGrid->GetWidth() || info[1] != m_pCorrelationGrid->GetHeight() ||
        info[2] != m_pCorrelationGrid->GetWidthStep()) {
      lslam_matcher_destroy(g.h);
      g.h = nullptr;
      throw std::runtime_error("lslam: device correlation grid geometry differs from karto::CorrelationGrid");
    }
    g.cfg = cfg;
    g.laser = laser;
    g.grid_w = info[0];
    g.grid_h = info[1];
  }

  g.sequential = sequential;
  const int nb = lslam_matcher_num_beams(g.h);
  const size_t stride = static_cast<size_t>(nb > 0 ? nb : 1);
  const size_t n_base = rBaseScans.size();
  if (static_cast<int>(pScan->GetNumberOfRangeReadings()) < nb) throw std::runtime_error("lslam: scan has too few readings");
  bool managed = cache_enabled();
  for (size_t i = 0; i < n_base; i++) {
    LocalizedRangeScan* b = rBaseScans[i];
    if (b->GetLaserRangeFinder() != lrf && !same_laser(laser_from(b->GetLaserRangeFinder()), laser))
      throw std::runtime_error("lslam: base scans of a different LaserRangeFinder are not supported");
    if (static_cast<int>(b->GetNumberOfRangeReadings()) < nb) throw std::runtime_error("lslam: base scan has too few readings");
    managed = managed && b->GetUniqueId() >= 0;
  }
  g.poses.resize(n_base * 3);
  for (size_t i = 0; i < n_base; i++) {
    const Pose2 sp = rBaseScans[i]->GetSensorPose();  // LocalizedRangeScan -> SENSOR pose (Karto.h:5020,5280)
    g.poses[3 * i] = sp.GetX();
    g.poses[3 * i + 1] = sp.GetY();
    g.poses[3 * i + 2] = sp.GetHeading();
  }
  const double q[3] = {scanPose.GetX(), scanPose.GetY(), scanPose.GetHeading()};
  lslam_match_result r;
  int rc;
  auto t_dev = t_in;
  if (managed) {
    // ---- base scans by id; the query's readings go up once (or not at all) ----------------------------------------
    Residents& R = residents_for(laser);
    const uint64_t call_clock = R.clock + 1;  // entries touched from here on are this call's: never evicted by it
    g.ids.resize(n_base);
    for (size_t i = 0; i < n_base; i++) g.ids[i] = resident_id(R, rBaseScans[i], nb);
    int flags = (doPenalize ? LSLAM_MATCH_PENALIZE : 0) | (doRefineMatch ? LSLAM_MATCH_REFINE : 0);
    int64_t qid = -1;
    const kt_double* qr = pScan->GetRangeReadings();
    bool process_call = false;
    if (ResidentScan* e = find_resident(R, pScan, nb)) {
      qid = e->id;  // a managed scan matched again (LinkNearChains, TryCloseLoop's coarse match): nothing to send
      qr = nullptr;
    } else if (pScan->GetUniqueId() < 0 && sequential && doPenalize && doRefineMatch) {
      // Mapper::Process (Mapper.cpp:2040-2044): the scan is added to the running window next; keep its readings
      // (provisionally: AddScan numbers it after this call)
      auto it = R.scans.find(pScan);
      ResidentScan e;
      e.id = it != R.scans.end() ? it->second.id : R.next_id++;
      e.unique_id = -1;
      e.readings = qr;
      e.checksum = checksum(qr, nb);
      e.fingerprint = fingerprint(qr, nb);
      e.last_use = ++R.clock;
      R.scans[pScan] = e;
      qid = e.id;
      if (lslam_scan_cache_contains(R.cache, qid)) lslam_scan_cache_forget(R.cache, qid);  // stale provisional entry
      process_call = true;
    }  // else: a temporary (TryCloseLoop's stack scan) or an unmanaged scan: anonymous query, nothing kept
    t_dev = std::chrono::steady_clock::now();
    rc = lslam_matcher_match_scan_cached(g.h, R.cache, static_cast<int>(n_base), g.ids.data(), g.poses.data(), qid, qr, q,
                                         flags, &r);
    if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_matcher_match_scan_cached: ") + lslam_last_error(context()));
    g.cached_calls++;
    evict_beyond_capacity(R, call_clock);
    if (process_call && r.status == LSLAM_OK) {
      // prepare the scan (world points + FindValidPoints anchors) at the pose AddEdges is about to give it, behind this
      // call: the kernel runs while the reference's host code (AddScan, AddEdges, the graph search) does
      Matrix3 cov = rCovariance;
      cov(0, 0) = r.covariance[0];
      cov(0, 1) = r.covariance[1];
      cov(1, 0) = r.covariance[3];
      cov(1, 1) = r.covariance[4];
      cov(2, 2) = r.covariance[8];
      const Pose2 next = PredictPoseAfterAddEdges(Pose2(r.pose[0], r.pose[1], r.pose[2]), cov);
      // SetSensorPose(next) followed by GetSensorPose() (Karto.h:5280-5313): the identity for a laser at the robot's centre, a
      // round trip through the robot pose otherwise
      double robot[3], sensor[3];
      const double np[3] = {next.GetX(), next.GetY(), next.GetHeading()};
      lslam_robot_pose_from_sensor(&laser, np, robot);
      lslam_sensor_pose_from_robot(&laser, robot, sensor);
      (void)lslam_scan_cache_prepare(R.cache, qid, sensor);
    }
  } else {
    // ---- literal forwarding: every base scan's readings cross the bus ------------------------------------------------
    g.ranges.resize(n_base * stride);
    for (size_t i = 0; i < n_base; i++)
      std::memcpy(&g.ranges[i * stride], rBaseScans[i]->GetRangeReadings(), sizeof(double) * static_cast<size_t>(nb));
    t_dev = std::chrono::steady_clock::now();
    rc = lslam_matcher_match_scan(g.h, static_cast<int>(n_base), g.ranges.data(), static_cast<int>(stride), g.poses.data(),
                                  pScan->GetRangeReadings(), q, doPenalize ? 1 : 0, doRefineMatch ? 1 : 0, &r);
    if (rc != LSLAM_OK) throw std::runtime_error(std::string("lslam_matcher_match_scan: ") + lslam_last_error(context()));
  }
  {
    const auto t_out = std::chrono::steady_clock::now();
    g.ns_device += std::chrono::duration_cast<std::chrono::nanoseconds>(t_out - t_dev).count();
    g.ns_total += std::chrono::duration_cast<std::chrono::nanoseconds>(t_out - t_in).count();
  }
  g.calls++;
  // the reference throws from inside CorrelateScan (Mapper.cpp:444-447, 484-487)
  if (r.status == LSLAM_ERR_PROBABILITY_SEARCH)
    throw std::runtime_error("Mapper FATAL ERROR - Index out of range in probability search!");
  if (r.status == LSLAM_ERR_NO_BEST_POSE) throw std::runtime_error("Mapper FATAL ERROR - Unable to find best position");
  if (r.status != LSLAM_OK) throw std::runtime_error("lslam: scan match failed");
  rMean = Pose2(r.pose[0], r.pose[1], r.pose[2]);
  // the entries CorrelateScan writes (Mapper.cpp:535-692); the others keep what the caller put there
  rCovariance(0, 0) = r.covariance[0];
  rCovariance(0, 1) = r.covariance[1];
  rCovariance(1, 0) = r.covariance[3];
  rCovariance(1, 1) = r.covariance[4];
  rCovariance(2, 2) = r.covariance[8];
  return r.response;
}

}  // namespace karto
