// Streaming front-end: karto::Mapper::Process (Mapper.cpp:1999-2079) with every processed scan resident in
// HBM.  Included at the end of scan_matcher.hip (it drives that file's kernels).
//
// Per processed scan:
//   host   lastTransform propagation (Mapper.cpp:2021-2025), HasMovedEnough (:2087-2120)
//   device MatchScan against the running scans (:2037-2045): AddScans of the window around the query
//          (find_valid, mark + smear; parity planes refreshed), coarse + fine search
//   host   MapperGraph::AddEdges (:901-973): LinkScans to the previous scan, LinkChainToScan to the running scans,
//          LinkNearChains -- breadth-first FindNearLinkedScans over the pose graph, FindNearChains, one device
//          MatchScan per near chain -- and the closing SetSensorPose(ComputeWeightedMean(means, covariances))
//   device world points of the scan at its final pose (LocalizedRangeScan::Update, Karto.h:5362-5428) -- computed
//          once, reused by every later grid rebuild (running window, near chains, loop-closure chains)
//   host   AddRunningScan window policy (Mapper.h:1365-1386)
//   both   MapperGraph::TryCloseLoop (:976-1051): FindPossibleLoopClosure on the host, coarse MatchScan of each
//          candidate chain on the LOOP matcher (second lslam_matcher: 8 m search space -> 81 x 81 x 21 lattice),
//          fine MatchScan on the sequential matcher, SetSensorPose + LinkChainToScan when both pass
// What stays out: the ScanSolver behind CorrectPoses() (spa / g2o / ceres / gtsam are third-party back-ends; the
// reference's own library ships none, Mapper.cpp:1392-1412).  Without a solver a closed loop re-poses the closing
// scan and links it, exactly what karto::Mapper does when no optimizer is attached.
//
// Chains are always contiguous ranges of scan ids (running window, FindNearChains, FindPossibleLoopClosure all grow
// them over consecutive state ids), so a chain is (first id, count) into the resident world-point array.
//
// Numerics: the graph decisions compare squared distances between scan barycentres with thresholds.  The barycentre
// (mean of the filtered world points) is evaluated as sensor + R(heading) * mean(r_i (cos a_i, sin a_i)) instead of
// summing the world points one by one; it differs from the reference's sum by ~1e-15 m, which can change a decision
// only when a distance sits within that of a threshold.
#pragma once

#include <cfloat>
#include <queue>

struct lslam_frontend_scan {
  double odom[3], robot[3], sensor[3];
  double bary[2];           // GetBarycenterPose position, or the sensor position when the scan has no filtered reading
  double blx = 0, bly = 0;  // sum of r_i (cos a_i, sin a_i) over the filtered readings (scan frame)
  int nfilt = 0;
  double time = 0;
  std::vector<std::pair<int, int>> edges;  // Vertex::m_Edges: (source, target) in insertion order
};

// One more loop matcher on its own HIP stream: the candidate chains of one TryCloseLoop are matched side by side
struct lslam_loop_slot {
  lslam_context* ctx = nullptr;  // owned: a second stream on the matcher's device
  lslam_matcher* m = nullptr;    // owned: same parameters as loop_m
  DevBuf<double> d_q;
  lslam_match_result* h_res = nullptr;  // pinned
};

struct lslam_frontend {
  lslam_matcher* m = nullptr;
  lslam_matcher* loop_m = nullptr;  // MapperGraph::m_pLoopScanMatcher (owned)
  std::vector<lslam_loop_slot> loop_pool;  // slots 1..P-1 of the speculative loop search (slot 0 = loop_m itself)
  hipEvent_t ev_ready = nullptr;           // "everything the main stream wrote so far", awaited by the pool streams
  int64_t n_loop_discarded = 0;            // speculative coarse matches thrown away because an earlier chain closed
  lslam_frontend_config cfg;
  std::vector<lslam_frontend_scan> scans;  // MapperSensorManager::GetScans, by state id
  int run_start = 0, run_count = 0;        // running scans = ids [run_start, run_start + run_count)
  int cap = 0;                             // scans the device arrays hold
  double2* d_world = nullptr;              // [cap][n] world points at the scans' current poses
  double* d_ranges = nullptr;              // [cap][n] readings (kept to re-pose a scan after a closed loop)
  int* d_next = nullptr;                   // [cap][n + 1] k_anchor_chain rows: FindValidPoints' anchors of the world points
  // the newest scan's anchor chain worked out beside its own match (anchor_spec_block): [ok, count, anchors (n), classes
  // (n bytes, padded), fall-backs]; spec_chain_id = the scan it belongs to (-1: none)
  int* d_spec_chain = nullptr;
  int spec_chain_id = -1;
  bool cfg_spec_chain = true;              // LSLAM_FE_SPEC_CHAIN=0 in the environment: off (A/B)
  int64_t n_spec_chain_used = 0;           // k_anchor_chain launches that were handed a speculative chain
  DevBuf<double> d_q;                      // query pose (3)
  DevBuf<lslam_match_result> d_res;
  lslam_match_result* h_res = nullptr;     // pinned host memory the match's last kernel writes its record to
  double* h_ranges = nullptr;              // pinned staging of the new scan's readings (n)
  const double* pending_ranges = nullptr;  // staged readings the next fe_match carries into HBM (its first kernel)
  std::vector<double> cos_a, sin_a;        // cos / sin of minimum_angle + i * angular_resolution
  bool have_last = false;
  int64_t n_chain_matches = 0, n_loop_coarse = 0, n_loop_fine = 0, n_loops_closed = 0, n_edges = 0;
  // ---- look-ahead (lslam_frontend_process_many) ---------------------------------------------------------------------
  // The loop search of scan t is a chain of lone, latency-bound matches on the loop matchers, and 98 % of them close
  // nothing.  When the caller has handed over scan t + 1 already, its running-window match -- which reads the poses as
  // they stand, exactly what the sequential walk reads unless a loop closes -- is enqueued on the main matcher WHILE the
  // loop candidates of scan t are in flight.  Process(t + 1) then finds its match done.  A loop that does close re-poses
  // scan t: the speculative match is discarded and redone (every accepted result is the sequential walk's).
  struct LookAhead {
    const double* ranges;
    const double* odom;
    double time;
  };
  const LookAhead* la = nullptr;  // the scan after the one being processed (process_many only)
  struct Spec {
    bool started = false, finished = false, invalid = false;
    bool chain_launched = false;       // its launch carried the scan's speculative anchor chain
    const double* ranges = nullptr;
    double odom[3] = {0, 0, 0}, time = 0;
    int id = -1;
    double sp[3] = {0, 0, 0};          // the sensor pose the match was started from
    double last_robot[3] = {0, 0, 0};  // the previous scan's robot pose it was derived from
    lslam_match_result r;
    int rc = 0;
  } spec;
  lslam_match_result* h_res_spec = nullptr;  // pinned: the speculative match's record
  DevBuf<double> d_q_spec;
  int64_t n_spec_started = 0, n_spec_used = 0, n_spec_discarded = 0;
};

namespace {

inline double sq_dist2(const double* a, const double* b) { return ksq(a[0] - b[0]) + ksq(a[1] - b[1]); }
inline size_t spec_chain_bytes(int n) { return (size_t)(2 + std::max(n, 1)) * 4 + (((size_t)std::max(n, 1) + 3) & ~(size_t)3) + 4; }
inline size_t fe_anchor_lds(int n) { return (size_t)n * (sizeof(double2) + 13) + 16; }  // points, three tables, reach

int fe_grow(lslam_frontend* f, int need) {
  if (need <= f->cap) return LSLAM_OK;
  lslam_context* ctx = f->m->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));  // a pool may have left the calling thread on another device
  const size_t n = (size_t)std::max(f->m->g.n_beams, 1);
  int cap = std::max(256, f->cap);
  while (cap < need) cap *= 2;
  double2* w = nullptr;
  double* r = nullptr;
  int* nx = nullptr;
  if (hipMalloc((void**)&w, (size_t)cap * n * sizeof(double2)) != hipSuccess ||
      hipMalloc((void**)&r, (size_t)cap * n * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&nx, (size_t)cap * (n + 1) * sizeof(int)) != hipSuccess) {
    (void)hipGetLastError();
    if (w) (void)hipFree(w);
    if (r) (void)hipFree(r);
    return ctx->fail(LSLAM_ERR_HIP, "cannot keep %d scans resident in HBM", cap);
  }
  if (f->cap > 0) {
    LSLAM_HIP(ctx, hipMemcpyAsync(w, f->d_world, (size_t)f->cap * n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
    LSLAM_HIP(ctx, hipMemcpyAsync(r, f->d_ranges, (size_t)f->cap * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    LSLAM_HIP(ctx, hipMemcpyAsync(nx, f->d_next, (size_t)f->cap * (n + 1) * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(f->d_world);
    (void)hipFree(f->d_ranges);
    (void)hipFree(f->d_next);
  }
  f->d_world = w;
  f->d_ranges = r;
  f->d_next = nx;
  f->cap = cap;
  return LSLAM_OK;
}

// SetSensorPose (Karto.h:5297-5313) + the barycentre of the filtered readings at that pose (Karto.h:5362-5416)
void fe_set_sensor_pose(const lslam_frontend* f, lslam_frontend_scan& s, const double sensor[3]) {
  for (int i = 0; i < 3; i++) s.sensor[i] = sensor[i];
  lslam_robot_pose_from_sensor(&f->m->laser, sensor, s.robot);
  if (s.nfilt > 0) {
    const double c = cos(sensor[2]), sn = sin(sensor[2]);
    s.bary[0] = sensor[0] + (c * s.blx - sn * s.bly) / (double)s.nfilt;
    s.bary[1] = sensor[1] + (sn * s.blx + c * s.bly) / (double)s.nfilt;
  } else {
    s.bary[0] = sensor[0];
    s.bary[1] = sensor[1];
  }
}
// GetReferencePose(m_pUseScanBarycenter) position
inline const double* fe_ref(const lslam_frontend* f, const lslam_frontend_scan& s) {
  return f->cfg.use_scan_barycenter ? s.bary : s.sensor;
}

// world points of scan `id` at its current sensor pose into its resident slot
int fe_update_world(lslam_frontend* f, int id) {
  lslam_matcher* m = f->m;
  lslam_context* ctx = m->ctx;
  const int n = m->g.n_beams;
  if (n <= 0) return LSLAM_OK;
  // No host synchronisation: the pose goes through its own slot of a small ring (the copy of a pageable host buffer is
  // staged before hipMemcpyAsync returns; the stream orders it before the kernel, and the kernel before the next
  // grid rebuild that reads these points).
  PoseArg pv;  // the pose travels as a kernel argument: no copy on the stream in front of the kernel
  for (int i = 0; i < 3; i++) pv.v[i] = f->scans[id].sensor[i];
  if (fe_anchor_lds(n) <= 60 * 1024) {
    // world points AND FindValidPoints' anchor chain of the scan at this pose in one launch: the chain once here, not once
    // per rebuild of every window the scan will be part of
    // the chain worked out beside the scan's own match, if it was (the kernel takes it over only when it is provably the
    // chain of the final points: see k_anchor_chain)
    const int* spec = (f->spec_chain_id == id && f->d_spec_chain) ? f->d_spec_chain : (const int*)nullptr;
    f->spec_chain_id = -1;
    if (spec) f->n_spec_chain_used++;
    launch(ctx, "scan_prep_base", k_anchor_chain, dim3(1), dim3(n > 512 ? 1024 : 256), fe_anchor_lds(n), n,
           f->d_world + (size_t)id * n, f->d_next + (size_t)id * (n + 1), (const double*)(f->d_ranges + (size_t)id * n), pv,
           m->g, spec);
  } else {
    launch(ctx, "scan_prep_base", k_scan_prep<double>, dim3((n + 255) / 256, 1), dim3(256), 0,
           (const double*)(f->d_ranges + (size_t)id * n), n, (const double*)nullptr, m->g, (double2*)nullptr,
           f->d_world + (size_t)id * n, PassCfg{}, (Lattice*)nullptr, (double2*)nullptr, 0, pv);
  }
  return LSLAM_OK;
}

// ScanMatcher::MatchScan(pScan, chain, mean, covariance, doPenalize, doRefineMatch) (Mapper.cpp:184-291) of resident
// scan `id`, posed at `sensor`, against the resident scans [first, first + count): ENQUEUE on m's stream (grid rebuild +
// search; the last kernel writes the 112-byte record into the pinned h_res), fe_match_finish waits for it.
int fe_match_enqueue(lslam_frontend* f, lslam_matcher* m, double* d_q, lslam_match_result* h_res, int id, const double sensor[3],
                     int first, int count, int do_penalize, int do_refine) {
  const int n = f->m->g.n_beams;
  // the grid rebuild's first kernel also carries the query pose into device memory and clears the numerators of the
  // match's first (beam-sliced) pass: two stream operations fewer per scan
  RebuildExtras x{};
  for (int i = 0; i < 3; i++) x.pose[i] = sensor[i];
  x.pose_dst = d_q;
  x.zero = m->d_resp.p;
  x.anchor_ring = fe_anchor_lds(n) <= 60 * 1024 ? f->d_next : nullptr;
  x.zero_words = (int)std::min<size_t>(m->d_resp.cap, (size_t)1 << 16);
  x.prep_ranges = f->d_ranges + (size_t)id * n;  // the match's k_scan_prep rides along too (extra blocks)
  if (f->pending_ranges) {  // the scan's readings were only staged: the same kernel brings them into HBM
    x.ranges_src = f->pending_ranges;
    x.ranges_dst = f->d_ranges + (size_t)id * n;
    x.n_ranges = n;
    x.prep_ranges = f->pending_ranges;  // ... and its prep blocks read the staged copy (the row is being written)
    f->pending_ranges = nullptr;
  }
  int rc = rebuild_grid_dev(m, f->d_world, first, count, f->cap, sensor, &x);
  if (rc) return rc;
  // the last kernel of the match writes the 112-byte record straight into pinned host memory -- no copy operation -- and
  // posts a ticket behind it, on which fe_match_finish spins (arm_done_ticket / wait_record)
  arm_done_ticket(m);
  return match_batch_impl<double>(m, 1, f->d_ranges + (size_t)id * n, n, d_q, do_penalize, do_refine, h_res, nullptr, 0);
}
int fe_match_finish(lslam_matcher* m, const lslam_match_result* h_res, lslam_match_result* out) {
  lslam_context* ctx = m->ctx;
  int rcw = wait_record(m);
  if (rcw) return rcw;
  *out = *h_res;
  if (out->status != LSLAM_OK) return ctx->fail(out->status, "scan matcher: the reference throws here");
  return LSLAM_OK;
}
int fe_match(lslam_frontend* f, lslam_matcher* m, int id, const double sensor[3], int first, int count, int do_penalize,
             int do_refine, lslam_match_result* out) {
  const int n = f->m->g.n_beams;
  if (n <= 0) {  // scan without readings (Mapper.cpp:199-209): rMean = scanPose
    memset(out, 0, sizeof *out);
    for (int i = 0; i < 3; i++) out->pose[i] = sensor[i];
    out->covariance[0] = out->covariance[4] = kMaxVariance;
    out->covariance[8] = 4 * ksq(m->cfg.coarse_angle_resolution);
    return LSLAM_OK;
  }
  int rc = fe_match_enqueue(f, m, f->d_q.p, f->h_res, id, sensor, first, count, do_penalize, do_refine);
  if (rc) return rc;
  rc = fe_match_finish(m, f->h_res, out);
  if (rc && m->ctx != f->m->ctx) f->m->ctx->last_error = m->ctx->last_error;
  return rc;
}

// MapperGraph::AddEdge + LinkScans (Mapper.cpp:1072-1121): a new edge unless `from` already has one whose target is `to`
void fe_link_scans(lslam_frontend* f, int from, int to) {
  for (const auto& e : f->scans[from].edges)
    if (e.second == to) return;
  f->scans[from].edges.emplace_back(from, to);  // Edge ctor: source->AddEdge, target->AddEdge (Mapper.h:304-309)
  f->scans[to].edges.emplace_back(from, to);
  f->n_edges++;
}

// LinkChainToScan (Mapper.cpp:1152-1167) with GetClosestScanToPose (:1054-1070)
void fe_link_chain_to_scan(lslam_frontend* f, int first, int count, int id) {
  const double* pose = fe_ref(f, f->scans[id]);
  int closest = -1;
  double best = DBL_MAX;
  for (int i = first; i < first + count; i++) {
    const double d = sq_dist2(pose, fe_ref(f, f->scans[i]));
    if (d < best) {
      best = d;
      closest = i;
    }
  }
  if (closest < 0) return;
  if (best < ksq(f->cfg.link_scan_maximum_distance) + kTol) fe_link_scans(f, closest, id);
}

// FindNearLinkedScans (Mapper.cpp:1277-1286): BreadthFirstTraversal + NearScanVisitor (Mapper.h:542-643)
std::vector<int> fe_near_linked(const lslam_frontend* f, int id, double max_distance) {
  const double* center = fe_ref(f, f->scans[id]);
  const double max2 = ksq(max_distance);
  std::queue<int> to_visit;
  std::vector<char> seen(f->scans.size(), 0);
  std::vector<int> valid;
  to_visit.push(id);
  seen[id] = 1;
  do {
    const int v = to_visit.front();
    to_visit.pop();
    if (sq_dist2(fe_ref(f, f->scans[v]), center) <= max2 - kTol) {
      valid.push_back(v);
      for (const auto& e : f->scans[v].edges) {  // GetAdjacentVertices (Mapper.h:251-272)
        const int adj[2] = {e.first, e.second};
        for (int k = 0; k < 2; k++)
          if (adj[k] != v && !seen[adj[k]]) {
            to_visit.push(adj[k]);
            seen[adj[k]] = 1;
          }
      }
    }
  } while (!to_visit.empty());
  return valid;
}

// FindNearChains (Mapper.cpp:1170-1274): (first id, count) of every valid chain
std::vector<std::pair<int, int>> fe_near_chains(const lslam_frontend* f, int id) {
  std::vector<std::pair<int, int>> chains;
  const double* scan_pose = fe_ref(f, f->scans[id]);
  const double lim = ksq(f->cfg.link_scan_maximum_distance) + kTol;
  std::vector<char> processed(f->scans.size(), 0);
  const std::vector<int> near = fe_near_linked(f, id, f->cfg.link_scan_maximum_distance);
  const int end = (int)f->scans.size();
  for (int near_id : near) {
    if (near_id == id) continue;
    if (processed[near_id]) continue;
    processed[near_id] = 1;
    bool valid = true;
    int lo = near_id, hi = near_id;
    for (int c = near_id - 1; c >= 0; c--) {
      if (c == id) valid = false;
      if (sq_dist2(scan_pose, fe_ref(f, f->scans[c])) < lim) {
        lo = c;
        processed[c] = 1;
      } else {
        break;
      }
    }
    for (int c = near_id + 1; c < end; c++) {
      if (c == id) valid = false;
      if (sq_dist2(scan_pose, fe_ref(f, f->scans[c])) < lim) {
        hi = c;
        processed[c] = 1;
      } else {
        break;
      }
    }
    if (valid) chains.emplace_back(lo, hi - lo + 1);
  }
  return chains;
}

// FindPossibleLoopClosure (Mapper.cpp:1332-1390); rStartNum advances across calls
std::pair<int, int> fe_possible_loop_closure(const lslam_frontend* f, int id, const std::vector<char>& is_near_linked,
                                             int& start_num) {
  const double* pose = fe_ref(f, f->scans[id]);
  const double lim = ksq(f->cfg.loop_search_maximum_distance) + kTol;
  const int n_scans = (int)f->scans.size();
  int first = 0, count = 0;
  for (; start_num < n_scans; start_num++) {
    const int c = start_num;
    if (sq_dist2(fe_ref(f, f->scans[c]), pose) < lim) {
      if (is_near_linked[c]) {
        count = 0;  // chain.clear(): a linked scan cannot be in the chain
      } else {
        if (count == 0) first = c;
        count++;
      }
    } else {
      if (count >= f->cfg.loop_match_minimum_chain_size) return {first, count};
      count = 0;
    }
  }
  return {first, count};  // the reference returns whatever is left when the scans run out
}

// Mapper::Process up to the match (Mapper.cpp:2021-2031, 2087-2120): the odometry-corrected robot pose of a new scan and
// HasMovedEnough, from the last processed scan as it stands
void fe_prologue(const lslam_frontend* f, const double odom_pose[3], double time_s, double corrected[3], bool* moved) {
  const lslam_laser* laser = &f->m->laser;
  const lslam_frontend_scan& last = f->scans.back();
  PoseXform t = pose_xform(last.odom, last.robot);  // :2021-2025
  pose_xform_apply(t, odom_pose, corrected);
  // HasMovedEnough (:2087-2120): time first, then the ODOMETRIC sensor poses
  bool mv = (time_s - last.time) >= f->cfg.minimum_time_interval;
  if (!mv) {
    double lsp[3], csp[3];
    lslam_sensor_pose_from_robot(laser, last.odom, lsp);
    lslam_sensor_pose_from_robot(laser, odom_pose, csp);
    const double dh = normalize_angle(csp[2] - lsp[2]);
    mv = fabs(dh) >= f->cfg.minimum_travel_heading;
    if (!mv) mv = sq_dist2(lsp, csp) >= ksq(f->cfg.minimum_travel_distance) - kTol;
  }
  *moved = mv;
}

// wait for the speculative match (if one is in flight) and keep its record
void fe_spec_finish(lslam_frontend* f) {
  if (!f->spec.started || f->spec.finished) return;
  f->spec.rc = fe_match_finish(f->m, f->h_res_spec, &f->spec.r);
  f->spec.finished = true;
}

// Enqueue the running-window match of the NEXT scan (f->la) on the main matcher, now -- called by the loop search once
// its first candidates are in flight.  Nothing of the graph or the scan list is touched: Process(t + 1) repeats the
// (cheap) prologue and accepts the record only if it started from the very same poses.
void fe_spec_try_start(lslam_frontend* f) {
  if (!f->la || f->spec.started || !f->have_last || !f->h_res_spec) return;
  lslam_matcher* m = f->m;
  const int n = m->g.n_beams;
  const int id = (int)f->scans.size();
  if (n <= 0 || id + 1 > f->cap) return;  // (growing the resident arrays frees the old ones: not under matches in flight)
  double corrected[3];
  bool moved = false;
  fe_prologue(f, f->la->odom, f->la->time, corrected, &moved);
  if (!moved) return;  // Process(t + 1) will reject the scan: nothing to match
  lslam_frontend::Spec& sp = f->spec;
  sp = lslam_frontend::Spec{};
  sp.ranges = f->la->ranges;
  for (int i = 0; i < 3; i++) sp.odom[i] = f->la->odom[i];
  sp.time = f->la->time;
  sp.id = id;
  for (int i = 0; i < 3; i++) sp.last_robot[i] = f->scans.back().robot[i];
  lslam_sensor_pose_from_robot(&m->laser, corrected, sp.sp);
  memcpy(f->h_ranges, f->la->ranges, (size_t)n * sizeof(double));  // staged: the match's first kernel moves them into HBM
  f->pending_ranges = f->h_ranges;
  if (f->d_spec_chain && f->cfg_spec_chain) {  // ... and its speculative anchor chain (see k_anchor_chain)
    m->spec_req.ranges = f->d_ranges + (size_t)id * n;
    for (int i = 0; i < 3; i++) m->spec_req.pose[i] = sp.sp[i];
    m->spec_req.out = f->d_spec_chain;
    m->spec_armed = true;
  }
  const int rc = fe_match_enqueue(f, m, f->d_q_spec.p, f->h_res_spec, id, sp.sp, f->run_start, f->run_count, 1, 1);
  sp.chain_launched = rc == 0 && m->spec_launched;
  f->pending_ranges = nullptr;
  if (rc) {  // could not even be enqueued: Process(t + 1) does it the plain way (and reports whatever is wrong)
    (void)hipStreamSynchronize(m->ctx->stream);
    return;
  }
  sp.started = true;
  f->n_spec_started++;
}

// TryCloseLoop (Mapper.cpp:976-1051), one sensor.
//
// The reference matches the candidate chains strictly one after another; 98.7 % of those coarse matches close nothing
// (10 000-scan run), and each is a lone ~0.1 ms chain of small kernels on an otherwise idle chip.  A chain's coarse
// match reads only the graph and the poses as they stand, and neither changes unless a loop CLOSES.  So the next few
// chains FindPossibleLoopClosure would return are enumerated up front, their coarse matches run side by side on a small
// pool of loop matchers (own grids and workspaces, own streams), and the results are consumed in the reference's order.
// When a chain closes (scan re-posed, edge added) the speculative results behind it are discarded and the search resumes
// from the reference's rStartNum with the new state -- every accepted result is the one the sequential walk computes.
int fe_close_after_coarse(lslam_frontend* f, int id, const std::pair<int, int>& chain, const lslam_match_result& coarse,
                          bool* closed) {
  *closed = false;
  if (coarse.response > f->cfg.loop_match_minimum_response_coarse &&
      coarse.covariance[0] < f->cfg.loop_match_maximum_variance_coarse &&
      coarse.covariance[4] < f->cfg.loop_match_maximum_variance_coarse) {
    lslam_match_result fine;  // tmpScan.SetSensorPose(bestPose); MatchScan(&tmpScan, chain, ..., false)
    fe_spec_finish(f);  // the fine match runs on the sequential matcher: a look-ahead match in flight there completes first
    int rc = fe_match(f, f->m, id, coarse.pose, chain.first, chain.second, 0, 1, &fine);
    if (rc) return rc;
    f->n_loop_fine++;
    if (!(fine.response < f->cfg.loop_match_minimum_response_fine)) {
      fe_set_sensor_pose(f, f->scans[id], fine.pose);  // pScan->SetSensorPose(bestPose)
      rc = fe_update_world(f, id);
      if (rc) return rc;
      fe_link_chain_to_scan(f, chain.first, chain.second, id);
      f->n_loops_closed++;  // CorrectPoses(): no ScanSolver attached
      *closed = true;
      if (f->spec.started) f->spec.invalid = true;  // the look-ahead match read this scan at its old pose
    }
  }
  return LSLAM_OK;
}

int fe_try_close_loop(lslam_frontend* f, int id) {
  lslam_context* ctx = f->m->ctx;
  int start_num = 0;
  // speculation is off while kernels are being timed (the pool's streams are not the profiled one)
  const size_t width = ctx->timer.enabled ? 1 : 1 + f->loop_pool.size();
  for (;;) {
    // FindPossibleLoopClosure recomputes the near-linked set on every call (the graph may have gained an edge)
    std::vector<char> linked(f->scans.size(), 0);
    for (int v : fe_near_linked(f, id, f->cfg.loop_search_maximum_distance)) linked[v] = 1;
    std::vector<std::pair<int, int>> chains;
    std::vector<int> resume;  // rStartNum after each chain was found
    int sn = start_num;
    while (chains.size() < width) {
      const std::pair<int, int> c = fe_possible_loop_closure(f, id, linked, sn);
      if (c.second <= 0) break;
      chains.push_back(c);
      resume.push_back(sn);
    }
    if (chains.empty()) return LSLAM_OK;
    bool closed = false;
    // which matcher takes candidate k of a round.  With a look-ahead scan waiting (process_many) the pool's matchers --
    // own streams -- go first and the loop matcher proper, which shares the main stream with the sequential matcher, last:
    // the look-ahead match is enqueued on that stream and must not queue up behind a candidate.
    const bool ahead = f->la != nullptr && !f->loop_pool.empty() && !ctx->timer.enabled;
    struct Slot { lslam_matcher* m; double* d_q; lslam_match_result* h_res; bool pooled; };
    auto slot_of = [&](size_t k) -> Slot {
      const size_t P = f->loop_pool.size();
      if (ahead) {
        if (k < P) return Slot{f->loop_pool[k].m, f->loop_pool[k].d_q.p, f->loop_pool[k].h_res, true};
        return Slot{f->loop_m, f->d_q.p, f->h_res, false};
      }
      if (k == 0) return Slot{f->loop_m, f->d_q.p, f->h_res, false};
      return Slot{f->loop_pool[k - 1].m, f->loop_pool[k - 1].d_q.p, f->loop_pool[k - 1].h_res, true};
    };
    if (chains.size() == 1 && !ahead) {
      lslam_match_result coarse;
      int rc = fe_match(f, f->loop_m, id, f->scans[id].sensor, chains[0].first, chains[0].second, 0, 0, &coarse);
      if (rc) return rc;
      f->n_loop_coarse++;
      rc = fe_close_after_coarse(f, id, chains[0], coarse, &closed);
      if (rc) return rc;
      start_num = resume[0];
      continue;  // closed or not: the next round re-reads the graph, like the reference's next FindPossibleLoopClosure
    }
    // ---- several chains (or one, with a look-ahead scan waiting): all coarse matches in flight at once ----
    LSLAM_HIP(ctx, hipEventRecord(f->ev_ready, ctx->stream));  // the scan's world points etc. are ordered before this
    for (size_t k = 0; k < chains.size(); k++) {
      const Slot sl = slot_of(k);
      if (sl.pooled) LSLAM_HIP(ctx, hipStreamWaitEvent(sl.m->ctx->stream, f->ev_ready, 0));
      int rc = fe_match_enqueue(f, sl.m, sl.d_q, sl.h_res, id, f->scans[id].sensor, chains[k].first, chains[k].second, 0, 0);
      if (rc) {
        if (sl.pooled) ctx->last_error = sl.m->ctx->last_error;
        for (size_t j = 0; j < k; j++) (void)hipStreamSynchronize(slot_of(j).m->ctx->stream);
        return rc;
      }
    }
    fe_spec_try_start(f);  // the next scan's running-window match goes out now, under the candidates (no-op without look-ahead)
    // wait for all of them (the slowest sets the pace either way), then consume in the reference's order
    std::vector<lslam_match_result> res(chains.size());
    std::vector<int> rcs(chains.size(), LSLAM_OK);
    for (size_t k = 0; k < chains.size(); k++) {
      const Slot sl = slot_of(k);
      rcs[k] = fe_match_finish(sl.m, sl.h_res, &res[k]);
      if (rcs[k] && sl.pooled) ctx->last_error = sl.m->ctx->last_error;
    }
    size_t used = 0;
    for (size_t k = 0; k < chains.size(); k++) {
      if (closed) {  // a chain before this one closed the loop: the sequential walk would match this one against the new state
        f->n_loop_discarded++;
        continue;
      }
      if (rcs[k]) return rcs[k];
      f->n_loop_coarse++;
      used = k;
      int rc = fe_close_after_coarse(f, id, chains[k], res[k], &closed);
      if (rc) return rc;
    }
    start_num = resume[used];
  }
}

}  // namespace

extern "C" {

void lslam_frontend_config_defaults(lslam_frontend_config* c) {
  if (!c) return;
  memset(c, 0, sizeof *c);
  c->scan_buffer_size = 70;                      // Mapper.cpp:1501-1515
  c->scan_buffer_maximum_scan_distance = 20.0;
  c->minimum_travel_distance = 0.2;              // :1480-1499
  c->minimum_travel_heading = 10.0 * kPi180;
  c->minimum_time_interval = 3600.0;             // :1468-1478
  c->use_scan_barycenter = 1;                    // :1462-1466
  c->link_match_minimum_response_fine = 0.8;     // :1517-1521
  c->link_scan_maximum_distance = 10.0;          // :1523-1527
  c->loop_search_maximum_distance = 4.0;         // :1529-1533
  c->do_loop_closing = 1;                        // :1535-1538
  c->loop_match_minimum_chain_size = 10;         // :1540-1545
  c->loop_match_maximum_variance_coarse = 0.4 * 0.4;  // :1547-1552
  c->loop_match_minimum_response_coarse = 0.8;   // :1554-1558
  c->loop_match_minimum_response_fine = 0.8;     // :1560-1564
  c->loop_search_space_dimension = 8.0;          // :1586-1589
  c->loop_search_space_resolution = 0.05;        // :1591-1594
  c->loop_search_space_smear_deviation = 0.03;   // :1596-1600
}

int lslam_frontend_create_ex(lslam_matcher* m, const lslam_frontend_config* cfg, lslam_frontend** out) {
  if (!m || !cfg || !out || cfg->scan_buffer_size < 1) return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  lslam_context* ctx = m->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));  // every allocation below belongs to the matcher's device
  lslam_frontend* f = new lslam_frontend();
  f->m = m;
  f->cfg = *cfg;
  if (cfg->do_loop_closing) {  // MapperGraph ctor (Mapper.cpp:862-871): the loop matcher shares every other parameter
    lslam_matcher_config lc = m->cfg;
    lc.search_size = cfg->loop_search_space_dimension;
    lc.resolution = cfg->loop_search_space_resolution;
    lc.smear_deviation = cfg->loop_search_space_smear_deviation;
    int rc = lslam_matcher_create(ctx, &lc, &m->laser, &f->loop_m);
    if (rc) {
      delete f;
      return rc;
    }
    // the speculative loop search: P - 1 more loop matchers, each on its own stream of the same device
    // (LSLAM_FE_LOOP_POOL = P, default 4; 1 = the strictly sequential walk)
    int P = 4;
    if (const char* e = getenv("LSLAM_FE_LOOP_POOL")) P = std::max(1, std::min(16, atoi(e)));
    for (int k = 1; k < P; k++) {
      lslam_loop_slot sl;
      rc = lslam_create(ctx->device, &sl.ctx);
      if (rc == LSLAM_OK) rc = lslam_matcher_create(sl.ctx, &lc, &m->laser, &sl.m);
      if (rc == LSLAM_OK && (sl.d_q.reserve(4 + 4 * 8) != hipSuccess ||
                             hipHostMalloc((void**)&sl.h_res, sizeof(lslam_match_result), hipHostMallocDefault) != hipSuccess)) {
        (void)hipGetLastError();
        rc = ctx->fail(LSLAM_ERR_HIP, "cannot allocate a loop-matcher slot");
      }
      f->loop_pool.push_back(std::move(sl));  // pushed even when incomplete: lslam_frontend_destroy releases what exists
      if (rc) {
        lslam_frontend_destroy(f);
        return rc;
      }
    }
    if (hipEventCreateWithFlags(&f->ev_ready, hipEventDisableTiming) != hipSuccess) {
      (void)hipGetLastError();
      lslam_frontend_destroy(f);
      return ctx->fail(LSLAM_ERR_HIP, "hipEventCreate failed");
    }
  }
  if (f->d_q.reserve(4 + 4 * 8) != hipSuccess || f->d_res.reserve(1) != hipSuccess || f->d_q_spec.reserve(4 + 4 * 8) != hipSuccess ||
      hipHostMalloc((void**)&f->h_res, sizeof(lslam_match_result), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&f->h_res_spec, sizeof(lslam_match_result), hipHostMallocDefault) != hipSuccess ||
      hipHostMalloc((void**)&f->h_ranges, sizeof(double) * (size_t)std::max(m->g.n_beams, 1), hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void**)&f->d_spec_chain, spec_chain_bytes(m->g.n_beams)) != hipSuccess ||
      hipMemset(f->d_spec_chain, 0, spec_chain_bytes(m->g.n_beams)) != hipSuccess) {
    (void)hipGetLastError();
    lslam_frontend_destroy(f);
    return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the front-end scratch in HBM");
  }
  int rc = fe_grow(f, 256);
  if (rc) {
    lslam_frontend_destroy(f);
    return rc;
  }
  if (const char* e = getenv("LSLAM_FE_SPEC_CHAIN")) f->cfg_spec_chain = atoi(e) != 0;
  const int n = m->g.n_beams;
  f->cos_a.resize(n);
  f->sin_a.resize(n);
  for (int i = 0; i < n; i++) {
    const double a = m->laser.minimum_angle + (double)i * m->laser.angular_resolution;
    f->cos_a[i] = cos(a);
    f->sin_a[i] = sin(a);
  }
  *out = f;
  return LSLAM_OK;
}

// The four-parameter form of round 1: library defaults for the pose-graph side, loop closing off
int lslam_frontend_create(lslam_matcher* m, int scan_buffer_size, double scan_buffer_max_distance,
                          double min_travel_distance, double min_travel_heading, lslam_frontend** out) {
  lslam_frontend_config c;
  lslam_frontend_config_defaults(&c);
  c.scan_buffer_size = scan_buffer_size;
  c.scan_buffer_maximum_scan_distance = scan_buffer_max_distance;
  c.minimum_travel_distance = min_travel_distance;
  c.minimum_travel_heading = min_travel_heading;
  c.do_loop_closing = 0;
  return lslam_frontend_create_ex(m, &c, out);
}

void lslam_frontend_destroy(lslam_frontend* f) {
  if (!f) return;
  (void)hipSetDevice(f->m->ctx->device);
  (void)hipStreamSynchronize(f->m->ctx->stream);
  if (f->loop_m) lslam_matcher_destroy(f->loop_m);
  for (auto& sl : f->loop_pool) {
    if (sl.ctx) (void)hipStreamSynchronize(sl.ctx->stream);
    if (sl.m) lslam_matcher_destroy(sl.m);
    sl.d_q.release();
    if (sl.h_res) (void)hipHostFree(sl.h_res);
    if (sl.ctx) lslam_destroy(sl.ctx);
  }
  if (f->ev_ready) (void)hipEventDestroy(f->ev_ready);
  if (f->d_world) (void)hipFree(f->d_world);
  if (f->d_ranges) (void)hipFree(f->d_ranges);
  if (f->d_next) (void)hipFree(f->d_next);
  if (f->d_spec_chain) (void)hipFree(f->d_spec_chain);
  f->d_q.release();
  f->d_res.release();
  f->d_q_spec.release();
  if (f->h_res) (void)hipHostFree(f->h_res);
  if (f->h_res_spec) (void)hipHostFree(f->h_res_spec);
  if (f->h_ranges) (void)hipHostFree(f->h_ranges);
  delete f;
}

int lslam_frontend_reset(lslam_frontend* f) {
  if (!f) return LSLAM_ERR_INVALID_ARGUMENT;
  f->scans.clear();
  f->run_start = f->run_count = 0;
  f->have_last = false;
  f->n_chain_matches = f->n_loop_coarse = f->n_loop_fine = f->n_loops_closed = f->n_edges = f->n_loop_discarded = 0;
  if (f->spec.started) fe_spec_finish(f);
  f->spec = lslam_frontend::Spec{};
  f->spec_chain_id = -1;
  f->n_spec_started = f->n_spec_used = f->n_spec_discarded = 0;
  return LSLAM_OK;
}

int lslam_frontend_running_scans(const lslam_frontend* f) { return f ? f->run_count : LSLAM_ERR_INVALID_ARGUMENT; }
int lslam_frontend_num_scans(const lslam_frontend* f) { return f ? (int)f->scans.size() : LSLAM_ERR_INVALID_ARGUMENT; }

int lslam_frontend_scan_pose(const lslam_frontend* f, int scan_id, double robot_pose[3]) {
  if (!f || !robot_pose || scan_id < 0 || scan_id >= (int)f->scans.size()) return LSLAM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < 3; i++) robot_pose[i] = f->scans[scan_id].robot[i];
  return LSLAM_OK;
}

int lslam_frontend_stats(const lslam_frontend* f, int64_t out[6]) {
  if (!f || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  out[0] = (int64_t)f->scans.size();
  out[1] = f->n_edges;
  out[2] = f->n_chain_matches;
  out[3] = f->n_loop_coarse;
  out[4] = f->n_loop_fine;
  out[5] = f->n_loops_closed;
  return LSLAM_OK;
}

int lslam_frontend_process_stamped(lslam_frontend* f, const double* ranges, int n_ranges, const double odom_pose[3],
                                   double time_s, int* processed, double corrected_pose[3], double covariance[9],
                                   double* response) {
  if (!f || !ranges || !odom_pose || !processed || !corrected_pose) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_matcher* m = f->m;
  lslam_context* ctx = m->ctx;
  const int n = m->g.n_beams;
  if (n_ranges < n) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "n_ranges %d < num_beams %d", n_ranges, n);
  LSLAM_NOT_REENTRANT(m);  // the front-end drives its matcher's grid and workspaces
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  const lslam_laser* laser = &m->laser;
  double corrected[3] = {odom_pose[0], odom_pose[1], odom_pose[2]};
  double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // Mapper.cpp:2033-2034
  double resp = 0.0;
  *processed = 0;
  if (covariance) memcpy(covariance, cov, sizeof cov);  // also what a rejected scan reports
  if (response) *response = 0.0;
  if (f->have_last) {
    bool moved = false;
    fe_prologue(f, odom_pose, time_s, corrected, &moved);  // :2021-2031, HasMovedEnough :2087-2120
    if (!moved) {
      for (int i = 0; i < 3; i++) corrected_pose[i] = corrected[i];
      return LSLAM_OK;
    }
  }
  const int id = (int)f->scans.size();
  double sp[3];
  lslam_sensor_pose_from_robot(laser, corrected, sp);
  // A look-ahead match of THIS scan may be in flight (process_many: enqueued under the previous scan's loop search).  It
  // counts only if it is this scan's, started from the very poses the plain walk has now; anything else is waited for
  // and dropped.
  bool have_spec = false;
  lslam_match_result spec_r;
  if (f->spec.started) {
    fe_spec_finish(f);
    const lslam_frontend::Spec& q = f->spec;
    have_spec = f->have_last && !q.invalid && q.rc == LSLAM_OK && q.id == id && q.ranges == ranges && q.time == time_s &&
                memcmp(q.odom, odom_pose, sizeof q.odom) == 0 && memcmp(q.sp, sp, sizeof sp) == 0 &&
                memcmp(q.last_robot, f->scans.back().robot, sizeof q.last_robot) == 0;
    if (have_spec) {
      spec_r = q.r;
      f->n_spec_used++;
      f->spec_chain_id = q.chain_launched ? id : -1;
    } else {
      f->spec_chain_id = -1;
      f->n_spec_discarded++;
    }
    f->spec.started = false;
  }
  int rc = fe_grow(f, id + 1);
  if (rc) return rc;
  if (n > 0 && !have_spec) {  // (an accepted look-ahead match has already moved the readings into their resident row)
    if (f->have_last) {  // a match follows at once: stage the readings, its first kernel moves them (one operation fewer)
      memcpy(f->h_ranges, ranges, (size_t)n * sizeof(double));
      f->pending_ranges = f->h_ranges;
    } else {
      LSLAM_HIP(ctx, hipMemcpyAsync(f->d_ranges + (size_t)id * n, ranges, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
  }
  lslam_frontend_scan s;
  for (int i = 0; i < 3; i++) s.odom[i] = odom_pose[i];
  s.time = time_s;
  for (int i = 0; i < n; i++) {  // the filtered readings (Karto.h:5381-5405): minimum range <= r <= range threshold
    const double r = ranges[i];
    if (r >= laser->minimum_range && r <= laser->range_threshold) {
      s.blx += r * f->cos_a[i];
      s.bly += r * f->sin_a[i];
      s.nfilt++;
    }
  }
  if (f->have_last) {  // MatchScan(pScan, runningScans) + SetSensorPose(bestPose) (:2037-2045)
    lslam_match_result r;
    if (have_spec) {
      r = spec_r;
    } else {
      // ask the match for the scan's speculative anchor chain (an extra block of its lone coarse reduce)
      if (f->d_spec_chain && f->cfg_spec_chain) {
        m->spec_req.ranges = f->d_ranges + (size_t)id * n;
        for (int i = 0; i < 3; i++) m->spec_req.pose[i] = sp[i];
        m->spec_req.out = f->d_spec_chain;
        m->spec_armed = true;
      }
      rc = fe_match(f, m, id, sp, f->run_start, f->run_count, 1, 1, &r);
      if (rc) return rc;
      f->spec_chain_id = m->spec_launched ? id : -1;
    }
    resp = r.response;
    for (int i = 0; i < 9; i++) cov[i] = r.covariance[i];
    for (int i = 0; i < 3; i++) sp[i] = r.pose[i];
  } else if (n > 0) {
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the caller's ranges are free again when we return
  }
  fe_set_sensor_pose(f, s, sp);
  f->scans.push_back(s);  // AddScan + AddVertex (:2048-2054)
  // ---- MapperGraph::AddEdges (:901-973) ----
  if (f->have_last) {
    fe_link_scans(f, id - 1, id);
    std::vector<double> means(f->scans[id].sensor, f->scans[id].sensor + 3), covs(cov, cov + 9);
    fe_link_chain_to_scan(f, f->run_start, f->run_count, id);
    // LinkNearChains (:1124-1149)
    for (const auto& chain : fe_near_chains(f, id)) {
      if (chain.second < f->cfg.loop_match_minimum_chain_size) continue;
      lslam_match_result r;
      rc = fe_match(f, m, id, f->scans[id].sensor, chain.first, chain.second, 0, 1, &r);
      if (rc) return rc;
      f->n_chain_matches++;
      if (r.response > f->cfg.link_match_minimum_response_fine - kTol) {
        means.insert(means.end(), r.pose, r.pose + 3);
        covs.insert(covs.end(), r.covariance, r.covariance + 9);
        fe_link_chain_to_scan(f, chain.first, chain.second, id);
      }
    }
    double wm[3];
    weighted_mean((int)(means.size() / 3), means.data(), covs.data(), wm);
    fe_set_sensor_pose(f, f->scans[id], wm);  // pScan->SetSensorPose(ComputeWeightedMean(means, covariances))
  }
  rc = fe_update_world(f, id);
  if (rc) return rc;
  // ---- AddRunningScan (Mapper.h:1365-1386): push, then trim the front ----
  if (f->run_count == 0) f->run_start = id;
  f->run_count++;
  for (;;) {
    const double d2 = sq_dist2(f->scans[f->run_start].sensor, f->scans[f->run_start + f->run_count - 1].sensor);
    if (!((uint32_t)f->run_count > (uint32_t)f->cfg.scan_buffer_size ||
          d2 > ksq(f->cfg.scan_buffer_maximum_scan_distance) - kTol))
      break;
    f->run_start++;
    f->run_count--;
  }
  if (f->cfg.do_loop_closing && f->loop_m) {  // :2063-2070
    rc = fe_try_close_loop(f, id);
    if (rc) return rc;
  }
  f->have_last = true;  // SetLastScan (:2074)
  for (int i = 0; i < 3; i++) corrected_pose[i] = f->scans[id].robot[i];
  if (covariance) memcpy(covariance, cov, sizeof cov);
  if (response) *response = resp;
  *processed = 1;
  return LSLAM_OK;
}

int lslam_frontend_process(lslam_frontend* f, const double* ranges, int n_ranges, const double odom_pose[3],
                           int* processed, double corrected_pose[3], double covariance[9], double* response) {
  return lslam_frontend_process_stamped(f, ranges, n_ranges, odom_pose, 0.0, processed, corrected_pose, covariance, response);
}

// Mapper::Process for n scans the caller already holds (offline / batch use), one after the other, with ONE scan of
// look-ahead: see lslam_frontend::la.  Scan for scan the same poses, edges and graph as n calls of
// lslam_frontend_process_stamped -- the loop search of scan t and the running-window match of scan t + 1 merely share
// the device.  Stops at the first error.
int lslam_frontend_process_many(lslam_frontend* f, int n_scans, const double* ranges, int ranges_stride, const double* odom_poses,
                                const double* times_s, int32_t* processed, double* corrected_poses, double* covariances,
                                double* responses) {
  if (!f || n_scans < 0 || (n_scans > 0 && (!ranges || !odom_poses || !processed || !corrected_poses)))
    return LSLAM_ERR_INVALID_ARGUMENT;
  if (ranges_stride < f->m->g.n_beams) return f->m->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "ranges_stride < num_beams");
  int rc = LSLAM_OK;
  for (int t = 0; t < n_scans && rc == LSLAM_OK; t++) {
    lslam_frontend::LookAhead next;
    if (t + 1 < n_scans) {
      next.ranges = ranges + (size_t)(t + 1) * ranges_stride;
      next.odom = odom_poses + 3 * (size_t)(t + 1);
      next.time = times_s ? times_s[t + 1] : 0.0;
      f->la = &next;
    }
    int done = 0;
    rc = lslam_frontend_process_stamped(f, ranges + (size_t)t * ranges_stride, ranges_stride, odom_poses + 3 * (size_t)t,
                                        times_s ? times_s[t] : 0.0, &done, corrected_poses + 3 * (size_t)t,
                                        covariances ? covariances + 9 * (size_t)t : nullptr, responses ? responses + t : nullptr);
    processed[t] = done;
    f->la = nullptr;
  }
  if (f->spec.started) {  // an error left a look-ahead match behind: nothing of it is used
    fe_spec_finish(f);
    f->spec.started = false;
    f->n_spec_discarded++;
  }
  return rc;
}

// out[0] = look-ahead matches started, [1] = accepted, [2] = discarded (a loop closed in between, or the scan was not the one
// announced)
// diagnostics: FindValidPoints' anchors of a resident scan at its current pose, row = [count, indices...] (n + 1 ints)
int lslam_debug_frontend_anchor_row(lslam_frontend* f, int scan_id, int32_t* out) {
  if (!f || !out || scan_id < 0 || scan_id >= (int)f->scans.size()) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = f->m->ctx;
  const int n = f->m->g.n_beams;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  LSLAM_HIP(ctx, hipMemcpy(out, f->d_next + (size_t)scan_id * (n + 1), (size_t)(n + 1) * sizeof(int), hipMemcpyDeviceToHost));
  return LSLAM_OK;
}

// diagnostics: k_anchor_chain launches that were handed a speculative chain / of those, the ones that could not take it
// over (a comparison inside the band, a point whose class changed) and worked the chain out themselves
int lslam_frontend_spec_chain_stats(lslam_frontend* f, int64_t out[2]) {
  if (!f || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = f->m->ctx;
  out[0] = f->n_spec_chain_used;
  out[1] = 0;
  if (f->d_spec_chain) {
    const int n = std::max(f->m->g.n_beams, 1);
    int fb = 0;
    LSLAM_HIP(ctx, hipSetDevice(ctx->device));
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    LSLAM_HIP(ctx, hipMemcpy(&fb, f->d_spec_chain + 2 + n + ((n + 3) >> 2), sizeof fb, hipMemcpyDeviceToHost));
    out[1] = fb;
  }
  return LSLAM_OK;
}

int lslam_frontend_lookahead_stats(const lslam_frontend* f, int64_t out[3]) {
  if (!f || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  out[0] = f->n_spec_started;
  out[1] = f->n_spec_used;
  out[2] = f->n_spec_discarded;
  return LSLAM_OK;
}

}  // extern "C"
