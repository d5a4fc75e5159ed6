// scan_cache_impl.hpp -- device-side scan cache behind seam B1 (included at the end of scan_matcher.hip).
//
// karto::ScanMatcher::MatchScan (Mapper.h:1155-1159) receives its base scans as a LocalizedRangeScanVector on EVERY call
// (Mapper.cpp:2040 the running window, :942/:1140 near chains, :991/:1015 loop chains); a C ABI that mirrors that
// signature literally (lslam_matcher_match_scan) re-uploads ~70 x 8.6 KB of readings per call and re-derives their world
// points and FindValidPoints anchors every time, although a scan's readings never change and its pose changes about once
// in its life (when its own match is accepted, or when a closed loop re-poses it).  The cache keeps, per scan id chosen by
// the caller: the readings, the world points at the pose they were last matched against, and the anchor chain of
// FindValidPoints at that pose (k_anchor_chain) -- what the streaming front-end keeps for its own scans
// (frontend_impl.hpp).  lslam_matcher_match_scan_cached then names its base scans by id and sends only their 24-byte
// poses; a scan whose pose differs BITWISE from the cached one is refreshed first (one launch for all of them), so results
// are identical to lslam_matcher_match_scan by construction: the same kernels on the same world points and anchors.
#pragma once

#include <unordered_map>

struct lslam_scan_cache {
  lslam_context* ctx = nullptr;
  lslam_laser laser;
  Geom g;  // only n_beams, min_angle, ang_res are read (beam_world_point)
  int cap = 0;
  double* d_ranges = nullptr;  // [cap][n]
  double2* d_world = nullptr;  // [cap][n] at slot.pose
  int* d_anchor = nullptr;     // [cap][n + 1] k_anchor_chain rows at slot.pose
  struct Slot {
    int64_t id = -1;
    bool posed = false;  // world / anchors are valid for `pose`
    double pose[3] = {0, 0, 0};
  };
  std::vector<Slot> slots;
  std::vector<int> free_slots;
  int next_slot = 0;  // slots [0, next_slot) have been handed out at least once
  std::unordered_map<int64_t, int> slot_of;
  // pinned staging, grown on demand: the slot list + refresh table of one call, the query's readings, the result record
  int* h_slots = nullptr;
  CacheRefresh* h_refresh = nullptr;
  int h_cap = 0;
  double* h_query = nullptr;
  lslam_match_result* h_result = nullptr;
  DevBuf<double> d_qpose;
  int64_t n_uploads = 0, n_refreshed = 0, n_speculated = 0, n_matches = 0;
  std::atomic<bool> busy{false};
};

namespace {

inline size_t sc_anchor_lds(int n) { return (size_t)n * (sizeof(double2) + 13) + 16; }

int sc_grow(lslam_scan_cache* c, int need) {
  if (need <= c->cap) return LSLAM_OK;
  lslam_context* ctx = c->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  const size_t n = (size_t)std::max(c->g.n_beams, 1);
  int cap = std::max(256, c->cap);
  while (cap < need) cap *= 2;
  double2* w = nullptr;
  double* r = nullptr;
  int* a = nullptr;
  if (hipMalloc((void**)&w, (size_t)cap * n * sizeof(double2)) != hipSuccess ||
      hipMalloc((void**)&r, (size_t)cap * n * sizeof(double)) != hipSuccess ||
      hipMalloc((void**)&a, (size_t)cap * (n + 1) * sizeof(int)) != hipSuccess) {
    (void)hipGetLastError();
    if (w) (void)hipFree(w);
    if (r) (void)hipFree(r);
    return ctx->fail(LSLAM_ERR_HIP, "scan cache: cannot keep %d scans resident in HBM", cap);
  }
  if (c->cap > 0) {
    LSLAM_HIP(ctx, hipMemcpyAsync(w, c->d_world, (size_t)c->cap * n * sizeof(double2), hipMemcpyDeviceToDevice, ctx->stream));
    LSLAM_HIP(ctx, hipMemcpyAsync(r, c->d_ranges, (size_t)c->cap * n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    LSLAM_HIP(ctx, hipMemcpyAsync(a, c->d_anchor, (size_t)c->cap * (n + 1) * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // nothing reads the old arrays any more: free them
    (void)hipFree(c->d_world);
    (void)hipFree(c->d_ranges);
    (void)hipFree(c->d_anchor);
  }
  c->d_world = w;
  c->d_ranges = r;
  c->d_anchor = a;
  c->cap = cap;
  c->slots.resize((size_t)cap);
  return LSLAM_OK;
}

int sc_staging(lslam_scan_cache* c, int n_base) {
  lslam_context* ctx = c->ctx;
  if (!c->h_query) {
    const size_t n = (size_t)std::max(c->g.n_beams, 1);
    if (hipHostMalloc((void**)&c->h_query, sizeof(double) * n, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&c->h_result, sizeof(lslam_match_result), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return ctx->fail(LSLAM_ERR_HIP, "scan cache: cannot allocate the pinned staging");
    }
  }
  if (n_base + 1 > c->h_cap) {
    int cap = std::max(256, c->h_cap);
    while (cap < n_base + 1) cap *= 2;
    int* hs = nullptr;
    CacheRefresh* hr = nullptr;
    if (hipHostMalloc((void**)&hs, sizeof(int) * (size_t)cap, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&hr, sizeof(CacheRefresh) * (size_t)cap, hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      if (hs) (void)hipHostFree(hs);
      return ctx->fail(LSLAM_ERR_HIP, "scan cache: cannot allocate the pinned slot tables");
    }
    // the previous call has synchronised: nothing on the stream reads the old tables
    if (c->h_slots) (void)hipHostFree(c->h_slots);
    if (c->h_refresh) (void)hipHostFree(c->h_refresh);
    c->h_slots = hs;
    c->h_refresh = hr;
    c->h_cap = cap;
  }
  return LSLAM_OK;
}

// a free slot for `id` (no readings yet)
int sc_new_slot(lslam_scan_cache* c, int64_t id, int* slot_out) {
  int slot;
  if (!c->free_slots.empty()) {
    slot = c->free_slots.back();
    c->free_slots.pop_back();
  } else {
    slot = c->next_slot;
    int rc = sc_grow(c, slot + 1);
    if (rc) return rc;
    c->next_slot++;
  }
  lslam_scan_cache::Slot& s = c->slots[(size_t)slot];
  s.id = id;
  s.posed = false;
  c->slot_of[id] = slot;
  *slot_out = slot;
  return LSLAM_OK;
}

inline bool sc_same_pose(const double* a, const double* b) { return memcmp(a, b, 3 * sizeof(double)) == 0; }

struct CacheBusy {
  lslam_scan_cache* c;
  bool ok;
  explicit CacheBusy(lslam_scan_cache* cc) : c(cc), ok(!cc->busy.exchange(true)) {}
  ~CacheBusy() {
    if (ok) c->busy.store(false);
  }
};

}  // namespace

extern "C" {

int lslam_scan_cache_create(lslam_context* ctx, const lslam_laser* laser, lslam_scan_cache** out) {
  if (!ctx || !laser || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!(laser->angular_resolution > 0) || !(laser->maximum_angle >= laser->minimum_angle))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "scan cache: laser angles");
  lslam_scan_cache* c = new lslam_scan_cache();
  c->ctx = ctx;
  c->laser = *laser;
  memset(&c->g, 0, sizeof c->g);
  c->g.n_beams = (int)(uint32_t)kround((laser->maximum_angle - laser->minimum_angle) / laser->angular_resolution);  // Karto.h:4158
  c->g.min_angle = laser->minimum_angle;
  c->g.ang_res = laser->angular_resolution;
  *out = c;
  return LSLAM_OK;
}

void lslam_scan_cache_destroy(lslam_scan_cache* c) {
  if (!c) return;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->d_world) (void)hipFree(c->d_world);
  if (c->d_ranges) (void)hipFree(c->d_ranges);
  if (c->d_anchor) (void)hipFree(c->d_anchor);
  if (c->h_slots) (void)hipHostFree(c->h_slots);
  if (c->h_refresh) (void)hipHostFree(c->h_refresh);
  if (c->h_query) (void)hipHostFree(c->h_query);
  if (c->h_result) (void)hipHostFree(c->h_result);
  c->d_qpose.release();
  delete c;
}

int lslam_scan_cache_put(lslam_scan_cache* c, int64_t scan_id, const double* ranges) {
  if (!c || scan_id < 0 || !ranges) return LSLAM_ERR_INVALID_ARGUMENT;
  CacheBusy guard(c);
  if (!guard.ok) return c->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "scan cache used concurrently");
  lslam_context* ctx = c->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  int slot;
  auto it = c->slot_of.find(scan_id);
  if (it != c->slot_of.end()) {
    slot = it->second;
  } else {
    int rc = sc_new_slot(c, scan_id, &slot);
    if (rc) return rc;
  }
  c->slots[(size_t)slot].posed = false;
  const int n = c->g.n_beams;
  if (n > 0) {
    // pageable source: the runtime stages the copy before it returns, so the caller's buffer is free again
    LSLAM_HIP(ctx, hipMemcpyAsync(c->d_ranges + (size_t)slot * n, ranges, sizeof(double) * (size_t)n, hipMemcpyHostToDevice,
                                  ctx->stream));
  }
  c->n_uploads++;
  return LSLAM_OK;
}

int lslam_scan_cache_prepare(lslam_scan_cache* c, int64_t scan_id, const double sensor_pose[3]) {
  if (!c || scan_id < 0 || !sensor_pose) return LSLAM_ERR_INVALID_ARGUMENT;
  CacheBusy guard(c);
  if (!guard.ok) return c->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "scan cache used concurrently");
  lslam_context* ctx = c->ctx;
  auto it = c->slot_of.find(scan_id);
  if (it == c->slot_of.end()) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "scan %lld is not in the scan cache", (long long)scan_id);
  const int n = c->g.n_beams;
  lslam_scan_cache::Slot& s = c->slots[(size_t)it->second];
  if (n <= 0 || (s.posed && sc_same_pose(s.pose, sensor_pose))) return LSLAM_OK;
  if (sc_anchor_lds(n) > 60 * 1024) return LSLAM_OK;  // very long scans are refreshed in front of their match only
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  CacheRefresh one;
  one.slot = it->second;
  one.pad = 0;
  for (int k = 0; k < 3; k++) one.pose[k] = sensor_pose[k];
  launch(ctx, "cache_refresh", k_anchor_chain_list, dim3(1), dim3(n > 512 ? 1024 : 256), sc_anchor_lds(n), n, c->d_world,
         c->d_anchor, (const double*)c->d_ranges, (const CacheRefresh*)nullptr, (const lslam_match_result*)nullptr, one, c->g);
  LSLAM_HIP(ctx, hipGetLastError());
  s.posed = true;
  for (int k = 0; k < 3; k++) s.pose[k] = sensor_pose[k];
  c->n_speculated++;
  return LSLAM_OK;
}

int lslam_scan_cache_contains(const lslam_scan_cache* c, int64_t scan_id) {
  if (!c) return 0;
  return c->slot_of.count(scan_id) ? 1 : 0;
}

int lslam_scan_cache_forget(lslam_scan_cache* c, int64_t scan_id) {
  if (!c) return LSLAM_ERR_INVALID_ARGUMENT;
  CacheBusy guard(c);
  if (!guard.ok) return c->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "scan cache used concurrently");
  if (scan_id < 0) {
    for (auto& kv : c->slot_of) {
      c->slots[(size_t)kv.second].id = -1;
      c->slots[(size_t)kv.second].posed = false;
    }
    c->slot_of.clear();
    c->free_slots.clear();
    c->next_slot = 0;
    return LSLAM_OK;
  }
  auto it = c->slot_of.find(scan_id);
  if (it == c->slot_of.end()) return LSLAM_OK;
  c->slots[(size_t)it->second].id = -1;
  c->slots[(size_t)it->second].posed = false;
  c->free_slots.push_back(it->second);
  c->slot_of.erase(it);
  return LSLAM_OK;
}

int lslam_scan_cache_size(const lslam_scan_cache* c) { return c ? (int)c->slot_of.size() : LSLAM_ERR_INVALID_ARGUMENT; }

int lslam_scan_cache_counters(const lslam_scan_cache* c, int64_t out[5]) {
  if (!c || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  out[0] = c->n_matches;
  out[1] = c->n_uploads;
  out[2] = c->n_refreshed;
  out[3] = c->n_speculated;
  out[4] = (int64_t)c->cap * ((int64_t)std::max(c->g.n_beams, 1) * 24 + (std::max(c->g.n_beams, 1) + 1) * 4);  // resident bytes
  return LSLAM_OK;
}

int lslam_matcher_match_scan_cached(lslam_matcher* m, lslam_scan_cache* c, int n_base, const int64_t* base_ids,
                                    const double* base_poses, int64_t query_id, const double* q_ranges,
                                    const double q_pose[3], int flags, lslam_match_result* out) {
  if (!m || !c || !q_pose || !out || n_base < 0 || (n_base > 0 && (!base_ids || !base_poses)))
    return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  if (c->ctx != ctx) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "scan cache and matcher belong to different contexts");
  const int n = m->g.n_beams;
  if (n != c->g.n_beams || m->g.min_angle != c->g.min_angle || m->g.ang_res != c->g.ang_res)
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "scan cache and matcher were created for different lasers");
  const int do_penalize = (flags & LSLAM_MATCH_PENALIZE) ? 1 : 0, do_refine = (flags & LSLAM_MATCH_REFINE) ? 1 : 0;
  const bool q_cached = query_id >= 0 && c->slot_of.count(query_id) != 0;
  if (!q_ranges && !q_cached) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "query scan %lld is not cached and no readings were given", (long long)query_id);
  if (n == 0) {  // a scan without readings returns before AddScans (Mapper.cpp:199-209)
    const double none = 0.0;
    return lslam_matcher_match_batch(m, 1, q_ranges ? q_ranges : &none, 1, q_pose, do_penalize, do_refine, out);
  }
  LSLAM_NOT_REENTRANT(m);
  CacheBusy guard(c);
  if (!guard.ok) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "scan cache used concurrently");
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  int rc = sc_staging(c, n_base);
  if (rc) return rc;
  LSLAM_HIP(ctx, c->d_qpose.reserve(4));

  // ---- base scans: slot list + the ones whose cached world points are not at today's pose ------------------------------
  // (nothing is written into the cache's bookkeeping until every id has been checked: an error leaves it as it was)
  int n_refresh = 0;
  for (int i = 0; i < n_base; i++) {
    auto it = c->slot_of.find(base_ids[i]);
    if (it == c->slot_of.end()) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "base scan %lld is not in the scan cache", (long long)base_ids[i]);
    const int slot = it->second;
    c->h_slots[i] = slot;
    const double* want = base_poses + 3 * i;
    int listed = -1;  // the same scan may be named twice
    for (int k = 0; k < n_refresh && listed < 0; k++)
      if (c->h_refresh[k].slot == slot) listed = k;
    if (listed >= 0) {
      if (!sc_same_pose(c->h_refresh[listed].pose, want)) {
        return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "base scan %lld is named twice with different poses", (long long)base_ids[i]);
      }
      continue;
    }
    const lslam_scan_cache::Slot& s = c->slots[(size_t)slot];
    if (!s.posed || !sc_same_pose(s.pose, want)) {
      CacheRefresh& r = c->h_refresh[n_refresh++];
      r.slot = slot;
      r.pad = 0;
      for (int k = 0; k < 3; k++) r.pose[k] = want[k];
    } else {
      // posed at today's pose -- unless an earlier entry of this very list asked for another pose of the same scan
      for (int j = 0; j < i; j++)
        if (c->h_slots[j] == slot && !sc_same_pose(base_poses + 3 * j, want)) {
          return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "base scan %lld is named twice with different poses", (long long)base_ids[i]);
        }
    }
  }
  // ---- the query's slot: its readings ride into HBM in the rebuild's first launch (pinned staging), straight into their
  // resident row when the caller wants the scan kept (an uncached id cannot be among the base ids: checked above) ----------
  int q_slot = -1;
  if (q_cached) {
    q_slot = c->slot_of[query_id];
  } else if (query_id >= 0) {
    rc = sc_new_slot(c, query_id, &q_slot);
    if (rc) return rc;
    c->n_uploads++;
  }
  const bool lds_ok = sc_anchor_lds(n) <= 60 * 1024;
  if (n_refresh > 0) {
    if (lds_ok) {
      launch(ctx, "cache_refresh", k_anchor_chain_list, dim3(n_refresh), dim3(n > 512 ? 1024 : 256), sc_anchor_lds(n), n,
             c->d_world, c->d_anchor, (const double*)c->d_ranges, (const CacheRefresh*)c->h_refresh,
             (const lslam_match_result*)nullptr, CacheRefresh{}, c->g);
    } else {  // very long scans: world points only (k_find_valid walks the chain in global scratch)
      for (int k = 0; k < n_refresh; k++) {
        PoseArg pv;
        for (int i = 0; i < 3; i++) pv.v[i] = c->h_refresh[k].pose[i];
        launch(ctx, "cache_refresh", k_scan_prep<double>, dim3((n + 255) / 256, 1), dim3(256), 0,
               (const double*)(c->d_ranges + (size_t)c->h_refresh[k].slot * n), n, (const double*)nullptr, c->g,
               (double2*)nullptr, c->d_world + (size_t)c->h_refresh[k].slot * n, PassCfg{}, (Lattice*)nullptr,
               (double2*)nullptr, 0, pv);
      }
    }
    if (hipGetLastError() != hipSuccess) {  // the refresh is not on the stream: the slots stay as they were
      for (int k = 0; k < n_refresh; k++) c->slots[(size_t)c->h_refresh[k].slot].posed = false;
      return ctx->fail(LSLAM_ERR_HIP, "scan cache: the refresh launch failed");
    }
    for (int k = 0; k < n_refresh; k++) {
      lslam_scan_cache::Slot& s = c->slots[(size_t)c->h_refresh[k].slot];
      s.posed = true;
      for (int j = 0; j < 3; j++) s.pose[j] = c->h_refresh[k].pose[j];
    }
    c->n_refreshed += n_refresh;
  }

  // ---- rebuild around the query pose + search, exactly lslam_matcher_match_scan's stream ---------------------------------
  double* q_row = q_slot >= 0 ? c->d_ranges + (size_t)q_slot * n : m->d_query.p;
  if (q_slot < 0) {
    LSLAM_HIP(ctx, m->d_query.reserve((size_t)n));
    q_row = m->d_query.p;
  }
  RebuildExtras x{};
  for (int i = 0; i < 3; i++) x.pose[i] = q_pose[i];
  x.pose_dst = c->d_qpose.p;
  x.zero = m->d_resp.p;
  x.zero_words = (int)std::min<size_t>(m->d_resp.cap, (size_t)1 << 16);
  x.prep_ranges = q_row;
  if (!q_cached) {
    memcpy(c->h_query, q_ranges, sizeof(double) * (size_t)n);
    x.ranges_src = c->h_query;
    x.ranges_dst = q_row;
    x.n_ranges = n;
    x.prep_ranges = c->h_query;  // the prep blocks read the staged copy (the row is being written in the same launch)
  }
  x.anchor_ring = lds_ok ? c->d_anchor : nullptr;
  x.slot_list = c->h_slots;
  rc = rebuild_grid_dev(m, c->d_world, 0, n_base, std::max(c->cap, 1), q_pose, &x);
  if (rc == kRebuildNeedsContiguous) {
    // smear kernels / scan lengths the clear-free rebuild does not cover: gather the window's world points into the
    // matcher's own contiguous workspace and take the listed path (rare configurations; correctness over speed)
    LSLAM_HIP(ctx, m->d_world.reserve((size_t)std::max(n_base, 1) * n));
    for (int i = 0; i < n_base; i++)
      LSLAM_HIP(ctx, hipMemcpyAsync(m->d_world.p + (size_t)i * n, c->d_world + (size_t)c->h_slots[i] * n, sizeof(double2) * (size_t)n,
                                    hipMemcpyDeviceToDevice, ctx->stream));
    x.anchor_ring = nullptr;
    x.slot_list = nullptr;
    rc = rebuild_grid_dev(m, m->d_world.p, 0, n_base, n_base > 0 ? n_base : 1, q_pose, &x);
  }
  if (rc == LSLAM_OK) {
    arm_done_ticket(m);  // the last kernel posts a ticket behind the record: the host spins on it, not on the stream
    rc = match_batch_impl<double>(m, 1, q_row, n, c->d_qpose.p, do_penalize, do_refine, c->h_result, nullptr, 0);
  }
  if (rc) {
    // the query's readings may never have reached the slot this call handed out for them: an id that names garbage must
    // not stay behind
    if (q_slot >= 0 && !q_cached) {
      c->slot_of.erase(query_id);
      c->slots[(size_t)q_slot].id = -1;
      c->slots[(size_t)q_slot].posed = false;
      c->free_slots.push_back(q_slot);
      c->n_uploads--;
    }
    return rc;
  }
  // The caller announced that the query scan will take the match's mean as its sensor pose (Mapper::Process does:
  // Mapper.cpp:2040-2044) and, as the newest scan of the running window, be a base scan of the very next call: refresh
  // its world points + anchors at that pose BEHIND the match.  The host waits for the record only (the ticket the match's
  // last kernel posts), so the refresh overlaps the caller's own bookkeeping.
  bool speculate = (flags & LSLAM_MATCH_QUERY_TAKES_RESULT_POSE) && q_slot >= 0 && lds_ok;
  if (speculate) {
    launch(ctx, "cache_refresh", k_anchor_chain_list, dim3(1), dim3(n > 512 ? 1024 : 256), sc_anchor_lds(n), n, c->d_world,
           c->d_anchor, (const double*)c->d_ranges, (const CacheRefresh*)nullptr, (const lslam_match_result*)c->h_result,
           CacheRefresh{q_slot, 0, {0, 0, 0}}, c->g);
    // a refresh that was never enqueued must not leave the slot marked as posed at the result: the next call would
    // rasterise stale world points and anchors.  (The match itself is fine: its record is still delivered below.)
    if (hipGetLastError() != hipSuccess) speculate = false;
  }
  rc = wait_record(m);
  if (rc) {  // the stream reported an error: nothing about the slot's pose can be trusted any more
    if (q_slot >= 0) c->slots[(size_t)q_slot].posed = false;
    return rc;
  }
  *out = *c->h_result;
  c->n_matches++;
  if (speculate && out->status == LSLAM_OK) {
    lslam_scan_cache::Slot& s = c->slots[(size_t)q_slot];
    s.posed = true;
    for (int k = 0; k < 3; k++) s.pose[k] = out->pose[k];
    c->n_speculated++;
  } else if (q_slot >= 0 && ((flags & LSLAM_MATCH_QUERY_TAKES_RESULT_POSE) || !q_cached)) {
    // a refresh that failed to launch (or a failed match) may have left the slot half-way: recomputed on next use
    c->slots[(size_t)q_slot].posed = false;
  }
  return LSLAM_OK;
}

}  // extern "C"
