// Batched many-scan mode across the GPUs of one node, in ONE host process (north star "C++ host ...
// shards independent scan-match problems across the 8 GPUs").  One context + matcher per device; the shared
// correlation grid is built once on the first device and copied to the others over xGMI (hipMemcpyPeerAsync -- 4 MB,
// the only inter-GPU traffic of the path); scans [r*B/W, (r+1)*B/W) go to device r, one host thread per device drives
// its upload -> match -> download; results land in the caller's array in scan order (the "gather" is the D2H copy).
// No collective: the units are independent.  (The one-process-per-GPU form with RCCL is bench.py / shard.py.)
#include <set>
#include <thread>
#include <vector>

#include "common.hpp"
#include "karto_math.hpp"
#include "rccl_dyn.hpp"

struct lslam_pool {
  std::vector<lslam_context*> ctx;
  std::vector<lslam_matcher*> m;
  std::string last_error;
  bool peer_copy = true;
  std::vector<ncclComm_t> comms;  // one RCCL communicator per device (ncclCommInitAll), created on first use
};

extern "C" {

int lslam_pool_create_on(const int* devices, int n_devices, const lslam_matcher_config* cfg, const lslam_laser* laser,
                         lslam_pool** out) {
  if (!devices || n_devices < 1 || !cfg || !laser || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  lslam_pool* p = new lslam_pool();
  for (int i = 0; i < n_devices; i++) {
    lslam_context* c = nullptr;
    int rc = lslam_create(devices[i], &c);
    if (rc != LSLAM_OK) {
      lslam_pool_destroy(p);
      return rc;
    }
    p->ctx.push_back(c);
    lslam_matcher* mm = nullptr;
    rc = lslam_matcher_create(c, cfg, laser, &mm);
    if (rc != LSLAM_OK) {
      lslam::g_last_error = lslam_last_error(c);
      lslam_pool_destroy(p);
      return rc;
    }
    p->m.push_back(mm);
    // (the matchers stay at the default LSLAM_OPT_PIPELINE_DEPTH 1: lslam_pool_matcher hands them out BORROWED, and a
    // borrower who orders its own work with an event on lslam_stream() relies on the plain contract -- lslam_pool_match_batch
    // raises the depth around its own call only)
  }
  *out = p;
  return LSLAM_OK;
}

// n_devices = 0: every visible GPU.  More than are visible is an error, never a wrap-around.
int lslam_pool_create(int n_devices, const lslam_matcher_config* cfg, const lslam_laser* laser, lslam_pool** out) {
  int have = 0;
  if (hipGetDeviceCount(&have) != hipSuccess || have <= 0) {
    lslam::g_last_error = "lslam_pool_create: no HIP device visible (no CPU fallback)";
    return LSLAM_ERR_NO_DEVICE;
  }
  if (n_devices == 0) n_devices = have;
  if (n_devices < 0 || n_devices > have) {
    lslam::g_last_error = "lslam_pool_create: more devices requested than are visible";
    return LSLAM_ERR_INVALID_ARGUMENT;
  }
  std::vector<int> ids(n_devices);
  for (int i = 0; i < n_devices; i++) ids[i] = i;
  return lslam_pool_create_on(ids.data(), n_devices, cfg, laser, out);
}

void lslam_pool_destroy(lslam_pool* p) {
  if (!p) return;
  for (auto* mm : p->m) lslam_matcher_destroy(mm);
  if (!p->comms.empty() && lslam::Rccl::get().ok())
    for (auto c : p->comms) (void)lslam::Rccl::get().CommDestroy(c);
  for (auto* c : p->ctx) lslam_destroy(c);
  delete p;
}

int lslam_pool_devices(const lslam_pool* p) { return p ? (int)p->ctx.size() : LSLAM_ERR_INVALID_ARGUMENT; }
lslam_matcher* lslam_pool_matcher(lslam_pool* p, int i) { return (p && i >= 0 && i < (int)p->m.size()) ? p->m[i] : nullptr; }
const char* lslam_pool_last_error(const lslam_pool* p) { return p ? p->last_error.c_str() : lslam::g_last_error.c_str(); }

// AddScans once, on the first device; the grid bytes then travel device-to-device (xGMI) -- or, with
// replicate_by_rebuild != 0, every device rasterises the same base scans itself (no inter-GPU traffic at all)
int lslam_pool_set_base_scans(lslam_pool* p, int n_scans, const double* ranges, int ranges_stride, const double* sensor_poses,
                              const double center_pose[3], int replicate_by_rebuild) {
  if (!p || p->m.empty()) return LSLAM_ERR_INVALID_ARGUMENT;
  int rc = lslam_matcher_set_base_scans(p->m[0], n_scans, ranges, ranges_stride, sensor_poses, center_pose);
  if (rc) {
    p->last_error = lslam_last_error(p->ctx[0]);
    return rc;
  }
  int32_t gi[8];
  double off[2];
  lslam_matcher_grid_info(p->m[0], gi, off);
  const size_t bytes = (size_t)gi[1] * gi[2];  // height * widthStep
  for (size_t d = 1; d < p->m.size(); d++) {
    if (replicate_by_rebuild) {
      rc = lslam_matcher_set_base_scans(p->m[d], n_scans, ranges, ranges_stride, sensor_poses, center_pose);
    } else {
      rc = lslam_synchronize(p->ctx[0]);
      if (rc == LSLAM_OK) {
        void* dst = lslam_matcher_grid_dev_ptr(p->m[d]);
        void* src = lslam_matcher_grid_dev_ptr(p->m[0]);
        hipError_t e = hipSetDevice(p->ctx[d]->device);
        if (e == hipSuccess)
          e = hipMemcpyPeerAsync(dst, p->ctx[d]->device, src, p->ctx[0]->device, bytes, p->ctx[d]->stream);
        if (e != hipSuccess) {
          p->last_error = std::string("hipMemcpyPeerAsync: ") + hipGetErrorString(e);
          return LSLAM_ERR_HIP;
        }
        // the grid bytes changed behind the matcher's back: re-install them from their own address (refreshes the
        // derived planes / bitmaps lazily, like any grid change)
        rc = lslam_matcher_set_grid_u8_dev(p->m[d], (const uint8_t*)dst, off);
      }
    }
    if (rc) {
      p->last_error = lslam_last_error(p->ctx[d]);
      return rc;
    }
  }
  // the peer copies read device 0's grid: nobody may rebuild it (a second set_base_scans, direct use of matcher 0)
  // before they have landed
  for (size_t d = 1; d < p->m.size(); d++) {
    rc = lslam_synchronize(p->ctx[d]);
    if (rc) {
      p->last_error = lslam_last_error(p->ctx[d]);
      return rc;
    }
  }
  return LSLAM_OK;
}

// n_scans independent scans against the shared grid, sharded [r*B/W, (r+1)*B/W); out[n_scans] in scan order
int lslam_pool_match_batch(lslam_pool* p, int n_scans, const double* ranges, int ranges_stride, const double* sensor_poses,
                           int do_penalize, int do_refine, lslam_match_result* out) {
  if (!p || n_scans < 0 || (n_scans > 0 && (!ranges || !sensor_poses || !out))) return LSLAM_ERR_INVALID_ARGUMENT;
  const int W = (int)p->m.size();
  std::vector<int> rcs(W, LSLAM_OK);
  std::vector<std::thread> th;
  for (int r = 0; r < W; r++) {
    const long long lo = (long long)r * n_scans / W, hi = (long long)(r + 1) * n_scans / W;
    if (hi <= lo) continue;
    th.emplace_back([=, &rcs]() {
      // every device's share goes through as two pipelined sub-batches (upload of one under the kernels of the other, reduce
      // chains overlapped: lslam_matcher_match_batch does the splitting and joins before it returns); the depth the matcher
      // had -- a borrower may have set its own -- is restored afterwards, and a refused option is an error, not ignored
      lslam_matcher* mm = p->m[r];
      const int depth_before = lslam_matcher_get_option(mm, LSLAM_OPT_PIPELINE_DEPTH);
      int rc = depth_before < 2 ? lslam_matcher_set_option(mm, LSLAM_OPT_PIPELINE_DEPTH, 2) : LSLAM_OK;
      if (rc == LSLAM_OK)
        rc = lslam_matcher_match_batch(mm, (int)(hi - lo), ranges + (size_t)lo * ranges_stride, ranges_stride, sensor_poses + 3 * lo,
                                       do_penalize, do_refine, out + lo);
      if (depth_before >= 1 && depth_before < 2) {
        const int rc2 = lslam_matcher_set_option(mm, LSLAM_OPT_PIPELINE_DEPTH, depth_before);
        if (rc == LSLAM_OK) rc = rc2;
      }
      rcs[r] = rc;
    });
  }
  for (auto& t : th) t.join();
  for (int r = 0; r < W; r++)
    if (rcs[r] != LSLAM_OK) {
      p->last_error = lslam_last_error(p->ctx[r]);
      return rcs[r];
    }
  return LSLAM_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// Offline map build from known poses, sharded (SURVEY.md 8(e) last row; karto::OccupancyGrid::CreateFromScans,
// Karto.h:5659-5673): the one path with a real exchange step, with RCCL called directly from the C++ host.
// ------------------------------------------------------------------------------------------------------------------
namespace {
int rccl_fail(lslam_context* ctx, const char* what, ncclResult_t r) {
  lslam::Rccl& R = lslam::Rccl::get();
  return ctx->fail(LSLAM_ERR_HIP, "%s: %s", what, R.GetErrorString ? R.GetErrorString(r) : "rccl error");
}
}  // namespace

// ONE RANK's part (one process per GPU, or one thread per device inside ncclGroupStart/End): this rank's scans in,
// the grid of ALL ranks' scans out.  nccl_comm is the caller's ncclComm_t for ctx's device.  Everything runs on the
// context stream: box of the local scans -> all-reduce(max) of (-minx, -miny, maxx, maxy) -> local counters on the
// merged box -> ONE all-reduce(sum) over both counter planes, in place in HBM.
// Arguments every rank shares -- laser, resolution, the row stride -- are checked BEFORE the first collective,
// independently of how many scans this rank holds (an empty shard must fail like a full one: a rank that returned early
// would leave its peers waiting in ncclAllReduce for ever).  A failure that is genuinely local to one rank (its scans'
// box, an allocation) does not return early either: the rank stays in the protocol with an error flag in the reduced
// vector, so EVERY rank learns of it and all leave together -- after the box exchange, or after a one-word status
// exchange in front of the counter all-reduce.
namespace {
// what every rank passes alike -- the laser, the resolution, the row stride: a rank-independent verdict, so checking it
// BEFORE the first collective cannot strand a peer
int sharded_args_ok(lslam_context* ctx, const lslam_laser* laser, int ranges_stride, double resolution) {
  if (!laser) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "sharded map build: laser");
  if (!(laser->angular_resolution > 0.0) || !(laser->maximum_angle >= laser->minimum_angle))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "sharded map build: laser angles");
  if (resolution == 0.0 || (resolution > -1e-6 && resolution < 1e-6) || !(resolution == resolution))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "Resolution cannot be 0");  // Karto.h:5627-5630
  const int n = (int)(uint32_t)lslam::kround((laser->maximum_angle - laser->minimum_angle) / laser->angular_resolution);
  if (ranges_stride < n)  // whatever n_scans is: every rank must come to the same verdict
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "ranges_stride %d < num_beams %d", ranges_stride, n);
  return LSLAM_OK;
}
// what only THIS rank can see -- its own shard's pointers and count: never a reason to return before a collective
// (the rank joins exchange 1 with its error flag set, like any other local failure)
int shard_args_local(lslam_context* ctx, int n_scans, const double* ranges, const double* sensor_poses) {
  if (n_scans < 0 || (n_scans > 0 && (!ranges || !sensor_poses)))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "sharded map build: this rank's scans (count %d, ranges %p, poses %p)", n_scans,
                     (const void*)ranges, (const void*)sensor_poses);
  return LSLAM_OK;
}
}  // namespace

int lslam_occgrid_create_sharded(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges,
                                 int ranges_stride, const double* sensor_poses, double resolution, void* nccl_comm,
                                 lslam_occgrid** out) {
  if (!ctx || !out || !nccl_comm) return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  lslam::Rccl& R = lslam::Rccl::get();
  if (!R.ok()) return ctx->fail(LSLAM_ERR_UNSUPPORTED, "%s", R.error.c_str());
  int rc = sharded_args_ok(ctx, laser, ranges_stride, resolution);
  if (rc) return rc;  // arguments every rank shares: the same verdict everywhere, nobody enters a collective
  ncclComm_t comm = (ncclComm_t)nccl_comm;
  // ---- exchange 1: union of the boxes + "some rank failed" -----------------------------------------------------------
  std::string local_error;
  double box[4] = {1e300, 1e300, -1e300, -1e300};  // the identity of the union: what a rank contributes when it failed
  int local_rc = shard_args_local(ctx, n_scans, ranges, sensor_poses);
  if (!local_rc) local_rc = lslam_occgrid_scan_bounds(ctx, laser, n_scans, ranges, ranges_stride, sensor_poses, box);
  if (local_rc) {
    local_error = ctx->last_error;
    box[0] = box[1] = 1e300;
    box[2] = box[3] = -1e300;
  }
  if (hipSetDevice(ctx->device) != hipSuccess && !local_rc) {
    local_rc = LSLAM_ERR_HIP;
    local_error = "hipSetDevice failed";
  }
  double* d_x = (double*)ctx->d_small;  // allocated with the context: joining a collective never depends on a hipMalloc here
  // negation and max are exact: the union box bit for bit; [4] = 1 where a rank failed locally
  const double send[5] = {-box[0], -box[1], box[2], box[3], local_rc ? 1.0 : 0.0};
  double merged[5] = {0, 0, 0, 0, 1.0};
  hipError_t e = hipMemcpyAsync(d_x, send, sizeof send, hipMemcpyHostToDevice, ctx->stream);
  ncclResult_t nr = R.AllReduce(d_x, d_x, 5, ncclDouble, ncclMax, comm, ctx->stream);  // joined even after a failed copy
  if (e == hipSuccess && nr == ncclSuccess) e = hipMemcpyAsync(merged, d_x, sizeof merged, hipMemcpyDeviceToHost, ctx->stream);
  if (nr == ncclSuccess) {
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = es;
  }
  if (nr != ncclSuccess) return rccl_fail(ctx, "ncclAllReduce(box)", nr);
  if (local_rc) return ctx->fail(local_rc, "%s", local_error.c_str());
  if (e != hipSuccess) return ctx->fail(LSLAM_ERR_HIP, "box exchange failed: %s", hipGetErrorString(e));
  if (merged[4] != 0.0) return ctx->fail(LSLAM_ERR_HIP, "sharded map build: another rank failed before the box exchange");
  const double ubox[4] = {-merged[0], -merged[1], merged[2], merged[3]};
  if (!(ubox[0] <= ubox[2] && ubox[1] <= ubox[3]))  // no scans on any rank: the reference's NULL -- on every rank alike
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "empty bounding box (no scans on any shard: the reference returns NULL)");
  // ---- local counters on the merged box, then exchange 2: one status word, so that a rank whose build failed (its
  // counter planes did not fit) does not leave the others alone in the big all-reduce ----------------------------------
  lslam_occgrid* part = nullptr;
  local_rc = lslam_occgrid_create_partial(ctx, laser, n_scans, ranges, ranges_stride, sensor_poses, resolution, ubox, &part);
  if (local_rc) local_error = ctx->last_error;
  const double st = local_rc ? 1.0 : 0.0;
  double st_all = 1.0;
  e = hipMemcpyAsync(d_x, &st, sizeof st, hipMemcpyHostToDevice, ctx->stream);
  nr = R.AllReduce(d_x, d_x, 1, ncclDouble, ncclMax, comm, ctx->stream);
  if (e == hipSuccess && nr == ncclSuccess) e = hipMemcpyAsync(&st_all, d_x, sizeof st_all, hipMemcpyDeviceToHost, ctx->stream);
  if (nr == ncclSuccess) {
    const hipError_t es = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) e = es;
  }
  if (nr != ncclSuccess || local_rc || e != hipSuccess || st_all != 0.0) {
    if (part) lslam_occgrid_destroy(part);
    if (nr != ncclSuccess) return rccl_fail(ctx, "ncclAllReduce(status)", nr);
    if (local_rc) return ctx->fail(local_rc, "%s", local_error.c_str());
    if (e != hipSuccess) return ctx->fail(LSLAM_ERR_HIP, "status exchange failed: %s", hipGetErrorString(e));
    return ctx->fail(LSLAM_ERR_HIP, "sharded map build: another rank could not build its partial grid");
  }
  // ---- exchange 3: ONE all-reduce(sum) over both counter planes, in place in HBM ------------------------------------------
  size_t words = 0;
  lslam_occgrid_counter_words(part, &words);
  if (words) {
    void* cnt = lslam_occgrid_counters_dev_ptr(part);
    nr = R.AllReduce(cnt, cnt, words, ncclUint32, ncclSum, comm, ctx->stream);
    if (nr != ncclSuccess) {
      lslam_occgrid_destroy(part);
      return rccl_fail(ctx, "ncclAllReduce(counters)", nr);
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      lslam_occgrid_destroy(part);
      return ctx->fail(LSLAM_ERR_HIP, "counter all-reduce failed: %s", hipGetErrorString(e));
    }
  }
  *out = part;
  return LSLAM_OK;
}

// The same over the devices of a pool, in one process: scans [r*n/W, (r+1)*n/W) to device r, one host thread per
// device, the communicators from ncclCommInitAll.  A pool that names one GPU several times (how the 1-GPU tests
// shard) cannot form an RCCL clique; its partial grids are merged on the first device instead (exact all the same).
// The grid of ALL scans is returned on the pool's first device.
int lslam_pool_occgrid_from_scans(lslam_pool* p, const lslam_laser* laser, int n_scans, const double* ranges,
                                  int ranges_stride, const double* sensor_poses, double resolution, lslam_occgrid** out) {
  if (!p || p->ctx.empty() || !laser || !out || n_scans < 0 || (n_scans > 0 && (!ranges || !sensor_poses)))
    return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  {  // what every shard shares, checked ONCE before any thread (or collective) starts
    int rc = sharded_args_ok(p->ctx[0], laser, ranges_stride, resolution);
    if (rc) {
      p->last_error = lslam_last_error(p->ctx[0]);
      return rc;
    }
  }
  const int W = (int)p->ctx.size();
  std::set<int> distinct;
  for (auto* c : p->ctx) distinct.insert(c->device);
  lslam::Rccl& R = lslam::Rccl::get();
  const bool use_rccl = (int)distinct.size() == W && R.ok();
  auto lo_of = [&](int r) { return (long long)r * n_scans / W; };
  std::vector<lslam_occgrid*> parts(W, nullptr);
  std::vector<int> rcs(W, LSLAM_OK);
  if (use_rccl) {
    if (p->comms.empty()) {
      std::vector<int> devs;
      for (auto* c : p->ctx) devs.push_back(c->device);
      p->comms.resize(W);
      ncclResult_t nr = R.CommInitAll(p->comms.data(), W, devs.data());
      if (nr != ncclSuccess) {
        p->comms.clear();
        p->last_error = std::string("ncclCommInitAll: ") + R.GetErrorString(nr);
        return LSLAM_ERR_HIP;
      }
    }
    std::vector<std::thread> th;
    for (int r = 0; r < W; r++)
      th.emplace_back([&, r]() {
        const long long lo = lo_of(r), hi = lo_of(r + 1);
        rcs[r] = lslam_occgrid_create_sharded(p->ctx[r], laser, (int)(hi - lo), ranges + (size_t)lo * ranges_stride, ranges_stride,
                                              sensor_poses + 3 * lo, resolution, (void*)p->comms[r], &parts[r]);
      });
    for (auto& t : th) t.join();
    for (int r = 0; r < W; r++)
      if (rcs[r]) {
        p->last_error = lslam_last_error(p->ctx[r]);
        for (auto* g : parts) lslam_occgrid_destroy(g);
        return rcs[r];
      }
    for (int r = 1; r < W; r++) lslam_occgrid_destroy(parts[r]);  // every rank holds the full grid; keep the first
    *out = parts[0];
    return LSLAM_OK;
  }
  // ---- no RCCL clique: boxes merged on the host, partial counters added on the first device ----
  double ubox[4] = {1e300, 1e300, -1e300, -1e300};
  for (int r = 0; r < W; r++) {
    const long long lo = lo_of(r), hi = lo_of(r + 1);
    double b[4];
    int rc = lslam_occgrid_scan_bounds(p->ctx[r], laser, (int)(hi - lo), ranges + (size_t)lo * ranges_stride, ranges_stride,
                                       sensor_poses + 3 * lo, b);
    if (rc) {
      p->last_error = lslam_last_error(p->ctx[r]);
      return rc;
    }
    ubox[0] = std::min(ubox[0], b[0]); ubox[1] = std::min(ubox[1], b[1]);
    ubox[2] = std::max(ubox[2], b[2]); ubox[3] = std::max(ubox[3], b[3]);
  }
  for (int r = 0; r < W; r++) {
    const long long lo = lo_of(r), hi = lo_of(r + 1);
    int rc = lslam_occgrid_create_partial(p->ctx[r], laser, (int)(hi - lo), ranges + (size_t)lo * ranges_stride, ranges_stride,
                                          sensor_poses + 3 * lo, resolution, ubox, &parts[r]);
    if (rc == LSLAM_OK && r > 0) {
      size_t words = 0;
      lslam_occgrid_counter_words(parts[r], &words);
      std::vector<uint32_t> host(words);
      rc = lslam_occgrid_export_counters(parts[r], host.data(), 0);
      if (rc == LSLAM_OK) rc = lslam_occgrid_import_counters(parts[0], host.data(), 0, 1);
    }
    if (rc) {
      p->last_error = lslam_last_error(p->ctx[r]);
      for (auto* g : parts) lslam_occgrid_destroy(g);
      return rc;
    }
  }
  for (int r = 1; r < W; r++) lslam_occgrid_destroy(parts[r]);
  *out = parts[0];
  return LSLAM_OK;
}

}  // extern "C"
