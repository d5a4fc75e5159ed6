// Batched many-scan mode across the GPUs of one node, in ONE host process (north star "C++ host ...
// shards independent scan-match problems across the 8 GPUs").  One context + matcher per device; the shared
// correlation grid is built once on the first device and copied to the others over xGMI (hipMemcpyPeerAsync -- 4 MB,
// the only inter-GPU traffic of the path); scans [r*B/W, (r+1)*B/W) go to device r, one host thread per device drives
// its upload -> match -> download; results land in the caller's array in scan order (the "gather" is the D2H copy).
// No collective: the units are independent.  (The one-process-per-GPU form with RCCL is bench.py / shard.py.)
#include <thread>
#include <vector>

#include "common.hpp"

struct lslam_pool {
  std::vector<lslam_context*> ctx;
  std::vector<lslam_matcher*> m;
  std::string last_error;
  bool peer_copy = true;
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
      rcs[r] = lslam_matcher_match_batch(p->m[r], (int)(hi - lo), ranges + (size_t)lo * ranges_stride, ranges_stride,
                                         sensor_poses + 3 * lo, do_penalize, do_refine, out + lo);
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

}  // extern "C"
