// Karto hit/pass-counter occupancy grid on MI355X (gfx950): lesson6's published map.
//
// Reference behaviour reproduced (never copied): karto::OccupancyGrid::CreateFromScans and its
// helpers ComputeDimensions / AddScan / RayTrace / UpdateCell (lesson6/lib/open_karto/include/
// open_karto/Karto.h:5659-5990), Grid<T>::TraceLine (Karto.h:4680-4745), as called every
// map_update_interval by SlamKarto::updateMap (lesson6/src/karto_slam.cc:507-581) over ALL
// processed scans -- O(scans x beams x ray length), growing without bound on the CPU.
//
// Device formulation: integer work only, exact by construction.
//   k_occ_points   thread per (scan, beam): LocalizedRangeScan::Update in fp64, beam class, end
//                  point (shortened to rangeThreshold), bounding-box reduction
//   k_occ_trace    one wave per beam: TraceLine cells in closed form, atomicAdd on the pass plane,
//                  hit+pass on a valid end point
//   k_occ_update   thread per cell: occupied / free / unknown from the two counters
//
// Multi-GPU ("offline map build from known poses"): the scan boxes combine by min/max and the counters
// by integer addition, both exact, so ranks build partial grids of disjoint scan subsets on the grid of the merged
// box (lslam_occgrid_scan_bounds / _create_partial), all-reduce(sum) the counter planes -- one contiguous buffer,
// one collective (lslam_occgrid_export_counters / _import_counters; shard.py drives RCCL) -- and classify locally.
#include <algorithm>
#include <cmath>
#include <vector>

#include "common.hpp"
#include "karto_math.hpp"

using namespace lslam;

namespace {

struct OccLaser {
  double min_angle, ang_res, min_range, max_range, range_threshold;
  int n_beams;
};

// flags: bit0 = traced, bit1 = end point valid
__global__ void __launch_bounds__(256)
k_occ_points(const double* __restrict__ ranges, int stride, const double* __restrict__ poses, OccLaser l,
             double2* __restrict__ ends, uint8_t* __restrict__ flags, double* __restrict__ bbox /* minx,miny,maxx,maxy */) {
  __shared__ double sh[4][256];
  const int b = blockIdx.x * blockDim.x + threadIdx.x, s = blockIdx.y, tid = threadIdx.x;
  double mnx = 999999999999999999.99999, mny = mnx, mxx = -mnx, mxy = -mnx;  // BoundingBox2() (Karto.h:2765)
  if (b < l.n_beams) {
    const double sx = poses[3 * s], sy = poses[3 * s + 1], sh_ = poses[3 * s + 2];
    const double r = ranges[(size_t)s * stride + b];
    double px, py;
    beam_world_point(sx, sy, sh_, l.min_angle, l.ang_res, (uint32_t)b, r, px, py);
    if (r >= l.min_range && r <= l.range_threshold) {  // filtered reading -> scan bounding box (Karto.h:5382,5421-5424)
      mnx = mxx = px;
      mny = mxy = py;
    }
    if (b == 0) {  // the box also holds the sensor position (Karto.h:5420)
      mnx = fmin(mnx, sx); mxx = fmax(mxx, sx);
      mny = fmin(mny, sy); mxy = fmax(mxy, sy);
    }
    uint8_t f = 0;
    // AddScan (Karto.h:5866-5885)
    if (!(r <= l.min_range || r >= l.max_range || isnan(r))) {
      f = 1;
      if (r < (l.range_threshold - kTol)) f |= 2;
      if (r >= l.range_threshold) {  // trace up to the range threshold only
        double ratio = l.range_threshold / r;
        double dx = px - sx, dy = py - sy;
        px = sx + ratio * dx;
        py = sy + ratio * dy;
      }
    }
    const size_t o = (size_t)s * l.n_beams + b;
    ends[o] = make_double2(px, py);
    flags[o] = f;
  }
  sh[0][tid] = mnx; sh[1][tid] = mny; sh[2][tid] = mxx; sh[3][tid] = mxy;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) {
      sh[0][tid] = fmin(sh[0][tid], sh[0][tid + o]);
      sh[1][tid] = fmin(sh[1][tid], sh[1][tid + o]);
      sh[2][tid] = fmax(sh[2][tid], sh[2][tid + o]);
      sh[3][tid] = fmax(sh[3][tid], sh[3][tid + o]);
    }
    __syncthreads();
  }
  if (tid < 4) {  // exact min/max: order of combination is irrelevant
    unsigned long long* a = (unsigned long long*)&bbox[tid];
    double v = sh[tid][0];
    unsigned long long old = *a;
    for (;;) {
      double cur = __longlong_as_double((long long)old);
      bool better = tid < 2 ? v < cur : v > cur;
      if (!better) break;
      unsigned long long prev = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
      if (prev == old) break;
      old = prev;
    }
  }
}

struct OccGeom {
  int w, h, stride;
  double scale, ox, oy;
};

// Grid<T>::TraceLine (Karto.h:4680-4745) in closed form: with deltaY <= deltaX the error recurrence
// "error += deltaY; if (2*error >= deltaX) { y += ystep; error -= deltaX; }" has taken
// q(k) = floor((2*k*deltaY + deltaX) / (2*deltaX)) minor steps before point k (k = 0..deltaX).
__global__ void __launch_bounds__(256)
k_occ_trace(int S, OccLaser l, const double* __restrict__ poses, const double2* __restrict__ ends,
            const uint8_t* __restrict__ flags, OccGeom g, uint32_t* __restrict__ pass, uint32_t* __restrict__ hit) {
  const int lane = threadIdx.x & 63;
  const long long beam = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (beam >= (long long)S * l.n_beams) return;
  const uint8_t f = flags[beam];
  if (!(f & 1)) return;
  const int s = (int)(beam / l.n_beams);
  // RayTrace (Karto.h:5907-5942)
  int x0 = world_to_grid(poses[3 * s], g.ox, g.scale), y0 = world_to_grid(poses[3 * s + 1], g.oy, g.scale);
  const double2 e = ends[beam];
  int x1 = world_to_grid(e.x, g.ox, g.scale), y1 = world_to_grid(e.y, g.oy, g.scale);
  const int tx = x1, ty = y1;
  const bool steep = abs(y1 - y0) > abs(x1 - x0);
  if (steep) { int t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
  if (x0 > x1) { int t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
  const int dX = x1 - x0, dY = abs(y1 - y0), ystep = y0 < y1 ? 1 : -1;
  for (int k = lane; k <= dX; k += 64) {
    const int q = dX > 0 ? (int)((2LL * k * dY + dX) / (2LL * dX)) : 0;
    const int x = x0 + k, y = y0 + ystep * q;
    const int px = steep ? y : x, py = steep ? x : y;
    if (px >= 0 && px < g.w && py >= 0 && py < g.h) atomicAdd(&pass[px + (size_t)py * g.stride], 1u);
  }
  if (lane == 0 && (f & 2) && tx >= 0 && tx < g.w && ty >= 0 && ty < g.h) {  // :5923-5938
    atomicAdd(&pass[tx + (size_t)ty * g.stride], 1u);
    atomicAdd(&hit[tx + (size_t)ty * g.stride], 1u);
  }
}

// UpdateCell (Karto.h:5950-5965): MinPassThrough = 2, OccupancyThreshold = 0.1 (:5636-5637);
// ros = 1 applies karto_slam.cc:546-569 (0 -> -1, 100 -> 100, 255 -> 0)
__global__ void __launch_bounds__(256)
k_occ_update(OccGeom g, const uint32_t* __restrict__ pass, const uint32_t* __restrict__ hit, uint8_t* __restrict__ out,
             int ros) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= g.w || y >= g.h) return;
  const uint32_t pc = pass[x + (size_t)y * g.stride], hc = hit[x + (size_t)y * g.stride];
  uint8_t v = 0;  // GridStates_Unknown
  if (pc > 2u) v = ((double)hc / (double)pc > 0.1) ? 100 : 255;
  if (ros) v = v == 0 ? (uint8_t)(int8_t)-1 : (v == 100 ? 100 : 0);
  out[(size_t)y * g.w + x] = v;
}

}  // namespace

struct lslam_occgrid {
  lslam_context* ctx = nullptr;
  OccGeom g{};
  uint32_t* d_pass = nullptr;  // one allocation: pass plane, then the hit plane
  uint32_t* d_hit = nullptr;
  size_t cells = 0;            // stride * h words per plane
  DevBuf<uint8_t> d_out;
};

namespace {

__global__ void __launch_bounds__(256)
k_occ_add(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}

constexpr double kBoxBig = 999999999999999999.99999;  // BoundingBox2() (Karto.h:2765)

// Shared body of the whole and the sharded build.  forced_box == nullptr: the grid is sized from these scans
// (ComputeDimensions); otherwise from the given box (the union over all shards).  out == nullptr: bounds only.
int occ_build(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges, int ranges_stride,
              const double* sensor_poses, double resolution, const double* forced_box, double* box_out,
              lslam_occgrid** out) {
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  OccLaser l;
  l.min_angle = laser->minimum_angle;
  l.ang_res = laser->angular_resolution;
  l.min_range = laser->minimum_range;
  l.max_range = laser->maximum_range;
  l.range_threshold = laser->range_threshold;
  l.n_beams = (int)(uint32_t)kround((laser->maximum_angle - laser->minimum_angle) / laser->angular_resolution);
  const int n = l.n_beams;
  if (n_scans > 0 && ranges_stride < n)
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "ranges_stride %d < num_beams %d", ranges_stride, n);
  const size_t total = (size_t)n_scans * std::max(n, 1);
  DevBuf<double> d_ranges, d_poses, d_bbox;
  DevBuf<double2> d_ends;
  DevBuf<uint8_t> d_flags;
  auto cleanup = [&]() { d_ranges.release(); d_poses.release(); d_bbox.release(); d_ends.release(); d_flags.release(); };
  double bbox[4] = {kBoxBig, kBoxBig, -kBoxBig, -kBoxBig};
  if (n_scans > 0) {
    if (d_ranges.reserve(total) != hipSuccess || d_poses.reserve((size_t)n_scans * 3) != hipSuccess ||
        d_bbox.reserve(4) != hipSuccess || d_ends.reserve(total) != hipSuccess || d_flags.reserve(total) != hipSuccess) {
      cleanup();
      return ctx->fail(LSLAM_ERR_HIP, "cannot allocate occupancy-grid workspaces");
    }
    hipError_t e = hipSuccess;
    if (n > 0)
      e = hipMemcpy2DAsync(d_ranges.p, (size_t)n * sizeof(double), ranges, (size_t)ranges_stride * sizeof(double),
                           (size_t)n * sizeof(double), n_scans, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_poses.p, sensor_poses, (size_t)n_scans * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_bbox.p, bbox, sizeof bbox, hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { cleanup(); return ctx->fail(LSLAM_ERR_HIP, "upload failed: %s", hipGetErrorString(e)); }
    if (n > 0)
      launch(ctx, "occ_points", k_occ_points, dim3((n + 255) / 256, n_scans), dim3(256), 0, (const double*)d_ranges.p, n,
             (const double*)d_poses.p, l, d_ends.p, d_flags.p, d_bbox.p);
    if (!forced_box) {
      e = hipMemcpyAsync(bbox, d_bbox.p, sizeof bbox, hipMemcpyDeviceToHost, ctx->stream);
      if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
      if (e != hipSuccess) { cleanup(); return ctx->fail(LSLAM_ERR_HIP, "bounding box failed: %s", hipGetErrorString(e)); }
      if (n == 0)  // scans without readings: the boxes hold the sensor positions only
        for (int s = 0; s < n_scans; s++) {
          bbox[0] = std::min(bbox[0], sensor_poses[3 * s]); bbox[1] = std::min(bbox[1], sensor_poses[3 * s + 1]);
          bbox[2] = std::max(bbox[2], sensor_poses[3 * s]); bbox[3] = std::max(bbox[3], sensor_poses[3 * s + 1]);
        }
    }
  }
  if (forced_box)
    for (int i = 0; i < 4; i++) bbox[i] = forced_box[i];
  if (box_out)
    for (int i = 0; i < 4; i++) box_out[i] = bbox[i];
  if (!out) {
    hipError_t e = hipStreamSynchronize(ctx->stream);
    cleanup();
    if (e != hipSuccess) return ctx->fail(LSLAM_ERR_HIP, "bounding box failed: %s", hipGetErrorString(e));
    return LSLAM_OK;
  }
  lslam_occgrid* og = new lslam_occgrid();
  og->ctx = ctx;
  OccGeom& g = og->g;
  // ComputeDimensions (Karto.h:5799-5817)
  g.scale = 1.0 / resolution;
  g.w = (int)kround((bbox[2] - bbox[0]) * g.scale);
  g.h = (int)kround((bbox[3] - bbox[1]) * g.scale);
  g.ox = bbox[0];
  g.oy = bbox[1];
  g.stride = (g.w + 7) & ~7;  // Grid<kt_int32u>::Resize (Karto.h:4442)
  const size_t cells = (size_t)g.stride * std::max(g.h, 0);
  // one allocation, pass plane then hit plane: the sharded build all-reduces both with ONE collective
  if (hipMalloc((void**)&og->d_pass, std::max<size_t>(cells, 1) * 8) != hipSuccess) {
    (void)hipGetLastError();
    cleanup();
    delete og;
    return ctx->fail(LSLAM_ERR_HIP, "cannot allocate %d x %d counters", g.w, g.h);
  }
  og->d_hit = og->d_pass + std::max<size_t>(cells, 1);
  og->cells = cells;
  (void)hipMemsetAsync(og->d_pass, 0, std::max<size_t>(cells, 1) * 8, ctx->stream);
  if (n > 0 && n_scans > 0 && cells > 0) {
    const long long beams = (long long)n_scans * n;
    launch(ctx, "occ_trace", k_occ_trace, dim3((unsigned)((beams + 3) / 4)), dim3(256), 0, n_scans, l,
           (const double*)d_poses.p, (const double2*)d_ends.p, (const uint8_t*)d_flags.p, g, og->d_pass, og->d_hit);
  }
  hipError_t e = hipStreamSynchronize(ctx->stream);
  cleanup();
  if (e != hipSuccess) { lslam_occgrid_destroy(og); return ctx->fail(LSLAM_ERR_HIP, "trace failed: %s", hipGetErrorString(e)); }
  *out = og;
  return LSLAM_OK;
}

int occ_check(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges, const double* sensor_poses) {
  if (!ctx || !laser || n_scans < 0 || (n_scans > 0 && (!ranges || !sensor_poses))) return LSLAM_ERR_INVALID_ARGUMENT;
  return LSLAM_OK;
}

}  // namespace

extern "C" {

int lslam_occgrid_create_from_scans(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges,
                                    int ranges_stride, const double* sensor_poses, double resolution,
                                    lslam_occgrid** out) {
  if (!out || occ_check(ctx, laser, n_scans, ranges, sensor_poses) != LSLAM_OK) return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (n_scans == 0) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "no scans (the reference returns NULL)");
  if (resolution == 0.0 || (resolution > -kTol && resolution < kTol))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "Resolution cannot be 0");  // Karto.h:5627-5630
  return occ_build(ctx, laser, n_scans, ranges, ranges_stride, sensor_poses, resolution, nullptr, nullptr, out);
}

int lslam_occgrid_scan_bounds(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges,
                              int ranges_stride, const double* sensor_poses, double box[4]) {
  if (!box || occ_check(ctx, laser, n_scans, ranges, sensor_poses) != LSLAM_OK) return LSLAM_ERR_INVALID_ARGUMENT;
  return occ_build(ctx, laser, n_scans, ranges, ranges_stride, sensor_poses, 1.0, nullptr, box, nullptr);
}

int lslam_occgrid_create_partial(lslam_context* ctx, const lslam_laser* laser, int n_scans, const double* ranges,
                                 int ranges_stride, const double* sensor_poses, double resolution, const double box[4],
                                 lslam_occgrid** out) {
  if (!out || !box || occ_check(ctx, laser, n_scans, ranges, sensor_poses) != LSLAM_OK) return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (resolution == 0.0 || (resolution > -kTol && resolution < kTol))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "Resolution cannot be 0");  // Karto.h:5627-5630
  if (!(box[0] <= box[2] && box[1] <= box[3]))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "empty bounding box (no scans on any shard: the reference returns NULL)");
  return occ_build(ctx, laser, n_scans, ranges, ranges_stride, sensor_poses, resolution, box, nullptr, out);
}

int lslam_occgrid_counter_words(const lslam_occgrid* og, size_t* words) {
  if (!og || !words) return LSLAM_ERR_INVALID_ARGUMENT;
  *words = 2 * og->cells;
  return LSLAM_OK;
}

void* lslam_occgrid_counters_dev_ptr(lslam_occgrid* og) { return og ? (void*)og->d_pass : nullptr; }

int lslam_occgrid_export_counters(lslam_occgrid* og, uint32_t* out, int on_device) {
  if (!og || (!out && og->cells)) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = og->ctx;
  if (og->cells == 0) return LSLAM_OK;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  LSLAM_HIP(ctx, hipMemcpyAsync(out, og->d_pass, 2 * og->cells * sizeof(uint32_t),
                                on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_occgrid_import_counters(lslam_occgrid* og, const uint32_t* in, int on_device, int accumulate) {
  if (!og || (!in && og->cells)) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = og->ctx;
  if (og->cells == 0) return LSLAM_OK;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  const size_t words = 2 * og->cells;
  if (!accumulate) {
    LSLAM_HIP(ctx, hipMemcpyAsync(og->d_pass, in, words * sizeof(uint32_t),
                                  on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  } else {
    const uint32_t* src = in;
    DevBuf<uint32_t> staged;
    if (!on_device) {
      LSLAM_HIP(ctx, staged.reserve(words));
      LSLAM_HIP(ctx, hipMemcpyAsync(staged.p, in, words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
      src = staged.p;
    }
    launch(ctx, "occ_add", k_occ_add, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, og->d_pass, src, words);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    staged.release();
    if (e != hipSuccess) return ctx->fail(LSLAM_ERR_HIP, "counter merge failed: %s", hipGetErrorString(e));
    return LSLAM_OK;
  }
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

void lslam_occgrid_destroy(lslam_occgrid* og) {
  if (!og) return;
  (void)hipStreamSynchronize(og->ctx->stream);
  if (og->d_pass) (void)hipFree(og->d_pass);
  og->d_out.release();
  delete og;
}

int lslam_occgrid_info(const lslam_occgrid* og, int32_t dims[2], double offset_xy[2], double* resolution) {
  if (!og) return LSLAM_ERR_INVALID_ARGUMENT;
  if (dims) { dims[0] = og->g.w; dims[1] = og->g.h; }
  if (offset_xy) { offset_xy[0] = og->g.ox; offset_xy[1] = og->g.oy; }
  if (resolution) *resolution = 1.0 / og->g.scale;
  return LSLAM_OK;
}

static int occ_read(lslam_occgrid* og, uint8_t* out, int ros) {
  if (!og || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = og->ctx;
  const OccGeom& g = og->g;
  const size_t n = (size_t)std::max(g.w, 0) * std::max(g.h, 0);
  if (n == 0) return LSLAM_OK;
  LSLAM_HIP(ctx, og->d_out.reserve(n));
  launch(ctx, "occ_update", k_occ_update, dim3((g.w + 255) / 256, g.h), dim3(256), 0, g, (const uint32_t*)og->d_pass,
         (const uint32_t*)og->d_hit, og->d_out.p, ros);
  LSLAM_HIP(ctx, hipMemcpyAsync(out, og->d_out.p, n, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_occgrid_read_u8(lslam_occgrid* og, uint8_t* out_host) { return occ_read(og, out_host, 0); }
int lslam_occgrid_read_ros_i8(lslam_occgrid* og, int8_t* out_host) { return occ_read(og, (uint8_t*)out_host, 1); }

}  // extern "C"
