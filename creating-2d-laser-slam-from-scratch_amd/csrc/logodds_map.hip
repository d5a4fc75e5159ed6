// Hector-style Bresenham ray-casting log-odds occupancy grid update on MI355X (gfx950).
//
// Reference behaviour reproduced (never copied): hectorslam::OccGridMapBase::updateByScan and
// friends (lesson4/include/lesson4/hector_mapping/map/OccGridMapBase.h:118-330), LogOddsCell /
// GridMapLogOddsFunctions (.../map/GridMapLogOdds.h:37-161), GridMapBase::setMapTransformation
// (.../map/GridMapBase.h:270-286), MapRepMultiMap pyramid (.../slam_main/MapRepMultiMap.h:57-93,
// 174-191).  H/ below = lesson4/include/lesson4/hector_mapping/.
//
// The reference walks the beams one after another and uses a per-cell updateIndex so that, per
// scan, every traversed cell gets logOddsFree ONCE and every hit cell gets logOddsOccupied ONCE,
// un-doing a free mark made earlier in the same scan (H/map/OccGridMapBase.h:302-330).  The only
// order-dependent bit is whether (v + free) - free is applied to a hit cell, i.e. whether some
// beam with a smaller index crossed the cell before the first beam that ends in it.
// Device formulation (one WAVE per beam, closed-form Bresenham cells, two kernels, no atomics on
// the float plane):
//   k_logodds_mark   walks the Bresenham line; atomicMax(free_key[cell]) along the ray and
//                    atomicMax(occ_key[end]) where key = epoch<<12 | (4095-beam): for the current
//                    epoch the max key is the SMALLEST beam index that touched the cell.
//   k_logodds_apply  walks again; the owner of a cell (min beam ending in it, else min beam crossing
//                    it) applies exactly the float operations the sequential reference applies.
// HBM-bound integer/byte work: coalescing comes from neighbouring beams crossing neighbouring
// cells; nothing here is GEMM-shaped.
#include <cmath>
#include <vector>

#include "common.hpp"

using namespace lslam;

namespace {

constexpr int kBeamBits = 12;
constexpr int kMaxBeams = 1 << kBeamBits;
constexpr uint32_t kBeamMask = kMaxBeams - 1;
constexpr uint32_t kMaxEpoch = (1u << (32 - kBeamBits)) - 1;

struct LevelGeom {
  int sx, sy;
  float c, s, tx, ty;      // pose transform: Translation(tx,ty) * Rotation (H/map/OccGridMapBase.h:127-129)
  float factor;            // DataPointContainer::setFrom factor of this level
  float ox, oy;            // origo (level-0 units)
  float lo_free, lo_occ;
  uint32_t epoch;
  int just_once;           // updateByScanJustOnce end-point rule
  int bx, by;              // begin cell
  double metres_per_cell;
};

struct Line {
  int x0, y0, x1, y1;
  bool valid;
};

// end cell of beam i and the in-map test of updateLineBresenhami (H/map/OccGridMapBase.h:143-161,
// 220-238)
__device__ __forceinline__ Line beam_line(const LevelGeom& g, const float* __restrict__ pts, int i) {
  Line l;
  l.x0 = g.bx; l.y0 = g.by;
  float px = pts[2 * i], py = pts[2 * i + 1];
  if (g.just_once) {
    // scanBeginMapi + (int)round(p / 0.05) with p in metres (:202-203); float / double -> double
    l.x1 = g.bx + (int)round((double)px / g.metres_per_cell);
    l.y1 = g.by + (int)round((double)py / g.metres_per_cell);
  } else {
    px = px * g.factor;  // setFrom (H/scan/DataPointContainer.h:54-57); factor 1 on level 0 is exact
    py = py * g.factor;
    float ex = (g.c * px + (-g.s) * py) + g.tx;
    float ey = (g.s * px + g.c * py) + g.ty;
    ex += 0.5f;
    ey += 0.5f;
    l.x1 = (int)ex;
    l.y1 = (int)ey;
  }
  l.valid = !(l.x0 == l.x1 && l.y0 == l.y1) && l.x0 >= 0 && l.x0 < g.sx && l.y0 >= 0 && l.y0 < g.sy &&
            l.x1 >= 0 && l.x1 < g.sx && l.y1 >= 0 && l.y1 < g.sy;
  return l;
}

// Bresenham traversal of H/map/OccGridMapBase.h:240-299 in CLOSED FORM.  The reference loop adds
// abs_db to an error term that starts at abs_da/2 and takes a minor-axis step whenever it reaches
// abs_da; since abs_db <= abs_da that is at most one minor step per major step, so after i major
// steps exactly q(i) = floor((abs_da/2 + i*abs_db) / abs_da) minor steps were taken.  Cell i of the
// ray (i = 0 .. abs_da-1, end point excluded) is therefore start + i*offset_a + q(i)*offset_b --
// independent of the other cells, so a whole wave walks ONE ray, 64 cells at a time, instead of one
// thread crawling it cell by cell with a dependent atomic per step.
struct Ray {
  unsigned start, abs_da, abs_db;
  int offset_a, offset_b;
};
__device__ __forceinline__ Ray ray_of(const Line& l, int sx) {
  int dx = l.x1 - l.x0, dy = l.y1 - l.y0;
  unsigned abs_dx = (unsigned)abs(dx), abs_dy = (unsigned)abs(dy);
  int offset_dx = dx > 0 ? 1 : -1;  // util::sign: sign(0) = -1
  int offset_dy = (dy > 0 ? 1 : -1) * sx;
  Ray r;
  r.start = (unsigned)(l.y0 * sx + l.x0);
  if (abs_dx >= abs_dy) {
    r.abs_da = abs_dx; r.abs_db = abs_dy; r.offset_a = offset_dx; r.offset_b = offset_dy;
  } else {
    r.abs_da = abs_dy; r.abs_db = abs_dx; r.offset_a = offset_dy; r.offset_b = offset_dx;
  }
  return r;
}
__device__ __forceinline__ unsigned ray_cell(const Ray& r, unsigned i) {
  unsigned q = (unsigned)(((unsigned long long)(r.abs_da / 2) + (unsigned long long)i * r.abs_db) / r.abs_da);
  return r.start + (unsigned)((int)i * r.offset_a) + (unsigned)((int)q * r.offset_b);
}

// one wave per beam
__global__ void __launch_bounds__(256)
k_logodds_mark(LevelGeom g, const float* __restrict__ pts, int n, uint32_t* __restrict__ free_key,
               uint32_t* __restrict__ occ_key) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  Line l = beam_line(g, pts, i);
  if (!l.valid) return;
  const uint32_t key = (g.epoch << kBeamBits) | (kBeamMask - (uint32_t)i);
  const Ray r = ray_of(l, g.sx);
  for (unsigned c = lane; c < r.abs_da; c += 64) atomicMax(&free_key[ray_cell(r, c)], key);
  if (lane == 0) atomicMax(&occ_key[(unsigned)(l.y1 * g.sx + l.x1)], key);
}

__global__ void __launch_bounds__(256)
k_logodds_apply(LevelGeom g, const float* __restrict__ pts, int n, const uint32_t* __restrict__ free_key,
                const uint32_t* __restrict__ occ_key, float* __restrict__ logodds) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  Line l = beam_line(g, pts, i);
  if (!l.valid) return;
  const uint32_t me = kBeamMask - (uint32_t)i;
  const uint32_t ep = g.epoch;
  const Ray r = ray_of(l, g.sx);
  // crossed cells: free once per scan unless some beam ends here (bresenhamCellFree, :302-313)
  for (unsigned c = lane; c < r.abs_da; c += 64) {
    const unsigned off = ray_cell(r, c);
    uint32_t fk = free_key[off];
    if ((fk & kBeamMask) != me) continue;             // not the first beam crossing this cell
    if ((occ_key[off] >> kBeamBits) == ep) continue;  // a hit cell: handled by its occ owner
    logodds[off] += g.lo_free;
  }
  // end cell (bresenhamCellOcc, :316-330)
  if (lane == 0) {
    unsigned eoff = (unsigned)(l.y1 * g.sx + l.x1);
    uint32_t ok = occ_key[eoff];
    if ((ok & kBeamMask) == me) {  // first beam ending here
      float v = logodds[eoff];
      uint32_t fk = free_key[eoff];
      if ((fk >> kBeamBits) == ep && (fk & kBeamMask) > me) {  // crossed by an EARLIER beam: free then unset
        v += g.lo_free;
        v -= g.lo_free;
      }
      if (v < 50.0f) v += g.lo_occ;  // updateSetOccupied (H/map/GridMapLogOdds.h:108-114)
      logodds[eoff] = v;
    }
  }
}

__global__ void k_occupancy_i8(const float* __restrict__ v, int8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = v[i];
  out[i] = x < 0.0f ? (int8_t)0 : (x > 0.0f ? (int8_t)100 : (int8_t)-1);  // hector_slam.cc:287-304
}

float prob_to_logodds(float prob) {  // H/map/GridMapLogOdds.h:151-155 (log() is the double overload)
  float odds = prob / (1.0f - prob);
  return (float)log((double)odds);
}

struct Level {
  int sx = 0, sy = 0;
  float cell_length = 0.f, scale_to_map = 0.f, t_x = 0.f, t_y = 0.f;
  float* d_logodds = nullptr;
  uint32_t* d_free = nullptr;
  uint32_t* d_occ = nullptr;
  uint32_t epoch = 0;
};

}  // namespace

struct lslam_map {
  lslam_context* ctx = nullptr;
  std::vector<Level> levels;
  float off_x = 0.f, off_y = 0.f;
  float lo_free = 0.f, lo_occ = 0.f;
  DevBuf<float> d_pts;
  DevBuf<int8_t> d_i8;
};

namespace {

int clear_marks(lslam_map* map, Level& L) {
  lslam_context* ctx = map->ctx;
  size_t n = (size_t)L.sx * L.sy;
  LSLAM_HIP(ctx, hipMemsetAsync(L.d_free, 0, n * sizeof(uint32_t), ctx->stream));
  LSLAM_HIP(ctx, hipMemsetAsync(L.d_occ, 0, n * sizeof(uint32_t), ctx->stream));
  L.epoch = 0;
  return LSLAM_OK;
}

int update_impl(lslam_map* map, const float* d_pts, int n, const float origo[2], const float pose[3],
                int just_once, float begin_x, float begin_y, double metres_per_cell) {
  lslam_context* ctx = map->ctx;
  if (n > kMaxBeams)
    return ctx->fail(LSLAM_ERR_UNSUPPORTED, "at most %d points per scan are supported (got %d)", kMaxBeams, n);
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  const int n_levels = just_once ? 1 : (int)map->levels.size();
  for (int li = 0; li < n_levels; li++) {
    Level& L = map->levels[li];
    if (L.epoch >= kMaxEpoch) {
      int rc = clear_marks(map, L);
      if (rc) return rc;
    }
    L.epoch++;
    LevelGeom g;
    g.sx = L.sx; g.sy = L.sy;
    g.lo_free = map->lo_free; g.lo_occ = map->lo_occ;
    g.epoch = L.epoch;
    g.just_once = just_once;
    g.metres_per_cell = metres_per_cell;
    // DataPointContainer::setFrom factor (H/slam_main/MapRepMultiMap.h:161)
    g.factor = li == 0 ? 1.0f : (float)(1.0 / pow(2.0, (double)li));
    float ox = li == 0 ? origo[0] : origo[0] * g.factor;
    float oy = li == 0 ? origo[1] : origo[1] * g.factor;
    g.ox = ox; g.oy = oy;
    float mx, my, ang;
    if (just_once) {
      mx = begin_x; my = begin_y; ang = 0.0f;  // mapPose(800, 800, 0) (H/map/OccGridMapBase.h:182)
    } else {
      // getMapCoordsPose (H/map/GridMapBase.h:238-242), mapTworld = Scale*Translate (:278)
      float s = L.scale_to_map;
      mx = (s * pose[0] + 0.0f * pose[1]) + L.t_x;
      my = (0.0f * pose[0] + s * pose[1]) + L.t_y;
      ang = pose[2];
    }
    // host libm, exactly what the reference's Eigen::Rotation2Df evaluates
    g.c = cosf(ang);
    g.s = sinf(ang);
    g.tx = mx; g.ty = my;
    float bxf = (g.c * ox + (-g.s) * oy) + mx;  // :132
    float byf = (g.s * ox + g.c * oy) + my;
    g.bx = (int)(bxf + 0.5f);                   // :135
    g.by = (int)(byf + 0.5f);
    if (n > 0) {
      dim3 grid((n + 3) / 4), block(256);  // 4 waves per block, one wave per beam
      launch(ctx, "logodds_mark", k_logodds_mark, grid, block, 0, g, d_pts, n, L.d_free, L.d_occ);
      launch(ctx, "logodds_apply", k_logodds_apply, grid, block, 0, g, d_pts, n, (const uint32_t*)L.d_free,
             (const uint32_t*)L.d_occ, L.d_logodds);
    }
  }
  LSLAM_HIP(ctx, hipGetLastError());
  return LSLAM_OK;
}

}  // namespace

extern "C" {

int lslam_map_create(lslam_context* ctx, int size_x, int size_y, float cell_length, float offset_x,
                     float offset_y, int levels, lslam_map** out) {
  if (!ctx || !out || size_x <= 0 || size_y <= 0 || !(cell_length > 0.f) || levels < 1 || levels > 16)
    return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (hipSetDevice(ctx->device) != hipSuccess) return ctx->fail(LSLAM_ERR_HIP, "hipSetDevice failed");
  lslam_map* map = new lslam_map();
  map->ctx = ctx;
  map->off_x = offset_x;
  map->off_y = offset_y;
  map->lo_free = prob_to_logodds(0.4f);  // GridMapLogOddsFunctions ctor (H/map/GridMapLogOdds.h:98-102)
  map->lo_occ = prob_to_logodds(0.6f);
  int sx = size_x, sy = size_y;
  float cl = cell_length;
  for (int i = 0; i < levels; i++) {
    if (sx <= 0 || sy <= 0) break;
    Level L;
    L.sx = sx; L.sy = sy;
    L.cell_length = cl;
    L.scale_to_map = 1.0f / cl;            // H/map/GridMapBase.h:276
    L.t_x = L.scale_to_map * offset_x;     // Scale * Translate (:278)
    L.t_y = L.scale_to_map * offset_y;
    size_t n = (size_t)sx * sy;
    if (hipMalloc((void**)&L.d_logodds, n * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&L.d_free, n * sizeof(uint32_t)) != hipSuccess ||
        hipMalloc((void**)&L.d_occ, n * sizeof(uint32_t)) != hipSuccess) {
      map->levels.push_back(L);
      lslam_map_destroy(map);
      return ctx->fail(LSLAM_ERR_HIP, "cannot allocate map level %d (%dx%d) in HBM", i, sx, sy);
    }
    (void)hipMemsetAsync(L.d_logodds, 0, n * sizeof(float), ctx->stream);
    (void)hipMemsetAsync(L.d_free, 0, n * sizeof(uint32_t), ctx->stream);
    (void)hipMemsetAsync(L.d_occ, 0, n * sizeof(uint32_t), ctx->stream);
    map->levels.push_back(L);
    sx /= 2;     // resolution /= 2 (H/slam_main/MapRepMultiMap.h:83)
    sy /= 2;
    cl *= 2.0f;  // :84
  }
  (void)hipStreamSynchronize(ctx->stream);
  *out = map;
  return LSLAM_OK;
}

void lslam_map_destroy(lslam_map* map) {
  if (!map) return;
  (void)hipSetDevice(map->ctx->device);
  (void)hipStreamSynchronize(map->ctx->stream);
  for (auto& L : map->levels) {
    if (L.d_logodds) (void)hipFree(L.d_logodds);
    if (L.d_free) (void)hipFree(L.d_free);
    if (L.d_occ) (void)hipFree(L.d_occ);
  }
  map->d_pts.release();
  map->d_i8.release();
  delete map;
}

int lslam_map_reset(lslam_map* map) {
  if (!map) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  for (auto& L : map->levels) {
    LSLAM_HIP(ctx, hipMemsetAsync(L.d_logodds, 0, (size_t)L.sx * L.sy * sizeof(float), ctx->stream));
    int rc = clear_marks(map, L);
    if (rc) return rc;
  }
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_map_set_update_factor_free(lslam_map* map, float p) {
  if (!map) return LSLAM_ERR_INVALID_ARGUMENT;
  map->lo_free = prob_to_logodds(p);
  return LSLAM_OK;
}
int lslam_map_set_update_factor_occupied(lslam_map* map, float p) {
  if (!map) return LSLAM_ERR_INVALID_ARGUMENT;
  map->lo_occ = prob_to_logodds(p);
  return LSLAM_OK;
}
int lslam_map_levels(const lslam_map* map) { return map ? (int)map->levels.size() : LSLAM_ERR_INVALID_ARGUMENT; }
int lslam_map_size(const lslam_map* map, int level, int* sx, int* sy) {
  if (!map || level < 0 || level >= (int)map->levels.size()) return LSLAM_ERR_INVALID_ARGUMENT;
  if (sx) *sx = map->levels[level].sx;
  if (sy) *sy = map->levels[level].sy;
  return LSLAM_OK;
}
float lslam_map_scale_to_map(const lslam_map* map, int level) {
  if (!map || level < 0 || level >= (int)map->levels.size()) return 0.f;
  return map->levels[level].scale_to_map;
}

int lslam_map_update_by_scan_dev(lslam_map* map, const float* pts_dev, int n, const float origo[2],
                                 const float pose[3]) {
  if (!map || n < 0 || (n > 0 && !pts_dev) || !origo || !pose) return LSLAM_ERR_INVALID_ARGUMENT;
  return update_impl(map, pts_dev, n, origo, pose, 0, 0.f, 0.f, 0.0);
}

int lslam_map_update_by_scan(lslam_map* map, const float* pts, int n, const float origo[2], const float pose[3]) {
  if (!map || n < 0 || (n > 0 && !pts) || !origo || !pose) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  LSLAM_HIP(ctx, map->d_pts.reserve((size_t)2 * (n > 0 ? n : 1)));
  if (n > 0)
    LSLAM_HIP(ctx, hipMemcpyAsync(map->d_pts.p, pts, (size_t)2 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  int rc = update_impl(map, map->d_pts.p, n, origo, pose, 0, 0.f, 0.f, 0.0);
  if (rc) return rc;
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_map_update_just_once(lslam_map* map, const float* pts, int n, const float origo[2], float begin_x,
                               float begin_y, double metres_per_cell) {
  if (!map || n < 0 || (n > 0 && !pts) || !origo || !(metres_per_cell > 0)) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  LSLAM_HIP(ctx, map->d_pts.reserve((size_t)2 * (n > 0 ? n : 1)));
  if (n > 0)
    LSLAM_HIP(ctx, hipMemcpyAsync(map->d_pts.p, pts, (size_t)2 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  const float pose[3] = {0.f, 0.f, 0.f};
  int rc = update_impl(map, map->d_pts.p, n, origo, pose, 1, begin_x, begin_y, metres_per_cell);
  if (rc) return rc;
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_map_read_logodds(lslam_map* map, int level, float* out) {
  if (!map || !out || level < 0 || level >= (int)map->levels.size()) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  Level& L = map->levels[level];
  LSLAM_HIP(ctx, hipMemcpyAsync(out, L.d_logodds, (size_t)L.sx * L.sy * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_map_read_occupancy_i8(lslam_map* map, int level, int8_t* out) {
  if (!map || !out || level < 0 || level >= (int)map->levels.size()) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  Level& L = map->levels[level];
  size_t n = (size_t)L.sx * L.sy;
  LSLAM_HIP(ctx, map->d_i8.reserve(n));
  launch(ctx, "occupancy_i8", k_occupancy_i8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
         (const float*)L.d_logodds, map->d_i8.p, n);
  LSLAM_HIP(ctx, hipMemcpyAsync(out, map->d_i8.p, n, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

void* lslam_map_cells_dev_ptr(lslam_map* map, int level) {
  if (!map || level < 0 || level >= (int)map->levels.size()) return nullptr;
  return map->levels[level].d_logodds;
}

}  // extern "C"
