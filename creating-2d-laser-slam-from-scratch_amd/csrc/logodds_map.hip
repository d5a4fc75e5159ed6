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
//                    atomicMax(occ_key[end]) where key = epoch<<16 | (65535-beam): for the current
//                    epoch the max key is the SMALLEST beam index that touched the cell.
//   k_logodds_apply  walks again; the owner of a cell (min beam ending in it, else min beam crossing
//                    it) applies exactly the float operations the sequential reference applies.
// HBM-bound integer/byte work: coalescing comes from neighbouring beams crossing neighbouring
// cells; nothing here is GEMM-shaped.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "common.hpp"

using namespace lslam;

namespace {

constexpr int kBeamBits = 16;  // 65536 points per container; the epoch keeps 16 bits (marks cleared every 65535 scans)
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
  bool representable;
  float px = pts[2 * i], py = pts[2 * i + 1];
  if (g.just_once) {
    // scanBeginMapi + (int)round(p / 0.05) with p in metres (:202-203); float / double -> double
    const double rx = round((double)px / g.metres_per_cell), ry = round((double)py / g.metres_per_cell);
    // x86 cvttsd2si yields INT_MIN for NaN / Inf / out-of-range and the in-map test then drops the beam; the
    // gfx950 conversion saturates instead (NaN -> 0), so reject those end points explicitly
    representable = fabs(rx) < 2147483648.0 && fabs(ry) < 2147483648.0;
    l.x1 = g.bx + (int)rx;
    l.y1 = g.by + (int)ry;
  } else {
    px = px * g.factor;  // setFrom (H/scan/DataPointContainer.h:54-57); factor 1 on level 0 is exact
    py = py * g.factor;
    float ex = (g.c * px + (-g.s) * py) + g.tx;
    float ey = (g.s * px + g.c * py) + g.ty;
    ex += 0.5f;
    ey += 0.5f;
    representable = fabsf(ex) < 2147483648.0f && fabsf(ey) < 2147483648.0f;  // false for NaN / Inf too
    l.x1 = (int)ex;
    l.y1 = (int)ey;
  }
  l.valid = representable && !(l.x0 == l.x1 && l.y0 == l.y1) && l.x0 >= 0 && l.x0 < g.sx && l.y0 >= 0 && l.y0 < g.sy &&
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
__device__ __forceinline__ unsigned ray_minor(const Ray& r, unsigned i) {  // minor-axis steps taken after i major steps
  return (unsigned)(((unsigned long long)(r.abs_da / 2) + (unsigned long long)i * r.abs_db) / r.abs_da);
}
__device__ __forceinline__ unsigned ray_cell(const Ray& r, unsigned i) {
  return r.start + (unsigned)((int)i * r.offset_a) + (unsigned)((int)ray_minor(r, i) * r.offset_b);
}

// "does the traversal of beam P (same origin) visit the cell at signed offset (ox, oy) from the origin?"  The visited
// cells of a ray are exactly { A major steps, q_P(A) minor steps : 0 <= A < abs_da } (end point excluded), so membership
// is one closed-form test.  Used by the mark pass: a beam whose PREDECESSOR also crosses a cell cannot be the smallest
// beam index crossing it, so its atomicMax on that cell cannot change the result and is skipped.  Near the sensor every
// cell is crossed by hundreds of beams; without this the same-address atomics of those cells serialise in L2 and set the
// duration of the whole pass.
struct RayShape {
  unsigned abs_da, abs_db;
  int sgn_x, sgn_y;  // util::sign: sign(0) = -1
  bool x_major, valid;
};
__device__ __forceinline__ RayShape shape_of(const Line& l) {
  RayShape p;
  const int dx = l.x1 - l.x0, dy = l.y1 - l.y0;
  const unsigned adx = (unsigned)abs(dx), ady = (unsigned)abs(dy);
  p.sgn_x = dx > 0 ? 1 : -1;
  p.sgn_y = dy > 0 ? 1 : -1;
  p.x_major = adx >= ady;
  p.abs_da = p.x_major ? adx : ady;
  p.abs_db = p.x_major ? ady : adx;
  p.valid = l.valid;
  return p;
}
__device__ __forceinline__ bool ray_visits(const RayShape& p, int ox, int oy) {
  const int A = p.x_major ? ox * p.sgn_x : oy * p.sgn_y;
  const int M = p.x_major ? oy * p.sgn_y : ox * p.sgn_x;
  if (A < 0 || (unsigned)A >= p.abs_da || M < 0) return false;
  // A < abs_da <= 2^16 on any map this library allocates, so the product stays below 2^32
  return (unsigned)M == (p.abs_da / 2u + (unsigned)A * p.abs_db) / p.abs_da;
}

// one wave per beam; `wave` = index of this wave among the kernel's mark (resp. apply) waves
__device__ __forceinline__ void logodds_mark_wave(const LevelGeom& g, const float* __restrict__ pts, int n, int wave, int lane,
                                                  uint32_t* __restrict__ free_key, uint32_t* __restrict__ occ_key,
                                                  float* __restrict__ pts_copy) {
  const int i = wave;
  if (i >= n) return;
  if (pts_copy && lane < 2) pts_copy[2 * i + lane] = pts[2 * i + lane];  // kept for the deferred apply (pipelined path)
  Line l = beam_line(g, pts, i);
  if (!l.valid) return;
  const uint32_t key = (g.epoch << kBeamBits) | (kBeamMask - (uint32_t)i);
  const Ray r = ray_of(l, g.sx);
  const RayShape me = shape_of(l);
  RayShape prev;
  prev.valid = false;
  if (i > 0 && me.abs_da < 65536u) prev = shape_of(beam_line(g, pts, i - 1));
  prev.valid = prev.valid && prev.abs_da < 65536u;
  for (unsigned c = lane; c < r.abs_da; c += 64) {
    const unsigned q = ray_minor(r, c);
    if (prev.valid) {
      const int ox = me.x_major ? (int)c * me.sgn_x : (int)q * me.sgn_x;
      const int oy = me.x_major ? (int)q * me.sgn_y : (int)c * me.sgn_y;
      if (ray_visits(prev, ox, oy)) continue;  // beam i-1 marks this cell with a larger key
    }
    atomicMax(&free_key[r.start + (unsigned)((int)c * r.offset_a) + (unsigned)((int)q * r.offset_b)], key);
  }
  if (lane == 0) atomicMax(&occ_key[(unsigned)(l.y1 * g.sx + l.x1)], key);
}

__device__ __forceinline__ void logodds_apply_wave(const LevelGeom& g, const float* __restrict__ pts, int n, int wave, int lane,
                                                   const uint32_t* __restrict__ free_key, const uint32_t* __restrict__ occ_key,
                                                   float* __restrict__ logodds) {
  const int i = wave;
  if (i >= n) return;
  Line l = beam_line(g, pts, i);
  if (!l.valid) return;
  const uint32_t me = kBeamMask - (uint32_t)i;
  const uint32_t ep = g.epoch;
  const Ray r = ray_of(l, g.sx);
  // crossed cells: free once per scan unless some beam ends here (bresenhamCellFree, :302-313).
  // Four cells per lane and pass, their three reads issued together: the kernel is a chain of dependent
  // L2 round trips, not bandwidth, so the reads of non-owners are the cheaper evil.
  constexpr int kIlp = 4;
  for (unsigned c0 = lane; c0 < r.abs_da; c0 += 64 * kIlp) {
    unsigned off[kIlp];
    uint32_t fk[kIlp], ok[kIlp];
    float lo[kIlp];
#pragma unroll
    for (int u = 0; u < kIlp; u++) {
      const unsigned c = c0 + 64u * u;
      off[u] = ray_cell(r, c < r.abs_da ? c : c0);
    }
#pragma unroll
    for (int u = 0; u < kIlp; u++) {
      fk[u] = free_key[off[u]];
      ok[u] = occ_key[off[u]];
      lo[u] = logodds[off[u]];
    }
#pragma unroll
    for (int u = 0; u < kIlp; u++) {
      if (c0 + 64u * u >= r.abs_da) continue;
      if ((fk[u] & kBeamMask) != me) continue;       // not the first beam crossing this cell
      if ((ok[u] >> kBeamBits) == ep) continue;      // a hit cell: handled by its occ owner
      logodds[off[u]] = lo[u] + g.lo_free;           // the owner is the only writer of this cell in this kernel
    }
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

__global__ void __launch_bounds__(256)
k_logodds_mark(LevelGeom g, const float* __restrict__ pts, int n, uint32_t* __restrict__ free_key,
               uint32_t* __restrict__ occ_key) {
  logodds_mark_wave(g, pts, n, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), threadIdx.x & 63, free_key, occ_key, nullptr);
}

__global__ void __launch_bounds__(256)
k_logodds_apply(LevelGeom g, const float* __restrict__ pts, int n, const uint32_t* __restrict__ free_key,
                const uint32_t* __restrict__ occ_key, float* __restrict__ logodds) {
  logodds_apply_wave(g, pts, n, blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), threadIdx.x & 63, free_key, occ_key, logodds);
}

// PIPELINED single-scan update: ONE launch per scan in steady state.  The two kernels above are two dependent ~10 us
// launches whose only link is "apply needs every mark of its scan".  But the apply of scan t-1 and the mark of scan t are
// independent once the two scans keep their keys in different planes: the mark of scan t goes into key-plane set t & 1,
// while the apply of scan t-1 reads set (t-1) & 1 and is the launch's only writer of the float plane.  So each call
// launches [apply blocks of the previous scan | mark blocks of this scan] together and leaves this scan's apply pending;
// a reader (lslam_map_read_*, matchData, a batched update, lslam_synchronize, ...) first flushes the pending apply with the
// stand-alone kernel.  Per cell the float operations and their order are exactly those of the sequential reference:
// apply(t-1) completes in the launch before apply(t) starts (stream order).  The mark blocks also copy their points
// aside (the caller's buffer may be reused before the deferred apply runs).
// As ONE launch for EVERY pyramid level (round 4; a kernel per level until then): a 3-level MapRepMultiMap update was three pipe launches (and,
// when a reader follows every scan -- matchData in lesson4's loop -- three more stand-alone applies), each a ~8 us
// dependent step of a latency-bound chain.  A job = the apply of a level's pending scan or the mark of its new one; the
// blocks of all jobs share the grid, a block finds its job by its index (a uniform select chain over the kernel
// arguments: no dynamic indexing, no table in memory to keep alive behind an asynchronous call).  Within a level the
// invariants of k_logodds_pipe hold unchanged; levels are independent planes.
struct PipeJob {
  LevelGeom g;
  const float* pts;
  int n, first_block, apply;
  const uint32_t* free_r;
  const uint32_t* occ_r;
  uint32_t* free_w;
  uint32_t* occ_w;
  float* logodds;
  float* pts_copy;
};
constexpr int kPipeMaxJobs = 12;
struct PipeJobs {
  int n_jobs;
  PipeJob j[kPipeMaxJobs];
};
__global__ void __launch_bounds__(256)
k_logodds_pipe_ml(PipeJobs J) {
  const int b = (int)blockIdx.x;
  PipeJob cur = J.j[0];
#pragma unroll
  for (int k = 1; k < kPipeMaxJobs; k++)
    if (k < J.n_jobs && b >= J.j[k].first_block) cur = J.j[k];
  // Workgroups go to the eight XCDs round robin (block b -> XCD b % 8) and every XCD has its OWN L2: with beams dealt
  // out in block order, each XCD walked every eighth group of four beams -- all eight of them touched nearly every cell of
  // the scan's fan and fetched its lines separately (measured 4.3x the algorithmic bytes at 1000^2, 5.8x at 4000^2).  A
  // job's blocks are a multiple of 8 and start at a multiple of 8, so block (r, q) = (lb % 8, lb / 8) runs on XCD r:
  // XCD r takes the r-th CONTIGUOUS eighth of the beams -- one 34-degree sector of the fan per L2.
  const int lb = b - cur.first_block, per_xcd = ((cur.n + 3) / 4 + 7) / 8;
  const int lane = threadIdx.x & 63, wave = ((lb & 7) * per_xcd + (lb >> 3)) * (int)(blockDim.x >> 6) + (int)(threadIdx.x >> 6);
  if (cur.apply)
    logodds_apply_wave(cur.g, cur.pts, cur.n, wave, lane, cur.free_r, cur.occ_r, cur.logodds);
  else
    logodds_mark_wave(cur.g, cur.pts, cur.n, wave, lane, cur.free_w, cur.occ_w, cur.pts_copy);
}

// ------------------------------------------------------------------------------------------
// BATCHED update: K scans, each with its own pose, applied to the map exactly as K successive
// updateByScan calls would (the float operations of a cell are applied in scan order; cells are
// independent of each other).  The single-scan path above is bound by two dependent ~10 us kernels
// per scan; here the K scans are MARKED in parallel and APPLIED by one pass over the touched tiles:
//   k_lo_batch_hits    thread per (scan, beam): end cell -> byte plane[s][cell] = HIT, and the cell is
//                      entered in scan s's small hash with atomicMin(first beam ending there);
//   k_lo_batch_rays    wave per (scan, beam), closed-form Bresenham cells: a crossed cell whose byte is not
//                      HIT gets plane[s][cell] = CROSSED (plain byte store, every writer writes the same
//                      value); a crossed HIT cell records atomicMin(first beam crossing it) in the hash;
//   k_lo_batch_resolve thread per hash entry: a hit cell crossed by an earlier beam of its scan -> byte HIT_UNDO;
//   k_lo_batch_apply   wave per 8x8-cell tile: reads the tile's 64 "touched by scan s" flags in one load, then
//                      walks ONLY the planes that touched the tile, in scan order, and applies
//                      +free / [(v+free)-free] +occ-if-<50, the float sequence of the sequential reference
//                      (H/map/OccGridMapBase.h:302-330).
// Layout: the byte planes are TILED and WINDOWED.  A tile is 8x8 cells = one 64-byte line, so 64 consecutive cells of a
// ray lie in ~8 lines whichever way the ray runs (instead of 1 line for an x-major and 64 for a y-major ray in a
// row-major plane).  Scan s does not own a plane of the whole map but a WINDOW of tiles -- the square of tiles its rays
// can reach from its begin cell (host: begin cell +- the scan's longest point, ScanHdr::tx0 / ty0 / tw / th) -- stored
// densely in a slot pool: cell (x, y) of scan s is byte
//     pool[(base_s + (y>>3 - ty0_s) * tw_s + (x>>3 - tx0_s)) * 64 + (y&7)*8 + (x&7)].
// A 1081-beam scan with 20 m of range on a 0.025 m map owns 202 x 202 tiles = 2.6 MB whatever the size of the map (a
// plane of a 4000 x 4000 map is 16 MB, of an 8000 x 8000 map 64 MB); a window clipped by a small map IS the plane.
// The scans of a call are cut into ROUNDS whose windows fit the scratch budget (LSLAM_MAP_OPT_BATCH_SCRATCH_MB,
// default 192 MB): normally one.  flags[tile][s] (64 bytes per tile) is what lets the apply pass skip the ~90 % of
// (tile, scan) pairs no ray went through.  Pool and flag bytes carry a 6-bit round epoch, so they are cleared once per
// 63 rounds, not per batch.  HBM-bound byte work; no atomics on the traversal path.
// ------------------------------------------------------------------------------------------
struct ScanHdr {     // one scan of a batch on one pyramid level (host-computed like LevelGeom)
  float c, s, tx, ty;
  int bx, by;        // begin cell
  int n, pts_off;    // points of this scan: pts[2*(pts_off + i)]
  // the scan's window of tiles (clipped to the map; tw = 0: nothing of it is inside) and its first tile slot in the pool:
  // one aligned 16-byte record at offset 32, which k_lo_batch_apply reads with one vector load (lane = scan)
  int tx0, ty0, tw;
  uint32_t base;     // < 2^26: byte offsets into the pool stay below 2^32
  int th, pad[3];
};
static_assert(sizeof(ScanHdr) == 64, "ScanHdr: 64 bytes, window record at offset 32");
struct BatchGeom {
  int sx, sy, K;
  int tiles_x, n_tiles;
  float factor, lo_free, lo_occ;
  uint32_t tag;      // epoch << 2
  uint32_t hash_mask;  // slots per scan - 1 (power of two >= 2 * max points per scan)
};
constexpr uint32_t kCodeCrossed = 1u, kCodeHit = 2u, kCodeHitUndo = 3u;
constexpr uint32_t kHashEmpty = 0xFFFFFFFFu;
constexpr int kBatchSlots = 64;  // scans per batch = flag bytes per tile
#if !defined(LSLAM_TUNE_APPLY_CHUNK)
#define LSLAM_TUNE_APPLY_CHUNK 8
#endif
constexpr int kApplyChunk = LSLAM_TUNE_APPLY_CHUNK;  // planes whose bytes k_lo_batch_apply has in flight at a time (A/B, round 6:
                                                     // 8 / 16 / 32 / 64 -> 0.271 / 0.261 / 0.292 / 0.327 ms per 1000 scans: the tiles every
                                                     // scan touches are NOT what the launch waits for; the unrolled slot bookkeeping is)

__device__ __forceinline__ Line batch_line(const BatchGeom& g, const ScanHdr& h, const float* __restrict__ pts, int i) {
  LevelGeom lg;
  lg.sx = g.sx; lg.sy = g.sy; lg.c = h.c; lg.s = h.s; lg.tx = h.tx; lg.ty = h.ty; lg.factor = g.factor;
  lg.just_once = 0; lg.bx = h.bx; lg.by = h.by; lg.metres_per_cell = 0.0;
  return beam_line(lg, pts + 2 * (size_t)h.pts_off, i);
}
__device__ __forceinline__ uint32_t hash_slot0(uint32_t cell, uint32_t mask) { return (cell * 2654435761u) >> 7 & mask; }
__device__ __forceinline__ uint32_t tile_of(const BatchGeom& g, int x, int y) { return (uint32_t)((y >> 3) * g.tiles_x + (x >> 3)); }
__device__ __forceinline__ uint32_t in_tile(int x, int y) { return (uint32_t)(((y & 7) << 3) | (x & 7)); }
// byte offset of cell (x, y) in the window of scan h; false when the cell lies outside it (a point farther from the
// begin cell than the radius the host was given: counted in `misses`, never written -- nothing outside a window exists)
__device__ __forceinline__ bool window_off(const ScanHdr& h, int x, int y, uint32_t& off) {
  const unsigned wx = (unsigned)((x >> 3) - h.tx0), wy = (unsigned)((y >> 3) - h.ty0);
  off = ((h.base + wy * (unsigned)h.tw + wx) << 6) + in_tile(x, y);  // 32-bit: the host keeps a round below 2^26 slots
  return wx < (unsigned)h.tw && wy < (unsigned)h.th;
}

__global__ void __launch_bounds__(256)
k_lo_batch_hits(BatchGeom g, const ScanHdr* __restrict__ hdr, const float* __restrict__ pts, uint8_t* __restrict__ pool,
                uint8_t* __restrict__ flags, uint32_t* __restrict__ hkey, uint32_t* __restrict__ hhit,
                unsigned long long* __restrict__ misses) {
  const int sidx = blockIdx.y;
  const ScanHdr h = hdr[sidx];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= h.n) return;
  const Line l = batch_line(g, h, pts, i);
  if (!l.valid) return;
  const uint32_t cell = (uint32_t)(l.y1 * g.sx + l.x1);
  const uint32_t t = tile_of(g, l.x1, l.y1);
  uint32_t off;
  if (!window_off(h, l.x1, l.y1, off)) {
    atomicAdd(misses, 1ull);
    return;
  }
  pool[off] = (uint8_t)(g.tag | kCodeHit);
  flags[(size_t)t * kBatchSlots + sidx] = (uint8_t)(g.tag | 1u);
  const size_t hb = (size_t)sidx * (g.hash_mask + 1);
  uint32_t slot = hash_slot0(cell, g.hash_mask);
  for (;;) {
    const uint32_t old = atomicCAS(&hkey[hb + slot], kHashEmpty, cell);
    if (old == kHashEmpty || old == cell) break;
    slot = (slot + 1) & g.hash_mask;
  }
  atomicMin(&hhit[hb + slot], (uint32_t)i);
}

// The same marking with the scan's hash built in LDS: one block per scan.  The compare-and-swap that claims a slot
// returns a value, and a returning device-scope atomic is executed at the memory side -- a microsecond-class round trip
// per probe; in LDS it is a few dozen cycles.  The finished table is written out with plain stores for k_lo_batch_rays.
__global__ void __launch_bounds__(1024)
k_lo_batch_hits_lds(BatchGeom g, const ScanHdr* __restrict__ hdr, const float* __restrict__ pts, uint8_t* __restrict__ pool,
                    uint8_t* __restrict__ flags, uint32_t* __restrict__ hkey, uint32_t* __restrict__ hhit,
                    unsigned long long* __restrict__ misses) {
  extern __shared__ uint32_t sh_hash[];
  const uint32_t slots = g.hash_mask + 1;
  uint32_t* key = sh_hash;
  uint32_t* first = sh_hash + slots;
  const int sidx = blockIdx.x, tid = threadIdx.x;
  const ScanHdr h = hdr[sidx];
  for (uint32_t i = tid; i < 2 * slots; i += 1024) sh_hash[i] = kHashEmpty;
  __syncthreads();
  for (int i = tid; i < h.n; i += 1024) {
    const Line l = batch_line(g, h, pts, i);
    if (!l.valid) continue;
    const uint32_t cell = (uint32_t)(l.y1 * g.sx + l.x1);
    const uint32_t t = tile_of(g, l.x1, l.y1);
    uint32_t off;
    if (!window_off(h, l.x1, l.y1, off)) {
      atomicAdd(misses, 1ull);
      continue;
    }
    pool[off] = (uint8_t)(g.tag | kCodeHit);
    flags[(size_t)t * kBatchSlots + sidx] = (uint8_t)(g.tag | 1u);
    uint32_t slot = hash_slot0(cell, g.hash_mask);
    for (;;) {
      const uint32_t old = atomicCAS(&key[slot], kHashEmpty, cell);
      if (old == kHashEmpty || old == cell) break;
      slot = (slot + 1) & g.hash_mask;
    }
    atomicMin(&first[slot], (uint32_t)i);
  }
  __syncthreads();
  const size_t hb = (size_t)sidx * slots;
  for (uint32_t i = tid; i < slots; i += 1024) {
    hkey[hb + i] = key[i];
    hhit[hb + i] = first[i];
  }
}

#if !defined(LSLAM_TUNE_RAY_BEAMS)
#define LSLAM_TUNE_RAY_BEAMS 4
#endif
LSLAM_STAMP_TABLE(g_map_stamps)  // kernel 0 = k_lo_batch_rays: 0 header, 1 beam line, 2 cell addresses, 3 plane bytes back, 4 marks issued
constexpr int kRayBeamsPerWave = LSLAM_TUNE_RAY_BEAMS;  // consecutive beams one wave walks: fewer, longer waves (dispatch-rate bound otherwise)
__global__ void __launch_bounds__(256)
k_lo_batch_rays(BatchGeom g, const ScanHdr* __restrict__ hdr, const float* __restrict__ pts, uint8_t* __restrict__ pool,
                uint8_t* __restrict__ flags, const uint32_t* __restrict__ hkey, uint32_t* __restrict__ hcross, int groups_per_scan) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // wave index = scan * groups_per_scan + beam group
  const int sidx = w / groups_per_scan, grp = w - sidx * groups_per_scan;
  if (sidx >= g.K) return;
  LSLAM_PHASE_CLOCK(pck);
  const ScanHdr h = hdr[sidx];
  const uint32_t crossed = g.tag | kCodeCrossed, hit = g.tag | kCodeHit;
  const size_t hb = (size_t)sidx * (g.hash_mask + 1);
  LSLAM_PHASE_MARK(pck, 0);
  // The wave's points in ONE load (lane j = beam j of the group), handed out by v_readlane: the per-beam point load was a
  // round trip of its own in front of every beam's cells -- 1 850 of a beam's ~5 400 cycles (tools/phase_stamps.py rays,
  // profiles/r06/phase_stamps_rays.json).  The line itself is evaluated per beam from the broadcast point, bit for bit as before.
  const int i_first = grp * kRayBeamsPerWave, i_end = min(h.n, (grp + 1) * kRayBeamsPerWave);
  float2 my_pt = make_float2(0.f, 0.f);
  if (lane < kRayBeamsPerWave && i_first + lane < i_end) my_pt = ((const float2*)pts)[(size_t)h.pts_off + i_first + lane];
  for (int i = i_first; i < i_end; i++) {
    const float bpt[2] = {__int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_pt.x), i - i_first)),
                          __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_pt.y), i - i_first))};
    LevelGeom lg;
    lg.sx = g.sx; lg.sy = g.sy; lg.c = h.c; lg.s = h.s; lg.tx = h.tx; lg.ty = h.ty; lg.factor = g.factor;
    lg.just_once = 0; lg.bx = h.bx; lg.by = h.by; lg.metres_per_cell = 0.0;
    const Line l = beam_line(lg, bpt, 0);  // (batch_line with the point already in registers)
    LSLAM_PHASE_MARK(pck, 1);
    if (!l.valid) continue;
    // the reference's traversal (H/map/OccGridMapBase.h:240-299) in closed form, here as (x, y): cell c of the ray is c
    // major steps and q(c) = floor((abs_da/2 + c*abs_db) / abs_da) minor steps from the begin cell (see ray_cell)
    const int dx = l.x1 - l.x0, dy = l.y1 - l.y0;
    const unsigned abs_dx = (unsigned)abs(dx), abs_dy = (unsigned)abs(dy);
    const int sgx = dx > 0 ? 1 : -1, sgy = dy > 0 ? 1 : -1;  // util::sign: sign(0) = -1
    const bool xmajor = abs_dx >= abs_dy;
    const unsigned abs_da = xmajor ? abs_dx : abs_dy, abs_db = xmajor ? abs_dy : abs_dx;
    // q(c): the numerator stays below 2^31 (map sides <= 32768) and the quotient below 2^16, so a float estimate is
    // within +-1 of it and two integer corrections make it exact (a 64-bit integer division per cell otherwise)
    const float rcp_da = 1.0f / (float)abs_da;
    for (unsigned c = lane; c < abs_da; c += 64) {
      LSLAM_PHASE_MARK(pck, 2);
      const unsigned num = abs_da / 2 + c * abs_db;
      unsigned q = (unsigned)((float)num * rcp_da);
      int rem = (int)(num - q * abs_da);
      if (rem < 0) { q -= 1; rem += (int)abs_da; }
      if (rem < 0) { q -= 1; rem += (int)abs_da; }
      if (rem >= (int)abs_da) { q += 1; rem -= (int)abs_da; }
      if (rem >= (int)abs_da) { q += 1; }
      const int x = l.x0 + (xmajor ? (int)c * sgx : (int)q * sgx);
      const int y = l.y0 + (xmajor ? (int)q * sgy : (int)c * sgy);
      const uint32_t t = tile_of(g, x, y);
      uint32_t off;
      if (!window_off(h, x, y, off)) {  // (the end cell of this beam is outside too: k_lo_batch_hits counted it)
        continue;
      }
      uint8_t* plane = pool;  // (the window's bytes: `off` is an offset into the pool)
      const uint32_t b = plane[off];
      LSLAM_PHASE_MARK(pck, 3);
      if (b == hit) {  // some beam of this scan ends here: remember the first beam that crosses it
        const uint32_t cell = (uint32_t)(y * g.sx + x);
        uint32_t slot = hash_slot0(cell, g.hash_mask);
        while (hkey[hb + slot] != cell) slot = (slot + 1) & g.hash_mask;  // entered by k_lo_batch_hits
        atomicMin(&hcross[hb + slot], (uint32_t)i);
      } else if (b != crossed) {
        plane[off] = (uint8_t)crossed;
        flags[(size_t)t * kBatchSlots + sidx] = (uint8_t)(g.tag | 1u);
      }
      LSLAM_PHASE_MARK(pck, 4);
    }
  }
  LSLAM_PHASE_FLUSH(pck, g_map_stamps, 0, (unsigned)w, lane == 0);
}

// hash entry -> plane byte: a hit cell that an EARLIER beam of the same scan crossed becomes HIT_UNDO, so the apply
// pass needs nothing but the plane bytes (per-cell hash look-ups there were serialised dependent loads on the
// wall cells: 160 us per batch)
__global__ void __launch_bounds__(256)
k_lo_batch_resolve(BatchGeom g, const ScanHdr* __restrict__ hdr, const uint32_t* __restrict__ hkey,
                   const uint32_t* __restrict__ hhit, const uint32_t* __restrict__ hcross, uint8_t* __restrict__ pool) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t slots = (size_t)g.hash_mask + 1;
  if (idx >= slots * g.K) return;
  const uint32_t cell = hkey[idx];
  if (cell == kHashEmpty) return;
  if (hcross[idx] < hhit[idx]) {
    const int x = (int)(cell % (uint32_t)g.sx), y = (int)(cell / (uint32_t)g.sx);
    uint32_t off;
    if (window_off(hdr[idx / slots], x, y, off)) pool[off] = (uint8_t)(g.tag | kCodeHitUndo);  // (entered only from inside)
  }
}

// one wave per tile: lane = plane slot when reading the flags, lane = cell of the tile when applying
__global__ void __launch_bounds__(256)
k_lo_batch_apply(BatchGeom g, const ScanHdr* __restrict__ hdr, const uint8_t* __restrict__ pool,
                 const uint8_t* __restrict__ flags, float* __restrict__ logodds) {
  const int lane = threadIdx.x & 63;
  const uint32_t t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (t >= (uint32_t)g.n_tiles) return;
  const uint32_t f = flags[(size_t)t * kBatchSlots + lane];
  unsigned long long touched = __ballot(lane < g.K && (f & 0xFCu) == g.tag && (f & 3u) != 0u);  // wave-uniform
  if (!touched) return;
  const int x = (int)(t % (uint32_t)g.tiles_x) * 8 + (lane & 7), y = (int)(t / (uint32_t)g.tiles_x) * 8 + (lane >> 3);
  const bool in_map = x < g.sx && y < g.sy;
  float v = in_map ? logodds[(size_t)y * g.sx + x] : 0.f;
  bool dirty = false;
  // lane s: where THIS tile sits in the window of scan s (one 16-byte load per lane; read back per touched scan by
  // v_readlane with a uniform index)
  const int ttx = (int)(t % (uint32_t)g.tiles_x), tty = (int)(t / (uint32_t)g.tiles_x);
  uint32_t my_off = 0u;
  if (lane < g.K) {
    const int4 w = *(const int4*)((const char*)(hdr + lane) + 32);  // tx0, ty0, tw, base
    my_off = (((uint32_t)w.w + (uint32_t)(tty - w.y) * (uint32_t)w.z + (uint32_t)(ttx - w.x)) << 6);
  }
  // ascending slot = scan order; kApplyChunk planes' bytes in flight at a time (the mask is wave-uniform)
  while (touched) {
    int sl[kApplyChunk];
#pragma unroll
    for (int u = 0; u < kApplyChunk; u++) {
      sl[u] = touched ? __ffsll((long long)touched) - 1 : -1;
      touched &= touched - 1;  // 0 stays 0
    }
    uint32_t bv[kApplyChunk];
#pragma unroll
    for (int u = 0; u < kApplyChunk; u++) {
      bv[u] = 0u;
      if (sl[u] >= 0) {  // wave-uniform; a flagged tile lies inside the window of the scan that flagged it
        const uint32_t tile_off = (uint32_t)__builtin_amdgcn_readlane((int)my_off, sl[u]);
        bv[u] = (uint32_t)pool[tile_off + (uint32_t)lane];
      }
    }
#pragma unroll
    for (int u = 0; u < kApplyChunk; u++) {
      const uint32_t b = bv[u];
      if ((b & 0xFCu) != g.tag) continue;  // also the empty slots (tags start at 1 << 2)
      const uint32_t code = b & 3u;
      if (code == kCodeCrossed) {
        v += g.lo_free;  // bresenhamCellFree, once per scan (:302-313)
        dirty = true;
      } else if (code != 0u) {  // bresenhamCellOcc (:316-330)
        if (code == kCodeHitUndo) {  // crossed by an EARLIER beam: marked free, then un-marked
          v += g.lo_free;
          v -= g.lo_free;
        }
        if (v < 50.0f) v += g.lo_occ;  // updateSetOccupied (H/map/GridMapLogOdds.h:108-114)
        dirty = true;
      }
    }
  }
  if (dirty && in_map) logodds[(size_t)y * g.sx + x] = v;
}

// ------------------------------------------------------------------------------------------
// LaserScan -> Hector DataContainer on the device: what HectorMappingRos::scanCallback does
// before the map sees a scan -- laser_geometry's projectLaser(scan, cloud, 30.0) (hector_slam.cc:193) and
// rosPointCloudToDataContainer (hector_slam.cc:320-362).  One block; order-preserving compaction (the container
// keeps the beams' order, which the once-per-scan cell rule depends on).
//   projection   p = (double)r * (cos, sin)(angle_min + i * angle_increment), cast to float32 -- the cos/sin table is
//                built on the HOST once per scan geometry and cached, exactly as laser_geometry caches its
//                co_sine_map_: the device multiplies, it never evaluates a trigonometric function here
//   filters      r < cutoff && r >= range_min;  min_dist^2 < d2 < max_dist^2;  !(x < 0 && d2 < 0.5);
//                d2 <= use_max^2;  z window of the laser frame (hector_slam.cc:336-354)
//   transform    base_link <- laser (planar: yaw + translation), tf's double arithmetic, then float32 * scaleToMap
// ------------------------------------------------------------------------------------------
struct ProjectCfg {
  int n;
  float range_min, cutoff, sqr_min, sqr_max;
  double use_max_sq;
  float z_min, z_max;
  double cy, sy, tx, ty, tz;  // base_link -> laser transform: rotation about z (host cos/sin) + origin
  float scale_to_map;
};

__global__ void __launch_bounds__(1024)
k_hector_project(ProjectCfg c, const float* __restrict__ ranges, const double2* __restrict__ cossin,
                 float* __restrict__ out_xy, int* __restrict__ out_n) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int i0 = 0; i0 < c.n; i0 += 1024) {
    const int i = i0 + tid;
    bool keep = false;
    float ox = 0.f, oy = 0.f;
    if (i < c.n) {
      const float r = ranges[i];
      if (r < c.cutoff && r >= c.range_min) {  // laser_geometry::projectLaser_ (NaN fails both)
        const double2 cs = cossin[i];
        const float x = (float)((double)r * cs.x), y = (float)((double)r * cs.y);  // sensor_msgs/PointCloud: float32
        const float d2 = x * x + y * y;                                            // hector_slam.cc:336
        if (d2 > c.sqr_min && d2 < c.sqr_max && !(x < 0.0f && d2 < 0.50f) && !((double)d2 > c.use_max_sq)) {
          // tf::Transform * tf::Vector3 in double (:349): row . v, then + origin
          const double bx = (c.cy * (double)x + (-c.sy) * (double)y + 0.0 * 0.0) + c.tx;
          const double by = (c.sy * (double)x + c.cy * (double)y + 0.0 * 0.0) + c.ty;
          const double bz = (0.0 * (double)x + 0.0 * (double)y + 1.0 * 0.0) + c.tz;
          const float zl = (float)(bz - c.tz);  // pointPosLaserFrameZ (:352)
          if (zl > c.z_min && zl < c.z_max) {
            keep = true;
            ox = (float)bx * c.scale_to_map;  // Eigen::Vector2f(x, y) * scaleToMap (:357)
            oy = (float)by * c.scale_to_map;
          }
        }
      }
    }
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_wave[wv] = __popcll(bal);
    __syncthreads();
    int before = s_base;
    for (int w = 0; w < wv; w++) before += s_wave[w];
    if (keep) {
      const int pos = before + __popcll(bal & ((1ull << lane) - 1ull));
      out_xy[2 * pos] = ox;
      out_xy[2 * pos + 1] = oy;
    }
    __syncthreads();
    if (tid == 0) {
      int tot = 0;
      for (int w = 0; w < 16; w++) tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (tid == 0) *out_n = s_base;
}

__global__ void k_occupancy_i8(const float* __restrict__ v, int8_t* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = v[i];
  out[i] = x < 0.0f ? (int8_t)0 : (x > 0.0f ? (int8_t)100 : (int8_t)-1);  // hector_slam.cc:287-304
}

// ------------------------------------------------------------------------------------------
// Hector Gauss-Newton scan-to-map matcher (next-row #2): MapRepMultiMap::matchData
// (H/slam_main/MapRepMultiMap.h:144-167) -> ScanMatcher::matchData / estimateTransformationLogLh
// (H/matcher/ScanMatcher.h:60-139) -> OccGridMapUtil::getCompleteHessianDerivs /
// interpMapValueWithDerivatives (H/map/OccGridMapUtil.h:77-228).  One block per scan walks all
// pyramid levels and iterations: every thread evaluates the bilinear map value + gradient of its
// points and their 9 Hessian/gradient terms in fp32 (parallel), ONE thread then adds the terms in
// point order -- the reference accumulates H and dTr sequentially in fp32, so the order is part of
// the result -- and solves the 3x3 system.  The probability cache of the reference
// (GridMapCacheArray) is a pure memoisation and has no device counterpart.
// ------------------------------------------------------------------------------------------
constexpr int kGnMaxLevels = 8;
struct GnLevels {
  int n_levels;
  int sx[kGnMaxLevels], sy[kGnMaxLevels];
  float scale[kGnMaxLevels], t_x[kGnMaxLevels], t_y[kGnMaxLevels];
  const float* logodds[kGnMaxLevels];
};

__device__ __forceinline__ float gn_prob(const float* lo, int index) {  // getGridProbability (GridMapLogOdds.h:123-127)
  float odds = (float)exp((double)lo[index]);
  return odds / (odds + 1.0f);
}

__device__ __forceinline__ float gn_cof3(const float* m, int i, int j) {
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}

__global__ void __launch_bounds__(1024)
k_gn_match(GnLevels lv, const float* __restrict__ pts, int n, float bx, float by, float bth,
           float* __restrict__ out /* pose[3] + H[9] */) {
  extern __shared__ float terms[];  // [n][9]
  __shared__ float s_est[3], s_sum[9];
  const int tid = threadIdx.x, nt = blockDim.x;
  float tmp0 = bx, tmp1 = by, tmp2 = bth;
  float Hlast[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int L = lv.n_levels - 1; L >= 0; --L) {
    if (n == 0) continue;
    const float* lo = lv.logodds[L];
    const int sx = lv.sx[L], sy = lv.sy[L];
    const float sc = lv.scale[L];
    const float factor = L == 0 ? 1.0f : (float)(1.0 / pow(2.0, (double)L));
    const int iters = 1 + (L == 0 ? 5 : 3);
    if (tid == 0) {  // getMapCoordsPose (GridMapBase.h:238-242)
      s_est[0] = (sc * tmp0 + 0.0f * tmp1) + lv.t_x[L];
      s_est[1] = (0.0f * tmp0 + sc * tmp1) + lv.t_y[L];
      s_est[2] = tmp2;
    }
    __syncthreads();
    for (int it = 0; it < iters; it++) {
      const float e0 = s_est[0], e1 = s_est[1], e2 = s_est[2];
      const float c = (float)cos((double)e2), s = (float)sin((double)e2);  // Rotation2Df + sinRot/cosRot (:85-88)
      const float lim_x = (float)sx - 2.0f, lim_y = (float)sy - 2.0f;     // MapDimensionProperties.h:66-70
      for (int i = tid; i < n; i += nt) {
        const float px = pts[2 * i] * factor, py = pts[2 * i + 1] * factor;
        const float cx = (c * px + (-s) * py) + e0;
        const float cy = (s * px + c * py) + e1;
        float v = 0.0f, gxv = 0.0f, gyv = 0.0f;
        if (!(cx < 0.0f || cx > lim_x || cy < 0.0f || cy > lim_y)) {  // pointOutOfMapBounds (:60-63)
          const int ix = (int)cx, iy = (int)cy;
          const float fx = cx - (float)ix, fy = cy - (float)iy;
          const int index = iy * sx + ix;
          const float i0 = gn_prob(lo, index), i1 = gn_prob(lo, index + 1);
          const float i2 = gn_prob(lo, index + sx), i3 = gn_prob(lo, index + sx + 1);
          const float dx1 = i0 - i1, dx2 = i2 - i3, dy1 = i0 - i2, dy2 = i1 - i3;
          const float xi = 1.0f - fx, yi = 1.0f - fy;
          v = ((i0 * xi + i1 * fx) * (yi)) + ((i2 * xi + i3 * fx) * (fy));
          gxv = -((dx1 * yi) + (dx2 * fy));
          gyv = -((dy1 * xi) + (dy2 * fx));
        }
        const float funVal = 1.0f - v;
        const float rotDeriv = ((-s * px - c * py) * gxv + (c * px - s * py) * gyv);
        float* t = terms + 9 * i;
        t[0] = gxv * funVal; t[1] = gyv * funVal; t[2] = rotDeriv * funVal;
        t[3] = gxv * gxv; t[4] = gyv * gyv; t[5] = rotDeriv * rotDeriv;
        t[6] = gxv * gyv; t[7] = gxv * rotDeriv; t[8] = gyv * rotDeriv;
      }
      __syncthreads();
      // sequential fp32 accumulation in point order (:94-126): the nine sums are independent chains, so
      // nine lanes walk the points, one chain each -- same order, same roundings, a ninth of the latency
      if (tid < 9) {
        float acc = 0.0f;
#pragma unroll 8
        for (int i = 0; i < n; i++) acc += terms[9 * i + tid];
        s_sum[tid] = acc;
      }
      __syncthreads();
      if (tid == 0) {
        const float d0 = s_sum[0], d1 = s_sum[1], d2 = s_sum[2], h00 = s_sum[3], h11 = s_sum[4], h22 = s_sum[5],
                    h01 = s_sum[6], h02 = s_sum[7], h12 = s_sum[8];
        float H[9] = {h00, h01, h02, h01, h11, h12, h02, h12, h22};
        for (int q = 0; q < 9; q++) Hlast[q] = H[q];
        if (h00 != 0.0f && h11 != 0.0f) {  // estimateTransformationLogLh (ScanMatcher.h:113-133)
          const float c0 = gn_cof3(H, 0, 0), c1 = gn_cof3(H, 1, 0), c2 = gn_cof3(H, 2, 0);
          const float det = c0 * H[0] + (c1 * H[3] + c2 * H[6]);
          const float invdet = 1.0f / det;
          const float Hi[9] = {c0 * invdet, c1 * invdet, c2 * invdet,
                               gn_cof3(H, 0, 1) * invdet, gn_cof3(H, 1, 1) * invdet, gn_cof3(H, 2, 1) * invdet,
                               gn_cof3(H, 0, 2) * invdet, gn_cof3(H, 1, 2) * invdet, gn_cof3(H, 2, 2) * invdet};
          float sd[3];
          // H.inverse() * dTr: Eigen's coefficient-based product sums a 3-term row as a0 + (a1 + a2) (Core/Redux.h)
          for (int r = 0; r < 3; r++) sd[r] = Hi[3 * r] * d0 + (Hi[3 * r + 1] * d1 + Hi[3 * r + 2] * d2);
          if (sd[2] > 0.2f) sd[2] = 0.2f; else if (sd[2] < -0.2f) sd[2] = -0.2f;
          s_est[0] += sd[0]; s_est[1] += sd[1]; s_est[2] += sd[2];
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      // util::normalize_angle (UtilFunctions.h:36-48), double arithmetic with M_PI
      const double two_pi = 2.0f * 3.14159265358979323846;
      float a = (float)fmod(fmod((double)s_est[2], two_pi) + two_pi, two_pi);
      if ((double)a > 3.14159265358979323846) a = (float)((double)a - two_pi);
      // getWorldCoordsPose: worldTmap = mapTworld.inverse() (GridMapBase.h:229-233, 285)
      const float invdet = 1.0f / (sc * sc - 0.0f * 0.0f);
      const float l00 = sc * invdet, l01 = -0.0f * invdet, l10 = -0.0f * invdet, l11 = sc * invdet;
      const float wt0 = -(l00 * lv.t_x[L] + l01 * lv.t_y[L]), wt1 = -(l10 * lv.t_x[L] + l11 * lv.t_y[L]);
      tmp0 = (l00 * s_est[0] + l01 * s_est[1]) + wt0;
      tmp1 = (l10 * s_est[0] + l11 * s_est[1]) + wt1;
      tmp2 = a;
    }
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = tmp0; out[1] = tmp1; out[2] = tmp2;
    for (int q = 0; q < 9; q++) out[3 + q] = Hlast[q];
  }
}

// ------------------------------------------------------------------------------------------
// k_gn_match_fast -- the same Gauss-Newton matcher, summed in PARALLEL (round 4; the default).
// k_gn_match above adds the nine H / dTr sums in point order on nine lanes so that they round like the reference's
// sequential fp32 loop (it equals the CPU restatement bit for bit): 1081 dependent LDS adds per iteration, fourteen
// iterations, on one CU.  The north star grants 1e-4 m / 1e-4 rad for floating point, device libm already differs from
// glibc in the last bit, and a tree sum is -- if anything -- the more accurate one.  Here every thread keeps nine
// partial sums of ITS points in registers, a wave adds them with DPP row operations (no LDS round trip), the waves'
// partials meet in LDS and EVERY thread adds them up and solves the 3x3 system itself (same inputs, same operations:
// bitwise the same estimate in every thread), so an iteration has ONE barrier and no single-thread phase.  exp / sin /
// cos are the float32 ocml functions (<= 1 ulp; the ordered kernel evaluates them in double like the reference's host
// code).  Points are staged once in LDS; the kernel also writes the cached container (MapRepMultiMap.h:161) when the
// points come from pinned host memory, and its result lands in pinned host memory: no copy operation on the stream.
// LSLAM_MAP_OPT_ORDERED_SUMS selects the ordered kernel (the bit comparison with the restatement).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float gn_row16_sum(float v) {  // every lane of a 16-lane row ends up with the row's sum
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}
__device__ __forceinline__ float gn_wave_sum(float v) {
  v = gn_row16_sum(v);
  const int b = __float_as_int(v);
  return (__int_as_float(__builtin_amdgcn_readlane(b, 0)) + __int_as_float(__builtin_amdgcn_readlane(b, 16))) +
         (__int_as_float(__builtin_amdgcn_readlane(b, 32)) + __int_as_float(__builtin_amdgcn_readlane(b, 48)));
}
// getGridProbability (GridMapLogOdds.h:123-127) on the hardware's exp2 / reciprocal units (v_exp_f32, v_rcp_f32: ~1e-7
// relative, far inside the 1e-4 the matcher's result is held to).  Log-odds stay below ~52 (the occupied clamp,
// GridMapLogOdds.h:110), so odds + 1 neither overflows nor leaves __fdividef's range; -inf gives 0.
__device__ __forceinline__ float gn_prob_f(float lo) {
  const float odds = __expf(lo);
  return odds * __builtin_amdgcn_rcpf(odds + 1.0f);  // (__fdividef compiles to the full IEEE division sequence here)
}
// Nine wave sums at once: the four row steps of all nine values interleaved (no DPP read-after-write stalls), then the two
// cross-row broadcasts of GFX9's wave64 DPP; the totals are valid in LANE 63 only.
__device__ __forceinline__ void gn_wave_sums9(float* v) {
#define LSLAM_DPP_STEP(CTRL)                                                                                            \
  _Pragma("unroll") for (int q = 0; q < 9; q++)                                                                         \
      v[q] += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v[q]), CTRL, 0xF, 0xF, true));
  LSLAM_DPP_STEP(0xB1)   // quad_perm [1,0,3,2]
  LSLAM_DPP_STEP(0x4E)   // quad_perm [2,3,0,1]
  LSLAM_DPP_STEP(0x141)  // row_half_mirror
  LSLAM_DPP_STEP(0x140)  // row_mirror: every lane of a 16-lane row holds the row's sum
#undef LSLAM_DPP_STEP
#pragma unroll
  for (int q = 0; q < 9; q++)  // row_bcast:15 into rows 1 and 3: row 1 = rows 0+1, row 3 = rows 2+3
    v[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[q]), 0x142, 0xA, 0xF, false));
#pragma unroll
  for (int q = 0; q < 9; q++)  // row_bcast:31 into rows 2 and 3: row 3 = all four rows
    v[q] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[q]), 0x143, 0xC, 0xF, false));
}
// sin / cos of the pose heading: |x| stays within a few radians (a normalised start estimate plus steps clamped to 0.2),
// so the quadrant reduction is two fused steps of x - k * pi/2 with pi/2 split in two floats (exact to ~1e-8 for |k| <= 8)
// and the rest the classic degree-7 / degree-8 kernels on [-pi/4, pi/4] (~1 ulp).  ocml's sincosf carries the
// large-argument path along: four times the instructions, every thread, every iteration.
__device__ __forceinline__ void gn_sincos(float x, float* sn, float* cs) {
  const float k = rintf(x * 0.636619772367581343f);
  const int q = (int)k;
  float r = fmaf(-k, 1.57079601287841796875f, x);   // pi/2 hi (24 bits)
  r = fmaf(-k, 3.1391647326017846e-7f, r);            // pi/2 lo
  const float r2 = r * r;
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, r2, -1.6666654611e-1f);
  ps = fmaf(ps * r2, r, r);                           // sin r
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, r2, 4.166664568298827e-2f);
  pc = fmaf(pc, r2, -0.5f);
  pc = fmaf(pc, r2, 1.0f);                            // cos r
  const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}

__device__ __forceinline__ void gn_solve_step_fwd(const float* sum, float* H, float& e0, float& e1, float& e2);
template <int NT>
__global__ void __launch_bounds__(NT)
k_gn_match_fast(GnLevels lv, const float* __restrict__ pts, float* __restrict__ cache_dst, int n, int pts_in_lds, float bx,
                float by, float bth, float* __restrict__ out /* pose[3] + H[9], [15] = ticket */, int ticket) {
  extern __shared__ float s_pts[];  // [2n] when pts_in_lds
  constexpr int NW = NT / 64;
  __shared__ float s_part[2][NW][12];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 2 * n; i += NT) {
    const float v = pts[i];
    if (pts_in_lds) s_pts[i] = v;
    if (cache_dst) cache_dst[i] = v;
  }
  const float* P = pts_in_lds ? s_pts : (cache_dst ? cache_dst : pts);
  __syncthreads();  // (a thread reads points other threads staged; cache_dst is only re-read by its own writers' block)
  float tmp0 = bx, tmp1 = by, tmp2 = bth;
  float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int flip = 0;
  for (int L = lv.n_levels - 1; L >= 0; --L) {
    if (n == 0) continue;
    const float* lo = lv.logodds[L];
    const int sx = lv.sx[L], sy = lv.sy[L];
    const float sc = lv.scale[L];
    const float factor = L == 0 ? 1.0f : 1.0f / (float)(1 << L);
    const int iters = 1 + (L == 0 ? 5 : 3);
    // getMapCoordsPose (GridMapBase.h:238-242)
    float e0 = (sc * tmp0 + 0.0f * tmp1) + lv.t_x[L];
    float e1 = (0.0f * tmp0 + sc * tmp1) + lv.t_y[L];
    float e2 = tmp2;
    const float lim_x = (float)sx - 2.0f, lim_y = (float)sy - 2.0f;  // MapDimensionProperties.h:66-70
    for (int it = 0; it < iters; it++) {
      float s, c;
      sincosf(e2, &s, &c);
      float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
      for (int i = tid; i < n; i += NT) {
        const float px = P[2 * i] * factor, py = P[2 * i + 1] * factor;
        const float cx = (c * px + (-s) * py) + e0;
        const float cy = (s * px + c * py) + e1;
        float v = 0.0f, gxv = 0.0f, gyv = 0.0f;
        if (!(cx < 0.0f || cx > lim_x || cy < 0.0f || cy > lim_y)) {  // pointOutOfMapBounds (:60-63)
          const int ix = (int)cx, iy = (int)cy;
          const float fx = cx - (float)ix, fy = cy - (float)iy;
          const int index = iy * sx + ix;
          const float l0 = lo[index], l1 = lo[index + 1], l2 = lo[index + sx], l3 = lo[index + sx + 1];
          const float i0 = gn_prob_f(l0), i1 = gn_prob_f(l1), i2 = gn_prob_f(l2), i3 = gn_prob_f(l3);
          const float dx1 = i0 - i1, dx2 = i2 - i3, dy1 = i0 - i2, dy2 = i1 - i3;
          const float xi = 1.0f - fx, yi = 1.0f - fy;
          v = ((i0 * xi + i1 * fx) * (yi)) + ((i2 * xi + i3 * fx) * (fy));
          gxv = -((dx1 * yi) + (dx2 * fy));
          gyv = -((dy1 * xi) + (dy2 * fx));
        }
        const float funVal = 1.0f - v;
        const float rotDeriv = ((-s * px - c * py) * gxv + (c * px - s * py) * gyv);
        acc[0] += gxv * funVal; acc[1] += gyv * funVal; acc[2] += rotDeriv * funVal;
        acc[3] += gxv * gxv; acc[4] += gyv * gyv; acc[5] += rotDeriv * rotDeriv;
        acc[6] += gxv * gyv; acc[7] += gxv * rotDeriv; acc[8] += gyv * rotDeriv;
      }
#pragma unroll
      for (int q = 0; q < 9; q++) {
        const float t = gn_wave_sum(acc[q]);
        if (lane == 0) s_part[flip][wv][q] = t;
      }
      __syncthreads();
      // lane q < 9 of every wave adds the NW partials of sum q (NW LDS reads instead of 9 NW per lane), readlane hands
      // the nine totals to all lanes
      float mine = 0.0f;
      if (lane < 9) {
        mine = s_part[flip][0][lane];
#pragma unroll
        for (int w = 1; w < NW; w++) mine += s_part[flip][w][lane];
      }
      float sum[9];
#pragma unroll
      for (int q = 0; q < 9; q++) sum[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), q));
      flip ^= 1;  // the next iteration writes the other buffer: no second barrier needed
      gn_solve_step_fwd(sum, H, e0, e1, e2);
    }
    {
      // util::normalize_angle (UtilFunctions.h:36-48), double arithmetic with M_PI
      const double two_pi = 2.0f * 3.14159265358979323846;
      float a = (float)fmod(fmod((double)e2, two_pi) + two_pi, two_pi);
      if ((double)a > 3.14159265358979323846) a = (float)((double)a - two_pi);
      // getWorldCoordsPose: worldTmap = mapTworld.inverse() (GridMapBase.h:229-233, 285)
      const float invdet = 1.0f / (sc * sc - 0.0f * 0.0f);
      const float l00 = sc * invdet, l01 = -0.0f * invdet, l10 = -0.0f * invdet, l11 = sc * invdet;
      const float wt0 = -(l00 * lv.t_x[L] + l01 * lv.t_y[L]), wt1 = -(l10 * lv.t_x[L] + l11 * lv.t_y[L]);
      tmp0 = (l00 * e0 + l01 * e1) + wt0;
      tmp1 = (l10 * e0 + l11 * e1) + wt1;
      tmp2 = a;
    }
  }
  if (tid == 0) {
    out[0] = tmp0; out[1] = tmp1; out[2] = tmp2;
    for (int q = 0; q < 9; q++) out[3 + q] = H[q];
    // `out` is pinned host memory: a system-scope fence, then the caller's ticket -- the host spins on it instead of waiting
    // for the stream (the completion signal + wake-up cost several microseconds of a 34 us match)
    __threadfence_system();
    ((volatile int*)out)[15] = ticket;
  }
}

// The solve + pose update every thread performs after the sums (estimateTransformationLogLh, ScanMatcher.h:113-133)
__device__ __forceinline__ void gn_solve_step(const float* sum, float* H, float& e0, float& e1, float& e2);
__device__ __forceinline__ void gn_solve_step_fwd(const float* sum, float* H, float& e0, float& e1, float& e2) { gn_solve_step(sum, H, e0, e1, e2); }
__device__ __forceinline__ void gn_solve_step(const float* sum, float* H, float& e0, float& e1, float& e2) {
  const float d0 = sum[0], d1 = sum[1], d2 = sum[2], h00 = sum[3], h11 = sum[4], h22 = sum[5], h01 = sum[6], h02 = sum[7],
              h12 = sum[8];
  H[0] = h00; H[1] = h01; H[2] = h02; H[3] = h01; H[4] = h11; H[5] = h12; H[6] = h02; H[7] = h12; H[8] = h22;
  if (h00 != 0.0f && h11 != 0.0f) {
    const float c0 = gn_cof3(H, 0, 0), c1 = gn_cof3(H, 1, 0), c2 = gn_cof3(H, 2, 0);
    const float det = c0 * H[0] + (c1 * H[3] + c2 * H[6]);
    const float invdet = 1.0f / det;
    const float Hi[9] = {c0 * invdet, c1 * invdet, c2 * invdet,
                         gn_cof3(H, 0, 1) * invdet, gn_cof3(H, 1, 1) * invdet, gn_cof3(H, 2, 1) * invdet,
                         gn_cof3(H, 0, 2) * invdet, gn_cof3(H, 1, 2) * invdet, gn_cof3(H, 2, 2) * invdet};
    float sd[3];
    for (int r = 0; r < 3; r++) sd[r] = Hi[3 * r] * d0 + (Hi[3 * r + 1] * d1 + Hi[3 * r + 2] * d2);
    if (sd[2] > 0.2f) sd[2] = 0.2f; else if (sd[2] < -0.2f) sd[2] = -0.2f;
    e0 += sd[0]; e1 += sd[1]; e2 += sd[2];
  }
}

// k_gn_match_reg<NT, PMAX> -- k_gn_match_fast for scans of at most NT * PMAX points (every real LaserScan: 1081 beams =
// 512 x 3), the form that runs.  The generic kernel above walks its points one after the other, and each costs a
// dependent LDS read, then a dependent round trip to L2 for the four map cells: ~3.4 us per Gauss-Newton iteration of
// mostly waiting.  Here a thread's <= PMAX points live in REGISTERS for the whole match (no LDS, no barrier in front of
// the first iteration), and an iteration is straight-line code: all cell addresses first -- an out-of-map point reads
// cell 0 and is masked afterwards, so no load sits behind a branch --, then the 4 * PMAX loads back to back, then the
// arithmetic.  One L2 round trip per iteration instead of PMAX.
template <int NT, int PMAX>
__global__ void __launch_bounds__(NT)
k_gn_match_reg(GnLevels lv, const float* __restrict__ pts, float* __restrict__ cache_dst, int n, float bx, float by, float bth,
               float* __restrict__ out /* pose[3] + H[9], [15] = ticket */, int ticket) {
  constexpr int NW = NT / 64;
  __shared__ float s_part[2][NW][12];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float px0[PMAX], py0[PMAX];
  bool have[PMAX];
#pragma unroll
  for (int p = 0; p < PMAX; p++) {
    const int i = tid + p * NT;
    have[p] = i < n;
    float2 v = make_float2(0.0f, 0.0f);
    if (have[p]) {
      v = reinterpret_cast<const float2*>(pts)[i];
      if (cache_dst) reinterpret_cast<float2*>(cache_dst)[i] = v;
    }
    px0[p] = v.x;
    py0[p] = v.y;
  }
  float tmp0 = bx, tmp1 = by, tmp2 = bth;
  float H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  int flip = 0;
  for (int L = lv.n_levels - 1; L >= 0; --L) {
    if (n == 0) continue;
    const float* __restrict__ lo = lv.logodds[L];
    const int sx = lv.sx[L], sy = lv.sy[L];
    const float sc = lv.scale[L];
    const float factor = L == 0 ? 1.0f : 1.0f / (float)(1 << L);
    const int iters = 1 + (L == 0 ? 5 : 3);
    float e0 = (sc * tmp0 + 0.0f * tmp1) + lv.t_x[L];  // getMapCoordsPose (GridMapBase.h:238-242)
    float e1 = (0.0f * tmp0 + sc * tmp1) + lv.t_y[L];
    float e2 = tmp2;
    const float lim_x = (float)sx - 2.0f, lim_y = (float)sy - 2.0f;  // MapDimensionProperties.h:66-70
    float px[PMAX], py[PMAX];
#pragma unroll
    for (int p = 0; p < PMAX; p++) {
      px[p] = px0[p] * factor;
      py[p] = py0[p] * factor;
    }
    for (int it = 0; it < iters; it++) {
      float s, c;
      gn_sincos(e2, &s, &c);
      int idx[PMAX];
      float fx[PMAX], fy[PMAX];
      bool inb[PMAX];
#pragma unroll
      for (int p = 0; p < PMAX; p++) {
        const float cx = (c * px[p] + (-s) * py[p]) + e0;
        const float cy = (s * px[p] + c * py[p]) + e1;
        inb[p] = have[p] && !(cx < 0.0f || cx > lim_x || cy < 0.0f || cy > lim_y);  // pointOutOfMapBounds (:60-63)
        const int ix = inb[p] ? (int)cx : 0, iy = inb[p] ? (int)cy : 0;
        fx[p] = cx - (float)ix;
        fy[p] = cy - (float)iy;
        idx[p] = iy * sx + ix;
      }
      float l0[PMAX], l1[PMAX], l2[PMAX], l3[PMAX];
#pragma unroll
      for (int p = 0; p < PMAX; p++) {
        l0[p] = lo[idx[p]];
        l1[p] = lo[idx[p] + 1];
        l2[p] = lo[idx[p] + sx];
        l3[p] = lo[idx[p] + sx + 1];
      }
      float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int p = 0; p < PMAX; p++) {
        const float i0 = gn_prob_f(l0[p]), i1 = gn_prob_f(l1[p]), i2 = gn_prob_f(l2[p]), i3 = gn_prob_f(l3[p]);
        const float dx1 = i0 - i1, dx2 = i2 - i3, dy1 = i0 - i2, dy2 = i1 - i3;
        const float xi = 1.0f - fx[p], yi = 1.0f - fy[p];
        const float v = inb[p] ? ((i0 * xi + i1 * fx[p]) * (yi)) + ((i2 * xi + i3 * fx[p]) * (fy[p])) : 0.0f;
        const float gxv = inb[p] ? -((dx1 * yi) + (dx2 * fy[p])) : 0.0f;
        const float gyv = inb[p] ? -((dy1 * xi) + (dy2 * fx[p])) : 0.0f;
        const float funVal = have[p] ? 1.0f - v : 0.0f;
        const float rotDeriv = ((-s * px[p] - c * py[p]) * gxv + (c * px[p] - s * py[p]) * gyv);
        acc[0] += gxv * funVal; acc[1] += gyv * funVal; acc[2] += rotDeriv * funVal;
        acc[3] += gxv * gxv; acc[4] += gyv * gyv; acc[5] += rotDeriv * rotDeriv;
        acc[6] += gxv * gyv; acc[7] += gxv * rotDeriv; acc[8] += gyv * rotDeriv;
      }
      gn_wave_sums9(acc);
      if (lane == 63) {
#pragma unroll
        for (int q = 0; q < 9; q++) s_part[flip][wv][q] = acc[q];
      }
      __syncthreads();
      float mine = 0.0f;
      if (lane < 9) {
        mine = s_part[flip][0][lane];
#pragma unroll
        for (int w = 1; w < NW; w++) mine += s_part[flip][w][lane];
      }
      float sum[9];
#pragma unroll
      for (int q = 0; q < 9; q++) sum[q] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mine), q));
      flip ^= 1;  // the next iteration writes the other buffer: one barrier per iteration
      gn_solve_step(sum, H, e0, e1, e2);
    }
    {
      // util::normalize_angle (UtilFunctions.h:36-48), double arithmetic with M_PI
      const double two_pi = 2.0f * 3.14159265358979323846;
      float a = (float)fmod(fmod((double)e2, two_pi) + two_pi, two_pi);
      if ((double)a > 3.14159265358979323846) a = (float)((double)a - two_pi);
      // getWorldCoordsPose: worldTmap = mapTworld.inverse() (GridMapBase.h:229-233, 285)
      const float invdet = 1.0f / (sc * sc - 0.0f * 0.0f);
      const float l00 = sc * invdet, l01 = -0.0f * invdet, l10 = -0.0f * invdet, l11 = sc * invdet;
      const float wt0 = -(l00 * lv.t_x[L] + l01 * lv.t_y[L]), wt1 = -(l10 * lv.t_x[L] + l11 * lv.t_y[L]);
      tmp0 = (l00 * e0 + l01 * e1) + wt0;
      tmp1 = (l10 * e0 + l11 * e1) + wt1;
      tmp2 = a;
    }
  }
  if (tid == 0) {
    out[0] = tmp0; out[1] = tmp1; out[2] = tmp2;
    for (int q = 0; q < 9; q++) out[3 + q] = H[q];
    // `out` is pinned host memory: a system-scope fence, then the caller's ticket -- the host spins on it instead of waiting
    // for the stream (the completion signal + wake-up cost several microseconds of a 34 us match)
    __threadfence_system();
    ((volatile int*)out)[15] = ticket;
  }
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
  // pipelined single-scan path: second key-plane set (scans alternate), the points of the scan whose apply is pending
  uint32_t* d_free2 = nullptr;
  uint32_t* d_occ2 = nullptr;
  DevBuf<float> pipe_pts[2];
  bool pending = false;  // marks of scan `pend_g.epoch` are in plane set (epoch & 1); its apply has not been launched
  LevelGeom pend_g;
  int pend_n = 0;
  // batched update (k_lo_batch_*): a pool of 64-byte tile slots holding one window of tiles per scan of a round,
  // allocated on first use and grown on demand (never beyond the scratch budget unless ONE scan's window needs more)
  DevBuf<uint8_t> d_pool;       // [pool_slots][64]
  size_t pool_slots = 0;        // tile slots the epoch bookkeeping below covers (= d_pool.cap / 64 at the last clear)
  uint8_t* d_flags = nullptr;   // [n_tiles][64]: tile touched by the scan in slot s of the round
  int tiles_x = 0, n_tiles = 0;
  uint32_t batch_epoch = 0;     // 6-bit round epoch in the pool / flag bytes
};

}  // namespace

struct lslam_map {
  lslam_context* ctx = nullptr;
  std::vector<Level> levels;
  float off_x = 0.f, off_y = 0.f;
  float lo_free = 0.f, lo_occ = 0.f;
  DevBuf<float> d_pts;
  // MapRepMultiMap::dataContainers (H/slam_main/MapRepMultiMap.h:161,186,220): the points + origo of the LAST
  // matchData call; updateByScan feeds the levels above 0 from these, whatever container it is handed
  DevBuf<float> d_cached;
  int n_cached = 0;
  float cached_origo[2] = {0.f, 0.f};
  DevBuf<float> d_gn_out;
  // matchData, parallel-sum kernel (the default): the container goes up through pinned memory and is read -- and cached in
  // d_cached -- by the kernel itself; its 12 result floats land in pinned memory.  ordered_sums: k_gn_match instead.
  // batched update: scratch budget of the tile-slot pools of ONE level (bytes), rounds used by the last call, cells
  // found outside their scan's window (device counter; must stay 0: see lslam_map_batch_stats)
  size_t batch_budget = (size_t)192 << 20;
  int batch_radius_hint = 0;   // LSLAM_MAP_OPT_BATCH_RADIUS_CELLS: bound for callers whose points the host never sees
  int batch_last_rounds = 0;
  unsigned long long* d_batch_misses = nullptr;  // PINNED HOST memory (device-visible): the kernels' atomicAdd lands where the
                                                 // host can read it after any synchronise, without a copy on the stream
  unsigned long long batch_misses_reported = 0;  // of those, how many a synchronise has already reported as an error
  bool ordered_sums = false;  // lslam_map_set_option(LSLAM_MAP_OPT_ORDERED_SUMS) / LSLAM_GN_ORDERED=1
  int gn_threads = 512;       // LSLAM_GN_THREADS = 256 | 512 | 1024
  float* h_gn_pts = nullptr;
  size_t h_gn_cap = 0;        // floats
  float* h_gn_out = nullptr;  // 12 floats + the ticket word [15] the kernel posts behind them
  int gn_ticket = 0;
  // h_gn_pts[0 .. 2 * gn_host_n) is a host copy of what d_cached holds (the last matchData's container, fed from the
  // host): updateByScan with the SAME points -- HectorSlamProcessor::update always updates with the container it has just
  // matched (HectorSlamProcessor.h:91-105) -- finds them already resident and skips its own staging copy
  int gn_host_n = -1;
  // resident container of lslam_map_set_scan (device-side LaserScan -> DataContainer)
  DevBuf<float> d_scan;         // projected points, then one int: their count
  DevBuf<float> d_scan_ranges;
  DevBuf<double2> d_cossin;     // laser_geometry's co_sine_map_, host-built, cached per scan geometry
  float cs_angle_min = 0.f, cs_angle_inc = 0.f;
  int cs_n = 0;
  int n_scan = 0;
  float scan_origo[2] = {0.f, 0.f};
  DevBuf<uint32_t> d_hash;      // [3][K][slots]: key, first hit beam, first crossing beam
  DevBuf<ScanHdr> d_hdr;
  DevBuf<int8_t> d_i8;
  bool force_two_kernels = false;  // LSLAM_MAP_TWO_KERNELS=1: mark + apply per call, nothing deferred (A/B measurements, tests)
  // host -> device staging of the per-scan points: a ring of pinned slots, so updateByScan only enqueues
  // (copy + two kernels per level) and returns; a slot is reused when its copy has completed
  static constexpr int kStageSlots = 8;
  float* h_stage = nullptr;
  size_t stage_cap = 0;  // floats per slot
  hipEvent_t stage_ev[kStageSlots] = {};
  bool stage_busy[kStageSlots] = {};
  unsigned stage_next = 0;
};

namespace {

int clear_marks(lslam_map* map, Level& L) {
  lslam_context* ctx = map->ctx;
  size_t n = (size_t)L.sx * L.sy;
  LSLAM_HIP(ctx, hipMemsetAsync(L.d_free, 0, n * sizeof(uint32_t), ctx->stream));
  LSLAM_HIP(ctx, hipMemsetAsync(L.d_occ, 0, n * sizeof(uint32_t), ctx->stream));
  if (L.d_free2) {
    LSLAM_HIP(ctx, hipMemsetAsync(L.d_free2, 0, n * sizeof(uint32_t), ctx->stream));
    LSLAM_HIP(ctx, hipMemsetAsync(L.d_occ2, 0, n * sizeof(uint32_t), ctx->stream));
  }
  L.epoch = 0;
  return LSLAM_OK;
}

// blocks of one job of k_logodds_pipe_ml: four beams per block, a multiple of 8 (the kernel's XCD-sector mapping)
inline int pipe_blocks(int n) { return ((n + 3) / 4 + 7) / 8 * 8; }

// apply job of level L's pending scan
PipeJob apply_job(Level& L, int first_block) {
  const int set = (int)(L.pend_g.epoch & 1u);
  PipeJob j{};
  j.g = L.pend_g;
  j.pts = L.pipe_pts[set].p;
  j.n = L.pend_n;
  j.first_block = first_block;
  j.apply = 1;
  j.free_r = set ? L.d_free2 : L.d_free;
  j.occ_r = set ? L.d_occ2 : L.d_occ;
  j.logodds = L.d_logodds;
  return j;
}

// the deferred apply of the pipelined path, as a launch of its own: every reader of the float planes calls this first.
// ONE launch for all levels with a pending scan (k_logodds_pipe_ml); a pyramid deeper than the job table: per level.
int flush_pending(lslam_map* map) {
  lslam_context* ctx = map->ctx;
  PipeJobs J{};
  int blocks = 0;
  for (auto& L : map->levels) {
    if (!L.pending) continue;
    if (J.n_jobs == kPipeMaxJobs) {
      launch(ctx, "logodds_apply", k_logodds_pipe_ml, dim3(blocks), dim3(256), 0, J);
      J = PipeJobs{};
      blocks = 0;
    }
    J.j[J.n_jobs++] = apply_job(L, blocks);
    blocks += pipe_blocks(L.pend_n);
    L.pending = false;
  }
  if (blocks > 0) launch(ctx, "logodds_apply", k_logodds_pipe_ml, dim3(blocks), dim3(256), 0, J);
  LSLAM_HIP(ctx, hipGetLastError());
  return LSLAM_OK;
}

int update_impl(lslam_map* map, const float* d_pts, int n, const float* origo, const float pose[3],
                int just_once, float begin_x, float begin_y, double metres_per_cell) {
  lslam_context* ctx = map->ctx;
  if (n > kMaxBeams || (!just_once && map->levels.size() > 1 && map->n_cached > kMaxBeams))
    return ctx->fail(LSLAM_ERR_UNSUPPORTED, "at most %d points per scan are supported (got %d)", kMaxBeams, n);
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  const bool pipelined = !just_once && !map->force_two_kernels;
  if (!pipelined) {
    int rc = flush_pending(map);
    if (rc) return rc;
  }
  const int n_levels = just_once ? 1 : (int)map->levels.size();
  const float* const d_pts0 = d_pts;
  const int n0 = n;
  const float origo0[2] = {origo[0], origo[1]};
  // ---- phase 1 (pipelined path): everything that can FAIL -- the second key-plane set, the point buffers of the set this
  // scan will use, the mark reset when an epoch wraps -- for every level, before any epoch is bumped: an error return
  // leaves every level's (epoch, pending) pair as it was
  if (pipelined) {
    for (int li = 0; li < n_levels; li++) {
      Level& L = map->levels[li];
      if (L.epoch >= kMaxEpoch) {
        int rc = flush_pending(map);  // the pending scan's keys are about to be cleared
        if (rc == LSLAM_OK) rc = clear_marks(map, L);
        if (rc) return rc;
      }
      const size_t cells = (size_t)L.sx * L.sy;
      if (!L.d_free2) {  // second key-plane set, on first use of the pipelined path
        if (hipMalloc((void**)&L.d_free2, cells * sizeof(uint32_t)) != hipSuccess ||
            hipMalloc((void**)&L.d_occ2, cells * sizeof(uint32_t)) != hipSuccess) {
          (void)hipGetLastError();
          if (L.d_free2) (void)hipFree(L.d_free2);
          L.d_free2 = L.d_occ2 = nullptr;
          return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the second key planes of map level %d", li);
        }
        LSLAM_HIP(ctx, hipMemsetAsync(L.d_free2, 0, cells * sizeof(uint32_t), ctx->stream));
        LSLAM_HIP(ctx, hipMemsetAsync(L.d_occ2, 0, cells * sizeof(uint32_t), ctx->stream));
      }
      const int n_li = li == 0 ? n0 : map->n_cached;
      const int set = (int)((L.epoch + 1u) & 1u);
      LSLAM_HIP(ctx, L.pipe_pts[set].reserve((size_t)2 * std::max(n_li, 1)));  // nothing in flight reads THIS set's points
    }
  }
  PipeJobs J{};
  int blocks = 0;
  auto submit = [&]() {
    if (blocks > 0) launch(ctx, "logodds_pipe", k_logodds_pipe_ml, dim3(blocks), dim3(256), 0, J);
    J = PipeJobs{};
    blocks = 0;
  };
  for (int li = 0; li < n_levels; li++) {
    Level& L = map->levels[li];
    // level 0 takes the container it is handed, level i > 0 takes dataContainers[i-1] = what the last matchData
    // cached (H/slam_main/MapRepMultiMap.h:174-191) -- empty until the first matchData
    d_pts = li == 0 ? d_pts0 : map->d_cached.p;
    n = li == 0 ? n0 : map->n_cached;
    origo = li == 0 ? origo0 : map->cached_origo;
    if (!pipelined && L.epoch >= kMaxEpoch) {
      int rc = clear_marks(map, L);  // (nothing is pending on this path)
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
    if (pipelined) {
      // [apply of this level's pending scan | mark of this one], the jobs of ALL levels in one launch
      const int set = (int)(L.epoch & 1u);
      if (J.n_jobs + 2 > kPipeMaxJobs) submit();
      if (L.pending) {
        J.j[J.n_jobs++] = apply_job(L, blocks);
        blocks += pipe_blocks(L.pend_n);
      }
      if (n > 0) {
        PipeJob j{};
        j.g = g;
        j.pts = d_pts;
        j.n = n;
        j.first_block = blocks;
        j.apply = 0;
        j.free_w = set ? L.d_free2 : L.d_free;
        j.occ_w = set ? L.d_occ2 : L.d_occ;
        j.pts_copy = L.pipe_pts[set].p;
        J.j[J.n_jobs++] = j;
        blocks += pipe_blocks(n);
      }
      L.pending = n > 0;
      L.pend_g = g;
      L.pend_n = n;
    } else if (n > 0) {
      dim3 grid((n + 3) / 4), block(256);  // 4 waves per block, one wave per beam
      launch(ctx, "logodds_mark", k_logodds_mark, grid, block, 0, g, d_pts, n, L.d_free, L.d_occ);
      launch(ctx, "logodds_apply", k_logodds_apply, grid, block, 0, g, d_pts, n, (const uint32_t*)L.d_free,
             (const uint32_t*)L.d_occ, L.d_logodds);
    }
  }
  if (pipelined) submit();
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
      map->levels.push_back(std::move(L));
      lslam_map_destroy(map);
      return ctx->fail(LSLAM_ERR_HIP, "cannot allocate map level %d (%dx%d) in HBM", i, sx, sy);
    }
    (void)hipMemsetAsync(L.d_logodds, 0, n * sizeof(float), ctx->stream);
    (void)hipMemsetAsync(L.d_free, 0, n * sizeof(uint32_t), ctx->stream);
    (void)hipMemsetAsync(L.d_occ, 0, n * sizeof(uint32_t), ctx->stream);
    map->levels.push_back(std::move(L));
    sx /= 2;     // resolution /= 2 (H/slam_main/MapRepMultiMap.h:83)
    sy /= 2;
    cl *= 2.0f;  // :84
  }
  {
    const char* e = getenv("LSLAM_MAP_TWO_KERNELS");
    map->force_two_kernels = e && e[0] == '1';
    e = getenv("LSLAM_GN_ORDERED");
    map->ordered_sums = e && e[0] == '1';
    e = getenv("LSLAM_GN_THREADS");
    if (e && atoi(e) >= 256) map->gn_threads = atoi(e);
  }
  (void)hipStreamSynchronize(ctx->stream);
  ctx->pre_sync.emplace_back((void*)map, [](void* m) { return lslam_map_flush((lslam_map*)m); });  // lslam_synchronize flushes
  // ... and, once the stream has drained, makes a window miss LOUD: cells that LSLAM_MAP_OPT_BATCH_RADIUS_CELLS left outside
  // their scan's window were dropped -- the map no longer equals the reference's -- so the synchronise that learns of it fails
  ctx->post_sync.emplace_back((void*)map, [](void* mp) {
    lslam_map* m = (lslam_map*)mp;
    if (!m->d_batch_misses) return (int)LSLAM_OK;
    const unsigned long long now = __atomic_load_n(m->d_batch_misses, __ATOMIC_ACQUIRE);
    if (now == m->batch_misses_reported) return (int)LSLAM_OK;
    const unsigned long long fresh = now - m->batch_misses_reported;
    m->batch_misses_reported = now;
    return m->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT,
                        "lslam_map_update_batch_dev: %llu hit / free cells lay outside their scan's window and were DROPPED "
                        "(LSLAM_MAP_OPT_BATCH_RADIUS_CELLS understates a scan's reach): the map has diverged from "
                        "updateByScan's (OccGridMapBase.h:118-168); raise the hint or set it to 0 (whole-map windows)", fresh);
  });
  *out = map;
  return LSLAM_OK;
}

void lslam_map_destroy(lslam_map* map) {
  if (!map) return;
  for (size_t i = 0; i < map->ctx->pre_sync.size(); i++)
    if (map->ctx->pre_sync[i].first == (void*)map) {
      map->ctx->pre_sync.erase(map->ctx->pre_sync.begin() + (long)i);
      break;
    }
  for (size_t i = 0; i < map->ctx->post_sync.size(); i++)
    if (map->ctx->post_sync[i].first == (void*)map) {
      map->ctx->post_sync.erase(map->ctx->post_sync.begin() + (long)i);
      break;
    }
  (void)hipSetDevice(map->ctx->device);
  (void)hipStreamSynchronize(map->ctx->stream);
  for (auto& L : map->levels) {
    if (L.d_logodds) (void)hipFree(L.d_logodds);
    if (L.d_free) (void)hipFree(L.d_free);
    if (L.d_occ) (void)hipFree(L.d_occ);
    if (L.d_free2) (void)hipFree(L.d_free2);
    if (L.d_occ2) (void)hipFree(L.d_occ2);
    L.pipe_pts[0].release();
    L.pipe_pts[1].release();
    L.d_pool.release();
    if (L.d_flags) (void)hipFree(L.d_flags);
  }
  if (map->d_batch_misses) (void)hipHostFree(map->d_batch_misses);
  map->d_pts.release();
  map->d_cached.release();
  map->d_gn_out.release();
  map->d_hash.release();
  map->d_hdr.release();
  map->d_scan.release();
  map->d_scan_ranges.release();
  map->d_cossin.release();
  map->d_i8.release();
  if (map->h_stage) (void)hipHostFree(map->h_stage);
  if (map->h_gn_pts) (void)hipHostFree(map->h_gn_pts);
  if (map->h_gn_out) (void)hipHostFree(map->h_gn_out);
  for (auto e : map->stage_ev)
    if (e) (void)hipEventDestroy(e);
  delete map;
}

int lslam_map_reset(lslam_map* map) {
  if (!map) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  for (auto& L : map->levels) {
    L.pending = false;  // a reset map has no use for the deferred apply
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

namespace {
// Copy the caller's points through a pinned ring slot to d_pts on the context stream.  The caller's
// buffer is free again when this returns; the device copy is ordered before the kernels that follow.
int stage_points(lslam_map* map, const float* pts, int n) {
  lslam_context* ctx = map->ctx;
  const size_t floats = (size_t)2 * (n > 0 ? n : 1);
  LSLAM_HIP(ctx, map->d_pts.reserve(floats));
  if (n <= 0) return LSLAM_OK;
  if (floats > map->stage_cap) {  // (re)build the ring; drains whatever is in flight first
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (map->h_stage) (void)hipHostFree(map->h_stage);
    map->h_stage = nullptr;
    map->stage_cap = 0;
    const size_t cap = floats + floats / 4 + 64;
    LSLAM_HIP(ctx, hipHostMalloc((void**)&map->h_stage, cap * lslam_map::kStageSlots * sizeof(float), hipHostMallocDefault));
    map->stage_cap = cap;
    for (int i = 0; i < lslam_map::kStageSlots; i++) {
      if (!map->stage_ev[i]) LSLAM_HIP(ctx, hipEventCreateWithFlags(&map->stage_ev[i], hipEventDisableTiming));
      map->stage_busy[i] = false;
    }
  }
  const int slot = (int)(map->stage_next++ % lslam_map::kStageSlots);
  if (map->stage_busy[slot]) LSLAM_HIP(ctx, hipEventSynchronize(map->stage_ev[slot]));
  float* h = map->h_stage + (size_t)slot * map->stage_cap;
  memcpy(h, pts, (size_t)2 * n * sizeof(float));
  LSLAM_HIP(ctx, hipMemcpyAsync(map->d_pts.p, h, (size_t)2 * n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  LSLAM_HIP(ctx, hipEventRecord(map->stage_ev[slot], ctx->stream));
  map->stage_busy[slot] = true;
  return LSLAM_OK;
}
}  // namespace

// Asynchronous like every device-side mutation of the map: the update is ENQUEUED on the context stream
// when this returns (the caller's `pts` may be reused at once); the readers -- lslam_map_read_*,
// lslam_map_match_data, lslam_synchronize -- are ordered after it.
int lslam_map_update_by_scan(lslam_map* map, const float* pts, int n, const float origo[2], const float pose[3]) {
  if (!map || n < 0 || (n > 0 && !pts) || !origo || !pose) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  if (n > 0 && n == map->gn_host_n && map->levels.size() >= 1 && memcmp(pts, map->h_gn_pts, (size_t)2 * n * sizeof(float)) == 0)
    return update_impl(map, map->d_cached.p, n, origo, pose, 0, 0.f, 0.f, 0.0);  // the container just matched: already in HBM
  int rc = stage_points(map, pts, n);
  if (rc) return rc;
  return update_impl(map, map->d_pts.p, n, origo, pose, 0, 0.f, 0.f, 0.0);
}

namespace {
constexpr int kBatchMaxScans = kBatchSlots;

// The window of tiles of one scan on one level: every cell of a ray lies between the begin cell and the end cell, and the
// end cell is within |p - origo| * factor of the begin cell (a rotation), + 1 for the two roundings, + 1 for float32
// slack.  reach < 0 (or absurdly large) = unknown: the whole map.  out = tx0, ty0, tw, th (tw = 0: nothing reaches the map).
void plan_window(int sx, int sy, int bx, int by, int n, double reach, double factor, int out[4]) {
  long long x0 = 0, y0 = 0, x1 = sx - 1, y1 = sy - 1;
  if (reach >= 0.0 && reach < 1e9) {
    const long long R = (long long)ceil(reach * factor) + 2;
    x0 = std::max<long long>(x0, (long long)bx - R); x1 = std::min<long long>(x1, (long long)bx + R);
    y0 = std::max<long long>(y0, (long long)by - R); y1 = std::min<long long>(y1, (long long)by + R);
  }
  if (n == 0 || x1 < x0 || y1 < y0 || bx < 0 || by < 0 || bx >= sx || by >= sy) {
    out[0] = out[1] = out[2] = out[3] = 0;  // (a begin cell outside the map drops every beam, H/map/OccGridMapBase.h:226-238)
  } else {
    out[0] = (int)(x0 >> 3); out[1] = (int)(y0 >> 3);
    out[2] = (int)(x1 >> 3) - out[0] + 1; out[3] = (int)(y1 >> 3) - out[1] + 1;
  }
}
// Rounds: consecutive scans whose windows fit the budget together; a scan whose window alone exceeds it gets a round
// -- and the memory -- of its own.  base[k] = first tile slot of scan k inside its round; round_of[k] (may be null).
// Returns the number of rounds, or -1 when one window has 2^26 tiles or more (byte offsets into the pool are 32-bit).
int plan_rounds(const int* tw, const int* th, int K, size_t budget_bytes, uint32_t* base, int* round_of, size_t* max_slots) {
  const size_t budget_slots = std::max<size_t>(1, budget_bytes / 64);
  int n_rounds = 0;
  size_t most = 0;
  for (int k0 = 0; k0 < K;) {
    size_t sum = 0;
    int k1 = k0;
    while (k1 < K) {
      const size_t need = (size_t)tw[k1] * (size_t)th[k1];
      if (k1 > k0 && sum + need > budget_slots) break;
      if (sum + need >= ((size_t)1 << 26)) break;
      base[k1] = (uint32_t)sum;
      if (round_of) round_of[k1] = n_rounds;
      sum += need;
      k1++;
    }
    if (k1 == k0) return -1;
    most = std::max(most, sum);
    n_rounds++;
    k0 = k1;
  }
  if (max_slots) *max_slots = most;
  return n_rounds;
}

// K successive MapRepMultiMap::updateByScan calls (every level fed the same scan, i.e. each scan matched first) in
// four launches per level and round.  d_pts: the K containers back to back; counts / origos / poses are host arrays;
// radius (host, may be null): per scan, the distance of its farthest point from its origo in level-0 cells.
int update_batch_impl(lslam_map* map, int K, const float* d_pts, const int32_t* counts, const float* origos,
                      const float* poses, const float* radius, bool use_hint) {
  lslam_context* ctx = map->ctx;
  {
    int rc = flush_pending(map);  // a single-scan update may still owe its apply
    if (rc) return rc;
  }
  int n_max = 0;
  for (int k = 0; k < K; k++) {
    if (counts[k] < 0 || counts[k] > kMaxBeams)
      return ctx->fail(LSLAM_ERR_UNSUPPORTED, "at most %d points per scan are supported (got %d)", kMaxBeams, counts[k]);
    n_max = std::max(n_max, counts[k]);
  }
  if (n_max == 0) return LSLAM_OK;
  for (const Level& L : map->levels)
    if (L.sx > 32768 || L.sy > 32768)
      return ctx->fail(LSLAM_ERR_UNSUPPORTED, "batched update: map sides up to 32768 cells (k_lo_batch_rays' 32-bit ray arithmetic)");
  uint32_t slots = 64;
  while (slots < 2u * (uint32_t)n_max) slots *= 2;
  LSLAM_HIP(ctx, map->d_hash.reserve((size_t)3 * K * slots));
  LSLAM_HIP(ctx, map->d_hdr.reserve((size_t)K * map->levels.size()));
  if (!map->d_batch_misses) {
    LSLAM_HIP(ctx, hipHostMalloc((void**)&map->d_batch_misses, sizeof(unsigned long long), hipHostMallocDefault));
    *map->d_batch_misses = 0ull;
  }
  std::vector<ScanHdr> hdr((size_t)K * map->levels.size());
  std::vector<int> off(K + 1, 0);
  for (int k = 0; k < K; k++) off[k + 1] = off[k] + counts[k];
  struct Round { int k0, k1; size_t slots; };
  std::vector<std::vector<Round>> rounds(map->levels.size());
  for (size_t li = 0; li < map->levels.size(); li++) {
    Level& L = map->levels[li];
    L.tiles_x = (L.sx + 7) / 8;
    L.n_tiles = L.tiles_x * ((L.sy + 7) / 8);
    const float factor = li == 0 ? 1.0f : (float)(1.0 / pow(2.0, (double)li));
    for (int k = 0; k < K; k++) {
      ScanHdr& h = hdr[li * K + k];
      const float* pose = poses + 3 * k;
      const float ox = li == 0 ? origos[2 * k] : origos[2 * k] * factor;
      const float oy = li == 0 ? origos[2 * k + 1] : origos[2 * k + 1] * factor;
      const float sc = L.scale_to_map;
      const float mx = (sc * pose[0] + 0.0f * pose[1]) + L.t_x;  // getMapCoordsPose (H/map/GridMapBase.h:238-242)
      const float my = (0.0f * pose[0] + sc * pose[1]) + L.t_y;
      h.c = cosf(pose[2]);
      h.s = sinf(pose[2]);
      h.tx = mx; h.ty = my;
      const float bxf = (h.c * ox + (-h.s) * oy) + mx;  // H/map/OccGridMapBase.h:132
      const float byf = (h.s * ox + h.c * oy) + my;
      h.bx = (int)(bxf + 0.5f);                         // :135
      h.by = (int)(byf + 0.5f);
      h.n = counts[k];
      h.pts_off = off[k];
      double reach = -1.0;  // unknown: the whole map
      if (radius) reach = (double)radius[k];
      else if (use_hint && map->batch_radius_hint > 0) reach = (double)map->batch_radius_hint;  // device-resident points only
      int w[4];
      plan_window(L.sx, L.sy, h.bx, h.by, h.n, reach, (double)factor, w);
      h.tx0 = w[0]; h.ty0 = w[1]; h.tw = w[2]; h.th = w[3];
      h.base = 0; h.pad[0] = h.pad[1] = h.pad[2] = 0;
    }
    {
      std::vector<int> tw(K), th(K), round_of(K);
      std::vector<uint32_t> base(K);
      for (int k = 0; k < K; k++) { tw[k] = hdr[li * K + k].tw; th[k] = hdr[li * K + k].th; }
      const int nr = plan_rounds(tw.data(), th.data(), K, map->batch_budget, base.data(), round_of.data(), nullptr);
      if (nr < 0) return ctx->fail(LSLAM_ERR_UNSUPPORTED, "batched update: a scan window of 2^26 tiles or more");
      for (int k = 0; k < K; k++) {
        hdr[li * K + k].base = base[k];
        if (k == 0 || round_of[k] != round_of[k - 1]) rounds[li].push_back(Round{k, k, 0});
        Round& r = rounds[li].back();
        r.k1 = k + 1;
        r.slots = std::max(r.slots, (size_t)base[k] + (size_t)tw[k] * (size_t)th[k]);
      }
    }
  }
  map->batch_last_rounds = (int)rounds[0].size();
  LSLAM_HIP(ctx, hipMemcpyAsync(map->d_hdr.p, hdr.data(), hdr.size() * sizeof(ScanHdr), hipMemcpyHostToDevice, ctx->stream));
  for (size_t li = 0; li < map->levels.size(); li++) {
    Level& L = map->levels[li];
    if (!L.d_flags) {
      if (hipMalloc((void**)&L.d_flags, (size_t)L.n_tiles * kBatchSlots) != hipSuccess) {
        (void)hipGetLastError();
        L.d_flags = nullptr;
        return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the tile flags of the batched update (%zu bytes)", (size_t)L.n_tiles * kBatchSlots);
      }
      L.batch_epoch = 0;
    }
    size_t need_slots = 1;
    for (const Round& r : rounds[li]) need_slots = std::max(need_slots, r.slots);
    if (need_slots * 64 > L.d_pool.cap) {
      // grow to what this call needs, rounded up to 1 MB (rare: the windows of a trajectory have a steady size).  The
      // old pool goes first -- nothing may still read it, so the stream is drained once -- instead of living on beside the new one
      LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
      L.d_pool.release();
      // a sixteenth of headroom: the windows of the next 64 scans are clipped a little differently by the map's edges, and
      // growing again means draining the stream again
      const size_t want = need_slots * 64 + need_slots * 4;
      const size_t bytes = ((want + ((size_t)1 << 20) - 1) >> 20) << 20;
      uint8_t* fresh = nullptr;
      if (hipMalloc((void**)&fresh, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return ctx->fail(LSLAM_ERR_HIP, "cannot allocate %zu bytes of tile slots for the batched update", bytes);
      }
      L.d_pool.p = fresh;
      L.d_pool.cap = bytes;
      L.batch_epoch = 0;  // fresh memory: cleared below
    }
    for (const Round& r : rounds[li]) {
      const int Kr = r.k1 - r.k0;
      if (L.batch_epoch == 0 || L.batch_epoch >= 63) {  // 6-bit epoch in the pool / flag bytes
        LSLAM_HIP(ctx, hipMemsetAsync(L.d_pool.p, 0, L.d_pool.cap, ctx->stream));
        LSLAM_HIP(ctx, hipMemsetAsync(L.d_flags, 0, (size_t)L.n_tiles * kBatchSlots, ctx->stream));
        L.batch_epoch = 0;
      }
      L.batch_epoch++;
      L.pool_slots = L.d_pool.cap / 64;
      const bool lds_hash = (size_t)slots * 8 <= 64 * 1024;  // k_lo_batch_hits_lds writes the key and first-hit tables whole
      if (lds_hash)
        LSLAM_HIP(ctx, hipMemsetAsync(map->d_hash.p + (size_t)2 * Kr * slots, 0xFF, (size_t)Kr * slots * sizeof(uint32_t), ctx->stream));
      else
        LSLAM_HIP(ctx, hipMemsetAsync(map->d_hash.p, 0xFF, (size_t)3 * Kr * slots * sizeof(uint32_t), ctx->stream));
      BatchGeom g;
      g.sx = L.sx; g.sy = L.sy; g.K = Kr;
      g.tiles_x = L.tiles_x; g.n_tiles = L.n_tiles;
      g.factor = li == 0 ? 1.0f : (float)(1.0 / pow(2.0, (double)li));
      g.lo_free = map->lo_free; g.lo_occ = map->lo_occ;
      g.tag = L.batch_epoch << 2;
      g.hash_mask = slots - 1;
      uint32_t* hkey = map->d_hash.p;
      uint32_t* hhit = hkey + (size_t)Kr * slots;
      uint32_t* hcross = hhit + (size_t)Kr * slots;
      const ScanHdr* d_h = map->d_hdr.p + li * K + r.k0;
      int nr_max = 0;
      for (int k = r.k0; k < r.k1; k++) nr_max = std::max(nr_max, counts[k]);
      if (nr_max == 0) continue;
      if (lds_hash)
        launch(ctx, "lo_batch_hits", k_lo_batch_hits_lds, dim3(Kr), dim3(1024), (size_t)slots * 8, g, d_h, d_pts, L.d_pool.p,
               L.d_flags, hkey, hhit, map->d_batch_misses);
      else
        launch(ctx, "lo_batch_hits", k_lo_batch_hits, dim3((nr_max + 255) / 256, Kr), dim3(256), 0, g, d_h, d_pts, L.d_pool.p,
               L.d_flags, hkey, hhit, map->d_batch_misses);
      const int groups = (nr_max + kRayBeamsPerWave - 1) / kRayBeamsPerWave;
      const long long waves = (long long)Kr * groups;
      launch(ctx, "lo_batch_rays", k_lo_batch_rays, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, g, d_h, d_pts, L.d_pool.p,
             L.d_flags, (const uint32_t*)hkey, hcross, groups);
      launch(ctx, "lo_batch_resolve", k_lo_batch_resolve, dim3((unsigned)(((size_t)Kr * slots + 255) / 256)), dim3(256), 0, g,
             d_h, (const uint32_t*)hkey, (const uint32_t*)hhit, (const uint32_t*)hcross, L.d_pool.p);
      launch(ctx, "lo_batch_apply", k_lo_batch_apply, dim3((unsigned)((L.n_tiles + 3) / 4)), dim3(256), 0, g, d_h,
             (const uint8_t*)L.d_pool.p, (const uint8_t*)L.d_flags, L.d_logodds);
    }
  }
  LSLAM_HIP(ctx, hipGetLastError());
  return LSLAM_OK;
}
}  // namespace

namespace {
// n_scans containers back to back in points_xy; batches larger than 64 scans are cut into groups of 64
int update_batch_dev_impl(lslam_map* map, int n_scans, const float* points_xy_dev, const int32_t* n_points,
                          const float* origos_xy, const float* poses_world, const float* radius, bool use_hint) {
  lslam_context* ctx = map->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  size_t done_pts = 0;
  for (int k0 = 0; k0 < n_scans; k0 += kBatchMaxScans) {
    const int K = std::min(kBatchMaxScans, n_scans - k0);
    int rc = update_batch_impl(map, K, points_xy_dev + 2 * done_pts, n_points + k0, origos_xy + 2 * k0, poses_world + 3 * k0,
                               radius ? radius + k0 : nullptr, use_hint);
    if (rc) return rc;
    for (int k = 0; k < K; k++) done_pts += (size_t)n_points[k0 + k];
  }
  if (n_scans > 0 && map->levels.size() > 1) {  // dataContainers now hold the last scan (as after its matchData)
    const int last = n_scans - 1;
    const int n = n_points[last];
    map->gn_host_n = -1;
    LSLAM_HIP(ctx, map->d_cached.reserve((size_t)2 * (n > 0 ? n : 1)));
    if (n > 0)
      LSLAM_HIP(ctx, hipMemcpyAsync(map->d_cached.p, points_xy_dev + 2 * (done_pts - (size_t)n), (size_t)2 * n * sizeof(float),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    map->n_cached = n;
    map->cached_origo[0] = origos_xy[2 * last];
    map->cached_origo[1] = origos_xy[2 * last + 1];
  }
  return LSLAM_OK;
}
}  // namespace

int lslam_map_update_batch_dev(lslam_map* map, int n_scans, const float* points_xy_dev, const int32_t* n_points,
                               const float* origos_xy, const float* poses_world) {
  if (!map || n_scans < 0 || !n_points || !origos_xy || !poses_world) return LSLAM_ERR_INVALID_ARGUMENT;
  // the host never sees these points: the windows come from LSLAM_MAP_OPT_BATCH_RADIUS_CELLS, or cover the whole map
  return update_batch_dev_impl(map, n_scans, points_xy_dev, n_points, origos_xy, poses_world, nullptr, true);
}

int lslam_map_update_batch(lslam_map* map, int n_scans, const float* points_xy, const int32_t* n_points,
                           const float* origos_xy, const float* poses_world) {
  if (!map || n_scans < 0 || !n_points || !origos_xy || !poses_world) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  size_t total = 0;
  for (int k = 0; k < n_scans; k++) {
    if (n_points[k] < 0) return LSLAM_ERR_INVALID_ARGUMENT;
    total += (size_t)n_points[k];
  }
  if (total > 0 && !points_xy) return LSLAM_ERR_INVALID_ARGUMENT;
  if (total > (size_t)INT32_MAX / 2) return ctx->fail(LSLAM_ERR_UNSUPPORTED, "batch too large");
  // The scans' reach, where it decides anything: on a map whose whole planes fit the budget every window is the map
  // anyway.  One pass over the points the host is about to copy: max |p - origo| per scan, in level-0 cells.
  std::vector<float> radius;
  if (!map->levels.empty()) {
    const Level& L0 = map->levels[0];
    const size_t plane = (size_t)((L0.sx + 7) / 8) * (size_t)((L0.sy + 7) / 8) * 64;
    if (plane * (size_t)std::min(n_scans, kBatchMaxScans) > map->batch_budget) {
      radius.resize((size_t)n_scans);
      size_t at = 0;
      for (int k = 0; k < n_scans; k++) {
        const float ox = origos_xy[2 * k], oy = origos_xy[2 * k + 1];
        float m2 = 0.f;
        bool finite = true;
        for (int i = 0; i < n_points[k]; i++) {
          const float dx = points_xy[2 * (at + i)] - ox, dy = points_xy[2 * (at + i) + 1] - oy;
          const float d2 = dx * dx + dy * dy;
          finite = finite && (d2 == d2) && d2 < 1e18f;
          m2 = d2 > m2 ? d2 : m2;
        }
        at += (size_t)n_points[k];
        radius[k] = finite ? sqrtf(m2) * 1.0001f + 1.0f : 2e9f;  // a non-finite point: no bound (such a beam is dropped anyway)
      }
    }
  }
  int rc = stage_points(map, points_xy, (int)total);
  if (rc) return rc;
  // (the hint is for points the host cannot see; these it has just measured, or their windows are the map anyway)
  return update_batch_dev_impl(map, n_scans, map->d_pts.p, n_points, origos_xy, poses_world, radius.empty() ? nullptr : radius.data(),
                               false);
}

// Host-only: what a batched update of these scans would allocate on a level of sx x sy cells -- the planner
// update_batch_impl itself uses.  No context, no GPU.
int lslam_map_plan_batch_windows(int sx, int sy, int n_scans, const int32_t* begin_cells_xy, const int32_t* n_points,
                                 const double* reach_cells, double level_factor, int64_t budget_bytes, int32_t* windows_out,
                                 uint32_t* base_out, int32_t* round_out, int64_t* pool_bytes_out) {
  if (sx <= 0 || sy <= 0 || n_scans < 0 || budget_bytes < 64 || (n_scans > 0 && (!begin_cells_xy || !n_points || !windows_out || !base_out)))
    return LSLAM_ERR_INVALID_ARGUMENT;
  std::vector<int> tw((size_t)n_scans), th((size_t)n_scans);
  for (int k = 0; k < n_scans; k++) {
    int w[4];
    plan_window(sx, sy, begin_cells_xy[2 * k], begin_cells_xy[2 * k + 1], n_points[k], reach_cells ? reach_cells[k] : -1.0,
                level_factor, w);
    for (int i = 0; i < 4; i++) windows_out[4 * k + i] = w[i];
    tw[k] = w[2]; th[k] = w[3];
  }
  size_t most = 0;
  const int nr = plan_rounds(tw.data(), th.data(), n_scans, (size_t)budget_bytes, base_out, round_out, &most);
  if (nr < 0) return LSLAM_ERR_UNSUPPORTED;
  if (pool_bytes_out) *pool_bytes_out = (int64_t)(most * 64);
  return nr;
}

// out[0] = bytes of scratch the batched update holds right now over all levels (tile-slot pools + tile flags),
// out[1] = rounds level 0 of the last batch needed, out[2] = cells found outside their scan's window since the map was
// created (0 unless LSLAM_MAP_OPT_BATCH_RADIUS_CELLS understated a scan's reach; synchronises), out[3] = the budget
int lslam_map_batch_stats(lslam_map* map, int64_t out[4]) {
  if (!map || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  size_t bytes = 0;
  for (const Level& L : map->levels) bytes += L.d_pool.cap + (L.d_flags ? (size_t)L.n_tiles * kBatchSlots : 0);
  out[0] = (int64_t)bytes;
  out[1] = map->batch_last_rounds;
  out[2] = 0;
  out[3] = (int64_t)map->batch_budget;
  if (map->d_batch_misses) {
    LSLAM_HIP(ctx, hipSetDevice(ctx->device));
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    out[2] = (int64_t)__atomic_load_n(map->d_batch_misses, __ATOMIC_ACQUIRE);
    map->batch_misses_reported = (unsigned long long)out[2];  // the caller has seen the count: not an error a second time
  }
  return LSLAM_OK;
}

int lslam_map_set_scan(lslam_map* map, const float* ranges, int n, const lslam_hector_scan* sp, int* n_points) {
  if (!map || n < 0 || (n > 0 && !ranges) || !sp) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  if (n > kMaxBeams) return ctx->fail(LSLAM_ERR_UNSUPPORTED, "at most %d readings per scan (got %d)", kMaxBeams, n);
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  LSLAM_HIP(ctx, map->d_scan.reserve((size_t)2 * std::max(n, 1) + 4));
  LSLAM_HIP(ctx, map->d_scan_ranges.reserve((size_t)std::max(n, 1)));
  if (n > map->cs_n || sp->angle_min != map->cs_angle_min || sp->angle_increment != map->cs_angle_inc) {
    // laser_geometry's co_sine_map_ (rebuilt when the scan geometry changes): host libm, like the reference's stack
    std::vector<double2> cs((size_t)std::max(n, 1));
    for (int i = 0; i < n; i++) {
      const double a = (double)sp->angle_min + (double)i * (double)sp->angle_increment;
      cs[i] = make_double2(cos(a), sin(a));
    }
    LSLAM_HIP(ctx, map->d_cossin.reserve(cs.size()));
    LSLAM_HIP(ctx, hipMemcpyAsync(map->d_cossin.p, cs.data(), cs.size() * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // cs dies with this scope
    map->cs_n = n;
    map->cs_angle_min = sp->angle_min;
    map->cs_angle_inc = sp->angle_increment;
  }
  if (n > 0)
    LSLAM_HIP(ctx, hipMemcpyAsync(map->d_scan_ranges.p, ranges, (size_t)n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
  ProjectCfg c;
  c.n = n;
  c.range_min = sp->range_min;
  c.cutoff = sp->range_cutoff < 0.f ? sp->range_max : sp->range_cutoff;  // laser_geometry: cutoff < 0 -> range_max
  c.sqr_min = sp->sqr_laser_min_dist;
  c.sqr_max = sp->sqr_laser_max_dist;
  c.use_max_sq = (double)sp->use_max_scan_range * (double)sp->use_max_scan_range;
  c.z_min = sp->laser_z_min;
  c.z_max = sp->laser_z_max;
  c.cy = cos((double)sp->laser_yaw);
  c.sy = sin((double)sp->laser_yaw);
  c.tx = sp->laser_x; c.ty = sp->laser_y; c.tz = sp->laser_z;
  c.scale_to_map = map->levels[0].scale_to_map;
  int* d_n = reinterpret_cast<int*>(map->d_scan.p + (size_t)2 * std::max(n, 1));
  launch(ctx, "hector_project", k_hector_project, dim3(1), dim3(1024), 0, c, (const float*)map->d_scan_ranges.p,
         (const double2*)map->d_cossin.p, map->d_scan.p, d_n);
  int host_n = 0;
  LSLAM_HIP(ctx, hipMemcpyAsync(&host_n, d_n, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  map->n_scan = host_n;
  // dataContainer.setOrigo(Eigen::Vector2f(laserPos.x(), laserPos.y()) * scaleToMap) (hector_slam.cc:331)
  map->scan_origo[0] = (float)(double)sp->laser_x * c.scale_to_map;
  map->scan_origo[1] = (float)(double)sp->laser_y * c.scale_to_map;
  if (n_points) *n_points = host_n;
  return LSLAM_OK;
}

int lslam_map_read_container(lslam_map* map, float* out_xy, int capacity, float origo_xy[2]) {
  if (!map || capacity < 0 || (capacity > 0 && !out_xy)) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  const int n = std::min(capacity, map->n_scan);
  if (n > 0) {
    LSLAM_HIP(ctx, hipMemcpyAsync(out_xy, map->d_scan.p, (size_t)2 * n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (origo_xy) { origo_xy[0] = map->scan_origo[0]; origo_xy[1] = map->scan_origo[1]; }
  return map->n_scan;
}

int lslam_map_update_by_container(lslam_map* map, const float pose_world[3]) {
  if (!map || !pose_world) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(map->ctx, hipSetDevice(map->ctx->device));
  return update_impl(map, map->d_scan.p, map->n_scan, map->scan_origo, pose_world, 0, 0.f, 0.f, 0.0);
}

int lslam_map_update_just_once(lslam_map* map, const float* pts, int n, const float origo[2], float begin_x,
                               float begin_y, double metres_per_cell) {
  if (!map || n < 0 || (n > 0 && !pts) || !origo || !(metres_per_cell > 0)) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  int rc = stage_points(map, pts, n);
  if (rc) return rc;
  const float pose[3] = {0.f, 0.f, 0.f};
  return update_impl(map, map->d_pts.p, n, origo, pose, 1, begin_x, begin_y, metres_per_cell);
}

namespace {
int match_data_impl(lslam_map* map, const float* pts, int n, bool pts_on_device, const float origo[2],
                    const float begin_world[3], float out_pose[3], float out_cov[9]) {
  lslam_context* ctx = map->ctx;
  if ((int)map->levels.size() > kGnMaxLevels)
    return ctx->fail(LSLAM_ERR_UNSUPPORTED, "at most %d pyramid levels", kGnMaxLevels);
  if (map->ordered_sums && (size_t)n * 9 * sizeof(float) > 150 * 1024)
    return ctx->fail(LSLAM_ERR_UNSUPPORTED, "at most %d points per scan in matchData", (int)(150 * 1024 / 36));
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  {
    int rc = flush_pending(map);  // the matcher reads the float planes
    if (rc) return rc;
  }
  LSLAM_HIP(ctx, map->d_cached.reserve((size_t)2 * (n > 0 ? n : 1)));
  LSLAM_HIP(ctx, map->d_gn_out.reserve(16));
  float* d_out = map->d_gn_out.p;
  map->gn_host_n = -1;  // d_cached is about to be rewritten; the fast host-fed path below re-validates its host copy
  if (map->levels.size() > 1) {
    map->n_cached = n;
    map->cached_origo[0] = origo ? origo[0] : 0.f;
    map->cached_origo[1] = origo ? origo[1] : 0.f;
  }
  GnLevels lv;
  lv.n_levels = (int)map->levels.size();
  for (int i = 0; i < lv.n_levels; i++) {
    const Level& L = map->levels[i];
    lv.sx[i] = L.sx; lv.sy[i] = L.sy; lv.scale[i] = L.scale_to_map; lv.t_x[i] = L.t_x; lv.t_y[i] = L.t_y;
    lv.logodds[i] = L.d_logodds;
  }
  if (!map->ordered_sums) {
    // ---- parallel sums (default): one launch, no copy operation on the stream --------------------------------------------
    if (!map->h_gn_out && hipHostMalloc((void**)&map->h_gn_out, 16 * sizeof(float), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the pinned result of matchData");
    }
    if (map->gn_ticket == 0) ((int*)map->h_gn_out)[15] = 0;  // fresh allocation: no stale word may look like ticket 1
    const float* src = pts;
    if (!pts_on_device && n > 0) {
      if ((size_t)2 * n > map->h_gn_cap) {
        if (map->h_gn_pts) (void)hipHostFree(map->h_gn_pts);
        map->h_gn_pts = nullptr;
        map->h_gn_cap = 0;
        const size_t want = (size_t)2 * n + 256;
        if (hipHostMalloc((void**)&map->h_gn_pts, want * sizeof(float), hipHostMallocDefault) != hipSuccess) {
          (void)hipGetLastError();
          return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the pinned staging of matchData");
        }
        map->h_gn_cap = want;
      }
      memcpy(map->h_gn_pts, pts, (size_t)2 * n * sizeof(float));
      src = map->h_gn_pts;
      map->gn_host_n = n;
    }
    const int in_lds = (size_t)2 * n * sizeof(float) <= 56 * 1024;
    const size_t lds = in_lds ? (size_t)2 * std::max(n, 1) * sizeof(float) : 0;
    float* cache_dst = n > 0 ? map->d_cached.p : (float*)nullptr;
    const int ticket = ++map->gn_ticket;
#define LSLAM_GN_FAST(NT)                                                                                                  \
  launch(ctx, "gn_match", k_gn_match_fast<NT>, dim3(1), dim3(NT), lds, lv, src, cache_dst, n, in_lds, begin_world[0],      \
         begin_world[1], begin_world[2], map->h_gn_out, ticket)
#define LSLAM_GN_REG(NT, PMAX)                                                                                             \
  launch(ctx, "gn_match", k_gn_match_reg<NT, PMAX>, dim3(1), dim3(NT), 0, lv, src, cache_dst, n, begin_world[0],           \
         begin_world[1], begin_world[2], map->h_gn_out, ticket)
    // the points of the scan in registers when they fit (3 per thread at 512 threads: a 1081-beam scan), else LDS / memory
    if (map->gn_threads >= 1024 && n <= 1024 * 2) LSLAM_GN_REG(1024, 2);
    else if (map->gn_threads >= 512 && map->gn_threads < 1024 && n <= 512 * 3) LSLAM_GN_REG(512, 3);
    else if (map->gn_threads < 512 && n <= 256 * 5) LSLAM_GN_REG(256, 5);
    else if (map->gn_threads >= 1024) LSLAM_GN_FAST(1024);
    else if (map->gn_threads >= 512) LSLAM_GN_FAST(512);
    else LSLAM_GN_FAST(256);
#undef LSLAM_GN_REG
#undef LSLAM_GN_FAST
    LSLAM_HIP(ctx, hipGetLastError());
    // bounded spin on the ticket (acquire); the stream itself if it does not show up (a failed launch, a device fault)
    if (!spin_for_ticket((const int*)map->h_gn_out + 15, ticket)) LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 3; i++) out_pose[i] = map->h_gn_out[i];
    if (out_cov) for (int i = 0; i < 9; i++) out_cov[i] = map->h_gn_out[3 + i];
    return LSLAM_OK;
  }
  // ---- ordered sums: the reference's sequential fp32 accumulation, bit for bit -----------------------------------------
  // dataContainers[index-1].setFrom(dataContainer, ...) (MapRepMultiMap.h:161): the container is cached for the
  // next updateByScan -- also when it is empty
  if (n > 0)
    LSLAM_HIP(ctx, hipMemcpyAsync(map->d_cached.p, pts, (size_t)2 * n * sizeof(float),
                                  pts_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
  const size_t lds = (size_t)std::max(n, 1) * 9 * sizeof(float);
  if (lds > 64 * 1024)
    LSLAM_HIP(ctx, hipFuncSetAttribute((const void*)k_gn_match, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  launch(ctx, "gn_match", k_gn_match, dim3(1), dim3(n > 512 ? 1024 : 256), lds, lv, (const float*)map->d_cached.p, n,
         begin_world[0], begin_world[1], begin_world[2], d_out);
  float host[12];
  LSLAM_HIP(ctx, hipMemcpyAsync(host, d_out, sizeof host, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 3; i++) out_pose[i] = host[i];
  if (out_cov) for (int i = 0; i < 9; i++) out_cov[i] = host[3 + i];
  return LSLAM_OK;
}
}  // namespace

int lslam_map_match_data(lslam_map* map, const float* pts, int n, const float origo[2], const float begin_world[3],
                         float out_pose[3], float out_cov[9]) {
  if (!map || n < 0 || (n > 0 && !pts) || !begin_world || !out_pose) return LSLAM_ERR_INVALID_ARGUMENT;
  return match_data_impl(map, pts, n, false, origo, begin_world, out_pose, out_cov);
}

// matchData on the resident container of lslam_map_set_scan (the container is cached like any other)
int lslam_map_match_container(lslam_map* map, const float begin_world[3], float out_pose[3], float out_cov[9]) {
  if (!map || !begin_world || !out_pose) return LSLAM_ERR_INVALID_ARGUMENT;
  return match_data_impl(map, map->d_scan.p, map->n_scan, true, map->scan_origo, begin_world, out_pose, out_cov);
}

int lslam_map_set_option(lslam_map* map, int option, int value) {
  if (!map) return LSLAM_ERR_INVALID_ARGUMENT;
  if (option == LSLAM_MAP_OPT_ORDERED_SUMS) {
    map->ordered_sums = value != 0;
    return LSLAM_OK;
  }
  if (option == LSLAM_MAP_OPT_BATCH_SCRATCH_MB) {
    if (value < 1) return map->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "batch scratch budget must be >= 1 MB");
    map->batch_budget = (size_t)value << 20;
    return LSLAM_OK;
  }
  if (option == LSLAM_MAP_OPT_BATCH_RADIUS_CELLS) {
    if (value < 0) return map->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "batch radius hint must be >= 0 (0 = none)");
    map->batch_radius_hint = value;
    return LSLAM_OK;
  }
  return map->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "unknown map option %d", option);
}

int lslam_map_cached_points(const lslam_map* map) { return map ? map->n_cached : LSLAM_ERR_INVALID_ARGUMENT; }

int lslam_map_read_logodds(lslam_map* map, int level, float* out) {
  if (!map || !out || level < 0 || level >= (int)map->levels.size()) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  Level& L = map->levels[level];
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  {
    int rc = flush_pending(map);
    if (rc) return rc;
  }
  LSLAM_HIP(ctx, hipMemcpyAsync(out, L.d_logodds, (size_t)L.sx * L.sy * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_map_read_occupancy_i8(lslam_map* map, int level, int8_t* out) {
  if (!map || !out || level < 0 || level >= (int)map->levels.size()) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  Level& L = map->levels[level];
  size_t n = (size_t)L.sx * L.sy;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  {
    int rc = flush_pending(map);
    if (rc) return rc;
  }
  LSLAM_HIP(ctx, map->d_i8.reserve(n));
  launch(ctx, "occupancy_i8", k_occupancy_i8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
         (const float*)L.d_logodds, map->d_i8.p, n);
  LSLAM_HIP(ctx, hipMemcpyAsync(out, map->d_i8.p, n, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

void* lslam_map_cells_dev_ptr(lslam_map* map, int level) {
  if (!map || level < 0 || level >= (int)map->levels.size()) return nullptr;
  if (lslam_map_flush(map) != LSLAM_OK) return nullptr;  // a plane that lacks the last scan's apply is not handed out (lslam_last_error says why)
  return map->levels[level].d_logodds;
}

// Enqueue whatever a single-scan update still owes the float planes (the deferred apply of the pipelined path).
// lslam_map_read_*, lslam_map_match_*, the batched update and lslam_synchronize do this themselves; a caller that reads
// the plane behind lslam_map_cells_dev_ptr from its OWN stream calls it before recording its event on lslam_stream().
#if defined(LSLAM_PHASE_STAMPS)
// diagnostic builds only: out[kernel][8 cycle sums | 8 visit counts] of this translation unit's stamp table (summed over the
// wave slots); reset != 0 clears it
int lslam_debug_map_stamps(lslam_context* ctx, unsigned long long* out, int reset) {
  if (!ctx || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const size_t n = (size_t)lslam::kStampKernels * lslam::kStampSlots * 16;
  std::vector<unsigned long long> h(n);
  LSLAM_HIP(ctx, hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_map_stamps_slots), n * sizeof(unsigned long long)));
  for (int k = 0; k < lslam::kStampKernels; k++)
    for (int i = 0; i < 16; i++) {
      unsigned long long sum = 0;
      for (int sl = 0; sl < lslam::kStampSlots; sl++) sum += h[((size_t)k * lslam::kStampSlots + sl) * 16 + i];
      out[k * 16 + i] = sum;
    }
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    LSLAM_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_map_stamps_slots), h.data(), n * sizeof(unsigned long long)));
  }
  return LSLAM_OK;
}
#endif

int lslam_map_flush(lslam_map* map) {
  if (!map) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = map->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  return flush_pending(map);
}

}  // extern "C"
