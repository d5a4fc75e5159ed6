// Karto-style correlative scan matcher on MI355X (gfx950): HIP kernels + C ABI.
//
// Reference behaviour being reproduced (never copied): karto::ScanMatcher
// (lesson6/lib/open_karto/src/Mapper.cpp:126-856), karto::CorrelationGrid (Mapper.h:900-1118),
// karto::GridIndexLookup (Karto.h:6359-6555).  See DESIGN.md for the data layout and the
// roofline of each kernel.
//
// Device pipeline of one batch of S independent scans against the resident correlation grid
// (five launches; response-expansion passes add k_pass_setup + k_resp_rows + k_reduce_coarse each):
//   k_scan_prep      S*N threads   ranges -> scan-frame points (Karto.h:5384-5388, 6423-6434); its first
//                                  wave per scan lays out the coarse lattice + cos/sin of the angles
//                                  (pass_setup_wave, Mapper.cpp:339-393)
//   k_resp_rows      S*nA waves    ** hot kernel ** all nX*nY response sums of one (scan, angle) for a
//                                  uniform lattice (Mapper.cpp:373-424, 819-856); big batches gather from
//                                  the 2-D tiled parity planes (k_tile_planes)
//   k_reduce_coarse  S blocks      penalties, max, tie average, positional covariance; then the fine
//                                  lattice of the scan
//   k_resp_tile3     S*nA/3 waves  fine pass: one 12-byte load per (beam, angle) from overlapping 4x4 blocks
//                                  (k_tile4), three angles per wave from 1536 scans up (one below);
//                                  k_resp_rows<1,4> for small batches
//   k_reduce_fine    S blocks      same reductions + angular covariance (Mapper.cpp:431-506, 535-692)
//   k_resp_generic                 the same sums for arbitrary lattices; as block_generic_fallback it
//                                  runs inside the reduce kernels for scans with a non-uniform lattice
// Response numerators stay integers end to end (sum of uint8 <= 255*N); the fp64 part follows
// the reference's expression order (built with -ffp-contract=off).  Where a cheaper evaluation is
// used -- the fp32 estimate of a beam's table cell, the reciprocal form of the response
// normalisation -- it is one whose result is PROVEN (error band, exhaustive check) to be the
// reference's, with the reference's own expression as the fallback.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "common.hpp"
#include "karto_math.hpp"

using namespace lslam;

// A block's ordered sums run on ONE thread while its other waves wait at the next barrier: that wave's instructions are the
// block's critical path, and on a SIMD it shares with seven other waves it gets an eighth of the issue slots.  Raising its
// priority for the length of the section hands it the slots the waiting waves do not need (s_setprio: arbitration only).
#if defined(LSLAM_EXP_SERIAL_PRIO)
#define LSLAM_SERIAL_BEGIN() __builtin_amdgcn_s_setprio(3)
#define LSLAM_SERIAL_END() __builtin_amdgcn_s_setprio(0)
#else
#define LSLAM_SERIAL_BEGIN()
#define LSLAM_SERIAL_END()
#endif

namespace {

// diagnostic builds (-DLSLAM_PHASE_STAMPS, tools/phase_stamps.py): kernel ids of this translation unit's stamp table --
// 0 k_find_valid (base-scan blocks), 1 k_smear_gather, 2 k_anchor_chain, 3 reduce_coarse_lds_block, 4 reduce_fine_block,
// 5 resp_rows_wave (SOLO), 6 scan_prep_block
LSLAM_STAMP_TABLE(g_sm_stamps)
constexpr int kMaxLattice = 128;  // nX, nY per pass (loop-closure matcher: 101 x 101, Mapper.cpp:862-871)
constexpr int kMaxAngles = 128;   // nA per pass
constexpr int kMaxProbsSide = 255; // search-space probability grid side (loop closure: 201)
constexpr int kDenseMaxNx = 112;   // widest lattice row the dense kernel covers (7 lanes x 16 candidates)
constexpr int kMaxKernel = 33;    // widest smear kernel k_smear_gather tabulates in LDS (wider: the listed scatter path)
constexpr int kGuard = 256;       // zero bytes before/after the grid: unaligned row loads may overhang

struct Geom {
  int width, height, stride, border, roi_w, roi_h, data_size, kernel_size;
  int n_beams, probs_side;
  double scale;         // 1/resolution (Mapper.h:1020)
  double off_x, off_y;  // CoordinateConverter offset of the correlation grid
  double min_angle, ang_res;
};

struct PassCfg {
  double off_x, off_y;  // rSearchSpaceOffset
  double res_x, res_y;  // rSearchSpaceResolution
  double ang_off, ang_res;
  int nx, ny, na;
  int mode;  // 0 coarse, 1 coarse expansion, 2 fine
};

struct SearchCfg {
  double dvp, avp, min_dp, min_ap;
  int do_penalize;
  // GetResponse's normalisation (Mapper.cpp:852) is response = sum / (nBeams * 100) in fp64.  The numerator is an integer in
  // [0, nBeams * 100], so whether q = sum * inv, q' = fma(fma(-q, d, sum), inv, q) (inv = RN(1/d); Markstein's correction
  // step) returns the correctly rounded quotient can be -- and is, at lslam_matcher_create -- checked for EVERY possible
  // numerator: fast_div is set only if all of them agree with the division bit for bit (three instructions instead of the
  // dozen, one of them a quarter-rate v_rcp_f64, of an IEEE division; 2 541 of them per scan in the coarse reduce).
  int fast_div;
  double inv_denom;
};
__device__ __forceinline__ double response_of_sum(int32_t sum, int n_beams, const SearchCfg& sc) {
  const double d = (double)((uint32_t)n_beams * (uint32_t)kOccupied), a = (double)sum;
  if (sc.fast_div) {
    const double q = a * sc.inv_denom;
    return __builtin_fma(__builtin_fma(-q, d, a), sc.inv_denom, q);
  }
  return a / d;
}

constexpr int kTilePad = 512;  // zero bytes in front of the tiled parity planes (a dead row j reads offset 32 j)
constexpr int kTileYOff = 32;  // tiled planes: class rows start at grid row y = -kTileYOff (>= 2*16 - 1 + 1)
constexpr int kRowZero = 64;  // k_resp_rows: row-load offset 0 = 64 zero guard bytes in front of plane 0
constexpr int kMaxBeamsPerLane = 32;  // k_resp_rows: 8 lanes x 32 beams x 255 < 2^16 (packed DPP reduce)
constexpr int kOccMinScans = 8;      // smaller batches do not rebuild the row-occupancy bitmap for themselves
constexpr int kTileMinWaves = 2048;  // below this the fine pass stays on k_resp_rows (beam slices fill the chip)
constexpr int kPipeMinChunk = 256;   // lslam_matcher_match_batch, pipelined: no sub-batch smaller than this
constexpr int kMaxGridSide = 32768;  // widthStep and height: dataSize <= 2^30, flat indices stay int32

struct Lattice {  // per scan, per pass
  double center[3];
  int gx[kMaxLattice], gy[kMaxLattice];  // full-grid cell coordinates (ROI offset included)
  int step_x, step_y;                    // uniform cell step, 0 if not uniform
  int status, active;
};

struct CoarseOut {
  double mean[3];
  double cov[9];
  double best;
  int status;
  int expand;  // response expansion requested for the next pass (Mapper.cpp:242-244)
  int flags;
  int pad;
};

// ------------------------------------------------------------------------------------------
// k_scan_prep: ranges -> world point (LocalizedRangeScan::Update, Karto.h:5384-5388) -> point in
// the scan frame (Transform::InverseTransformPose, Karto.h:6426-6434).  lx = NaN marks INVALID_SCAN
// (reading NaN/Inf, Karto.h:6478-6483).  One thread per (scan, beam); coalesced.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void pass_setup_wave(int s, int lane, const Geom& g, const PassCfg& pc,
                                                const double* center, int active, Lattice* lat,
                                                double2* cossin, int want_step);  // below

struct PoseArg {
  double v[3];
};
// One 256-thread block of the prep: block bx of nbx shares the beams of scan s, posed at (sx, sy, sh).
template <typename RT>
__device__ __forceinline__ void scan_prep_block(const RT* __restrict__ ranges, int stride, double sx, double sy, double sh,
                                                const Geom& g, double2* __restrict__ local, double2* __restrict__ world,
                                                const PassCfg& setup_pc, Lattice* setup_lat, double2* setup_cossin,
                                                int setup_step, int bx, int nbx, int s) {
  const int nt = (int)blockDim.x;  // 256 (k_scan_prep) or 1024 (riding in k_find_valid's launch)
  int b = bx * nt + threadIdx.x;
  LSLAM_PHASE_CLOCK(pck);
  // the scan's own transform (two rotation matrices, one normalised heading) is the same for all of
  // its beams: one thread of the block evaluates it
  __shared__ SensorXform s_t;
  __shared__ double s_h;
  if (local && threadIdx.x == 0) {
    LSLAM_SERIAL_BEGIN();
    s_t = sensor_xform(sx, sy, sh);
    // rSourcePose - m_Transform (Pose2 operator-, Karto.h:2138-2141), heading = normalize(0 - th)
    s_h = normalize_angle(0.0 - s_t.th);
    LSLAM_SERIAL_END();
  }
  // Meanwhile the LAST wave of the scan's first block lays out the coarse search lattice (k_pass_setup, mode 0), and
  // every thread evaluates its first world point: neither needs the transform thread 0 is working on.
  if (setup_lat && bx == 0 && (int)threadIdx.x >= nt - 64) {
    const double center[3] = {sx, sy, sh};
    pass_setup_wave(s, (int)threadIdx.x - (nt - 64), g, setup_pc, center, 1, setup_lat, setup_cossin, setup_step);
  }
  double r = 0.0, px = 0.0, py = 0.0;
  if (b < g.n_beams) {
    r = (double)ranges[(size_t)s * stride + b];
    beam_world_point(sx, sy, sh, g.min_angle, g.ang_res, (uint32_t)b, r, px, py);
  }
  LSLAM_PHASE_MARK(pck, 0);  // first world point (all) | the scan's transform (thread 0) | lattice + cos/sin (last wave)
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 1);  // waiting for the slowest of the three
  // nbx blocks share the beams of a scan: ONE for chip-filling batches (the single-thread transform above is
  // then paid once per scan, not once per 256 beams -- it was most of this kernel's time), ceil(n/256) otherwise
  for (bool first = true; b < g.n_beams; b += nbx * nt, first = false) {
    if (!first) {
      r = (double)ranges[(size_t)s * stride + b];
      beam_world_point(sx, sy, sh, g.min_angle, g.ang_res, (uint32_t)b, r, px, py);
    }
    size_t o = (size_t)s * g.n_beams + b;
    if (world) world[o] = make_double2(px, py);
    if (local) {
      double lx, ly;
      if (isnan(r) || isinf(r)) {
        lx = ly = __builtin_nan("");
      } else {
        rot_apply(s_t.inv, px - s_t.tx, py - s_t.ty, s_h, lx, ly);
      }
      local[o] = make_double2(lx, ly);
    }
  }
  LSLAM_PHASE_MARK(pck, 2);  // scan-frame points (+ further beams)
  LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 6, (unsigned)(blockIdx.x * (nt >> 6) + (threadIdx.x >> 6)), (threadIdx.x & 63) == 0);
}

template <typename RT>
__global__ void __launch_bounds__(256)
k_scan_prep(const RT* __restrict__ ranges, int stride, const double* __restrict__ poses, Geom g,
            double2* __restrict__ local, double2* __restrict__ world, PassCfg setup_pc, Lattice* setup_lat,
            double2* setup_cossin, int setup_step, PoseArg pose_val) {
  const int s = blockIdx.y;
  // poses == nullptr: ONE scan whose pose is the kernel argument (no copy on the stream in front of the kernel)
  const double sx = poses ? poses[3 * s] : pose_val.v[0], sy = poses ? poses[3 * s + 1] : pose_val.v[1],
               sh = poses ? poses[3 * s + 2] : pose_val.v[2];
  scan_prep_block(ranges, stride, sx, sy, sh, g, local, world, setup_pc, setup_lat, setup_cossin, setup_step, (int)blockIdx.x,
                  (int)gridDim.x, s);
}

// ------------------------------------------------------------------------------------------
// k_pass_setup: candidate-lattice cell coordinates of one CorrelateScan pass
// (Mapper.cpp:339-386): each lattice coordinate is rounded on its own, exactly like the reference
// (the fine pass is centred on a possibly off-lattice tie average).
// ------------------------------------------------------------------------------------------
// one wave (64 lanes) sets up scan s: lane i owns lattice coordinates i, i+64 and candidate angles i, i+64
__device__ __forceinline__ void pass_setup_wave(int s, int lane, const Geom& g, const PassCfg& pc,
                                                const double* center, int active, Lattice* lat,
                                                double2* cossin, int want_step) {
  const double start_x = -pc.off_x, start_y = -pc.off_y;
  auto cell_x = [&](int i) {
    double x = start_x + (uint32_t)i * pc.res_x;
    double npx = center[0] + x;
    return world_to_grid(npx, g.off_x, g.scale) + g.border;  // CorrelationGrid::GridIndex adds the ROI
  };
  auto cell_y = [&](int j) {
    double y = start_y + (uint32_t)j * pc.res_y;
    double npy = center[1] + y;
    return world_to_grid(npy, g.off_y, g.scale) + g.border;
  };
  const int stx0 = pc.nx > 1 ? cell_x(1) - cell_x(0) : want_step;
  const int sty0 = pc.ny > 1 ? cell_y(1) - cell_y(0) : want_step;
  bool out_of_range = false, uneven_x = false, uneven_y = false;
  Lattice& L = lat[s];
  for (int i = lane; i < kMaxLattice; i += 64) {
    int cx = 0, cy = 0;
    if (i < pc.nx) {
      cx = cell_x(i);
      out_of_range |= cx < 0 || cx >= g.width;
      if (i >= 1) uneven_x |= cx - cell_x(i - 1) != stx0;
    }
    if (i < pc.ny) {
      cy = cell_y(i);
      out_of_range |= cy < 0 || cy >= g.height;
      if (i >= 1) uneven_y |= cy - cell_y(i - 1) != sty0;
    }
    L.gx[i] = cx;
    L.gy[i] = cy;
  }
  // cos/sin of every candidate angle, computed once here instead of once per response wave
  // (Mapper.cpp:390-393, Karto.h:6465-6466)
  for (int a = lane; a < pc.na; a += 64) {
    const double angle = (center[2] - pc.ang_off) + (uint32_t)a * pc.ang_res;
    cossin[(size_t)s * kMaxAngles + a] = make_double2(cos(angle), sin(angle));
  }
  const int status = __any(out_of_range) ? LSLAM_ERR_INDEX_OUT_OF_RANGE : 0;
  const int stx = __any(uneven_x) ? 0 : stx0, sty = __any(uneven_y) ? 0 : sty0;
  if (lane == 0) {
    L.center[0] = center[0]; L.center[1] = center[1]; L.center[2] = center[2];
    L.step_x = stx;
    L.step_y = sty;
    L.status = status;
    L.active = active;
  }
}

__global__ void __launch_bounds__(64)
k_pass_setup(int S, Geom g, PassCfg pc, const double* __restrict__ poses, const CoarseOut* __restrict__ coarse,
             Lattice* __restrict__ lat, double2* __restrict__ cossin, int want_step) {
  const int s = blockIdx.x;
  if (s >= S) return;
  double center[3];
  int active = 1;
  if (pc.mode == 0) {
    center[0] = poses[3 * s]; center[1] = poses[3 * s + 1]; center[2] = poses[3 * s + 2];
  } else if (pc.mode == 1) {
    center[0] = poses[3 * s]; center[1] = poses[3 * s + 1]; center[2] = poses[3 * s + 2];
    active = coarse[s].expand && coarse[s].status == 0;
  } else {
    center[0] = coarse[s].mean[0]; center[1] = coarse[s].mean[1]; center[2] = coarse[s].mean[2];
    active = coarse[s].status == 0;
  }
  pass_setup_wave(s, threadIdx.x, g, pc, center, active, lat, cossin, want_step);
}

// ------------------------------------------------------------------------------------------
// k_deinterleave: F_q[m] = G[2m+q].  On the coarse lattice (2-cell steps) the candidates of one
// beam sit on every second byte of a grid row; splitting the grid by the parity of the FLAT index
// (widthStep is even, so this is also the x parity) makes them CONTIGUOUS bytes of F_q, and the
// reference's 1-D bounds rule idx in [0,dataSize) becomes m in [0,dataSize/2).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_deinterleave(const uint8_t* __restrict__ grid, uint8_t* __restrict__ f0, uint8_t* __restrict__ f1, int n8) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n8) return;
  uint2 v = ((const uint2*)grid)[t];
  // bytes b0..b7 -> even: b0 b2 b4 b6, odd: b1 b3 b5 b7
  uint32_t e = (v.x & 0xFFu) | ((v.x >> 8) & 0xFF00u) | ((v.y & 0xFFu) << 16) | ((v.y << 8) & 0xFF000000u);
  uint32_t o = ((v.x >> 8) & 0xFFu) | ((v.x >> 16) & 0xFF00u) | ((v.y << 8) & 0xFF0000u) | (v.y & 0xFF000000u);
  ((uint32_t*)f0)[t] = e;
  ((uint32_t*)f1)[t] = o;
}

// ------------------------------------------------------------------------------------------
// k_tile_planes: the parity planes again, laid out for the gather unit.  A vector-memory instruction
// costs about one cycle per distinct 128-byte line its lanes touch (tools/micro/ta_rate.hip), and in
// the linear planes the 64 beams of a row load sit on ~40 different lines: neighbouring beams of a
// wall that is not parallel to x fall into different grid rows.  Here the planes are split by row
// parity as well (lattice rows are 2 grid rows apart) and cut into vertical STRIPS 32 bytes wide that
// step 16 bytes in x, each strip stored row after row: a 128-byte line is a 2-D patch of 4
// lattice-consecutive rows x 32 bytes, every dword-aligned 16-byte row segment lies inside one
// strip (2x the plane bytes), and the rows of one beam are 32 bytes apart -- an immediate offset.
// Defined on the FLAT plane index like the planes themselves:
//   T[q][ry][tx][Y'][c] = F_q[((2 Y' + ry) - kTileYOff) * widthStep/2 + 16 tx + c],   c in [0, 32),
// zero outside [0, dataSize/2); c may run past the row end = the next row (flat wrap).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_tile_planes(const uint8_t* __restrict__ grid, int stride, int data_size, uint32_t* __restrict__ tiles,
              int tile_tx, int tile_rows) {
  const size_t d = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // output dword
  const size_t per_class = (size_t)tile_tx * tile_rows * 8;
  if (d >= 4 * per_class) return;
  const int cls = (int)(d / per_class);  // q * 2 + ry
  const size_t rem = d - (size_t)cls * per_class;
  const int c4 = (int)(rem & 7);
  const size_t t = rem >> 3;
  const int row = (int)(t % tile_rows), tx = (int)(t / tile_rows);
  const int q = cls >> 1, ry = cls & 1;
  const long long y = (long long)(2 * row + ry) - kTileYOff;
  const long long m = y * (stride / 2) + 16 * tx + 4 * c4;
  uint32_t v = 0;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const long long idx = 2 * (m + k) + q;
    if (m + k >= 0 && idx < data_size) v |= (uint32_t)grid[idx] << (8 * k);
  }
  tiles[kTilePad / 4 + d] = v;
}

// ------------------------------------------------------------------------------------------
// k_row_occupancy: exact skip mask for the row loads of k_resp_rows.  Only ~0.4 % of the grid is
// non-zero, so most candidate rows of most beams sum zeros.  bit(x, y) = any G_flat[f .. f+kOccWin-1]
// != 0 for f = x + y*widthStep (flat index, out-of-range bytes count as zero, y from -1).
// Stored TRANSPOSED (column-major, 32 consecutive y per word) so one 8-byte read gives a beam the
// bits of all its lattice rows (rows are 1 or 2 grid rows apart, same column).  A row whose bit is
// clear contributes nothing to any candidate -> skipping its load is exact, not an approximation.
// ------------------------------------------------------------------------------------------
constexpr int kOccWinMax = 63;  // widest row span one bit can summarise (window mask is 64-bit)

// pass A: flat non-zero bitmap, one bit per grid byte (thread per 32 bytes, coalesced)
__global__ void __launch_bounds__(256)
k_nonzero_bits(const uint8_t* __restrict__ grid, int data_size, uint32_t* __restrict__ nz, int n_words) {
  int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_words) return;
  uint32_t bits = 0;
  const long long f0 = (long long)w * 32;
  if (f0 + 32 <= data_size) {
    const uint4* p = (const uint4*)(grid + f0);  // grid base is 16-byte aligned (kGuard = 256)
    uint4 a = p[0], b = p[1];
    const uint32_t v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      uint32_t d = v[k];
      bits |= ((d & 0xFFu) ? 1u : 0u) << (4 * k);
      bits |= ((d & 0xFF00u) ? 1u : 0u) << (4 * k + 1);
      bits |= ((d & 0xFF0000u) ? 1u : 0u) << (4 * k + 2);
      bits |= ((d & 0xFF000000u) ? 1u : 0u) << (4 * k + 3);
    }
  } else {
    for (int k = 0; k < 32; k++)
      if (f0 + k < data_size && grid[f0 + k] != 0) bits |= 1u << k;
  }
  nz[w] = bits;
}

// pass B: bit(x, y) = any of the kOccWin flat bits starting at f = x + y*widthStep, written transposed
// and SPLIT BY ROW PARITY: column x holds two bitmaps, bit h of bitmap E <-> row y = 2h + E - 1, so
// the rows y0, y0+2, ... of the coarse lattice are consecutive bits (no bit gather in k_resp_rows).
__global__ void __launch_bounds__(256)
k_row_occupancy(const uint32_t* __restrict__ nz, int n_words, int stride, int height, int win,
                uint32_t* __restrict__ occ_t, int words_per_half) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;  // column (flat index mod widthStep)
  int wq = blockIdx.y;                            // word index: [parity][word along y]
  if (x >= stride || wq >= 2 * words_per_half) return;
  const int E = wq / words_per_half, w2 = wq - E * words_per_half;
  const unsigned long long wmask = (1ull << win) - 1ull;
  uint32_t bits = 0;
  for (int b = 0; b < 32; b++) {
    int y = 2 * (w2 * 32 + b) + E - 1;  // bit 0 of bitmap 0 is y = -1
    if (y > height) break;
    long long f = (long long)x + (long long)y * stride;  // may be negative for y = -1
    long long w0 = f >> 5;                                // floor
    int sh = (int)(f & 31);
    // 96-bit window of flat bits starting at word w0 (out-of-range words are zero)
    unsigned long long lo = 0, hi = 0;
    if (w0 >= 0 && w0 < n_words) lo = nz[w0];
    if (w0 + 1 >= 0 && w0 + 1 < n_words) lo |= (unsigned long long)nz[w0 + 1] << 32;
    if (w0 + 2 >= 0 && w0 + 2 < n_words) hi = nz[w0 + 2];
    unsigned long long window = (lo >> sh) | (sh ? (hi << (64 - sh)) : 0ull);
    if (window & wmask) bits |= 1u << b;
  }
  occ_t[(size_t)x * (2 * words_per_half) + wq] = bits;
}

// pass C: the same bits regrouped for the gather unit: P[w][x][E] = words (w, w+1) of column x's
// parity-E bitmap as one 64-bit value.  Column-major words put the 64 beams of a wave on 64 different
// 128-byte lines (one line = 1024 rows of ONE column); x-major pairs put neighbouring beams --
// neighbouring columns, either row parity -- on the same line, and the (w, w+1) overlap keeps it one 8-byte load.
__global__ void __launch_bounds__(256)
k_occ_pairs(const uint32_t* __restrict__ occ_t, int stride, int words_per_half, uint2* __restrict__ pairs) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int ew = blockIdx.y;  // E * words_per_half + w (the layout of occ_t's columns)
  if (x >= stride) return;
  const int w = ew % words_per_half, E = ew / words_per_half;
  const uint32_t* col = occ_t + (size_t)x * (2 * words_per_half) + ew;
  // [w][x][E]: the two row parities of a column side by side, columns next to each other -- neighbouring beams (neighbouring
  // columns, either parity) find their words in ONE 128-byte line; with the parities in separate planes they sat in two
  // (measured: 288 M -> 276 M L1 accesses per 4096-scan launch of k_resp_rows)
  pairs[((size_t)w * stride + x) * 2 + E] = make_uint2(col[0], w + 1 < words_per_half ? col[1] : 0u);
}

// wave64 sum, uniform result: four DPP adds give every lane its row-of-16 total, the four row totals
// are read back as scalars
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true);  // row_mirror
  return (uint32_t)__builtin_amdgcn_readlane((int)v, 0) + (uint32_t)__builtin_amdgcn_readlane((int)v, 16) +
         (uint32_t)__builtin_amdgcn_readlane((int)v, 32) + (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
}

// ------------------------------------------------------------------------------------------
// k_resp_rows -- THE HOT KERNEL.
// Exact response numerators (Mapper.cpp:819-856) of every candidate position of a uniform
// lattice for one (scan, angle): one wave64 per (scan, angle[, beam slice]).
//   STEP 2 (coarse pass): sources are the parity planes F_0/F_1 of the grid, a lattice row is nX
//                         contiguous bytes of F_(base&1) starting at base>>1;
//   STEP 1 (fine pass)  : the source is the grid itself.
// Phase A (every beam; lane = beam, two blocks of 64 beams in flight per iteration): the lookup-table
// cell on the fly (no table round trip through HBM; cos/sin of the angle come from k_pass_setup) --
// decided on an fp32 estimate with a proven error band, the few beams inside the band of a rounding
// boundary re-done on the reference's own fp64 expression tree --, the flat index in int32, the rows
// inside the reference's 1-D index range, and -- coarse pass -- the exact row-occupancy bits; a beam
// with at least one live row is pushed on a circular LDS queue (wave ballot compaction).
// Phase B (every 64 queued beams; lane = beam): one dword-aligned 16-byte load per lattice row
// (a dead row reads the zero guard band instead, so the loads are straight-line), realigned with
// v_perm_b32 into FOUR candidates per dword in packed 16-bit fields.
// Epilogue: three packed DPP adds reduce over groups of 8 lanes, 2 KB of LDS brings the 8 partials of
// each packed word to one lane, which writes exact int32 sums angle-major (coalesced).
// No MFMA: this is a gather/compare path over an L2-resident 4 MB grid.
// Block -> (scan, angle) mapping keeps all angles of a scan on one XCD (block b runs on XCD b%8),
// so a scan's 17 KB of scan-frame points is fetched into ONE L2 instead of eight.
// ------------------------------------------------------------------------------------------
// STATS = true is the instrumented twin used for ONE untimed launch by lslam_matcher_read_stats (bench.py's
// pruned_row_fraction): it counts, per launch, the lattice rows inside the reference's index range, the rows
// still live after the exact row-occupancy pruning, the readable beams and the beams queued for phase B.
// LDSB = true is the MEASURED-AND-DROPPED variant the north star's wording asks about ("grid pyramid staged in LDS
// tiles"): phase B stages, per drain of 64 queued beams, the bounding patch of their rows from the
// linear parity planes in LDS (one patch per parity) and reads the row words with ds_read instead of global gathers;
// drains whose patch does not fit 6 KB per parity take the global path.  Same bytes, same sums.  DESIGN_HISTORY.md (B, "LDS-staged experiment") has the numbers.
constexpr int kPatchDw = 1536;  // dwords per parity patch
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
  v = min(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));
  v = min(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));
  v = min(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true));
  v = min(v, (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true));
  return min(min((uint32_t)__builtin_amdgcn_readlane((int)v, 0), (uint32_t)__builtin_amdgcn_readlane((int)v, 16)),
             min((uint32_t)__builtin_amdgcn_readlane((int)v, 32), (uint32_t)__builtin_amdgcn_readlane((int)v, 48)));
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return ~wave_min_u32(~v); }
// the first anchor of FindValidPoints (Mapper.cpp:774-778): the lowest index whose point has no NaN.  One LDS atomic per
// WAVE that has such a point (its lowest lane: lanes hold ascending indices) -- an atomic per point on one address is
// served lane by lane
__device__ __forceinline__ void first_valid_min(int* s_first, bool valid, int i) {
  const unsigned long long m = __ballot(valid);
  if (m && (int)(threadIdx.x & 63) == __ffsll((long long)m) - 1) atomicMin(s_first, i);
}

constexpr int kRowsQueue = 256;  // >= 63 waiting + two blocks of 64 coming in
// LDS of ONE wave of the row kernel.  SOLO (k_resp_rows, a block = a wave): four arrays of their own.  In the scan-resident
// step kernel (k_match_step: several waves per block, each on its own angle) a wave's area is 2.4 KB: `red` (epilogue) and
// `queue` (phases A / B) are never live together and share their bytes, and the beams the fp32 estimate cannot decide are
// remembered as one 64-bit lane mask per block of beams (`ambm`) instead of a list.
struct RowsLds {
  uint32_t (*red)[8];            // [packed word][group of 8 lanes]
  int2* queue;                   // circular; .x = first row index m0, .y = row mask | parity << 31
  uint16_t* ambq;                // SOLO: beams whose fp32 estimate could not decide the rounding (phase A)
  uint32_t (*patch)[kPatchDw];   // LDSB: the drain's bounding patch of each parity plane
  unsigned long long* ambm;      // !SOLO: [kMaxBeamsPerLane] lane masks of the undecided beams
};
// barrier between a wave's own LDS writes and reads: the whole block when the block IS the wave, else the wave alone
// (LDS operations of one wave are performed in order; the fence keeps the compiler from reordering them)
template <bool SOLO>
__device__ __forceinline__ void wsync() {
  if constexpr (SOLO) {
    __syncthreads();
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// One wave64: all nX*nY numerators of (scan s, angle a[, beam slice]).  SOLO: written to resp (global, angle-major; beam
// slices accumulate with atomics); !SOLO: stored to lds_out[a * ncand + ...] (the block's numerators live in LDS).
template <int NXD, int NYC, bool TILED, bool STATS, bool LDSB, bool SOLO, bool OUT_LDS = !SOLO>
__device__ __forceinline__ void resp_rows_wave(
    const int s, const int a, const int slice, const uint8_t* __restrict__ src0, const uint8_t* __restrict__ src1, int step,
    int limit, const Geom& g, const PassCfg& pc, const Lattice* __restrict__ lat, const double2* __restrict__ cossin,
    const double2* __restrict__ local, int32_t* __restrict__ resp, size_t resp_stride, int beam_slices,
    const uint2* __restrict__ occ_t, int occ_wpc, int tile_rows, uint32_t tile_class_bytes,
    unsigned long long* __restrict__ stats, const RowsLds lds, int32_t* lds_out) {
  constexpr int NW = NXD * NYC * 2;
  constexpr bool EST = TILED;  // phase A on the fp32 estimate, two blocks in flight: the chip-filling batches (see phase A)
  constexpr int kQueue = kRowsQueue;
  static_assert(NYC <= 16 && 2 * (NYC - 1) < 32, "row mask / occupancy window width");
  static_assert(SOLO || (!STATS && !LDSB), "multi-wave blocks run the production variants only");
  static_assert(!(SOLO && OUT_LDS), "a lone wave writes its numerators to global memory");
  uint32_t (*const red)[8] = lds.red;
  int2* const queue = lds.queue;
  [[maybe_unused]] uint16_t* const ambq = lds.ambq;
  [[maybe_unused]] uint32_t (*const patch)[kPatchDw] = lds.patch;
  [[maybe_unused]] unsigned long long* const ambm = lds.ambm;
  LSLAM_PHASE_CLOCK(pck);
  const Lattice& L = lat[s];
  if (!L.active || L.status != 0 || L.step_x != step || L.step_y != step) return;

  const double2 cs = cossin[(size_t)s * kMaxAngles + a];  // of (center - ang_off) + a * ang_res (k_pass_setup)
  const double cosine = cs.x, sine = cs.y;
  // cos * scale and sin * scale of the fp32 estimate of phase A, pinned to SGPRs: the packed-fp32 forms the vectoriser
  // prefers would keep them in VGPRs, which this kernel (66 accumulators in a 128-VGPR budget) does not have
  auto uniform = [](float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); };
  const float cos_sc = uniform((float)cosine * (float)g.scale), sin_sc = uniform((float)sine * (float)g.scale);
  const int X0 = L.gx[0], Y0 = L.gy[0];
  const double2* lp = local + (size_t)s * g.n_beams;
  const int ncand = pc.nx * pc.ny;
  const int shift = step == 2 ? 1 : 0;
  const int bstride = 64 * beam_slices;
  // row loads address zbase + 32-bit offset: offset 0 is the zero guard band in front of plane 0
  // (TILED: src0 is the tiled-plane buffer itself, whose first kTilePad bytes are zero)
  const uint8_t* zbase = TILED ? src0 : src0 - kRowZero;
  const uint32_t plane_delta = TILED ? 0u : (uint32_t)(src1 - src0);
  const uint32_t strip_bytes = (uint32_t)tile_rows * 32u;  // one 32-byte-wide strip, all class rows
  [[maybe_unused]] const int occ_wph = occ_wpc >> 1;  // words per (column, row parity)

  LSLAM_PHASE_MARK(pck, 0);  // lattice record, cos/sin of the angle
  for (int j0 = 0; j0 < pc.ny; j0 += NYC) {
    // PEEL (the chip-filling variants): the accumulators are NOT zeroed -- phase A runs in two stages: until the first 64
    // beams are queued nothing is drained (the accumulators are not even live), the FIRST drain then WRITES them
    // (drain_as<true>: perm results stored, not added) and the second stage adds as before.  66 v_mov and 66 v_add fewer
    // per wave than "zero, then always add" (profiles/r05/inst_budget_resp_rows.md).
#if defined(LSLAM_EXP_NO_PEEL)  // A/B switch (tools/ab_variants.py): round 4's form
    constexpr bool PEEL = false;
#else
    constexpr bool PEEL = EST;
#endif
    uint32_t acc[NYC][NXD][2];
    if constexpr (!PEEL) {
#pragma unroll
      for (int j = 0; j < NYC; j++)
#pragma unroll
        for (int k = 0; k < NXD; k++) acc[j][k][0] = acc[j][k][1] = 0u;
    }
    // The lane index as this iteration sees it, opaque to the optimiser: everything derived from it (offsets of the first
    // point loads, LDS addresses of the epilogue) is then computed where it is used instead of being hoisted out of this
    // loop into registers that the kernel -- 66 accumulators in a 128-VGPR budget -- would have to spill to scratch.
    // (re-read from the hardware -- mbcnt over a full mask = the lane's index, the block being one wave -- so that not even
    //  the index itself has to stay in a register across the iteration)
    uint32_t all_lanes = ~0u;
    asm volatile("" : "+s"(all_lanes));  // (pins the re-read inside the iteration)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(all_lanes, __builtin_amdgcn_mbcnt_lo(all_lanes, 0u));

    // Phase B: one queued beam per lane -- load the rows its mask names, accumulate 4 candidates per dword
    auto drain_as = [&](auto first_drain, int head, int cnt) {
      constexpr bool FIRST = decltype(first_drain)::value;
      const int2 e = lane < cnt ? queue[(head + lane) & (kQueue - 1)] : make_int2(0, 0);
      // dword-ALIGNED loads of NXD+1 words covering the row, realigned in registers: the planes and
      // widthStep are multiples of 4, so every row of a beam has the same byte phase
      const uint32_t sh = (uint32_t)e.x & 3u;
      // v_perm_b32 selectors: bytes sh+0 / sh+2 (even candidates) and sh+1 / sh+3 (odd) of the word pair,
      // each zero-extended into a 16-bit field (0x0C selects the constant 0)
      const uint32_t sel_e = 0x0C020C00u + sh * 0x00010001u;
      const uint32_t sel_o = 0x0C030C01u + sh * 0x00010001u;
      uint32_t wv[NYC][NXD + 1];
      if constexpr (TILED) {
        // e.x = X' | parity << 15 | (y + kTileYOff) << 16 (k_tile_planes): lattice row j is class row
        // (yy >> 1) + j of class yy & 1, 32 bytes after row j - 1 in the beam's strip
        const uint32_t xa = (uint32_t)e.x & 0x7FFCu, yy = (uint32_t)e.x >> 16;
        const uint32_t base_off = (uint32_t)kTilePad + (((uint32_t)e.x >> 15) & 1u) * 2u * tile_class_bytes +
                                  (yy & 1u) * tile_class_bytes + (xa >> 4) * strip_bytes + (yy >> 1) * 32u +
                                  (xa & 15u);
#pragma unroll
        for (int j = 0; j < NYC; j++) {
          // a dead row reads the zero pad (offset 32 j < kTilePad); the row step is an immediate offset
          const uint32_t off = base_off & (uint32_t)__builtin_amdgcn_sbfe(e.y, j, 1);
          __builtin_memcpy(wv[j], __builtin_assume_aligned(zbase + (size_t)off + 32 * j, 4), 4 * (NXD + 1));
        }
      } else {
        bool staged = false;
        if constexpr (LDSB) {
          // plane coordinates of the beam's first row word: pitch g.stride (one lattice row = one pitch)
          const uint32_t rowmask = (uint32_t)e.y & ((1u << NYC) - 1u);
          const bool on = lane < cnt && rowmask != 0u && e.x >= 0;
          const uint32_t cur0 = ((uint32_t)e.x & ~3u) + (uint32_t)kRowZero;
          const uint32_t ry = cur0 / (uint32_t)g.stride, cx = cur0 - ry * (uint32_t)g.stride;
          const bool p1 = e.y < 0;
          const bool bad = lane < cnt && rowmask != 0u && e.x < 0;  // a beam hanging over the plane's start: global path
          uint32_t W[2], Hh[2], cmin[2], rmin[2];
          bool fit = !__any(bad);
#pragma unroll
          for (int q = 0; q < 2; q++) {
            const bool mine = on && (p1 == (q == 1));
            cmin[q] = wave_min_u32(mine ? cx : 0xFFFFFFFFu);
            rmin[q] = wave_min_u32(mine ? ry : 0xFFFFFFFFu);
            const uint32_t cmax = wave_max_u32(mine ? cx + 4u * (NXD + 1) : 0u), rmax = wave_max_u32(mine ? ry + NYC : 0u);
            W[q] = cmax > cmin[q] ? (cmax - cmin[q]) >> 2 : 0u;
            Hh[q] = rmax > rmin[q] ? rmax - rmin[q] : 0u;
            fit = fit && W[q] * Hh[q] <= (uint32_t)kPatchDw && W[q] < 65536u && Hh[q] < 65536u;
          }
          if constexpr (STATS)
            if (lane == 0 && stats) atomicAdd(&stats[fit ? 5 : 6], 1ull);
          if (fit) {
            staged = true;
            const uint32_t limit_bytes = (uint32_t)limit + (uint32_t)kRowZero;  // rows past the plane's end are zeros
#pragma unroll
            for (int q = 0; q < 2; q++) {
              const uint32_t total = W[q] * Hh[q];
              if (total == 0u) continue;
              const uint32_t inv = 0xFFFFFFFFu / W[q] + 1u;  // idx / W for idx < 2^16 by one multiply
              for (uint32_t idx = (uint32_t)lane; idx < total; idx += 64u) {
                const uint32_t r = __umulhi(idx, inv), c = idx - r * W[q];
                const uint32_t off = (rmin[q] + r) * (uint32_t)g.stride + cmin[q] + 4u * c;
                uint32_t v = 0u;
                if (off + 4u <= limit_bytes + 64u) v = *(const uint32_t*)(zbase + (size_t)off + (q ? plane_delta : 0u));
                patch[q][idx] = v;
              }
            }
            wsync<SOLO>();
            const int q = p1 ? 1 : 0;
            const uint32_t base = on ? (ry - rmin[q]) * W[q] + ((cx - cmin[q]) >> 2) : 0u;
#pragma unroll
            for (int j = 0; j < NYC; j++) {
              const bool live = on && ((rowmask >> j) & 1u);
#pragma unroll
              for (int k = 0; k <= NXD; k++) wv[j][k] = live ? patch[q][base + (uint32_t)j * W[q] + (uint32_t)k] : 0u;
            }
            wsync<SOLO>();  // the next drain restages the patches
          }
        }
        if (!staged) {
          uint32_t cur = ((uint32_t)e.x & ~3u) + (uint32_t)kRowZero + (e.y < 0 ? plane_delta : 0u);
#pragma unroll
          for (int j = 0; j < NYC; j++) {
            // a masked row reads the zeros at offset 0 instead: straight-line loads beat exec-masked ones
            const uint32_t off = cur & (uint32_t)__builtin_amdgcn_sbfe(e.y, j, 1);
            __builtin_memcpy(wv[j], __builtin_assume_aligned(zbase + off, 4), 4 * (NXD + 1));
            cur += (uint32_t)g.stride;
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NYC; j++)
#pragma unroll
        for (int k = 0; k < NXD; k++) {
          const uint32_t pe = __builtin_amdgcn_perm(wv[j][k + 1], wv[j][k], sel_e);  // candidates 4k, 4k+2
          const uint32_t po = __builtin_amdgcn_perm(wv[j][k + 1], wv[j][k], sel_o);  // candidates 4k+1, 4k+3
          if constexpr (FIRST) {
            acc[j][k][0] = pe;
            acc[j][k][1] = po;
          } else {
            acc[j][k][0] += pe;
            acc[j][k][1] += po;
          }
        }
    };
    auto drain = [&](int head, int cnt) { drain_as(std::false_type{}, head, cnt); };

    // Phase A: every beam -- table entry, row mask (bounds + exact row occupancy); survivors are queued.
    // The lattice lies inside the grid (k_pass_setup) and the grid has <= 2^30 cells, so for a beam whose
    // table cell is within +-2^15 cells every index below is exact in int32 (24-bit multiplies, unsigned
    // range compares); any other beam takes the 64-bit path.
    int qcount = 0, qhead = 0;
    uint32_t st_rows = 0, st_live = 0, st_beams = 0, st_queued = 0;  // STATS only
    const int rows_here = min(NYC, pc.ny - j0);
    const uint32_t all_rows = (1u << rows_here) - 1u;
    const int B0 = X0 + Y0 * g.stride + j0 * step * g.stride;
    const int Yb1 = Y0 + j0 * step + 1;
    const int m0_max = limit - ((rows_here - 1) * g.stride + 4 * NXD);  // whole neighbourhood in range
    const int y1_max = g.height + 1 - step * (NYC - 1);                 // y+1 range of the occupancy window
    // The table cell of a beam -- ComputeOffsets + WorldToGrid (Karto.h:6465-6494, 4237-4252) -- is decided in two steps:
    //  * every beam on an fp32 ESTIMATE (x, y) -> (cos*scale x - sin*scale y, sin*scale x + cos*scale y).  With u = 2^-24,
    //    the roundings of cos*scale and sin*scale (three each), of the point, the product and the fma leave the estimate
    //    within 6 u scale (|x| + |y|) cells of the exact value of the reference's expression, whose own fp64 roundings stay
    //    below 1e-9 cells; a coordinate farther than 10 u scale (|x| + |y|) + 1e-6 from a half-integer therefore rounds
    //    the way the reference's does.  A beam with a coordinate inside the band (a few in 10^4), or beyond +-32000 cells
    //    (the int32 form of the flat index does not cover it), is parked in `ambq`;
    //  * the parked beams, 64 at a time, on the reference's own fp64 expression tree.
    // The sums are integer, so the order in which beams reach the accumulators does not matter.
    // `emit`: flat index, rows in range, exact row occupancy, queue push, drain -- the part both steps share.  It takes TWO
    // blocks of 64 beams at a time: a block is a chain of dependent latencies (point -> cell -> occupancy word -> ballot ->
    // queue), and two chains in flight per wave keep the SIMD fed where four waves of one chain each did not.
    struct Cell {
      uint32_t mask, par, col, osh;
      int m0i;
      bool have_occ;
    };
    auto cell_of = [&](bool valid, bool small, int gx, int gy) {
      // Straight-line for the common case (every lane, also the ones without a beam: their (gx, gy) is harmless and their
      // mask is cleared at the end); only the rare cases branch.
      Cell c;
      const int base = B0 + gx + __mul24(gy, g.stride);  // Karto.h:6494 + Mapper.cpp:838; |gx|, |gy| < 2^15 (`small`)
      c.par = (uint32_t)(base & shift);
      const int m0 = base >> shift;  // arithmetic shift = floor
      // rows inside the valid index range (the reference's 1-D check, Mapper.cpp:841-845)
      uint32_t mask = all_rows;
      if (!(m0_max >= 0 && (uint32_t)m0 <= (uint32_t)m0_max)) {
        mask = 0u;
        for (int j = 0; j < rows_here; j++) {
          const int rs = m0 + j * g.stride;
          if (rs >= -(4 * NXD) && rs < limit) mask |= 1u << j;
        }
      }
      int x = X0 + gx, y1 = Yb1 + gy;
      if (occ_t || TILED) {
        if ((uint32_t)x >= (uint32_t)g.stride) {  // flat index wrapped into a neighbouring row
          const int y = base >= 0 ? base / g.stride : -((-base + g.stride - 1) / g.stride);
          x = base - y * g.stride;
          y1 = y + 1;
        }
      }
      // exact row occupancy (step 2): lattice rows are consecutive bits of one parity
      c.have_occ = occ_t != nullptr && (uint32_t)y1 <= (uint32_t)y1_max;
      c.col = c.have_occ ? (((uint32_t)__mul24(y1 >> 6, g.stride) + (uint32_t)x) << 1) + ((uint32_t)y1 & 1u) : 0u;  // k_occ_pairs
      c.osh = ((uint32_t)y1 >> 1) & 31u;
      // a live row has y + 2j >= -1, so y >= -(2 NYC - 1) > -kTileYOff whenever the mask is not empty
      c.m0i = TILED ? (int)(((uint32_t)x >> 1) | (c.par << 15) | ((uint32_t)(y1 + kTileYOff - 1) << 16)) : m0;
      c.mask = valid ? mask : 0u;
      if (valid && !small) {  // a table cell beyond +-2^15: 64-bit flat index, no occupancy pruning
        const int t = gx + gy * g.stride;  // int32 like the reference
        const long long base64 = (long long)B0 + t;
        c.par = (uint32_t)((int)(base64 & 1) & shift);
        const long long m064 = base64 >> shift;
        c.mask = 0u;
        for (int j = 0; j < rows_here; j++) {
          long long rs = m064 + (long long)j * g.stride;
          if (rs >= -(long long)(4 * NXD) && rs < (long long)limit) c.mask |= 1u << j;
        }
        if constexpr (TILED) {
          const long long yl = base64 >= 0 ? base64 / g.stride : -((-base64 + g.stride - 1) / g.stride);
          const long long xl = base64 - yl * g.stride;
          c.m0i = c.mask ? (int)(((uint32_t)xl >> 1) | (c.par << 15) | ((uint32_t)(yl + kTileYOff) << 16)) : 0;
        } else {
          c.m0i = (int)m064;
        }
        c.have_occ = false;
        c.col = 0u;
      }
      return c;
    };
    auto enqueue = [&](auto two_blocks, int bA, bool validA, bool smallA, int gxA, int gyA, int bB, bool validB, bool smallB,
                       int gxB, int gyB) {
      constexpr bool TWO = decltype(two_blocks)::value;  // false: block B does not exist (single-block callers)
      Cell cA = cell_of(validA, smallA, gxA, gyA), cB{0u, 0u, 0u, 0u, 0, false};
      if constexpr (TWO) cB = cell_of(validB, smallB, gxB, gyB);
      if constexpr (STATS) st_rows += (uint32_t)__popc(cA.mask) + (uint32_t)__popc(cB.mask);
      if (occ_t) {
        // x-major: neighbouring beams read neighbouring words (k_occ_pairs); both words are in flight together
        // (32-bit byte offsets from a uniform base: the scalar-base form of the load, no 64-bit address pairs in VGPRs)
        uint32_t offA = cA.col << 3, offB = cB.col << 3;
        asm("" : "+v"(offA));  // (keeps the compiler from widening the offsets to 64 bits inside cell_of's branches, which
        if constexpr (TWO) asm("" : "+v"(offB));  //  costs a register pair this kernel has to spill)
        const uint2 owA = *(const uint2*)((const char*)occ_t + offA);
        const uint32_t keepA = __builtin_amdgcn_alignbit(owA.y, owA.x, cA.osh);  // bit j <-> lattice row j0 + j
        cA.mask &= cA.have_occ ? keepA : 0xFFFFFFFFu;
        if constexpr (TWO) {
          const uint2 owB = *(const uint2*)((const char*)occ_t + offB);
          const uint32_t keepB = __builtin_amdgcn_alignbit(owB.y, owB.x, cB.osh);
          cB.mask &= cB.have_occ ? keepB : 0xFFFFFFFFu;
        }
      }
      if constexpr (STATS) {
        st_live += (uint32_t)__popc(cA.mask) + (uint32_t)__popc(cB.mask);
        st_queued += (cA.mask ? 1u : 0u) + (cB.mask ? 1u : 0u);
        // per (scan, beam), OR-ed over the scan's angles: bit 0 = readable, bit 1 = some angle has a live row for it
        // (the flag words sit behind the eight counters; stats[4] = scans the buffer was sized for)
        if (stats && (unsigned long long)s < stats[4]) {
          if (validA) atomicOr((uint32_t*)(stats + 8) + (size_t)s * g.n_beams + bA, cA.mask ? 3u : 1u);
          if (TWO && validB) atomicOr((uint32_t*)(stats + 8) + (size_t)s * g.n_beams + bB, cB.mask ? 3u : 1u);
        }
      }
      const unsigned long long votesA = __ballot(cA.mask != 0);
      if (cA.mask) {
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(votesA >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)votesA, 0u));
        queue[(qhead + qcount + rank) & (kQueue - 1)] = make_int2(cA.m0i, (int)(cA.mask | (cA.par << 31)));
      }
      qcount += __popcll(votesA);
      if constexpr (TWO) {
        const unsigned long long votesB = __ballot(cB.mask != 0);
        if (cB.mask) {
          const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(votesB >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)votesB, 0u));
          queue[(qhead + qcount + rank) & (kQueue - 1)] = make_int2(cB.m0i, (int)(cB.mask | (cB.par << 31)));
        }
        qcount += __popcll(votesB);
      }
      wsync<SOLO>();
    };
    auto drain_ready = [&](auto two_blocks) {
      constexpr bool TWO = decltype(two_blocks)::value;
      if (qcount >= 64) {
        drain(qhead, 64);
        qhead = (qhead + 64) & (kQueue - 1);
        qcount -= 64;
      }
      if constexpr (TWO) {  // at most 63 were waiting and at most 128 came in
        if (qcount >= 64) {
          drain(qhead, 64);
          qhead = (qhead + 64) & (kQueue - 1);
          qcount -= 64;
        }
      }
    };
    auto emit = [&](auto two_blocks, int bA, bool validA, bool smallA, int gxA, int gyA, int bB, bool validB, bool smallB,
                    int gxB, int gyB) {
      enqueue(two_blocks, bA, validA, smallA, gxA, gyA, bB, validB, smallB, gxB, gyB);
      drain_ready(two_blocks);
    };
    // one block of beams on the estimate: cell, and whether the estimate decides it -- otherwise the beam is parked
    int acount = 0;
    [[maybe_unused]] uint32_t amb_its = 0u;  // !SOLO: blocks of 64 beams that parked at least one (bit = block index `it`)
    auto estimate = [&](int it, float2 p, bool in_scan, bool& valid, int& gx, int& gy) {
      float t1, t2, fx, fy;
      asm("v_mul_f32 %0, %1, %2" : "=v"(t1) : "s"(sin_sc), "v"(p.y));
      asm("v_mul_f32 %0, %1, %2" : "=v"(t2) : "s"(cos_sc), "v"(p.y));
      asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(fx) : "s"(cos_sc), "v"(p.x), "v"(t1));
      asm("v_fma_f32 %0, %1, %2, %3" : "=v"(fy) : "s"(sin_sc), "v"(p.x), "v"(t2));
      const float rx = __builtin_rintf(fx), ry = __builtin_rintf(fy);
      // the band in cells: scale (|x| + |y|) <= sqrt(2) (|fx| + |fy|), so 10 u sqrt(2) (|fx| + |fy|) + 1e-6 covers E
      const float reach = fabsf(fx) + fabsf(fy);
      const float lim = (0.5f - 1e-6f) - reach * (14.2f / 16777216.0f);
      // NaN = INVALID_SCAN (k_scan_prep writes both coordinates): every comparison below is false for it
      valid = (int)in_scan & (int)(fabsf(fx - rx) < lim) & (int)(fabsf(fy - ry) < lim) & (int)(reach < 32000.0f);
      const unsigned long long unsure = __ballot(!valid);
      if (unsure) {
        const bool park = !valid && in_scan && reach == reach;  // a beam the estimate does not decide
        if constexpr (STATS) st_beams += (valid || park) ? 1u : 0u;
        const unsigned long long parked = __ballot(park);
        if constexpr (SOLO) {
          if (park) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(parked >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)parked, 0u));
            ambq[acount + rank] = (uint16_t)(it * 64 + lane);  // (block of the slice, lane): < 64 kMaxBeamsPerLane
          }
          acount += __popcll(parked);
        } else if (parked) {  // the block's lane mask; `amb_its` (uniform) remembers which blocks have one
          if (lane == 0) ambm[it] = parked;
          amb_its |= 1u << it;
        }
      } else {
        if constexpr (STATS) st_beams += 1u;
      }
      gx = (int)rx;
      gy = (int)ry;
    };
    auto point_f = [&](int bb) {  // converted as it arrives: two registers in flight per block, not four
      uint32_t off = (uint32_t)min(bb, g.n_beams - 1) << 4;
      asm volatile("" : "+v"(off));  // (a 32-bit offset from the scalar base, computed where it is used -- not an address
                                     //  pair, or an offset, hoisted out of the j0 loop into registers this kernel has to spill)
      const double2 t = *(const double2*)((const char*)lp + off);
      return make_float2((float)t.x, (float)t.y);
    };
    // the reference's own expression tree: identical fp64 operations; (int)math::Round(v) taken as trunc(copysign(|v| + 0.5, v))
    auto exact_cell = [&](double2 p, int& gx, int& gy, bool& small) {
      const double ox = cosine * p.x - sine * p.y, oy = sine * p.x + cosine * p.y;
      const double vx = ((ox + g.off_x) - g.off_x) * g.scale, vy = ((oy + g.off_y) - g.off_y) * g.scale;
      const double ax = fabs(vx) + 0.5, ay = fabs(vy) + 0.5;
      gx = (int)copysign(ax, vx);
      gy = (int)copysign(ay, vy);
      small = fmax(ax, ay) < 32768.0;  // |gx|, |gy| < 2^15
    };
    if constexpr (EST) {
      float2 pA_next = point_f(64 * slice + lane), pB_next = point_f(64 * slice + bstride + lane);
      int b0 = 64 * slice, it = 0;
      // one iteration of phase A up to the queue push (two blocks of 64 beams)
      auto two_blocks_in = [&]() {
        const int bA = b0 + lane, bB = bA + bstride;
        const float2 pA = pA_next, pB = pB_next;  // fetched an iteration ahead: the latency hides behind this one's arithmetic
        pA_next = point_f(bA + 2 * bstride);
        pB_next = point_f(bB + 2 * bstride);
        bool validA, validB;
        int gxA, gyA, gxB, gyB;
        estimate(it, pA, bA < g.n_beams, validA, gxA, gyA);
        estimate(it + 1, pB, bB < g.n_beams, validB, gxB, gyB);
        enqueue(std::true_type{}, bA, validA, true, gxA, gyA, bB, validB, true, gxB, gyB);
        b0 += 2 * bstride;
        it += 2;
      };
      // stage 1: nothing to drain yet -- the accumulators do not exist
      if constexpr (PEEL) {
        while (b0 < g.n_beams && qcount < 64) two_blocks_in();
        // the first drain writes them (a partial one when the scan ran out first; zeros when nothing is queued)
        const int c = min(qcount, 64);
        if (c > 0) {
          drain_as(std::true_type{}, qhead, c);
          qhead = (qhead + c) & (kQueue - 1);
          qcount -= c;
        } else {
#pragma unroll
          for (int j = 0; j < NYC; j++)
#pragma unroll
            for (int k = 0; k < NXD; k++) acc[j][k][0] = acc[j][k][1] = 0u;
        }
        if (qcount >= 64) {  // the iteration that crossed 64 may have brought in up to 128
          drain(qhead, 64);
          qhead = (qhead + 64) & (kQueue - 1);
          qcount -= 64;
        }
      }
      // stage 2: as before
      while (b0 < g.n_beams) {
        two_blocks_in();
        drain_ready(std::true_type{});
      }
      if constexpr (SOLO) {
        for (int a0 = 0; a0 < acount; a0 += 64) {
          const bool valid = a0 + lane < acount;
          const int parked_at = valid ? (int)ambq[a0 + lane] : 0;
          const int b = 64 * slice + (parked_at >> 6) * bstride + (parked_at & 63);
          int gx = 0, gy = 0;
          bool small = true;
          if (valid) exact_cell(*(const double2*)((const char*)lp + ((uint32_t)b << 4)), gx, gy, small);
          emit(std::false_type{}, b, valid, small, gx, gy, 0, false, true, 0, 0);
        }
      } else {
        wsync<SOLO>();  // lane 0's mask words
        while (amb_its) {  // one pass per block of beams that parked any (a few beams in 10^4 park at all)
          const int pit = __builtin_ctz(amb_its);
          amb_its &= amb_its - 1u;
          const bool valid = (ambm[pit] >> lane) & 1ull;
          const int b = 64 * slice + pit * bstride + lane;
          int gx = 0, gy = 0;
          bool small = true;
          if (valid) exact_cell(*(const double2*)((const char*)lp + ((uint32_t)b << 4)), gx, gy, small);
          emit(std::false_type{}, b, valid, small, gx, gy, 0, false, true, 0, 0);
        }
      }
    } else {
      // Launches that do not fill the chip (a lone MatchScan is 21 x 8 waves) end when their SLOWEST wave does: a parked
      // beam -- a dependent load and one more pass through `emit` on some wave of nearly every launch -- costs them more
      // than the estimate saves (measured: 11.9 -> 13.8 us for the coarse pass of one scan), so they evaluate every beam
      // on the fp64 tree, one block per iteration.
      double2 p_next = lp[min(64 * slice + lane, g.n_beams - 1)];
      for (int b0 = 64 * slice; b0 < g.n_beams; b0 += bstride) {
        const int b = b0 + lane;
        const double2 p = p_next;  // fetched one block ahead
        p_next = lp[min(b + bstride, g.n_beams - 1)];
        // NaN = INVALID_SCAN (k_scan_prep writes both coordinates); testing both keeps the point ONE 16-byte load
        const bool valid = (int)(b < g.n_beams) & (int)!isnan(p.x) & (int)!isnan(p.y);
        if constexpr (STATS) st_beams += valid ? 1u : 0u;
        int gx = 0, gy = 0;
        bool small = true;
        if (valid) exact_cell(p, gx, gy, small);
        emit(std::false_type{}, b, valid, small, gx, gy, 0, false, true, 0, 0);
      }
    }
    if (qcount > 0) drain(qhead, qcount);
    wsync<SOLO>();
    LSLAM_PHASE_MARK(pck, 1);  // phases A + B: every beam's cell, the queued beams' rows
    if constexpr (STATS) {
      const uint32_t t0 = wave_sum(st_rows), t1 = wave_sum(st_live), t2 = wave_sum(st_beams), t3 = wave_sum(st_queued);
      if (lane == 0 && stats) {
        atomicAdd(&stats[0], (unsigned long long)t0);
        atomicAdd(&stats[1], (unsigned long long)t1);
        atomicAdd(&stats[2], (unsigned long long)t2);
        atomicAdd(&stats[3], (unsigned long long)t3);
      }
    }

    // Reduce over the wave.  A lane saw at most kMaxBeamsPerLane beams (the host slices longer scans),
    // so three packed DPP adds -- lanes xor 1, xor 2, then the mirrored quad -- leave every group of 8
    // lanes holding its 16-bit field sums without overflow; the 8 group partials of each packed word
    // go through 2 KB of LDS to one lane, which unpacks them into exact int32 sums.
#pragma unroll
    for (int j = 0; j < NYC; j++)
#pragma unroll
      for (int k = 0; k < NXD; k++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
          uint32_t v = acc[j][k][q];
          v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
          v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
          // row_half_mirror; written as the fused DPP add the compiler emits for the two steps above but not for this one
          // (it used v_mov_b32_dpp + v_add_u32: one VALU instruction more per accumulator, 66 per wave)
#if defined(LSLAM_EXP_NO_FUSED_DPP)  // A/B switch: round 4's form
          v += (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true);
#else
          asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v));
#endif
          if ((lane & 7) == 0) red[(j * NXD + k) * 2 + q][lane >> 3] = v;
        }
    wsync<SOLO>();
    for (int idx = lane; idx < NW; idx += 64) {
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const uint32_t v = red[idx][k];
        lo += v & 0xFFFFu;
        hi += v >> 16;
      }
      const int par2 = idx & 1, jk = idx >> 1;
      const int j = j0 + jk / NXD, i = 4 * (jk % NXD) + par2;
      if (j < pc.ny) {
        if constexpr (OUT_LDS) {
          int32_t* o = lds_out + a * ncand + j * pc.nx + i;
          if (i < pc.nx) o[0] = (int32_t)lo;
          if (i + 2 < pc.nx) o[2] = (int32_t)hi;
        } else {
          int32_t* o = resp + (size_t)s * resp_stride + (size_t)a * ncand + (size_t)j * pc.nx + i;
          if (beam_slices == 1) {
            if (i < pc.nx) o[0] = (int32_t)lo;
            if (i + 2 < pc.nx) o[2] = (int32_t)hi;
          } else {
            if (i < pc.nx) atomicAdd(o, (int32_t)lo);
            if (i + 2 < pc.nx) atomicAdd(o + 2, (int32_t)hi);
          }
        }
      }
    }
    wsync<SOLO>();
    LSLAM_PHASE_MARK(pck, 2);  // wave reduction + stores
  }
  if constexpr (SOLO) LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 5, (unsigned)blockIdx.x, (threadIdx.x & 63) == 0);
}

// The row kernel proper: one wave64 per block.
template <int NXD, int NYC, bool TILED, bool STATS = false, bool LDSB = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8)))  // <= 128 VGPRs
k_resp_rows(const uint8_t* __restrict__ src0, const uint8_t* __restrict__ src1, int step, int limit, Geom g,
            PassCfg pc, const Lattice* __restrict__ lat, const double2* __restrict__ cossin,
            const double2* __restrict__ local, int32_t* __restrict__ resp, size_t resp_stride, int beam_slices,
            int S, const uint2* __restrict__ occ_t, int occ_wpc, int tile_rows, uint32_t tile_class_bytes,
            unsigned long long* __restrict__ stats) {
  constexpr int NW = NXD * NYC * 2;
  __shared__ __align__(16) uint32_t red[NW][8];
  __shared__ int2 queue[kRowsQueue];
  __shared__ uint16_t ambq[TILED ? 64 * kMaxBeamsPerLane : 1];
  __shared__ uint32_t patch[LDSB ? 2 : 1][LDSB ? kPatchDw : 1];
  int w = blockIdx.x;
  const int slice = w % beam_slices;
  w /= beam_slices;
  const int xcd = w & 7, r = w >> 3;
  const int s = (r / pc.na) * 8 + xcd;
  const int a = r % pc.na;
  if (s >= S) return;
  resp_rows_wave<NXD, NYC, TILED, STATS, LDSB, true>(s, a, slice, src0, src1, step, limit, g, pc, lat, cossin, local, resp,
                                                     resp_stride, beam_slices, occ_t, occ_wpc, tile_rows, tile_class_bytes,
                                                     stats, RowsLds{red, queue, ambq, (uint32_t (*)[kPatchDw])patch, nullptr},
                                                     nullptr);
}

// The same waves in blocks of WAVES: the waves of a block take WAVES consecutive angles of ONE scan, so they run on ONE CU
// and share its L1 -- the angles of a scan read the same 17 KB of points, neighbouring occupancy words and largely the
// same lines of the tiled planes (a 2-degree step moves a beam's end point by 7 cells at 10 m; a line is a 64 x 8 cell
// patch).  k_resp_rows' one-wave blocks of a scan are dealt round robin over the CUs of the scan's XCD and share only
// the L2: 17 % of its L1 lookups miss, and a miss costs the CU's gather pipe 2.3 clocks against 0.5 for a hit
// (profiles/r06/micro_ta_rate.txt).  No barrier anywhere: every wave has its own LDS area and its own angle.
template <int NXD>
constexpr int rows_wave_area() {  // bytes of LDS per wave: max(red, queue) + the undecided-beam masks
  return (NXD == 3 ? 3 * 11 * 2 * 8 * 4 : 4 * 8 * 2 * 8 * 4) > kRowsQueue * 8
             ? (NXD == 3 ? 3 * 11 * 2 * 8 * 4 : 4 * 8 * 2 * 8 * 4) + kMaxBeamsPerLane * 8
             : kRowsQueue * 8 + kMaxBeamsPerLane * 8;
}
template <int NXD, int NYC, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_resp_rows_mw(const uint8_t* __restrict__ ptiles, int limit, Geom g, PassCfg pc, const Lattice* __restrict__ lat,
               const double2* __restrict__ cossin, const double2* __restrict__ local, int32_t* __restrict__ resp,
               size_t resp_stride, int S, const uint2* __restrict__ occ_t, int occ_wpc, int tile_rows,
               uint32_t tile_class_bytes) {
  constexpr int kArea = rows_wave_area<NXD>();
  __shared__ __align__(16) unsigned char areas[WAVES * kArea];
  const int wave = (int)threadIdx.x >> 6;
  const int groups = (pc.na + WAVES - 1) / WAVES;
  const int w = blockIdx.x;
  const int xcd = w & 7, r = w >> 3;
  const int s = (r / groups) * 8 + xcd;  // all blocks of a scan on one XCD
  const int a = (r % groups) * WAVES + wave;
  if (s >= S || a >= pc.na) return;
  unsigned char* wa = areas + wave * kArea;
  const RowsLds lds{(uint32_t (*)[8])wa, (int2*)wa, nullptr, nullptr, (unsigned long long*)(wa + (kArea - kMaxBeamsPerLane * 8))};
  resp_rows_wave<NXD, NYC, true, false, false, false, false>(s, a, 0, ptiles, ptiles, 2, limit, g, pc, lat, cossin, local, resp,
                                                             resp_stride, 1, occ_t, occ_wpc, tile_rows, tile_class_bytes,
                                                             (unsigned long long*)nullptr, lds, (int32_t*)nullptr);
}

// ------------------------------------------------------------------------------------------
// Fine pass on 4x4 cell blocks.  The fine lattice of MatchScan is always 3x3 cells at 1-cell steps
// (Mapper.cpp:276-281: offset = coarse resolution / 2 = one cell), so a beam's nine candidates are a
// 3x3 patch of the grid.  k_tile4 stores, for every EVEN (X, Y), the 16 bytes
//     T[Y/2][X/2][r][c] = G_flat[(Y + r) * widthStep + X + c]   (0 outside [0, dataSize))
// -- overlapping 4x4 blocks, 4x the grid in HBM, built once per grid change -- so the patch with
// top-left cell (x, y) lies inside block (x & ~1, y & ~1) and ONE aligned 16-byte load per beam
// replaces three row loads.  Blocks are defined on the FLAT index, like the reference's 1-D bounds
// rule (Mapper.cpp:841-845): an x that runs past widthStep continues in the next row.
// ------------------------------------------------------------------------------------------
// Block (ux, uy) -> slot: a 128-byte line holds 4 x 2 blocks = 8 x 4 cells (+ the blocks' own overlap), so
// neighbouring beams share lines whichever way the wall runs (gathers cost ~1 cycle per distinct line).
__host__ __device__ __forceinline__ size_t tile4_slot(int ux, int uy, int cols4) {
  return ((size_t)((uy >> 1) * cols4 + (ux >> 2)) << 3) | (size_t)(((uy & 1) << 2) | (ux & 3));
}
constexpr int kTileYPad = 4;  // blocks start at Y = -4: flat indices of row y = -3 can wrap into row 0

__global__ void __launch_bounds__(256)
k_tile4(const uint8_t* __restrict__ grid, int stride, int data_size, uint4* __restrict__ tiles, int tile_cols,
        int tile_rows) {
  const int ux = blockIdx.x * blockDim.x + threadIdx.x, uy = blockIdx.y;
  if (ux >= tile_cols || uy >= tile_rows) return;
  const long long idx0 = (long long)(2 * uy - kTileYPad) * stride + 2 * ux;  // even: pairs are in or out together
  uint32_t w[4];
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const long long i = idx0 + (long long)r * stride;
    uint32_t lo = 0, hi = 0;
    if (i >= 0 && i < data_size) lo = *reinterpret_cast<const uint16_t*>(grid + i);
    if (i + 2 >= 0 && i + 2 < data_size) hi = *reinterpret_cast<const uint16_t*>(grid + i + 2);
    w[r] = lo | (hi << 16);
  }
  tiles[tile4_slot(ux, uy, (tile_cols + 3) / 4)] = make_uint4(w[0], w[1], w[2], w[3]);
}

// One wave64 per (scan, kTile3Angles angles); lanes stride over the beams.  Same exact numerators as
// k_resp_rows<1,4> at step 1 (Mapper.cpp:819-856), written resp[s][a][j*3+i].  Several angles per wave in the chip-filling
// batches because the one-angle form runs at the access rate of the L1s (PMC: 29 accesses per vector-memory instruction,
// gather unit 85 % busy; the coalesced 16-byte-per-lane point load is charged about as much as the block gather): the
// angles share the point load, and their block gathers are independent loads in flight together.  Measured per 4096-scan launch: 1 angle per wave
// 119 us, 2: 96, 3: 92, 4: 95; below ~1000 scans one angle per wave (more waves) is the faster form (256 scans: 14 vs 22 us).
#if !defined(LSLAM_TUNE_TILE3_MIN)
#define LSLAM_TUNE_TILE3_MIN 1536
#endif
constexpr int kTile3ManyAngles = 3, kTile3ManyMinScans = LSLAM_TUNE_TILE3_MIN;
// One wave64: the 9 numerators of each of the angles a0 .. a0 + kTile3Angles - 1 of scan s, written to r_scan[a * 9 + c]
// (r_scan = the scan's numerators: resp + s * resp_stride in k_resp_tile3, the block's LDS copy in k_match_step).
template <int kTile3Angles>
__device__ __forceinline__ void resp_tile3_wave(const int s, const int a0, const int lane, const uint4* __restrict__ tiles,
                                                int tile_cols, const Geom& g, const PassCfg& pc,
                                                const Lattice* __restrict__ lat, const double2* __restrict__ cossin,
                                                const double2* __restrict__ local, int32_t* r_scan) {
  const Lattice& L = lat[s];
  if (!L.active || L.status != 0 || L.step_x != 1 || L.step_y != 1) return;

  double cosine[kTile3Angles], sine[kTile3Angles];
  float cos_sc[kTile3Angles], sin_sc[kTile3Angles];  // of the fp32 estimate
  bool angle_on[kTile3Angles];
#pragma unroll
  for (int q = 0; q < kTile3Angles; q++) {
    angle_on[q] = a0 + q < pc.na;
    const double2 cs = cossin[(size_t)s * kMaxAngles + min(a0 + q, pc.na - 1)];  // of (center - ang_off) + a * ang_res (k_pass_setup)
    cosine[q] = cs.x, sine[q] = cs.y;
    cos_sc[q] = (float)cs.x * (float)g.scale, sin_sc[q] = (float)cs.y * (float)g.scale;
  }
  const int X0 = L.gx[0], Y0 = L.gy[0];
  const int B0 = X0 + Y0 * g.stride;
  const int cols4 = (tile_cols + 3) / 4;
  const double2* lp = local + (size_t)s * g.n_beams;
  // packed 16-bit fields: e[j] = candidates (0,j) | (2,j) << 16; o01 = (1,0) | (1,1) << 16; o2 = (1,2)
  uint32_t e0[kTile3Angles], e1[kTile3Angles], e2[kTile3Angles], o01[kTile3Angles], o2[kTile3Angles];
#pragma unroll
  for (int q = 0; q < kTile3Angles; q++) e0[q] = e1[q] = e2[q] = o01[q] = o2[q] = 0u;
  // the 4x4 block of one beam at one angle: byte offset of its 12 bytes (0 = nothing to read: the buffer starts with
  // block rows of the pad, which hold zeros) and the column parity inside the block
  auto block_of = [&](const double2& p, float x32, float y32, int q, uint32_t& off, uint32_t& dx) {
    // The table cell as k_resp_rows evaluates it: decided on the fp32 estimate wherever the estimate's error band
    // (10 u sqrt(2) (|fx| + |fy|) + 1e-6 cells, derived in k_resp_rows' phase A) leaves no doubt about the rounding;
    // the few beams inside the band take the reference's own fp64 expression tree (lookup_cell_i32's;
    // (int)math::Round(v) taken as trunc(copysign(|v| + 0.5, v))).
    const float fx = __builtin_fmaf(cos_sc[q], x32, -(sin_sc[q] * y32)), fy = __builtin_fmaf(sin_sc[q], x32, cos_sc[q] * y32);
    const float rx = __builtin_rintf(fx), ry = __builtin_rintf(fy);
    const float reach = fabsf(fx) + fabsf(fy);
    const float lim = (0.5f - 1e-6f) - reach * (14.2f / 16777216.0f);
    int gx = (int)rx, gy = (int)ry;
    bool small = true;
    double vx = 0.0, vy = 0.0;
    if (!((int)(fabsf(fx - rx) < lim) & (int)(fabsf(fy - ry) < lim) & (int)(reach < 32000.0f))) {
      const double ox = cosine[q] * p.x - sine[q] * p.y, oy = sine[q] * p.x + cosine[q] * p.y;
      vx = ((ox + g.off_x) - g.off_x) * g.scale, vy = ((oy + g.off_y) - g.off_y) * g.scale;
      const double ax = fabs(vx) + 0.5, ay = fabs(vy) + 0.5;
      gx = (int)copysign(ax, vx), gy = (int)copysign(ay, vy);
      small = fmax(ax, ay) < 32768.0;
    }
    int x, y;
    bool ok;
    if (small) {  // |gx|, |gy| < 2^15: all int32-exact, see k_resp_rows
      x = X0 + gx, y = Y0 + gy;
      if ((uint32_t)x >= (uint32_t)g.stride) {  // flat index wrapped into a neighbouring row
        const int base = B0 + gx + __mul24(gy, g.stride);
        y = base >= 0 ? base / g.stride : -((-base + g.stride - 1) / g.stride);
        x = base - y * g.stride;
      }
      ok = y >= -3 && y < g.height;
    } else {
      const int gxl = kround_i32(vx), gyl = kround_i32(vy);
      const long long base = (long long)B0 + (int)(gxl + gyl * g.stride);  // int32 table offset like the reference
      const long long yl = base >= 0 ? base / g.stride : -((-base + g.stride - 1) / g.stride);
      ok = yl >= -3 && yl < g.height;
      y = ok ? (int)yl : 0;
      x = ok ? (int)(base - yl * g.stride) : 0;
    }
    // rows y .. y+2 of the patch are 12 contiguous bytes of the block, 4 bytes in when y is odd: one 12-byte load
    // (32-bit offset from the buffer base: the host keeps the tiled copy below 4 GB)
    off = (uint32_t)tile4_slot(x >> 1, (y + kTileYPad) >> 1, cols4) * 16u + ((uint32_t)y & 1u) * 4u;
    dx = (uint32_t)x & 1u;
    return ok;
  };
  double2 p = lp[min(lane, g.n_beams - 1)];
  for (int b = lane; b < g.n_beams; b += 64) {
    const double2 pn = lp[min(b + 64, g.n_beams - 1)];  // next point in flight while this one is used
    if (!isnan(p.x)) {  // NaN = INVALID_SCAN
      const float x32 = (float)p.x, y32 = (float)p.y;
      uint32_t off[kTile3Angles], dx[kTile3Angles], rr[kTile3Angles][3];
      bool ok[kTile3Angles];
#pragma unroll
      for (int q = 0; q < kTile3Angles; q++) ok[q] = block_of(p, x32, y32, q, off[q], dx[q]) && angle_on[q];
#pragma unroll
      for (int q = 0; q < kTile3Angles; q++) {
        rr[q][0] = rr[q][1] = rr[q][2] = 0u;
        if (ok[q]) __builtin_memcpy(rr[q], __builtin_assume_aligned((const uint8_t*)tiles + off[q], 4), 12);
      }
#pragma unroll
      for (int q = 0; q < kTile3Angles; q++) {
        const uint32_t r0 = rr[q][0], r1 = rr[q][1], r2 = rr[q][2];
        const uint32_t sel_e = 0x0C020C00u + dx[q] * 0x00010001u;  // bytes dx, dx+2 of one row
        const uint32_t sel_o = 0x0C050C01u + dx[q] * 0x00010001u;  // byte dx+1 of src1 (low) and of src0 (high)
        e0[q] += __builtin_amdgcn_perm(r0, r0, sel_e);
        e1[q] += __builtin_amdgcn_perm(r1, r1, sel_e);
        e2[q] += __builtin_amdgcn_perm(r2, r2, sel_e);
        o01[q] += __builtin_amdgcn_perm(r1, r0, sel_o);
        o2[q] += __builtin_amdgcn_perm(r2, r2, 0x0C0C0C01u + dx[q]);
      }
    }
    p = pn;
  }
#pragma unroll
  for (int q = 0; q < kTile3Angles; q++) {
    uint32_t tot[9];
    tot[0] = wave_sum(e0[q] & 0xFFFFu), tot[1] = wave_sum(o01[q] & 0xFFFFu), tot[2] = wave_sum(e0[q] >> 16);
    tot[3] = wave_sum(e1[q] & 0xFFFFu), tot[4] = wave_sum(o01[q] >> 16), tot[5] = wave_sum(e1[q] >> 16);
    tot[6] = wave_sum(e2[q] & 0xFFFFu), tot[7] = wave_sum(o2[q]), tot[8] = wave_sum(e2[q] >> 16);
    uint32_t mine = 0;
#pragma unroll
    for (int c = 0; c < 9; c++)
      if (lane == c) mine = tot[c];
    if (lane < 9 && angle_on[q]) r_scan[(a0 + q) * 9 + lane] = (int32_t)mine;
  }
}

template <int kTile3Angles>
__global__ void __launch_bounds__(64)
k_resp_tile3(const uint4* __restrict__ tiles, int tile_cols, Geom g, PassCfg pc, const Lattice* __restrict__ lat,
             const double2* __restrict__ cossin, const double2* __restrict__ local, int32_t* __restrict__ resp, size_t resp_stride, int S) {
  const int w = blockIdx.x;
  const int xcd = w & 7, r = w >> 3;
  const int pairs = (pc.na + kTile3Angles - 1) / kTile3Angles;
  const int s = (r / pairs) * 8 + xcd;  // all angles of a scan on one XCD, like k_resp_rows
  const int a0 = (r % pairs) * kTile3Angles;
  if (s >= S) return;
  resp_tile3_wave<kTile3Angles>(s, a0, (int)threadIdx.x, tiles, tile_cols, g, pc, lat, cossin, local, resp + (size_t)s * resp_stride);
}

// ------------------------------------------------------------------------------------------
// k_resp_generic: exact response numerators for arbitrary candidate positions (fine pass 3x3,
// non-uniform lattices, anything the packed kernel does not cover).  Work item = (scan, angle,
// chunk of 16 positions); lanes stride over beams; per-byte bounds check exactly as
// Mapper.cpp:841-845.
// ------------------------------------------------------------------------------------------
constexpr int kPosChunk = 16;
#if !defined(LSLAM_TUNE_REDUCE_NARROW_MIN)
#define LSLAM_TUNE_REDUCE_NARROW_MIN 2048
#endif
constexpr int kReduceNarrowMinScans = LSLAM_TUNE_REDUCE_NARROW_MIN;  // batches from here on run k_reduce_coarse_lds with 128-thread blocks
// one work item = (angle a, chunk c of 16 lattice positions) of one scan, done by one wave
__device__ __forceinline__ void generic_item(const uint8_t* __restrict__ grid, const Geom& g, const PassCfg& pc,
                                             const Lattice& L, const double2* __restrict__ lp, int32_t* r, int a,
                                             int c, int lane) {
  const int np = pc.nx * pc.ny;
  const double angle = (L.center[2] - pc.ang_off) + (uint32_t)a * pc.ang_res;
  const double cosine = cos(angle), sine = sin(angle);
  int pos[kPosChunk];
#pragma unroll
  for (int q = 0; q < kPosChunk; q++) {
    int f = c * kPosChunk + q;
    pos[q] = f < np ? L.gx[f % pc.nx] + L.gy[f / pc.nx] * g.stride : -1;
  }
  int32_t acc[kPosChunk];
#pragma unroll
  for (int q = 0; q < kPosChunk; q++) acc[q] = 0;
  for (int b = lane; b < g.n_beams; b += 64) {
    double2 p = lp[b];
    if (isnan(p.x)) continue;
    int t = lookup_offset(p.x, p.y, cosine, sine, g.off_x, g.off_y, g.scale, g.stride);
#pragma unroll
    for (int q = 0; q < kPosChunk; q++) {
      long long idx = (long long)pos[q] + t;
      if (pos[q] >= 0 && idx >= 0 && idx < g.data_size) acc[q] += grid[idx];
    }
  }
#pragma unroll
  for (int q = 0; q < kPosChunk; q++) {
    int v = acc[q];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    int f = c * kPosChunk + q;
    if (lane == 0 && f < np) r[(size_t)a * np + f] = v;
  }
}

__global__ void __launch_bounds__(64)
k_resp_generic(const uint8_t* __restrict__ grid, Geom g, PassCfg pc, const Lattice* __restrict__ lat,
               const double2* __restrict__ local, int32_t* __restrict__ resp, size_t resp_stride, int S) {
  const int chunks = (pc.nx * pc.ny + kPosChunk - 1) / kPosChunk;
  const int per_scan = pc.na * chunks;
  const long long total = (long long)S * per_scan;
  for (long long w = blockIdx.x; w < total; w += gridDim.x) {
    const int s = (int)(w / per_scan), rem = (int)(w % per_scan);
    const Lattice& L = lat[s];
    if (!L.active || L.status != 0) continue;
    generic_item(grid, g, pc, L, local + (size_t)s * g.n_beams, resp + (size_t)s * resp_stride, rem / chunks,
                 rem % chunks, threadIdx.x);
  }
}

// The packed kernels (k_resp_rows, k_resp_tile3, k_resp_dense) skip a scan whose lattice is not
// uniform -- a lattice coordinate that rounds on a cell boundary (k_pass_setup) -- and the scan's own
// reduce block computes its numerators here before it reduces them: no work list, no extra launch.
// `fb_step` is the lattice step the packed kernel required, or 0 when the generic kernel did the pass.
__device__ __forceinline__ void block_generic_fallback(const uint8_t* __restrict__ grid, const Geom& g,
                                                       const PassCfg& pc, const Lattice& L,
                                                       const double2* __restrict__ lp, int32_t* r, int fb_step,
                                                       int tid, int nthreads) {
  if (fb_step == 0 || (L.step_x == fb_step && L.step_y == fb_step)) return;  // block-uniform
  const int chunks = (pc.nx * pc.ny + kPosChunk - 1) / kPosChunk;
  const int per_scan = pc.na * chunks;
  for (int w = tid >> 6; w < per_scan; w += nthreads >> 6) generic_item(grid, g, pc, L, lp, r, w / chunks, w % chunks, tid & 63);
  __syncthreads();  // the block reads these sums next
}

// ------------------------------------------------------------------------------------------
// shared pieces of the two reduce kernels
// ------------------------------------------------------------------------------------------
struct Cand {
  double x, y, angle;  // lattice offsets and absolute angle of candidate k
};

__device__ __forceinline__ Cand cand_of(int k, const PassCfg& pc, const double* center) {
  int a = k % pc.na, c = k / pc.na;
  int xi = c % pc.nx, yi = c / pc.nx;
  Cand cd;
  cd.x = -pc.off_x + (uint32_t)xi * pc.res_x;  // Mapper.cpp:342-345
  cd.y = -pc.off_y + (uint32_t)yi * pc.res_y;  // :353-356
  cd.angle = (center[2] - pc.ang_off) + (uint32_t)a * pc.ang_res;  // :390-393
  return cd;
}

// GetResponse normalisation + odometry penalty (Mapper.cpp:852, 399-414)
__device__ __forceinline__ double penalized(int32_t sum, const Cand& cd, const double* center, int n_beams,
                                            const SearchCfg& sc) {
  double r = response_of_sum(sum, n_beams, sc);
  if (sc.do_penalize && !double_equal(r, 0.0)) {
    double sd = ksq(cd.x) + ksq(cd.y);
    double dp = 1.0 - (kDistPenaltyGain * sd / sc.dvp);
    dp = dp > sc.min_dp ? dp : sc.min_dp;
    double sad = ksq(cd.angle - center[2]);
    double ap = 1.0 - (kAnglePenaltyGain * sad / sc.avp);
    ap = ap > sc.min_ap ? ap : sc.min_ap;
    r *= (dp * ap);
  }
  return r;
}

// max over the block: butterfly inside each wave, then the (<= 16) wave maxima through LDS.  Two barriers;
// the first also publishes whatever the callers wrote to LDS before the call.
__device__ __forceinline__ double block_max(double v, double* sh, int tid, int nthreads) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) {
    const double o = __shfl_xor(v, m);
    v = v > o ? v : o;
  }
  if ((tid & 63) == 0) sh[tid >> 6] = v;
  __syncthreads();
  double r = sh[0];
  for (int w = 1; w < (nthreads >> 6); w++) r = r > sh[w] ? r : sh[w];
  __syncthreads();
  return r;
}

// Average of all poses whose response equals the best within KT_TOLERANCE, accumulated in
// lattice order by ONE thread so the fp64 sums round exactly like the reference's sequential loop
// (Mapper.cpp:456-483).  `mask` holds the tie bits.
__device__ int tie_average(const uint32_t* mask, int total, const PassCfg& pc, const double* center,
                           double* avg) {
  double ax = 0, ay = 0, tx = 0, ty = 0;
  int cnt = 0;
  int words = (total + 31) / 32;
  for (int wd = 0; wd < words; wd++) {
    uint32_t m = mask[wd];
    while (m) {
      int bit = __ffs(m) - 1;
      m &= m - 1;
      int k = wd * 32 + bit;
      Cand cd = cand_of(k, pc, center);
      double h = normalize_angle(cd.angle);  // stored heading (:417-418)
      ax += center[0] + cd.x;
      ay += center[1] + cd.y;
      double sn, cs;  // one argument reduction for both (ocml's sincos returns sin's and cos's own values)
      sincos(h, &sn, &cs);
      tx += cs;
      ty += sn;
      cnt++;
    }
  }
  if (cnt == 0) return 0;
  ax /= cnt; ay /= cnt; tx /= cnt; ty /= cnt;
  avg[0] = ax; avg[1] = ay; avg[2] = atan2(ty, tx);
  return cnt;
}

// ------------------------------------------------------------------------------------------
// k_reduce_coarse: one block per scan (Mapper.cpp:431-501, 535-630).  Response numerators are
// stored angle-major (resp[a*ncand + c]); the reference's candidate order k = c*nA + a (y, x,
// angle) is what the tie mask and every ordered loop use.  Dynamic LDS:
//   [CACHE ? total : 0] doubles penalised responses | ncand doubles | side^2 doubles | mask words
// ------------------------------------------------------------------------------------------
template <bool CACHE>
__global__ void __launch_bounds__(256)
k_reduce_coarse(Geom g, PassCfg pc, SearchCfg sc, const Lattice* __restrict__ lat,
                int32_t* resp, size_t resp_stride, CoarseOut* __restrict__ out,
                int use_expansion, int pass_index, const uint8_t* __restrict__ grid,
                const double2* __restrict__ local, int fb_step) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ double sh[256];
  const int s = blockIdx.x, tid = threadIdx.x;
  const Lattice& L = lat[s];
  if (!L.active) return;
  if (L.status != 0) {
    if (tid == 0) { out[s].status = L.status; out[s].expand = 0; out[s].best = 0.0; }
    return;
  }
  block_generic_fallback(grid, g, pc, L, local + (size_t)s * g.n_beams, resp + (size_t)s * resp_stride, fb_step, tid, 256);
  const int ncand = pc.nx * pc.ny;
  const int total = ncand * pc.na;
  const int words = (total + 31) / 32;
  double* presp = (double*)smem;
  double* latmax = presp + (CACHE ? total : 0);
  double* probs = latmax + ncand;
  double* terms = probs + g.probs_side * g.probs_side;  // 4 per lattice cell
  uint32_t* mask = (uint32_t*)(terms + 4 * ncand);
  int* cell = (int*)(mask + words);
  __shared__ double s_avg[3];
  __shared__ int s_status;
  const int32_t* r = resp + (size_t)s * resp_stride;
  const double center[3] = {L.center[0], L.center[1], L.center[2]};
  auto value = [&](int k) -> double {
    if (CACHE) return presp[k];
    const int a = k % pc.na, c = k / pc.na;
    return penalized(r[a * ncand + c], cand_of(k, pc, center), center, g.n_beams, sc);
  };

  double lm = -1.0;  // bestResponse starts at -1 (Mapper.cpp:431)
  for (int t = tid; t < total; t += 256) {  // storage order: coalesced reads
    const int a = t / ncand, c = t - a * ncand;
    const int k = c * pc.na + a;
    double v = penalized(r[t], cand_of(k, pc, center), center, g.n_beams, sc);
    if (CACHE) presp[k] = v;
    lm = lm > v ? lm : v;
  }
  for (int wd = tid; wd < words; wd += 256) mask[wd] = 0u;
  for (int c = tid; c < g.probs_side * g.probs_side; c += 256) probs[c] = 0.0;  // Clear (:329)
  const double best = block_max(lm, sh, tid, 256);  // contains the barriers that publish presp/mask

  // best response per lattice cell over all angles (what the reference max-merges into
  // m_pSearchSpaceProbs, Mapper.cpp:437-450)
  for (int c = tid; c < ncand; c += 256) {
    double m = -1.0;
    for (int a = 0; a < pc.na; a++) {
      double v = value(c * pc.na + a);
      m = m > v ? m : v;
    }
    latmax[c] = m;
  }
  for (int k = tid; k < total; k += 256)
    if (double_equal(value(k), best)) atomicOr(&mask[k >> 5], 1u << (k & 31));
  // search-space probability cell of every lattice position (offset = searchCenter -
  // searchSpaceOffset, :332-333; WorldToGrid of the candidate position, :440) -- in parallel; only
  // the order-sensitive merges and sums below stay on one thread
  const double p_off_x = center[0] - pc.off_x, p_off_y = center[1] - pc.off_y;
  for (int c = tid; c < ncand; c += 256) {
    int xi = c % pc.nx, yi = c / pc.nx;
    double wx = center[0] + (-pc.off_x + (uint32_t)xi * pc.res_x);
    double wy = center[1] + (-pc.off_y + (uint32_t)yi * pc.res_y);
    int gx = world_to_grid(wx, p_off_x, g.scale), gy = world_to_grid(wy, p_off_y, g.scale);
    cell[c] = (gx < 0 || gx >= g.probs_side || gy < 0 || gy >= g.probs_side) ? -1 : gy * g.probs_side + gx;
  }
  __syncthreads();
  if (tid == 0) {
    int st = 0;
    for (int c = 0; c < ncand; c++) {  // *ptr = max(response, *ptr) in candidate order (:443-449)
      if (cell[c] < 0) { st = LSLAM_ERR_PROBABILITY_SEARCH; break; }
      double* p = &probs[cell[c]];
      *p = latmax[c] > *p ? latmax[c] : *p;
    }
    double avg[3] = {0, 0, 0};
    if (st == 0 && tie_average(mask, total, pc, center, avg) == 0) st = LSLAM_ERR_NO_BEST_POSE;
    s_avg[0] = avg[0]; s_avg[1] = avg[1]; s_avg[2] = avg[2];
    s_status = st;
  }
  __syncthreads();
  // ComputePositionalCovariance terms (Mapper.cpp:573-594), one lattice cell per thread; the
  // running sums are then accumulated in the reference's order by one thread
  const double dx = s_avg[0] - center[0], dy = s_avg[1] - center[1];
  for (int c = tid; c < ncand; c += 256) {
    int xi = c % pc.nx, yi = c / pc.nx;
    double x = -pc.off_x + (uint32_t)xi * pc.res_x;
    double y = -pc.off_y + (uint32_t)yi * pc.res_y;
    double rr = cell[c] >= 0 ? probs[cell[c]] : 0.0;
    terms[4 * c + 0] = rr;
    terms[4 * c + 1] = (ksq(x - dx) * rr);
    terms[4 * c + 2] = ((x - dx) * (y - dy) * rr);
    terms[4 * c + 3] = (ksq(y - dy) * rr);
  }
  __syncthreads();
  if (tid != 0) return;

  CoarseOut o;
  o.status = s_status;
  o.flags = pass_index > 0 ? 1 : 0;
  o.pad = 0;
  const double avg[3] = {s_avg[0], s_avg[1], s_avg[2]};
  double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (o.status == 0) {
    if (best < kTol) {
      cov[0] = kMaxVariance; cov[4] = kMaxVariance; cov[8] = 4 * ksq(pc.ang_res);
    } else {
      double axx = 0, axy = 0, ayy = 0, norm = 0;
      for (int c = 0; c < ncand; c++) {  // y outer, x inner = candidate-cell order
        double rr = terms[4 * c];
        if (rr >= (best - 0.1)) {
          norm += rr;
          axx += terms[4 * c + 1];
          axy += terms[4 * c + 2];
          ayy += terms[4 * c + 3];
        }
      }
      if (norm > kTol) {
        double vxx = axx / norm, vxy = axy / norm, vyy = ayy / norm;
        double vthth = 4 * ksq(pc.ang_res);
        double min_xx = 0.1 * ksq(pc.res_x), min_yy = 0.1 * ksq(pc.res_y);
        vxx = vxx > min_xx ? vxx : min_xx;
        vyy = vyy > min_yy ? vyy : min_yy;
        double mult = 1.0 / best;
        cov[0] = vxx * mult; cov[1] = vxy * mult; cov[3] = vxy * mult; cov[4] = vyy * mult;
        cov[8] = vthth;
      }
      if (double_equal(cov[0], 0.0)) cov[0] = kMaxVariance;
      if (double_equal(cov[4], 0.0)) cov[4] = kMaxVariance;
    }
  }
  o.mean[0] = avg[0]; o.mean[1] = avg[1]; o.mean[2] = avg[2];
  for (int i = 0; i < 9; i++) o.cov[i] = cov[i];
  o.best = best > 1.0 ? 1.0 : best;  // :514-517
  // Mapper.cpp:242-244,259: expand (again) while the best response is still zero
  o.expand = (use_expansion && o.status == 0 && pass_index < 3 && double_equal(o.best, 0.0)) ? 1 : 0;
  out[s] = o;
}

// ------------------------------------------------------------------------------------------
// k_reduce_coarse_lds: k_reduce_coarse<true> restructured for throughput (same results bit for bit;
// used when all penalised responses fit LDS and the tie mask has <= 256 words).  Differences:
//  * the two penalty factors depend on the lattice cell only (distance) and on the angle only, so
//    they are tabulated once (121 + 21 entries here) and a candidate costs one exact fp64 division
//    and two multiplications -- r *= (dp * ap), the reference's own grouping (Mapper.cpp:399-414);
//  * candidate indices advance incrementally (no integer divisions in the loops);
//  * the search-space-probability merge is a max (order-independent): 64-bit LDS atomicMax on the
//    bit patterns of non-negative doubles instead of a 121-step serial loop;
//  * the tie average visits only the non-zero mask words (wave ballots), still in lattice order on
//    one thread; the ordered covariance sums stay on one thread, with their LDS reads batched.
// ------------------------------------------------------------------------------------------
// NT threads per block: 128 when the batch fills the chip (a block is mostly single-thread ordered sums, so residency --
// 32 waves per CU = 16 such blocks -- buys more than lanes), 256 for small batches (latency of the one block that runs).
// The block's work as a function of (scan s, thread tid of NT): `r` = the scan's numerators (angle-major: global in
// k_reduce_coarse_lds, the block's LDS copy in k_match_step), `smem` = reduce_lds_nocache() bytes of LDS scratch.
template <int NT>
__device__ __forceinline__ void reduce_coarse_lds_block(
    const int s, const int tid, const Geom& g, const PassCfg& pc, const SearchCfg& sc, Lattice* lat, int32_t* r,
    CoarseOut* __restrict__ out, int use_expansion, int pass_index, const uint8_t* __restrict__ grid,
    const double2* __restrict__ local, int fb_step, const PassCfg& fine_pc, double2* fine_cossin, int fine_step,
    int zero_fine_words, int parts, unsigned char* smem) {
  __shared__ double sh[NT];
  __shared__ double s_ap[kMaxAngles];
  __shared__ unsigned long long s_nz[4];
  __shared__ double s_avg[3];
  __shared__ int s_status, s_bad, s_ntie, s_wcnt[(NT + 63) / 64];
  const Lattice& L = lat[s];
  if (!L.active) return;
  if (L.status != 0) {
    if (tid == 0) { out[s].status = L.status; out[s].expand = 0; out[s].best = 0.0; }
    return;
  }
  LSLAM_PHASE_CLOCK(pck);
  block_generic_fallback(grid, g, pc, L, local + (size_t)s * g.n_beams, r, fb_step, tid, NT);
  const int ncand = pc.nx * pc.ny;
  const int total = ncand * pc.na;
  const int words = (total + 31) / 32;  // <= 256 (host)
  // No per-candidate cache in LDS (it was 20 KB of the block's 29 KB and capped the kernel at 5 blocks per CU -- the
  // kernel is latency-bound, so residency is what it needs): each thread owns the angles of ONE lattice cell half and
  // keeps only the cell maximum; the few cells that can tie with the best response are re-evaluated below with the
  // same expression, hence the same bits.
  double* latmax = (double*)smem;  // `parts` rows of ncand: the angles of a cell are split over `parts` threads (host:
  double* probs = latmax + (size_t)parts * ncand;  // as many as fit the block, so that a thread's loads are one batch)
  double* terms = probs + g.probs_side * g.probs_side;  // 4 per lattice cell
  uint32_t* mask = (uint32_t*)(terms + 4 * ncand);
  int* cell = (int*)(mask + words);
  int* tie = cell + ncand;  // lattice cells that may hold a tie with the best response
  double* dpen = terms;  // distance penalty per lattice cell; dead before `terms` is written
  const double center[3] = {L.center[0], L.center[1], L.center[2]};

  // per-cell and per-angle penalty factors (Mapper.cpp:399-414) + search-space cell of every lattice
  // position (offset = searchCenter - searchSpaceOffset, :332-333; WorldToGrid of the position, :440)
  const double p_off_x = center[0] - pc.off_x, p_off_y = center[1] - pc.off_y;
  if (tid == 0) { s_bad = 0; s_ntie = 0; }
  for (int c = tid; c < ncand; c += NT) {
    const int xi = c % pc.nx, yi = c / pc.nx;
    const double x = -pc.off_x + (uint32_t)xi * pc.res_x;  // Mapper.cpp:342-345
    const double y = -pc.off_y + (uint32_t)yi * pc.res_y;  // :353-356
    const double sd = ksq(x) + ksq(y);
    double dp = 1.0 - (kDistPenaltyGain * sd / sc.dvp);
    dpen[c] = dp > sc.min_dp ? dp : sc.min_dp;
    const double wx = center[0] + x, wy = center[1] + y;
    const int gx = world_to_grid(wx, p_off_x, g.scale), gy = world_to_grid(wy, p_off_y, g.scale);
    cell[c] = (gx < 0 || gx >= g.probs_side || gy < 0 || gy >= g.probs_side) ? -1 : gy * g.probs_side + gx;
  }
  for (int a = tid; a < pc.na; a += NT) {
    const double angle = (center[2] - pc.ang_off) + (uint32_t)a * pc.ang_res;  // :390-393
    const double sad = ksq(angle - center[2]);
    double ap = 1.0 - (kAnglePenaltyGain * sad / sc.avp);
    s_ap[a] = ap > sc.min_ap ? ap : sc.min_ap;
  }
  for (int wd = tid; wd < words; wd += NT) mask[wd] = 0u;
  for (int c = tid; c < g.probs_side * g.probs_side; c += NT) probs[c] = 0.0;  // Clear (:329)
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 0);  // penalty tables, search-space cells, clears

  // penalised response of candidate (c, a): GetResponse normalisation (:852) and r *= (dp * ap) (:399-414)
  auto value_of = [&](int32_t sum, int c, int a) -> double {
    double v = response_of_sum(sum, g.n_beams, sc);
    if (sc.do_penalize && !double_equal(v, 0.0)) v *= (dpen[c] * s_ap[a]);
    return v;
  };
  // cell maxima: thread -> (cell, half of the angles); numerators are stored angle-major, so neighbouring threads read
  // neighbouring words; a thread's loads are issued eleven at a time
  double lm = -1.0;  // bestResponse starts at -1 (Mapper.cpp:431)
  for (int idx = tid; idx < ncand * parts; idx += NT) {
    const int part = idx / ncand, c = idx - part * ncand;
    const int a_lo = part * pc.na / parts, a_hi = (part + 1) * pc.na / parts;
    double m = -1.0;
#if !defined(LSLAM_TUNE_REDUCE_BATCH)
#define LSLAM_TUNE_REDUCE_BATCH 11
#endif
    constexpr int kBatch = LSLAM_TUNE_REDUCE_BATCH;
    for (int a0 = a_lo; a0 < a_hi; a0 += kBatch) {
      int32_t rv[kBatch];
#pragma unroll
      for (int i = 0; i < kBatch; i++) rv[i] = a0 + i < a_hi ? r[(a0 + i) * ncand + c] : 0;
#pragma unroll
      for (int i = 0; i < kBatch; i++)
        if (a0 + i < a_hi) {
          const double v = value_of(rv[i], c, a0 + i);
          m = m > v ? m : v;
        }
    }
    latmax[(size_t)part * ncand + c] = m;
    lm = lm > m ? lm : m;
  }
  LSLAM_PHASE_MARK(pck, 1);  // numerators -> penalised responses (one division each), cell maxima
  const double best = block_max(lm, sh, tid, NT);  // contains the barriers that publish the latmax rows
  LSLAM_PHASE_MARK(pck, 2);  // block maximum

  // best response per lattice cell over all angles, max-merged into the search-space probabilities
  // (Mapper.cpp:437-450); responses are >= +0, so the unsigned order of the bit patterns is the fp order
  for (int c = tid; c < ncand; c += NT) {
    double m = latmax[c];
    for (int q = 1; q < parts; q++) m = m > latmax[(size_t)q * ncand + c] ? m : latmax[(size_t)q * ncand + c];
    latmax[c] = m;
    if (cell[c] < 0) s_bad = 1;
    else atomicMax((unsigned long long*)&probs[cell[c]], (unsigned long long)__double_as_longlong(m < 0.0 ? 0.0 : m));
    // ties with the best response (:452-455): only a cell whose maximum reaches it can hold one
    if (!(m + 2.0 * kTol < best)) tie[atomicAdd(&s_ntie, 1)] = c;  // conservative filter; the test below is the reference's
  }
  __syncthreads();
  for (int idx = tid, n = s_ntie * pc.na; idx < n; idx += NT) {
    const int c = tie[idx / pc.na], a = idx % pc.na;
    if (double_equal(value_of(r[a * ncand + c], c, a), best)) {
      const int k = c * pc.na + a;
      atomicOr(&mask[k >> 5], 1u << (k & 31));
    }
  }
  __syncthreads();
  for (int w0 = 0; w0 < 256; w0 += NT) {  // which mask words are non-zero (words <= 256, host)
    const int wd = w0 + tid;
    const unsigned long long nz = __ballot(wd < words && mask[wd] != 0u);
    if ((tid & 63) == 0 && wd < 256) s_nz[wd >> 6] = nz;
  }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 3);  // probabilities merged, tie candidates re-evaluated, mask words
  if (tid == 0) {
    LSLAM_SERIAL_BEGIN();
    int st = s_bad ? LSLAM_ERR_PROBABILITY_SEARCH : 0;
    double avg[3] = {0, 0, 0};
    if (st == 0) {  // tie average in lattice order (Mapper.cpp:456-483), only over the words that hold ties
      double ax = 0, ay = 0, tx = 0, ty = 0;
      int cnt = 0;
      for (int wv = 0; wv < 4; wv++) {
        unsigned long long nzw = s_nz[wv];
        while (nzw) {
          const int wd = wv * 64 + (__ffsll((long long)nzw) - 1);
          nzw &= nzw - 1;
          uint32_t m = mask[wd];
          while (m) {
            const int bit = __ffs(m) - 1;
            m &= m - 1;
            const Cand cd = cand_of(wd * 32 + bit, pc, center);
            const double h = normalize_angle(cd.angle);  // stored heading (:417-418)
            ax += center[0] + cd.x;
            ay += center[1] + cd.y;
            double sn, cs;  // one argument reduction for both (ocml's sincos returns sin's and cos's own values)
            sincos(h, &sn, &cs);
            tx += cs;
            ty += sn;
            cnt++;
          }
        }
      }
      if (cnt == 0) {
        st = LSLAM_ERR_NO_BEST_POSE;
      } else {
        ax /= cnt; ay /= cnt; tx /= cnt; ty /= cnt;
        avg[0] = ax; avg[1] = ay; avg[2] = atan2(ty, tx);
      }
    }
    s_avg[0] = avg[0]; s_avg[1] = avg[1]; s_avg[2] = avg[2];
    s_status = st;
    LSLAM_SERIAL_END();
  }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 4);  // ordered tie average on thread 0 (cos, sin per tie, atan2)
  // No expansion passes follow: the block's LAST wave lays out the scan's fine lattice around the mean
  // just found (k_pass_setup, mode 2; one launch fewer) while the other waves go on to the covariance.
  // The lattice record of this pass is not read again below (centre and flags are in registers).
  if (fine_cossin != nullptr && tid >= NT - 64) {
    const double center2[3] = {s_avg[0], s_avg[1], s_avg[2]};
    pass_setup_wave(s, tid - (NT - 64), g, fine_pc, center2, s_status == 0, lat, fine_cossin, fine_step);
  }
  // ComputePositionalCovariance terms (Mapper.cpp:573-594), one lattice cell per thread.  Only the cells with
  // response >= best - 0.1 enter the sums (:580): they are compacted IN LATTICE ORDER (wave ballots), so the one thread
  // that adds them up in the reference's order walks a dozen entries instead of every cell of the lattice.
  const double dx = s_avg[0] - center[0], dy = s_avg[1] - center[1];
  const double thr = best - 0.1;
  int n_pass = 0;
  for (int c0 = 0; c0 < ncand; c0 += NT) {
    const int c = c0 + tid;
    double rr = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
    bool pass = false;
    if (c < ncand) {
      const int xi = c % pc.nx, yi = c / pc.nx;
      const double x = -pc.off_x + (uint32_t)xi * pc.res_x;
      const double y = -pc.off_y + (uint32_t)yi * pc.res_y;
      rr = cell[c] >= 0 ? probs[cell[c]] : 0.0;
      t1 = (ksq(x - dx) * rr);
      t2 = ((x - dx) * (y - dy) * rr);
      t3 = (ksq(y - dy) * rr);
      pass = rr >= thr;
    }
    const unsigned long long bal = __ballot(pass);
    if ((tid & 63) == 0) s_wcnt[tid >> 6] = __popcll(bal);
    __syncthreads();
    int off = n_pass;
    for (int w = 0; w < (tid >> 6); w++) off += s_wcnt[w];
    for (int w = 0; w < (NT + 63) / 64; w++) n_pass += s_wcnt[w];
    if (pass) {
      const int k = off + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
      terms[4 * k + 0] = rr; terms[4 * k + 1] = t1; terms[4 * k + 2] = t2; terms[4 * k + 3] = t3;
    }
    __syncthreads();
  }
  LSLAM_PHASE_MARK(pck, 5);  // fine lattice (last wave) | covariance terms compacted in lattice order
  if (tid == 0) {
    CoarseOut o;
    o.status = s_status;
    o.flags = pass_index > 0 ? 1 : 0;
    o.pad = 0;
    const double avg[3] = {s_avg[0], s_avg[1], s_avg[2]};
    double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (o.status == 0) {
      if (best < kTol) {
        cov[0] = kMaxVariance; cov[4] = kMaxVariance; cov[8] = 4 * ksq(pc.ang_res);
      } else {
        double axx = 0, axy = 0, ayy = 0, norm = 0;
        int c = 0;
        for (; c + 4 <= n_pass; c += 4) {  // lattice order (y outer, x inner); reads batched, adds in order
          double q[16];
  #pragma unroll
          for (int i = 0; i < 16; i++) q[i] = terms[4 * c + i];
  #pragma unroll
          for (int i = 0; i < 4; i++) { norm += q[4 * i]; axx += q[4 * i + 1]; axy += q[4 * i + 2]; ayy += q[4 * i + 3]; }
        }
        for (; c < n_pass; c++) { norm += terms[4 * c]; axx += terms[4 * c + 1]; axy += terms[4 * c + 2]; ayy += terms[4 * c + 3]; }
        if (norm > kTol) {
          double vxx = axx / norm, vxy = axy / norm, vyy = ayy / norm;
          double vthth = 4 * ksq(pc.ang_res);
          double min_xx = 0.1 * ksq(pc.res_x), min_yy = 0.1 * ksq(pc.res_y);
          vxx = vxx > min_xx ? vxx : min_xx;
          vyy = vyy > min_yy ? vyy : min_yy;
          double mult = 1.0 / best;
          cov[0] = vxx * mult; cov[1] = vxy * mult; cov[3] = vxy * mult; cov[4] = vyy * mult;
          cov[8] = vthth;
        }
        if (double_equal(cov[0], 0.0)) cov[0] = kMaxVariance;
        if (double_equal(cov[4], 0.0)) cov[4] = kMaxVariance;
      }
    }
    o.mean[0] = avg[0]; o.mean[1] = avg[1]; o.mean[2] = avg[2];
    for (int i = 0; i < 9; i++) o.cov[i] = cov[i];
    o.best = best > 1.0 ? 1.0 : best;  // :514-517
    // Mapper.cpp:242-244,259: expand (again) while the best response is still zero
    o.expand = (use_expansion && o.status == 0 && pass_index < 3 && double_equal(o.best, 0.0)) ? 1 : 0;
    out[s] = o;
  }
  // The fine pass follows at once and its beam-sliced form accumulates with atomics: clear its numerators here (every read
  // of this scan's coarse numerators is behind the barriers above) instead of a fill operation on the stream.
  for (int i = tid; i < zero_fine_words; i += NT) r[i] = 0;
  LSLAM_PHASE_MARK(pck, 6);  // ordered covariance sums + record on thread 0, fine numerators cleared
  LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 3, (unsigned)(s * (NT >> 6) + (tid >> 6)), (tid & 63) == 0);
}

// (anchor_spec_block: the speculative anchor chain of the scan being matched, see k_anchor_chain -- an extra block of the
//  lone launch, blockIdx.x == the number of scans)
struct SpecArgs {
  const double* ranges;  // the scan's readings (resident)
  double pose[3];        // sensor pose the match starts from
  int* out;              // nullptr: no speculative block in this launch
  const double2* local;  // nullptr, or the scan's scan-frame points (k_scan_prep) -- see anchor_spec_block
};
__device__ __forceinline__ void anchor_spec_block(int n, const SpecArgs& sa, const Geom& g, unsigned char* smem);
template <int NT>
__global__ void __launch_bounds__(NT)
k_reduce_coarse_lds(Geom g, PassCfg pc, SearchCfg sc, Lattice* lat,
                    int32_t* resp, size_t resp_stride, CoarseOut* __restrict__ out,
                    int use_expansion, int pass_index, const uint8_t* __restrict__ grid,
                const double2* __restrict__ local, int fb_step, PassCfg fine_pc,
                    double2* fine_cossin, int fine_step, int zero_fine_words, int parts, SpecArgs spec) {
  extern __shared__ __align__(16) unsigned char smem[];
  if constexpr (NT == 1024) {
    if (spec.out && blockIdx.x + 1 == gridDim.x) {
      anchor_spec_block(g.n_beams, spec, g, smem);
      return;
    }
  }
  reduce_coarse_lds_block<NT>((int)blockIdx.x, (int)threadIdx.x, g, pc, sc, lat, resp + (size_t)blockIdx.x * resp_stride, out,
                              use_expansion, pass_index, grid, local, fb_step, fine_pc, fine_cossin, fine_step, zero_fine_words,
                              parts, smem);
}


// ------------------------------------------------------------------------------------------
// Large lattices (the loop-closure matcher: search space 8-15 m -> 81..151 positions per side x 21
// angles = 140-480 k candidates per match, Mapper.cpp:862-871, 976-1051).  Here the candidate
// window of a beam is hundreds of cells wide and the windows of ALL candidates overlap almost
// completely, so the work is turned around: one wave owns a tile of lattice rows of one
// (scan, angle) and streams every beam's window through it.  Lane (r, k) loads 16 contiguous
// parity-plane bytes = 16 neighbouring candidates of lattice row r -- adjacent lanes read adjacent
// bytes, so a row is one coalesced 112-byte access -- and accumulates them in packed 16-bit
// fields, flushed to 32-bit registers every 256 beams (255 * 256 < 2^16).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_table_big(int S, Geom g, PassCfg pc, const Lattice* __restrict__ lat, const double2* __restrict__ local,
            int32_t* __restrict__ tbl, unsigned long long* __restrict__ best_bits) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  const int a = blockIdx.y, s = blockIdx.z;
  // k_big_latmax's atomicMax target starts from zero: cleared here (two launches ahead of its first use) instead of by a
  // fill operation on the stream -- a 5 us kernel of its own in a lone loop-closure match
  if (b == 0 && a == 0) best_bits[s] = 0ull;
  if (b >= g.n_beams) return;
  const Lattice& L = lat[s];
  int32_t v = kInvalidScan;
  if (L.active && L.status == 0) {
    const double angle = (L.center[2] - pc.ang_off) + (uint32_t)a * pc.ang_res;
    const double2 p = local[(size_t)s * g.n_beams + b];
    if (!isnan(p.x)) v = lookup_offset(p.x, p.y, cos(angle), sin(angle), g.off_x, g.off_y, g.scale, g.stride);
  }
  tbl[((size_t)s * pc.na + a) * g.n_beams + b] = v;
}

constexpr int kDenseT = 4;  // row slots per lane
// T = row slots per lane: 4 for batches (fewer table reads per candidate), 1 for a lone match (4x the waves: the kernel is
// a chain of dependent loads at 3 waves per CU otherwise)
template <int T>
__global__ void __launch_bounds__(64)
k_resp_dense(const uint8_t* __restrict__ src0, const uint8_t* __restrict__ src1, int limit, Geom g, PassCfg pc,
             const Lattice* __restrict__ lat, const int32_t* __restrict__ tbl, int32_t* __restrict__ resp,
             size_t resp_stride, int n_tiles, int beam_slices) {
  const int lane = threadIdx.x;
  int w = blockIdx.x;
  const int slice = w % beam_slices; w /= beam_slices;
  const int tile = w % n_tiles; w /= n_tiles;
  const int a = w % pc.na;
  const int s = w / pc.na;
  const Lattice& L = lat[s];
  if (!L.active || L.status != 0 || L.step_x != 2 || L.step_y != 2) return;
  const int lpr = (pc.nx + 15) / 16;  // lanes per lattice row
  const int rpw = 64 / lpr;           // lattice rows per wave-wide load
  const int r = lane / lpr, k = lane % lpr;
  const bool lane_on = r < rpw;
  const int j_base = tile * rpw * T;
  const long long pos00 = (long long)L.gx[0] + (long long)L.gy[0] * g.stride;
  const int32_t* trow = tbl + ((size_t)s * pc.na + a) * g.n_beams;
  const int per = (g.n_beams + beam_slices - 1) / beam_slices;
  const int b_lo = slice * per, b_hi = min(g.n_beams, b_lo + per);

  uint32_t acc[T][16];
#pragma unroll
  for (int t = 0; t < T; t++)
#pragma unroll
    for (int c = 0; c < 16; c++) acc[t][c] = 0u;
  // the planes are addressed like k_resp_rows addresses them: 32-bit offsets from the 64 zero bytes in front of plane 0
  const uint8_t* zbase = src0 - kRowZero;
  const uint32_t plane_delta = (uint32_t)(src1 - src0);
  int lane_off[T];   // of the lane's 16 candidates in row slot t, relative to the window's first half-index
  bool slot_on[T];
#pragma unroll
  for (int t = 0; t < T; t++) {
    const int j = j_base + t * rpw + r;
    slot_on[t] = lane_on && j < pc.ny;
    lane_off[t] = 16 * k + (slot_on[t] ? j : 0) * g.stride;
  }

  for (int b0 = b_lo; b0 < b_hi; b0 += 256) {
    uint32_t pe[T][4], po[T][4];
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int q = 0; q < 4; q++) pe[t][q] = po[t][q] = 0u;
    const int b1 = min(b_hi, b0 + 256);
    // The table entries of 64 beams arrive as ONE coalesced vector load (lane = beam; the next 64 are requested while these
    // are used) and are handed out with v_readlane: a scalar load per beam group sat in front of every group's row loads
    // as a second dependent round trip.  kGroup beams per iteration: all kGroup * T row loads are issued before the first
    // byte is used -- one beam per iteration was a chain of dependent loads (135 round trips for a single match).
    constexpr int kGroup = 4;
    int32_t tvec_next = b0 + lane < b1 ? trow[b0 + lane] : kInvalidScan;
    for (int c0 = b0; c0 < b1; c0 += 64) {
      const int32_t tvec = tvec_next;
      tvec_next = c0 + 64 + lane < b1 ? trow[c0 + 64 + lane] : kInvalidScan;
      const int cn = min(64, b1 - c0);
      for (int u0 = 0; u0 < cn; u0 += kGroup) {
        // Five dword-ALIGNED words per lane (a byte-aligned 16-byte access is split into single dwords by the address unit:
        // four times the cycles -- that, not latency or arithmetic, was this kernel's time), shifted into place below.  The
        // byte phase is wave-uniform: the planes, widthStep and the lanes' 16 k + j * widthStep are multiples of 4.
        uint32_t w5[kGroup][T][5];
        uint32_t phase[kGroup];
#pragma unroll
        for (int u = 0; u < kGroup; u++) {
          // per beam, on the scalar unit: flat index of the window's first cell, its parity plane, half index
          const int32_t tv = u0 + u < cn ? __builtin_amdgcn_readlane(tvec, u0 + u) : kInvalidScan;
          const long long base = pos00 + tv;  // the lattice lies inside the grid and the grid has <= 2^30 cells: |base| < 2^31
          const int half = (int)(base >> 1);
          const uint32_t poff = (uint32_t)kRowZero + ((base & 1) ? plane_delta : 0u);
          const bool beam_on = tv != kInvalidScan;
          phase[u] = ((uint32_t)half + poff) & 3u;
#pragma unroll
          for (int t = 0; t < T; t++) {
            // per lane: 32-bit offset from the zero guard in front of plane 0; a segment outside (-16, limit) -- and every
            // lane of a beam without a reading -- reads the guard instead: straight-line loads, no 64-bit address arithmetic
            const int mm = half + lane_off[t];
            const bool in = beam_on && slot_on[t] && (uint32_t)(mm + 15) < (uint32_t)(limit + 15);
            const uint32_t off = in ? (((uint32_t)mm + poff) & ~3u) : 0u;
            __builtin_memcpy(w5[u][t], __builtin_assume_aligned(zbase + off, 4), 20);
          }
        }
        uint4 d[kGroup][T];
#pragma unroll
        for (int u = 0; u < kGroup; u++)
#pragma unroll
          for (int t = 0; t < T; t++) {
            d[u][t].x = __builtin_amdgcn_alignbyte(w5[u][t][1], w5[u][t][0], phase[u]);
            d[u][t].y = __builtin_amdgcn_alignbyte(w5[u][t][2], w5[u][t][1], phase[u]);
            d[u][t].z = __builtin_amdgcn_alignbyte(w5[u][t][3], w5[u][t][2], phase[u]);
            d[u][t].w = __builtin_amdgcn_alignbyte(w5[u][t][4], w5[u][t][3], phase[u]);
          }
        if constexpr (T == 1) {
          // a lone match is a handful of waves per SIMD, each a serial instruction stream: one byte-select add (SDWA) per
          // candidate straight into its 32-bit sum is 16 instructions per beam where mask / shift / packed add / flush are 22
#pragma unroll
          for (int u = 0; u < kGroup; u++) {
            const uint32_t dw[4] = {d[u][0].x, d[u][0].y, d[u][0].z, d[u][0].w};
#pragma unroll
            for (int q = 0; q < 4; q++) {
#define LSLAM_ADD_BYTE(A, W, B) \
  asm("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_" #B : "+v"(A) : "v"(W))
              LSLAM_ADD_BYTE(acc[0][4 * q + 0], dw[q], 0);
              LSLAM_ADD_BYTE(acc[0][4 * q + 1], dw[q], 1);
              LSLAM_ADD_BYTE(acc[0][4 * q + 2], dw[q], 2);
              LSLAM_ADD_BYTE(acc[0][4 * q + 3], dw[q], 3);
#undef LSLAM_ADD_BYTE
            }
          }
        } else {
#pragma unroll
          for (int u = 0; u < kGroup; u++)
#pragma unroll
            for (int t = 0; t < T; t++) {
              const uint32_t dw[4] = {d[u][t].x, d[u][t].y, d[u][t].z, d[u][t].w};
#pragma unroll
              for (int q = 0; q < 4; q++) {
                pe[t][q] += dw[q] & 0x00FF00FFu;
                po[t][q] += (dw[q] >> 8) & 0x00FF00FFu;
              }
            }
        }
      }
    }
#pragma unroll
    for (int t = 0; t < T; t++)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        acc[t][4 * q + 0] += pe[t][q] & 0xFFFFu;
        acc[t][4 * q + 2] += pe[t][q] >> 16;
        acc[t][4 * q + 1] += po[t][q] & 0xFFFFu;
        acc[t][4 * q + 3] += po[t][q] >> 16;
      }
  }
  // The wave's rows j_base .. j_base + T*rpw - 1 are ONE contiguous run of the output (rows are nX ints apart): staged
  // through LDS and written with consecutive lanes on consecutive words.  Lane by lane (16 words each, 64 bytes apart)
  // every store instruction was ~60 separate write requests: 1.8 M of them per lone match, and THAT -- not the loads --
  // was this kernel's time.
  __shared__ int32_t stage[T * 1024];  // rpw * nX <= 64 * 16
  const int ncand = pc.nx * pc.ny;
  if (lane_on) {
#pragma unroll
    for (int t = 0; t < T; t++) {
      const int jl = t * rpw + r;
      if (j_base + jl >= pc.ny) continue;
#pragma unroll
      for (int c = 0; c < 16; c++) {
        const int i2 = 16 * k + c;
        if (i2 < pc.nx) stage[jl * pc.nx + i2] = (int32_t)acc[t][c];
      }
    }
  }
  __syncthreads();
  // beam slices write their partial sums side by side (slice q of scan s at resp + (s*slices + q) * resp_stride);
  // k_big_latmax adds them up -- integer atomics here were 2 M per single 101x101x21 match
  const int rows_here = min(T * rpw, pc.ny - j_base);
  int32_t* o = resp + ((size_t)s * beam_slices + slice) * resp_stride + (size_t)a * ncand + (size_t)j_base * pc.nx;
  for (int idx = lane, n_out = rows_here * pc.nx; idx < n_out; idx += 64) o[idx] = stage[idx];
}

// Reduce of a large coarse lattice (block per scan, global scratch instead of LDS): same steps as
// k_reduce_coarse.  scratch per scan: latmax[ncand] | probs[side^2] (as uint64 bit patterns: all
// responses are >= 0, so unsigned integer max == double max) | terms[4*ncand] | mask words
// k_big_latmax: the one part of the big-lattice reduce that is pure throughput -- one exact fp64 division per candidate
// (214 k of them for a 101 x 101 x 21 match) -- spread over as many blocks as the lattice has 256-cell slices instead of
// the single block (one CU) of k_reduce_coarse_big: per cell the best penalised response over all angles (latmax), per scan
// the best response (integer atomicMax on the bit pattern of a non-negative double: order-independent).  Scans whose
// lattice is not uniform are left to the reduce block, which computes their numerators first (block_generic_fallback).
__global__ void __launch_bounds__(256)
k_big_latmax(Geom g, PassCfg pc, SearchCfg sc, const Lattice* __restrict__ lat, int32_t* __restrict__ resp,
             size_t resp_stride, double* __restrict__ scratch, size_t scratch_stride,
             unsigned long long* __restrict__ best_bits, int fb_step, const int32_t* __restrict__ part, int slices) {
  __shared__ double s_ap[kMaxAngles];
  __shared__ double sh[4];
  const int s = blockIdx.y, tid = threadIdx.x;
  const Lattice& L = lat[s];
  if (!L.active || L.status != 0) return;
  if (fb_step != 0 && (L.step_x != fb_step || L.step_y != fb_step)) return;
  const int ncand = pc.nx * pc.ny;
  const double center2 = L.center[2];
  for (int a = tid; a < pc.na; a += 256) {
    const double angle = (center2 - pc.ang_off) + (uint32_t)a * pc.ang_res;  // Mapper.cpp:390-393
    const double sad = ksq(angle - center2);
    const double ap = 1.0 - (kAnglePenaltyGain * sad / sc.avp);
    s_ap[a] = ap > sc.min_ap ? ap : sc.min_ap;
  }
  __syncthreads();
  // 64 lattice cells x 4 groups of angles per block: a thread's loads (its angles x the beam slices) are ONE batch in
  // flight, and a lone 101x101 match spreads over 160 blocks instead of 40 (it was a chain of three dependent batches
  // per thread on a sixth of the chip)
  __shared__ double s_m[4][64];
  const int cl = tid & 63, grp = tid >> 6;
  const int c = blockIdx.x * 64 + cl;
  const int a_per = (pc.na + 3) / 4, a_lo = grp * a_per, a_hi = min(pc.na, a_lo + a_per);
  double m = -1.0;
  if (c < ncand) {
    int32_t* r = resp + (size_t)s * resp_stride;
    const int32_t* ps = part + (size_t)s * slices * resp_stride;
    double dp = 1.0;
    if (sc.do_penalize) {
      const int xi = c % pc.nx, yi = c / pc.nx;
      const double x = -pc.off_x + (uint32_t)xi * pc.res_x, y = -pc.off_y + (uint32_t)yi * pc.res_y;
      const double sd = ksq(x) + ksq(y);
      dp = 1.0 - (kDistPenaltyGain * sd / sc.dvp);
      dp = dp > sc.min_dp ? dp : sc.min_dp;
    }
    constexpr int kBatch = 8, kMaxSlices = 8;
    for (int a0 = a_lo; a0 < a_hi; a0 += kBatch) {
      int32_t rv[kBatch];
      if (slices > 1) {  // add up the beam slices of k_resp_dense (exact: integers) and publish the sums as the numerators;
        int32_t pv[kBatch][kMaxSlices];  // all loads of the batch are issued before the first add
#pragma unroll
        for (int i = 0; i < kBatch; i++)
#pragma unroll
          for (int q = 0; q < kMaxSlices; q++)
            pv[i][q] = (a0 + i < a_hi && q < slices) ? ps[(size_t)q * resp_stride + (size_t)(a0 + i) * ncand + c] : 0;
#pragma unroll
        for (int i = 0; i < kBatch; i++) {
          int32_t sum = 0;
#pragma unroll
          for (int q = 0; q < kMaxSlices; q++) sum += pv[i][q];
          rv[i] = sum;
          if (a0 + i < a_hi) r[(size_t)(a0 + i) * ncand + c] = sum;
        }
      } else {
#pragma unroll
        for (int i = 0; i < kBatch; i++) rv[i] = a0 + i < a_hi ? r[(a0 + i) * ncand + c] : 0;
      }
#pragma unroll
      for (int i = 0; i < kBatch; i++)
        if (a0 + i < a_hi) {
          double v = response_of_sum(rv[i], g.n_beams, sc);  // GetResponse normalisation (:852)
          if (sc.do_penalize && !double_equal(v, 0.0)) v *= (dp * s_ap[a0 + i]);
          m = m > v ? m : v;
        }
    }
  }
  s_m[grp][cl] = m;
  __syncthreads();
  if (grp == 0 && c < ncand) {
    double mm = m;
#pragma unroll
    for (int q = 1; q < 4; q++) mm = mm > s_m[q][cl] ? mm : s_m[q][cl];
    scratch[(size_t)s * scratch_stride + c] = mm;
  }
  const double bm = block_max(m, sh, tid, 256);
  if (tid == 0 && bm >= 0.0) atomicMax(&best_bits[s], (unsigned long long)__double_as_longlong(bm));
}

// NT threads per block: 256 for batches (one block per scan, many scans in flight), 1024 for a handful of scans
// (TryCloseLoop's single coarse match: the block is the only parallelism there is)
template <int NT>
__global__ void __launch_bounds__(NT)
k_reduce_coarse_big(Geom g, PassCfg pc, SearchCfg sc, const Lattice* __restrict__ lat,
                    int32_t* resp, size_t resp_stride, CoarseOut* __restrict__ out,
                    int use_expansion, int pass_index, double* __restrict__ scratch, size_t scratch_stride,
                    const uint8_t* __restrict__ grid, const double2* __restrict__ local, int fb_step,
                    const unsigned long long* __restrict__ best_bits, lslam_match_result* __restrict__ final_out,
                    int* done_flag, int done_ticket) {
  // final_out != nullptr: nothing follows this pass (no refinement, no response expansion -- the coarse match of a
  // loop-closure candidate, Mapper.cpp:991): the block writes the match record itself, and posts the caller's ticket behind
  // it, instead of leaving both to a k_reduce_fine launch that would only copy them
  auto publish = [&](const CoarseOut& co) {
    if (!final_out) return;
    lslam_match_result res;
    for (int i = 0; i < 3; i++) res.pose[i] = co.status ? 0.0 : co.mean[i];
    for (int i = 0; i < 9; i++) res.covariance[i] = co.status ? 0.0 : co.cov[i];
    res.response = co.status ? 0.0 : co.best;
    res.status = co.status;
    res.flags = co.flags;
    final_out[blockIdx.x] = res;
    if (done_flag) {
      __threadfence_system();
      *(volatile int*)done_flag = done_ticket;
    }
  };
  constexpr int kList = 2048;
  __shared__ double sh[NT];
  __shared__ double chunk[4 * NT];
  __shared__ int s_list[kList];
  __shared__ int s_nlist, s_status, s_ncell, s_npass;
  __shared__ int s_cells[kList];
  __shared__ unsigned long long s_bal2[16][NT / 64];  // covariance pass: ballots of up to 16 chunks of NT cells
  __shared__ int s_pref[16][NT / 64];
  __shared__ double s_avg[3];
  __shared__ unsigned long long s_bal[NT / 64];
  const int s = blockIdx.x, tid = threadIdx.x;
  const Lattice& L = lat[s];
  // (an inactive lattice belongs to an expansion pass that is not needed or to a fine pass behind a failed coarse one;
  // final_out is only ever set for pass 0 of a match without either, whose lattice is always active -- so no record
  // and no ticket is skipped here)
  if (!L.active) return;
  if (L.status != 0) {
    if (tid == 0) {
      out[s].status = L.status; out[s].expand = 0; out[s].best = 0.0;
      CoarseOut co{};
      co.status = L.status;
      co.flags = pass_index > 0 ? 1 : 0;
      publish(co);
    }
    return;
  }
  block_generic_fallback(grid, g, pc, L, local + (size_t)s * g.n_beams, resp + (size_t)s * resp_stride, fb_step, tid, NT);
  const int ncand = pc.nx * pc.ny, total = ncand * pc.na, words = (total + 31) / 32;
  const int side2 = g.probs_side * g.probs_side;
  double* latmax = scratch + (size_t)s * scratch_stride;
  unsigned long long* probs = (unsigned long long*)(latmax + ncand);
  double* terms = (double*)(probs + side2);
  uint32_t* mask = (uint32_t*)(terms + 4 * (size_t)ncand);
  const int32_t* r = resp + (size_t)s * resp_stride;
  const double center[3] = {L.center[0], L.center[1], L.center[2]};
  // penalized(sum, cand_of(c*nA + a)) (Mapper.cpp:399-414, 852) with the per-cell and per-angle factors hoisted: the
  // distance penalty depends on the cell only, the angle penalty on the angle only, and r *= (dp * ap) is the
  // reference's own grouping -- no integer division and no cos-free recomputation per candidate
  __shared__ double s_ap[kMaxAngles];
  for (int a = tid; a < pc.na; a += NT) {
    const double angle = (center[2] - pc.ang_off) + (uint32_t)a * pc.ang_res;  // :390-393
    const double sad = ksq(angle - center[2]);
    const double ap = 1.0 - (kAnglePenaltyGain * sad / sc.avp);
    s_ap[a] = ap > sc.min_ap ? ap : sc.min_ap;
  }
  __syncthreads();
  auto cell_dp = [&](int c) -> double {
    const int xi = c % pc.nx, yi = c / pc.nx;
    const double x = -pc.off_x + (uint32_t)xi * pc.res_x;  // Mapper.cpp:342-345
    const double y = -pc.off_y + (uint32_t)yi * pc.res_y;  // :353-356
    const double sd = ksq(x) + ksq(y);
    const double dp = 1.0 - (kDistPenaltyGain * sd / sc.dvp);
    return dp > sc.min_dp ? dp : sc.min_dp;
  };
  auto value_of = [&](int32_t sum, double dp, int a) -> double {
    double v = response_of_sum(sum, g.n_beams, sc);
    if (sc.do_penalize && !double_equal(v, 0.0)) v *= (dp * s_ap[a]);
    return v;
  };
  // k_big_latmax has done the per-cell maxima of every scan with a uniform lattice; only a scan whose numerators this
  // block had to compute itself (block_generic_fallback above) still needs them here
  const bool have_latmax = !(fb_step != 0 && (L.step_x != fb_step || L.step_y != fb_step));
  double lm = -1.0;
  for (int c = tid; c < ncand && !have_latmax; c += NT) {
    double m = -1.0;
    const double dp = sc.do_penalize ? cell_dp(c) : 1.0;
    // the numerators of a cell are fetched eight angles at a time (independent loads in flight)
    constexpr int kBatch = 8;
    for (int a0 = 0; a0 < pc.na; a0 += kBatch) {
      int32_t rv[kBatch];
#pragma unroll
      for (int i = 0; i < kBatch; i++) rv[i] = a0 + i < pc.na ? r[(a0 + i) * ncand + c] : 0;
#pragma unroll
      for (int i = 0; i < kBatch; i++)
        if (a0 + i < pc.na) {
          const double v = value_of(rv[i], dp, a0 + i);
          m = m > v ? m : v;
        }
    }
    latmax[c] = m;
    lm = lm > m ? lm : m;
  }
  // Lattice positions are >= 1.5 search-space cells apart (always so on the coarse pass: the step is two cells): the
  // lattice -> search-space cell map (:440) is injective, every cell's probability is the value its own lattice
  // position wrote, and m_pSearchSpaceProbs never has to exist -- no clear, no max-merge atomics, no fence.
  const bool injective = pc.res_x * g.scale >= 1.5 && pc.res_y * g.scale >= 1.5;
  if (!injective)
    for (int c = tid; c < side2; c += NT) probs[c] = 0ull;  // Clear (:329) (+0.0)
  if (tid == 0) { s_nlist = 0; s_status = 0; s_ncell = 0; s_npass = -1; }
  double best = block_max(lm, sh, tid, NT);
  if (have_latmax) best = __longlong_as_double((long long)best_bits[s]);  // >= +0: the unsigned order of the bits is the fp order
  // ties with the best response (:452-455), as candidate indices k = c * nA + a in an LDS list; only when more than
  // kList candidates tie (an all-zero response surface) the bitmap in global memory takes over
  // the few cells whose maximum reaches the best are listed first, then all threads share their (cell, angle) pairs:
  // one load each -- the thread that owns the best cell alone was a chain of nA dependent round trips
  for (int c = tid; c < ncand; c += NT)
    if (!(latmax[c] + kTol < best)) {
      const int pos = atomicAdd(&s_ncell, 1);
      if (pos < kList) s_cells[pos] = c;
    }
  __syncthreads();
  if (s_ncell <= kList) {
    for (int idx = tid, n = s_ncell * pc.na; idx < n; idx += NT) {
      const int c = s_cells[idx / pc.na], a = idx % pc.na;
      if (double_equal(value_of(r[a * ncand + c], sc.do_penalize ? cell_dp(c) : 1.0, a), best)) {
        const int pos = atomicAdd(&s_nlist, 1);
        if (pos < kList) s_list[pos] = c * pc.na + a;
      }
    }
  } else {  // an all-zero surface: every cell
    for (int c = tid; c < ncand; c += NT) {
      if (latmax[c] + kTol < best) continue;
      const double dp = sc.do_penalize ? cell_dp(c) : 1.0;
      for (int a = 0; a < pc.na; a++)
        if (double_equal(value_of(r[a * ncand + c], dp, a), best)) {
          const int pos = atomicAdd(&s_nlist, 1);
          if (pos < kList) s_list[pos] = c * pc.na + a;
        }
    }
  }
  __syncthreads();
  const bool listed = s_nlist <= kList;
  if (!listed) {
    for (int wd = tid; wd < words; wd += NT) mask[wd] = 0u;
    __threadfence();
    __syncthreads();
    for (int c = tid; c < ncand; c += NT) {
      if (latmax[c] + kTol < best) continue;
      const double dp = sc.do_penalize ? cell_dp(c) : 1.0;
      for (int a = 0; a < pc.na; a++)
        if (double_equal(value_of(r[a * ncand + c], dp, a), best)) {
          const int k = c * pc.na + a;
          atomicOr(&mask[k >> 5], 1u << (k & 31));
        }
    }
  }
  // search-space cell of every lattice position (:440); out of the search space -> the reference throws
  const double p_off_x = center[0] - pc.off_x, p_off_y = center[1] - pc.off_y;
  auto probs_cell = [&](int c) -> int {
    const int xi = c % pc.nx, yi = c / pc.nx;
    const double wx = center[0] + (-pc.off_x + (uint32_t)xi * pc.res_x);
    const double wy = center[1] + (-pc.off_y + (uint32_t)yi * pc.res_y);
    const int gx = world_to_grid(wx, p_off_x, g.scale), gy = world_to_grid(wy, p_off_y, g.scale);
    return (gx < 0 || gx >= g.probs_side || gy < 0 || gy >= g.probs_side) ? -1 : gy * g.probs_side + gx;
  };
  if (injective) {
    // world_to_grid is monotone in each coordinate: the lattice leaves the search space iff one of two opposite corners does
    if (tid == 0 && (probs_cell(0) < 0 || probs_cell(ncand - 1) < 0)) s_status = LSLAM_ERR_PROBABILITY_SEARCH;
  } else {
    for (int c = tid; c < ncand; c += NT) {
      const int cell = probs_cell(c);
      if (cell < 0) s_status = LSLAM_ERR_PROBABILITY_SEARCH;
      else  // max-merge is order independent -> parallel integer max
        atomicMax(&probs[cell], (unsigned long long)__double_as_longlong(latmax[c] > 0.0 ? latmax[c] : 0.0));
    }
  }
  // mask / probs were updated with device-scope atomics: agent-scope fence so the plain loads below cannot hit stale
  // lines of this CU's vector L1
  if (!injective || !listed) __threadfence();
  __syncthreads();
  if (tid == 0) {
    int st = s_status;
    double ax = 0, ay = 0, tx = 0, ty = 0;
    int cnt = 0;
    auto visit = [&](int k) {
      Cand cd = cand_of(k, pc, center);
      double h = normalize_angle(cd.angle);
      ax += center[0] + cd.x; ay += center[1] + cd.y;
      double sn, cs;
      sincos(h, &sn, &cs);
      tx += cs; ty += sn;
      cnt++;
    };
    if (listed) {
      const int nl = s_nlist;
      for (int i = 1; i < nl; i++) {  // insertion sort into lattice order: a handful of entries
        int v = s_list[i], j = i - 1;
        while (j >= 0 && s_list[j] > v) { s_list[j + 1] = s_list[j]; j--; }
        s_list[j + 1] = v;
      }
      for (int i = 0; i < nl; i++) visit(s_list[i]);
    } else {
      for (int wd = 0; wd < words; wd++) {
        uint32_t mbits = mask[wd];
        while (mbits) {
          const int bit = __ffs(mbits) - 1;
          mbits &= mbits - 1;
          visit(wd * 32 + bit);
        }
      }
    }
    if (st == 0 && cnt == 0) st = LSLAM_ERR_NO_BEST_POSE;
    if (cnt) { ax /= cnt; ay /= cnt; tx /= cnt; ty /= cnt; }
    s_avg[0] = ax; s_avg[1] = ay; s_avg[2] = cnt ? atan2(ty, tx) : 0.0;
    s_status = st;
  }
  __syncthreads();
  // ComputePositionalCovariance (:573-594): the cells that pass the (best - 0.1) test are few -- every thread evaluates
  // its cell's terms, a ballot picks the passing ones and thread 0 adds them up in lattice order: same cells, same
  // order, same sums as the reference's loop over all cells
  const double dx = s_avg[0] - center[0], dy = s_avg[1] - center[1];
  double axx = 0, axy = 0, ayy = 0, norm = 0;
  const int nch = (ncand + NT - 1) / NT;
  bool done = false;
  if (injective && nch <= 16 && s_status == 0) {
    // every chunk's ballot first (no serial work in between), one exclusive prefix over (chunk, wave), then each passing
    // cell writes its terms at its lattice-order rank and thread 0 adds the list up: a handful of barriers, not two per chunk
    uint32_t mine = 0u;  // bit ch: my cell of chunk ch passes
    for (int ch = 0; ch < nch; ch++) {
      const int c = ch * NT + tid;
      const bool sel = c < ncand && (latmax[c] > 0.0 ? latmax[c] : 0.0) >= (best - 0.1);
      const unsigned long long bal = __ballot(sel);
      if ((tid & 63) == 0) s_bal2[ch][tid >> 6] = bal;
      mine |= (uint32_t)sel << ch;
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int ch = 0; ch < nch; ch++)
        for (int w = 0; w < NT / 64; w++) { s_pref[ch][w] = run; run += __popcll(s_bal2[ch][w]); }
      s_npass = run;
    }
    __syncthreads();
    if (s_npass <= NT) {  // the list fits `chunk`
      for (int ch = 0; ch < nch; ch++) {
        if (!((mine >> ch) & 1u)) continue;
        const int c = ch * NT + tid;
        const int xi = c % pc.nx, yi = c / pc.nx;
        const double x = -pc.off_x + (uint32_t)xi * pc.res_x;
        const double y = -pc.off_y + (uint32_t)yi * pc.res_y;
        const double rr = latmax[c] > 0.0 ? latmax[c] : 0.0;
        const int rank = s_pref[ch][tid >> 6] + __popcll(s_bal2[ch][tid >> 6] & ((1ull << (tid & 63)) - 1ull));
        chunk[4 * rank + 0] = rr;
        chunk[4 * rank + 1] = (ksq(x - dx) * rr);
        chunk[4 * rank + 2] = ((x - dx) * (y - dy) * rr);
        chunk[4 * rank + 3] = (ksq(y - dy) * rr);
      }
      __syncthreads();
      if (tid == 0)
        for (int i = 0, n = s_npass; i < n; i++) {
          norm += chunk[4 * i];
          axx += chunk[4 * i + 1];
          axy += chunk[4 * i + 2];
          ayy += chunk[4 * i + 3];
        }
      done = true;
    }
  }
  for (int c0 = 0; c0 < ncand && !done; c0 += NT) {
    const int c = c0 + tid;
    bool sel = false;
    if (c < ncand) {
      const int xi = c % pc.nx, yi = c / pc.nx;
      const double x = -pc.off_x + (uint32_t)xi * pc.res_x;
      const double y = -pc.off_y + (uint32_t)yi * pc.res_y;
      const int cell = probs_cell(c);
      double rr = 0.0;
      if (cell >= 0) rr = injective ? (latmax[c] > 0.0 ? latmax[c] : 0.0) : __longlong_as_double((long long)probs[cell]);
      sel = rr >= (best - 0.1);
      if (sel) {
        chunk[4 * tid + 0] = rr;
        chunk[4 * tid + 1] = (ksq(x - dx) * rr);
        chunk[4 * tid + 2] = ((x - dx) * (y - dy) * rr);
        chunk[4 * tid + 3] = (ksq(y - dy) * rr);
      }
    }
    const unsigned long long bal = __ballot(sel);
    if ((tid & 63) == 0) s_bal[tid >> 6] = bal;
    __syncthreads();
    if (tid == 0) {
      for (int w = 0; w < NT / 64; w++) {
        unsigned long long mb = s_bal[w];
        while (mb) {
          const int i = w * 64 + __ffsll((long long)mb) - 1;
          mb &= mb - 1;
          norm += chunk[4 * i];
          axx += chunk[4 * i + 1];
          axy += chunk[4 * i + 2];
          ayy += chunk[4 * i + 3];
        }
      }
    }
    __syncthreads();
  }
  if (tid != 0) return;
  CoarseOut o;
  o.status = s_status;
  o.flags = pass_index > 0 ? 1 : 0;
  o.pad = 0;
  double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (o.status == 0) {
    if (best < kTol) {
      cov[0] = kMaxVariance; cov[4] = kMaxVariance; cov[8] = 4 * ksq(pc.ang_res);
    } else {
      if (norm > kTol) {
        double vxx = axx / norm, vxy = axy / norm, vyy = ayy / norm;
        double vthth = 4 * ksq(pc.ang_res);
        double min_xx = 0.1 * ksq(pc.res_x), min_yy = 0.1 * ksq(pc.res_y);
        vxx = vxx > min_xx ? vxx : min_xx;
        vyy = vyy > min_yy ? vyy : min_yy;
        double mult = 1.0 / best;
        cov[0] = vxx * mult; cov[1] = vxy * mult; cov[3] = vxy * mult; cov[4] = vyy * mult;
        cov[8] = vthth;
      }
      if (double_equal(cov[0], 0.0)) cov[0] = kMaxVariance;
      if (double_equal(cov[4], 0.0)) cov[4] = kMaxVariance;
    }
  }
  o.mean[0] = s_avg[0]; o.mean[1] = s_avg[1]; o.mean[2] = s_avg[2];
  for (int i = 0; i < 9; i++) o.cov[i] = cov[i];
  o.best = best > 1.0 ? 1.0 : best;
  o.expand = (use_expansion && o.status == 0 && pass_index < 3 && double_equal(o.best, 0.0)) ? 1 : 0;
  out[s] = o;
  publish(o);
}

// ------------------------------------------------------------------------------------------
// k_reduce_fine: one block per scan: max + tie average of the fine lattice, then
// ComputeAngularCovariance (Mapper.cpp:641-692): nA more response sums at the best cell,
// gathered by the whole block; final result record.  Dynamic LDS: mask words.
// ------------------------------------------------------------------------------------------
// The block's work as a function of (scan s, thread tid of NT): `r` = the scan's fine numerators (global in k_reduce_fine,
// the block's LDS copy in k_match_step), `smem` = the mask words.
template <int NT>
__device__ __forceinline__ void reduce_fine_block(
    const int s, const int tid, const uint8_t* __restrict__ grid, const Geom& g, const PassCfg& pc, const SearchCfg& sc,
    const Lattice* __restrict__ lat, int32_t* r, const double2* __restrict__ local, const CoarseOut* __restrict__ coarse,
    lslam_match_result* __restrict__ out, int do_refine, int fb_step, int* done_flag, int done_ticket, unsigned char* smem) {
  __shared__ double sh[NT];
  __shared__ int32_t asum[kMaxAngles];
  // done_flag (single-scan matches whose record goes to pinned host memory): after the record, a system-scope fence and the
  // caller's ticket -- the host spins on the ticket instead of waiting for the stream (the kernel-completion signal and the
  // runtime's wake-up cost more than the last kernel itself)
  auto signal_done = [&]() {
    if (done_flag) {
      __threadfence_system();
      *(volatile int*)done_flag = done_ticket;
    }
  };
  __shared__ double s_avg[3];
  __shared__ double s_best;
  __shared__ int s_status, s_pos;
  uint32_t* mask = (uint32_t*)smem;
  const CoarseOut& co = coarse[s];
  if (!do_refine || co.status != 0) {
    if (tid == 0) {
      lslam_match_result res;
      for (int i = 0; i < 3; i++) res.pose[i] = co.status ? 0.0 : co.mean[i];
      for (int i = 0; i < 9; i++) res.covariance[i] = co.status ? 0.0 : co.cov[i];
      res.response = co.status ? 0.0 : co.best;
      res.status = co.status;
      res.flags = co.flags;
      out[s] = res;
      signal_done();
    }
    return;
  }
  const Lattice& L = lat[s];
  if (L.status != 0) {
    if (tid == 0) {
      lslam_match_result res;
      memset(&res, 0, sizeof res);
      res.status = L.status;
      res.flags = co.flags;
      out[s] = res;
      signal_done();
    }
    return;
  }
  LSLAM_PHASE_CLOCK(pck);
  block_generic_fallback(grid, g, pc, L, local + (size_t)s * g.n_beams, r, fb_step, tid, NT);
  const int ncand = pc.nx * pc.ny;
  const int total = ncand * pc.na;
  const double center[3] = {L.center[0], L.center[1], L.center[2]};
  // every thread keeps its candidates' penalised responses (the usual fine lattice has 99 of them:
  // one per thread) instead of evaluating the division and the penalty twice
  constexpr int kKeep = 4;
  double mine[kKeep];
  auto value = [&](int k) -> double {
    const int a = k % pc.na, c = k / pc.na;
    return penalized(r[a * ncand + c], cand_of(k, pc, center), center, g.n_beams, sc);
  };
  double lm = -1.0;
#pragma unroll
  for (int i = 0; i < kKeep; i++) {
    const int k = tid + NT * i;
    mine[i] = k < total ? value(k) : -1.0;
    lm = lm > mine[i] ? lm : mine[i];
  }
  for (int k = tid + NT * kKeep; k < total; k += NT) {
    const double v = value(k);
    lm = lm > v ? lm : v;
  }
  const int words = (total + 31) / 32;
  for (int wd = tid; wd < words; wd += NT) mask[wd] = 0u;
  for (int a = tid; a < kMaxAngles; a += NT) asum[a] = 0;
  LSLAM_PHASE_MARK(pck, 0);  // penalised responses of the fine lattice
  const double best = block_max(lm, sh, tid, NT);
  LSLAM_PHASE_MARK(pck, 1);  // block maximum
#pragma unroll
  for (int i = 0; i < kKeep; i++) {
    const int k = tid + NT * i;
    if (k < total && double_equal(mine[i], best)) atomicOr(&mask[k >> 5], 1u << (k & 31));
  }
  for (int k = tid + NT * kKeep; k < total; k += NT)
    if (double_equal(value(k), best)) atomicOr(&mask[k >> 5], 1u << (k & 31));
  __syncthreads();
  if (tid == 0) {
    LSLAM_SERIAL_BEGIN();
    double avg[3] = {0, 0, 0};
    int st = 0;
    if (tie_average(mask, total, pc, center, avg) == 0) st = LSLAM_ERR_NO_BEST_POSE;
    int pos = 0;
    if (st == 0) {  // Mapper.cpp:653-654
      int gx = world_to_grid(avg[0], g.off_x, g.scale) + g.border;
      int gy = world_to_grid(avg[1], g.off_y, g.scale) + g.border;
      if (gx < 0 || gx >= g.width || gy < 0 || gy >= g.height) st = LSLAM_ERR_INDEX_OUT_OF_RANGE;
      pos = gx + gy * g.stride;
    }
    s_avg[0] = avg[0]; s_avg[1] = avg[1]; s_avg[2] = avg[2];
    s_best = best;
    s_status = st;
    s_pos = pos;
    LSLAM_SERIAL_END();
  }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 2);  // tie mask + ordered tie average on thread 0
  if (s_status == 0) {
    // GetResponse(angleIndex, gridIndex) for every fine angle at the best cell (:663-666).
    // Usually that cell is one of the fine lattice cells whose numerators are already there.
    int hit = -1;
    for (int yi = 0, c = 0; yi < pc.ny; yi++) {
      const int row = L.gy[yi] * g.stride;
      for (int xi = 0; xi < pc.nx; xi++, c++)
        if (L.gx[xi] + row == s_pos) hit = c;
    }
    if (hit >= 0) {
      for (int a = tid; a < pc.na; a += NT) asum[a] = r[a * ncand + hit];
    } else {
      const double2* lp = local + (size_t)s * g.n_beams;
      const int pos = s_pos;
      for (int a = 0; a < pc.na; a++) {
        double angle = (center[2] - pc.ang_off) + (uint32_t)a * pc.ang_res;
        double cosine = cos(angle), sine = sin(angle);
        int32_t part = 0;
        for (int b = tid; b < g.n_beams; b += NT) {
          double2 p = lp[b];
          if (isnan(p.x)) continue;
          int t = lookup_offset(p.x, p.y, cosine, sine, g.off_x, g.off_y, g.scale, g.stride);
          long long idx = (long long)pos + t;
          if (idx >= 0 && idx < g.data_size) part += grid[idx];
        }
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o);
        if ((tid & 63) == 0) atomicAdd(&asum[a], part);
      }
    }
  }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 3);  // angular-covariance numerators at the best cell
  LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 4, (unsigned)(s * (NT >> 6) + (tid >> 6)), (tid & 63) == 0 && tid != 0);
  if (tid != 0) return;
  LSLAM_SERIAL_BEGIN();
  lslam_match_result res;
  memset(&res, 0, sizeof res);
  res.flags = co.flags;
  res.status = s_status;
  if (s_status == 0) {
    double bestAngle = normalize_angle_difference(s_avg[2], center[2]);  // :651
    double norm = 0.0, acc = 0.0;
    double start = center[2] - pc.ang_off;
    for (int a = 0; a < pc.na; a++) {
      double angle = start + (uint32_t)a * pc.ang_res;
      double rr = response_of_sum(asum[a], g.n_beams, sc);
      if (rr >= (s_best - 0.1)) {
        norm += rr;
        acc += (ksq(angle - bestAngle) * rr);
      }
    }
    if (norm > kTol) {
      if (acc < kTol) acc = ksq(pc.ang_res);
      acc /= norm;
    } else {
      acc = 1000 * ksq(pc.ang_res);
    }
    for (int i = 0; i < 9; i++) res.covariance[i] = co.cov[i];  // NOTE: covariance is not reset (:648)
    res.covariance[8] = acc;
    res.pose[0] = s_avg[0]; res.pose[1] = s_avg[1]; res.pose[2] = s_avg[2];
    res.response = s_best > 1.0 ? 1.0 : s_best;
  }
  out[s] = res;
  signal_done();
  LSLAM_PHASE_MARK(pck, 4);  // angular covariance + the record + ticket (thread 0)
  LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 4, (unsigned)(s * (NT >> 6)), true);
}

template <int NT>  // threads per block: 128 for chip-filling batches (residency), 256 otherwise; see k_reduce_coarse_lds
__global__ void __launch_bounds__(NT)
k_reduce_fine(const uint8_t* __restrict__ grid, Geom g, PassCfg pc, SearchCfg sc,
              const Lattice* __restrict__ lat, int32_t* resp, size_t resp_stride,
              const double2* __restrict__ local, const CoarseOut* __restrict__ coarse,
              lslam_match_result* __restrict__ out, int do_refine, int fb_step, int* done_flag, int done_ticket) {
  extern __shared__ __align__(16) unsigned char smem[];
  reduce_fine_block<NT>((int)blockIdx.x, (int)threadIdx.x, grid, g, pc, sc, lat, resp + (size_t)blockIdx.x * resp_stride, local,
                        coarse, out, do_refine, fb_step, done_flag, done_ticket, smem);
}

// ------------------------------------------------------------------------------------------
// k_match_step -- ONE LAUNCH PER STEP: a scan-resident workgroup (round 6).
// The five kernels of a batched match are a dependent chain per SCAN only -- scans never exchange anything -- so a
// workgroup of WAVES waves takes ONE scan through all of it with __syncthreads() between the phases:
//   0  scan_prep_block        ranges -> scan-frame points (+ coarse lattice, cos/sin of the angles)      [all threads]
//   1  resp_rows_wave         the nA coarse angles, one angle per wave at a time, numerators -> LDS      [wave = angle]
//   2  reduce_coarse_lds_block penalties, max, tie average, covariance, fine lattice                     [all threads]
//   3  resp_tile3_wave        the fine angles, kStepFineAngles per wave, numerators -> LDS                [wave = angles]
//   4  reduce_fine_block      max, tie average, angular covariance, the 112-byte record                  [all threads]
// Same device functions, same arguments, same order of every rounding as the five-kernel path: records are byte-identical.
// What changes is where things live and what overlaps: the int32 numerators (10 KB per scan) never leave LDS (the
// five-kernel step writes and re-reads 41.6 MB of them per 4096 scans), four launches and their drain / fill gaps are gone,
// and the latency-bound phases of one scan run under the gathers of the other scans resident on the same SIMDs -- each
// block is in its own phase -- which stream pipelining could not arrange (profiles/r05/experiments/pipe_lockstep).
// LDS per block: numerators + max(WAVES x 2.4 KB of wave areas, the reduce scratch) ~ 21-24 KB; 128 VGPRs -> the waves of
// 16 / WAVES blocks per CU.  The reference has no counterpart (one ScanMatcher, one scan at a time: Mapper.cpp:184-291).
// ------------------------------------------------------------------------------------------
constexpr int kStepFineAngles = 4;  // fine angles per wave in phase 3 (11 fine angles = 4 + 4 + 3: three balanced items)
template <int NXD>
constexpr int step_wave_area() { return rows_wave_area<NXD>(); }  // bytes of LDS one wave of phase 1 owns
struct StepGrid {  // the read-only views of the correlation grid the phases gather from
  const uint8_t* ptiles;     // tiled parity planes (k_tile_planes)
  const uint2* occ;          // row-occupancy word pairs (k_occ_pairs); nullptr = no pruning
  const uint4* tiles4;       // overlapping 4x4 blocks (k_tile4)
  const uint8_t* grid;       // the grid itself (fallback paths of the reduce blocks)
  int limit, occ_wpc, ptile_rows, tile4_cols;
  uint32_t ptile_class_bytes;
};
template <int NXD, int NYC, int WAVES, typename RT>
__global__ void __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(4, 8)))
k_match_step(const RT* __restrict__ ranges, int stride, const double* __restrict__ poses, Geom g, PassCfg pc, PassCfg pf,
             SearchCfg sc, StepGrid v, Lattice* lat, double2* cossin, double2* local, CoarseOut* coarse,
             lslam_match_result* __restrict__ out, int parts, uint32_t phase_off, int32_t* dbg_coarse, int32_t* dbg_fine,
             size_t dbg_stride) {
  constexpr int NT = 64 * WAVES;
  constexpr int kArea = step_wave_area<NXD>();
  extern __shared__ __align__(16) unsigned char smem[];
  const int s = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int32_t* const num = (int32_t*)smem;        // the scan's numerators, angle-major: coarse pass, then fine pass
  unsigned char* const phase = smem + phase_off;  // wave areas (phase 1) / reduce scratch (phases 2, 4)

  // 0: LocalizedRangeScan::Update + InverseTransformPose (Karto.h:5384-5388, 6423-6434), coarse lattice (Mapper.cpp:339-393)
  scan_prep_block(ranges, stride, poses[3 * s], poses[3 * s + 1], poses[3 * s + 2], g, local, (double2*)nullptr, pc, lat, cossin, 2,
                  0, 1, s);
  __syncthreads();
  // 1: coarse response numerators (Mapper.cpp:373-424, 819-856)
  {
    unsigned char* wa = phase + wave * kArea;
    const RowsLds lds{(uint32_t (*)[8])wa, (int2*)wa, nullptr, nullptr,
                      (unsigned long long*)(wa + (kArea - kMaxBeamsPerLane * 8))};
    for (int a = wave; a < pc.na; a += WAVES)
      resp_rows_wave<NXD, NYC, true, false, false, false>(s, a, 0, v.ptiles, v.ptiles, 2, v.limit, g, pc, lat, cossin, local,
                                                          (int32_t*)nullptr, 0, 1, v.occ, v.occ_wpc, v.ptile_rows,
                                                          v.ptile_class_bytes, (unsigned long long*)nullptr, lds, num);
  }
  __syncthreads();
  // 2: Mapper.cpp:431-501, 535-630 (+ the fine lattice around the mean: :276-281)
  reduce_coarse_lds_block<NT>(s, tid, g, pc, sc, lat, num, coarse, 0, 0, v.grid, local, 2, pf, cossin, 1, 0, parts, phase);
  __syncthreads();
  if (dbg_coarse) {  // diagnostics (lslam_matcher_debug_coarse_sums_batch): the numerators as the reduce saw them
    for (int i = tid, n = pc.nx * pc.ny * pc.na; i < n; i += NT) dbg_coarse[(size_t)s * dbg_stride + i] = num[i];
    __syncthreads();
  }
  // 3: fine response numerators, 3 x 3 x nA (same function, one-cell steps)
  for (int a0 = wave * kStepFineAngles; a0 < pf.na; a0 += WAVES * kStepFineAngles)
    resp_tile3_wave<kStepFineAngles>(s, a0, lane, v.tiles4, v.tile4_cols, g, pf, lat, cossin, local, num);
  __syncthreads();
  // 4: Mapper.cpp:431-506, 641-692, the record
  reduce_fine_block<NT>(s, tid, v.grid, g, pf, sc, lat, num, local, coarse, out, 1, 1, (int*)nullptr, 0, phase);
  if (dbg_fine) {
    __syncthreads();
    for (int i = tid, n = pf.nx * pf.ny * pf.na; i < n; i += NT) dbg_fine[(size_t)s * dbg_stride + i] = num[i];
  }
}

// ------------------------------------------------------------------------------------------
// k_match_lone -- the four kernels of ONE scan's match (coarse responses, coarse reduce, fine responses, fine reduce) as
// ONE launch with device-side hand-overs (round 6; profiles/r06/lone_chain_phases.md).
// A lone MatchScan is a chain of small dependent launches, and a dependent launch costs 4-5 us on this stack against
// 3-11 us of work in each kernel.  Moving the chain onto ONE CU loses (k_match_tail: the fine pass does not fit there), so
// this kernel keeps the chain's SHAPE and replaces the launches by words in device memory:
//   * the response passes want one wave per CU (a wave's gathers are bound by its CU's lookup rate: four waves on a CU
//     take twice as long as four CUs), the reduces want a wide block (1024 threads: a cell's angles over 8 threads).  So a
//     block is WAVES waves of which only the first TW draw (angle, beam slice) tasks -- the others go straight to the barrier
//     and cost nothing there -- and the launch is as many blocks as there are tasks / TW;
//   * every block bumps `c_done` when its tasks are done (release); the block that arrives LAST runs the coarse reduce with all
//     its waves (nobody waits for it to start) and posts `fine_ready`; the other blocks wait on that one word (bounded; s_sleep
//     between polls); fine tasks the same way, the last block to arrive runs the fine reduce and writes the record (+ the
//     caller's ticket).
// One wait per block, bounded by the coarse reduce.  A wait that does not end (it would take a grid that does not fit the
// chip -- it is at most 168 blocks, one per CU -- and a dispatcher that starts nothing new) gives up after ~40 ms, the record
// carries LSLAM_ERR_HIP, nothing hangs.  The words of a launch live in slot seq % kLoneRing of a ring; the last block clears
// the slot half a ring ahead (launches of a matcher are ordered on its stream).  Same wave functions, same beam slices, same
// atomically accumulated integers as the four-kernel chain: records are byte-identical.
// (Tried and dropped: tasks drawn from a ticket counter -- two more dependent atomics per wave and pass, +6 us; all
//  participants on ONE XCD so that a hand-over needs no L2 write-back / invalidate -- 168 tasks on 32 CUs, +12 us.)
// ------------------------------------------------------------------------------------------
struct LoneSync {
  unsigned c_done, fine_ready, f_done, timeouts, pad[4];
};
constexpr int kLoneRing = 16;
// the word as memory holds it: a compare-and-swap that changes nothing (never served from an L1, never folded into a load)
__device__ __forceinline__ unsigned lone_peek(unsigned* w) {
  unsigned expected = 0xFFFFFFFFu;
  __hip_atomic_compare_exchange_strong(w, &expected, 0xFFFFFFFFu, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expected;
}
__device__ __forceinline__ bool lone_wait(unsigned* w) {
  for (int i = 0; i < (1 << 15); i++) {  // ~40 ms; the wait is for one block's coarse reduce (~10 us)
    if (lone_peek(w) != 0u) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  return false;
}
// the hand-overs' fences, as weak as each side allows (a full agent-scope fence is an L2 write-back AND an invalidate: ~2.5 us)
#if defined(LSLAM_EXP_LONE_FULL_FENCES)  // A/B switch: the first draft
#define LONE_ARRIVE_FENCE() __threadfence()
#define LONE_ACQUIRE() __threadfence()
#define LONE_RELEASE() __threadfence()
#else
#define LONE_ARRIVE_FENCE() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define LONE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#define LONE_RELEASE() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#endif
template <int NXD, int NYC, int WAVES, int TW>
__global__ void __launch_bounds__(64 * WAVES)
k_match_lone(const uint8_t* sub0, const uint8_t* sub1, const uint8_t* grid, Geom g, PassCfg pc, PassCfg pf, SearchCfg sc,
             Lattice* lat, double2* cossin, double2* local, int32_t* resp, CoarseOut* coarse, lslam_match_result* out,
             const uint2* occ, int occ_wpc, int c_slices, int f_slices, int parts, LoneSync* ring, unsigned seq,
             int* done_flag, int done_ticket) {
  constexpr int NT = 64 * WAVES;
  constexpr int kArea = rows_wave_area<NXD>();
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int s_last;
  const int tid = threadIdx.x, wave = tid >> 6;
  LoneSync* const sy = ring + (seq % kLoneRing);
  const int nwaves = (int)gridDim.x * TW, w = (int)blockIdx.x * TW + wave;
  LSLAM_PHASE_CLOCK(pck);

  // 1: coarse response numerators (Mapper.cpp:373-424, 819-856), one (angle, beam slice) per task wave
  if (wave < TW) {
    unsigned char* wa = smem + wave * kArea;
    const RowsLds lds{(uint32_t (*)[8])wa, (int2*)wa, nullptr, nullptr, nullptr};
    for (int t = w; t < pc.na * c_slices; t += nwaves)
      resp_rows_wave<NXD, NYC, false, false, false, false, false>(0, t / c_slices, t % c_slices, sub0, sub1, 2, g.data_size / 2, g, pc,
                                                                  lat, cossin, local, resp, 0, c_slices, occ, occ_wpc, 0, 0u,
                                                                  (unsigned long long*)nullptr, lds, (int32_t*)nullptr);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomics are performed
  LSLAM_PHASE_MARK(pck, 0);  // coarse tasks
  __syncthreads();
  if (tid == 0) {
    LONE_ARRIVE_FENCE();  // this block published nothing but atomics: they are performed (every wave waited before the barrier)
    s_last = __hip_atomic_fetch_add(&sy->c_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 1);  // the block's slowest wave, release fence, arrival
  if (s_last) {
    // 2: Mapper.cpp:431-501, 535-630 (+ the fine lattice around the mean, the fine numerators cleared)
    LONE_ACQUIRE();  // every block's numerators
    reduce_coarse_lds_block<NT>(0, tid, g, pc, sc, lat, resp, coarse, 0, 0, grid, local, 2, pf, cossin, 1, pf.nx * pf.ny * pf.na,
                                parts, smem);
    LONE_RELEASE();  // the fine lattice, cos/sin, the cleared numerators, the coarse record
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&sy->fine_ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if (tid == 0 && !lone_wait(&sy->fine_ready)) __hip_atomic_fetch_add(&sy->timeouts, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    LONE_ACQUIRE();  // the fine lattice, cos/sin, the cleared numerators
  }
  LSLAM_PHASE_MARK(pck, 2);  // the coarse reduce (last block) | the wait for it + acquire fence (the others)
  // 3: fine response numerators, one-cell steps on the grid itself
  if (wave < TW) {
    unsigned char* wa = smem + wave * kArea;
    const RowsLds lds{(uint32_t (*)[8])wa, (int2*)wa, nullptr, nullptr, nullptr};
    for (int t = w; t < pf.na * f_slices; t += nwaves)
      resp_rows_wave<1, 4, false, false, false, false, false>(0, t / f_slices, t % f_slices, grid, grid, 1, g.data_size, g, pf, lat,
                                                              cossin, local, resp, 0, f_slices, (const uint2*)nullptr, occ_wpc, 0, 0u,
                                                              (unsigned long long*)nullptr, lds, (int32_t*)nullptr);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LSLAM_PHASE_MARK(pck, 3);  // fine tasks
  __syncthreads();
  if (tid == 0) {
    LONE_ARRIVE_FENCE();
    s_last = __hip_atomic_fetch_add(&sy->f_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
  }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 4);  // slowest wave, release fence, arrival
  LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 7, (unsigned)blockIdx.x, tid == 0);
  if (!s_last) return;
  // 4: Mapper.cpp:431-506, 641-692, the record
  LONE_ACQUIRE();
  const bool timed_out = lone_peek(&sy->timeouts) != 0u;
  reduce_fine_block<NT>(0, tid, grid, g, pf, sc, lat, resp, local, coarse, out, 1, 1, timed_out ? (int*)nullptr : done_flag,
                        done_ticket, smem);
  if (tid < (int)(sizeof(LoneSync) / 4)) ((unsigned*)(ring + ((seq + kLoneRing / 2) % kLoneRing)))[tid] = 0u;
  if (timed_out) {  // a hand-over that did not arrive: the record must not pass for a result
    __syncthreads();
    if (tid == 0) {
      out[0].status = LSLAM_ERR_HIP;
      if (done_flag) {
        __threadfence_system();
        *(volatile int*)done_flag = done_ticket;
      }
    }
  }
}

// result for a laser with zero beams (Mapper.cpp:199-209)
__global__ void k_result_no_readings(int S, const double* poses, double coarse_ang_res,
                                     lslam_match_result* out) {
  int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S) return;
  lslam_match_result r;
  memset(&r, 0, sizeof r);
  r.pose[0] = poses[3 * s]; r.pose[1] = poses[3 * s + 1]; r.pose[2] = poses[3 * s + 2];
  r.covariance[0] = kMaxVariance; r.covariance[4] = kMaxVariance;
  r.covariance[8] = 4 * ksq(coarse_ang_res);
  out[s] = r;
}

// First kernel of a grid rebuild: Grid::Clear (Mapper.cpp:701) and, for the streaming front-end, two small jobs that
// would otherwise each be a separate copy / fill on the stream of a latency-bound chain: the query scan's sensor pose
// (a kernel argument) into device memory, zeros over the response numerators of the match that follows (its
// beam-sliced passes accumulate with atomics), and the new scan's readings from pinned host memory into their row.
struct RebuildExtras {
  double pose[3];
  double* pose_dst;          // nullptr: nothing to write
  int32_t* zero;             // nullptr: nothing to clear
  int zero_words;
  const double* ranges_src;  // nullptr, or the new scan's readings in pinned host memory (read over the bus) ...
  double* ranges_dst;        // ... and their resident row in HBM
  int n_ranges;
  const int* anchor_ring;    // nullptr, or k_anchor_chain's rows (n + 1 ints) of every resident scan, same ring as the world points
  const int* slot_list;      // nullptr: window entry b lives in slot (ring_start + b) % cap; else in slot_list[b] (scan cache:
                             // the base scans of a MatchScan are arbitrary resident scans; the list may be pinned host memory)
  // k_scan_prep of the ONE query scan of the match that follows (its pose is `pose`), run by extra blocks of the same
  // launch: the prep needs nothing of the grid, so it runs beside the clear instead of behind the whole rebuild
  const double* prep_ranges;  // nullptr: no prep
  double2* prep_local;
  Lattice* prep_lat;
  double2* prep_cossin;
  PassCfg prep_pc;
  Geom prep_g;
  int clear_blocks;
};
__global__ void __launch_bounds__(256)
k_rebuild_begin(uint4* __restrict__ grid16, size_t n16, RebuildExtras x) {
  if (x.prep_ranges && (int)blockIdx.x >= x.clear_blocks) {
    scan_prep_block(x.prep_ranges, x.prep_g.n_beams, x.pose[0], x.pose[1], x.pose[2], x.prep_g, x.prep_local,
                    (double2*)nullptr, x.prep_pc, x.prep_lat, x.prep_cossin, 2, (int)blockIdx.x - x.clear_blocks,
                    (int)gridDim.x - x.clear_blocks, 0);
    return;
  }
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n16) grid16[i] = make_uint4(0u, 0u, 0u, 0u);
  if (x.zero && i < (size_t)x.zero_words) x.zero[i] = 0;
  if (x.pose_dst && i < 3) x.pose_dst[i] = x.pose[i];
  if (x.ranges_src && i < (size_t)x.n_ranges) x.ranges_dst[i] = x.ranges_src[i];
}

// ------------------------------------------------------------------------------------------
// correlation-grid construction (AddScans, Mapper.cpp:699-748)
// ------------------------------------------------------------------------------------------
// FindValidPoints (Mapper.cpp:756-811) is a sequential scan with a lagging iterator.  One wave per
// base scan, points staged in LDS (coalesced); valid[] marks the points the reference would
// push_back.  Base scans live in a ring of `cap` slots (streaming front-end); slot of window entry
// b = (ring_start + b) % cap.
// Parallel form of the same walk.  The reference's loop only carries two things from one point to
// the next: the anchor `firstPoint` (replaced by the first later point farther than 10 cm) and the
// lagging iterator, which always ends up at the new anchor; the run [old anchor, new anchor) is kept
// iff the side test ss >= 0 (the very first run starts at index 0).  So:
//   (1) every thread finds the successor next[i] of "its" point as if it were an anchor,
//   (2) the anchors = the points reachable from the first valid point through next[]: pointer
//       doubling marks them in ceil(log2 n) rounds (LDS path; the global-scratch path for very long
//       scans walks the chain on one thread),
//   (3) every anchor evaluates its side test and marks its run [anchor, next[anchor]) -- in parallel.
// Every fp64 expression is the reference's.
// first later point farther than 10 cm from point i (Mapper.cpp:780-781), n if none; four candidates per step so that
// their LDS reads are in flight together (a block waits for its slowest thread: a wall 0.3 m away is 80 points away)
__device__ __forceinline__ int successor_of(const double2* p, int n, int i, double min_sq) {
  const double fx = p[i].x, fy = p[i].y;
  int j = i + 1;
  for (bool found = false; !found && j < n;) {
    double2 q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) q[u] = p[min(j + u, n - 1)];
    int hit = 4;
#pragma unroll
    for (int u = 3; u >= 0; u--) {
      const double dx = fx - q[u].x, dy = fy - q[u].y;
      if (ksq(dx) + ksq(dy) > min_sq) hit = u;
    }
    if (hit < 4 && j + hit < n) { j += hit; found = true; }
    else j = min(j + 4, n);
  }
  return j;
}
// reach[] = the points reachable from `first` through the successor table ja (= the anchors), by pointer doubling:
// round k marks chain distances [2^k, 2^(k+1)).  ja / jb are overwritten.  Ends with a barrier.
__device__ __forceinline__ void mark_reachable(int n, int first, int* ja, int* jb, uint8_t* reach, int tid, int nt) {
  if (tid == 0 && first < n) reach[first] = 1;
  __syncthreads();
  for (int span = 1; span < n; span <<= 1) {
    for (int i = tid; i < n; i += nt) {
      const int j = ja[i];
      if (reach[i] && j < n) reach[j] = 1;
      jb[i] = j < n ? ja[j] : n;
    }
    __syncthreads();
    int* t = ja; ja = jb; jb = t;
  }
}

__global__ void __launch_bounds__(1024)
k_find_valid(int n, const double2* __restrict__ world, int ring_start, int cap, double vx, double vy,
             uint8_t* __restrict__ valid, int use_lds, int* __restrict__ scratch, Geom g, uint8_t* __restrict__ grid,
             const int* __restrict__ anchor_ring, int mark_value, int n_scans, RebuildExtras x) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int s_first, s_len;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  if (b >= n_scans) {
    // Extra blocks behind the base scans (streaming front-end, clear-free rebuild): the jobs of k_rebuild_begin that
    // nothing in this launch depends on -- the query scan's prep, its pose and readings into device memory, zeros over
    // the next match's sliced numerators -- so the rebuild needs no launch of its own for them.
    const int e = b - n_scans, prep_blocks = x.prep_ranges ? (x.prep_g.n_beams + nt - 1) / nt : 0;
    if (e < prep_blocks) {
      scan_prep_block(x.prep_ranges, x.prep_g.n_beams, x.pose[0], x.pose[1], x.pose[2], x.prep_g, x.prep_local,
                      (double2*)nullptr, x.prep_pc, x.prep_lat, x.prep_cossin, 2, e, prep_blocks, 0);
    } else {
      if (x.zero)
        for (int i = tid; i < x.zero_words; i += nt) x.zero[i] = 0;
      if (x.ranges_src)
        for (int i = tid; i < x.n_ranges; i += nt) x.ranges_dst[i] = x.ranges_src[i];
      if (x.pose_dst && tid < 3) x.pose_dst[tid] = x.pose[tid];
    }
    return;
  }
  LSLAM_PHASE_CLOCK(pck);
  const int slot = x.slot_list ? x.slot_list[b] : (ring_start + b) % cap;
  const double2* gp = world + (size_t)slot * n;
  uint8_t* gv = valid + (size_t)b * n;
  const double2* p = gp;
  uint8_t* v = gv;
  int *next, *chain, *jump2 = nullptr;
  uint8_t* reach = nullptr;
  if (use_lds) {
    double2* lp = (double2*)smem;
    next = (int*)(lp + n);
    chain = next + n;   // jump table A
    jump2 = chain + n;  // jump table B
    uint8_t* lv = (uint8_t*)(jump2 + n);
    reach = lv + n;
    for (int i = tid; i < n; i += nt) { lp[i] = gp[i]; lv[i] = 0; reach[i] = 0; }
    p = lp;
    v = lv;
  } else {
    next = scratch + (size_t)b * 2 * n;
    chain = next + n;
    for (int i = tid; i < n; i += nt) gv[i] = 0;
  }
  if (tid == 0) { s_first = n; s_len = 0; }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 0);  // world points of the scan into LDS
  // the side test of anchor a against its successor f, and the run it keeps (:788-806)
  auto keep_run = [&](int a, int f, bool first) {
    const double fx = p[a].x, fy = p[a].y, cx = p[f].x, cy = p[f].y;
    const double aa = vy - fy;
    const double bb = fx - vx;
    const double cc = fy * vx - fx * vy;
    const double ss = cx * aa + cy * bb + cc;
    if (!(ss < 0.0))
      for (int t = (first ? 0 : a); t < f; t++) v[t] = 1;
  };
  // The anchors depend on the scan's points only, not on the viewpoint: the streaming front-end lists them once per
  // (re)posed scan (k_anchor_chain: row = count, anchors in order) instead of once per rebuild of every window the scan
  // is part of; only the side tests and the marking are left here.
  const int* ain = (anchor_ring && use_lds) ? anchor_ring + (size_t)slot * (n + 1) : nullptr;
  if (ain) {
    const int cnt = ain[0];
    for (int k = tid; k + 1 < cnt; k += nt) keep_run(ain[1 + k], ain[2 + k], k == 0);
  } else {
    const double min_sq = ksq(0.1);
    for (int i = tid; i < n; i += nt) {
      const double fx = p[i].x, fy = p[i].y;
      first_valid_min(&s_first, !isnan(fx) && !isnan(fy), i);
      next[i] = successor_of(p, n, i, min_sq);
      if (use_lds) chain[i] = next[i];
    }
    __syncthreads();
    if (use_lds) {
      const int first = s_first;
      mark_reachable(n, first, chain, jump2, reach, tid, nt);
      for (int a = tid; a < n; a += nt)
        if (reach[a] && next[a] < n) keep_run(a, next[a], a == first);
    } else {
      if (tid == 0) {
        int len = 0;
        for (int a = s_first; a < n; a = next[a]) chain[len++] = a;
        s_len = len;
      }
      __syncthreads();
      const int len = s_len;
      for (int k = tid; k + 1 < len; k += nt) keep_run(chain[k], chain[k + 1], k == 0);
    }
  }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 1);  // anchors' side tests, runs marked valid
  if (use_lds)
    for (int i = tid; i < n; i += nt) gv[i] = v[i];
  // Fused AddScan stage (the streaming front-end rebuilds the grid once per scan, so a launch and the valid[] round trip
  // through memory matter): each valid point tags its cell in the MARK plane with a PLAIN byte store of this rebuild's
  // epoch (1..255) -- every writer writes the same value, which point got there first does not matter because
  // k_smear_gather finds the centres in the plane itself, and neither the plane nor the grid is ever cleared (the gather
  // writes every grid byte; the plane is reset when the epoch wraps).  (Device-scope atomics that report the winner are executed at the memory side: ~40 k of them per rebuild,
  // most on cells other scans of the window had already set, were two thirds of this kernel's time.)
  if (grid && use_lds) {
    for (int i0 = 0; i0 < n; i0 += nt) {  // whole waves iterate together: the shuffle below needs every lane
      const int i = i0 + tid;
      uint32_t idx = 0xFFFFFFFFu;
      if (i < n && v[i]) {
        const double2 q = p[i];
        const int gx = world_to_grid(q.x, g.off_x, g.scale);
        const int gy = world_to_grid(q.y, g.off_y, g.scale);
        if (gx >= 0 && gx < g.roi_w && gy >= 0 && gy < g.roi_h)  // IsUpTo on the ROI (Mapper.cpp:724-729)
          idx = (uint32_t)((gx + g.border) + (gy + g.border) * g.stride);
      }
      const uint32_t left = (uint32_t)__shfl_up((int)idx, 1);  // neighbouring beams mostly hit the same cell
      if (idx != 0xFFFFFFFFu && ((tid & 63) == 0 || left != idx)) grid[idx] = (uint8_t)mark_value;
    }
  }
  LSLAM_PHASE_MARK(pck, 2);  // valid[] out, cells of the valid points tagged in the mark plane
  LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 0, (unsigned)(b * (nt >> 6) + (tid >> 6)), (tid & 63) == 0);
}

// The anchor chain of FindValidPoints for ONE scan (see k_find_valid), as row = [count, anchor indices in order]: launched
// by the streaming front-end when a scan's world points are (re)computed, off the per-scan critical path.
// With `ranges` it first evaluates the world points themselves (k_scan_prep's world branch: LocalizedRangeScan::Update,
// Karto.h:5384-5388, at the pose passed as a kernel argument) and stores them: one launch for both.
//
// SPECULATIVE ANCHORS (round 6).  The chain is 11 of this kernel's 13 us and sits between a match and the next rebuild.  But
// which point follows which depends on the scan's SHAPE only -- distances between its points -- and a rigid motion changes a
// squared distance by no more than the roundings of the coordinates (~3e-14 m^2 at 100 m).  So the chain is worked out EARLY,
// on the points at the pose the match STARTS from (anchor_spec_block: an extra block of the lone coarse reduce's launch, where
// it costs nothing), with every comparison that comes closer to the 0.1 m threshold than kSpecBand flagged, and this kernel
// -- which still evaluates the world points at the FINAL pose -- takes the speculative row over when (a) nothing was flagged
// and (b) the non-finite points cannot change the walk.  A comparison that involves a non-finite point is decided by the
// points' classes (finite / +inf / -inf / NaN per coordinate) alone: finite against infinite is always "farther" (inf^2), NaN
// never is, and two infinite points are farther only when BOTH coordinates differ in sign -- opposite quadrants, i.e. beams a
// quarter turn apart with nothing finite between them.  A scan whose non-finite readings together span less than a quadrant
// (1.5 rad of beams) therefore has the same chain at every heading, and only the KIND of each point (finite / infinite / NaN:
// a function of the reading alone) is checked; a scan with more of them is taken over only if every point's class, signs
// included, is the predicted one.  Otherwise the kernel computes the chain as before.  Either way the row is the one the
// reference's walk produces on the final points.
constexpr double kSpecBand = 1e-13;  // x (1 + 2 x the scan's largest |x| + |y|): >= 40 x the bound on |d^2(pose A) - d^2(pose B)|
__device__ __forceinline__ uint32_t point_class(double2 q) {
  auto c = [](double v) -> uint32_t { return isnan(v) ? 3u : isinf(v) ? (v > 0.0 ? 1u : 2u) : 0u; };
  return c(q.x) | (c(q.y) << 2);
}
// ordered compaction of the points marked in reach[] into row = [count, indices...]; ends with the row complete
__device__ __forceinline__ void anchors_to_row(int n, const uint8_t* reach, int* __restrict__ row, int* s_wc, int tid, int nt) {
  int base = 0;
  for (int i0 = 0; i0 < n; i0 += nt) {
    const int i = i0 + tid;
    const bool is_anchor = i < n && reach[i];
    const unsigned long long bal = __ballot(is_anchor);
    if ((tid & 63) == 0) s_wc[tid >> 6] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < (tid >> 6); w++) off += s_wc[w];
    for (int w = 0; w < (nt >> 6); w++) base += s_wc[w];
    if (is_anchor) row[1 + off + __popcll(bal & ((1ull << (tid & 63)) - 1ull))] = i;
    __syncthreads();
  }
  if (tid == 0) row[0] = base;
}
// spec = [ok, count, anchors (n), classes (n bytes)] (anchor_spec_block) or nullptr
__device__ __forceinline__ void anchor_chain_block(int n, double2* __restrict__ world, int* __restrict__ row,
                                                   const double* __restrict__ ranges, const PoseArg& pose, const Geom& g,
                                                   const int* __restrict__ spec) {
  extern __shared__ __align__(16) unsigned char smem[];
  __shared__ int s_first, s_wc[16], s_spec_ok;
  double2* p = (double2*)smem;
  int* next = (int*)(p + n);
  int* ja = next + n;
  int* jb = ja + n;
  uint8_t* reach = (uint8_t*)(jb + n);
  const int tid = threadIdx.x, nt = blockDim.x;
  LSLAM_PHASE_CLOCK(pck);
  if (tid == 0) { s_first = n; s_spec_ok = (spec && spec[0] >= 1) ? spec[0] : 0; }
  __syncthreads();
  const uint8_t* spec_cls = spec ? (const uint8_t*)(spec + 2 + n) : nullptr;
  // (1: the KIND of every point -- finite / infinite / NaN -- must be the predicted one; 2: its class, signs included)
  const bool kinds_only = s_spec_ok == 1;
  auto kind_of = [](uint32_t c) -> uint32_t { return ((c & 3u) == 3u || (c >> 2) == 3u) ? 2u : c ? 1u : 0u; };
  bool same = true;
  for (int i = tid; i < n; i += nt) {
    double2 q;
    if (ranges) {
      beam_world_point(pose.v[0], pose.v[1], pose.v[2], g.min_angle, g.ang_res, (uint32_t)i, ranges[i], q.x, q.y);
      world[i] = q;
    } else {
      q = world[i];
    }
    p[i] = q;
    reach[i] = 0;
    if (spec_cls) {
      const uint32_t c = point_class(q), c0 = (uint32_t)spec_cls[i];
      same = same && (kinds_only ? kind_of(c) == kind_of(c0) : c == c0);
    }
  }
  if (spec_cls && !same) s_spec_ok = 0;
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 0);  // world points (fp64 sincos per beam)
  if (spec && tid == 0 && !s_spec_ok) atomicAdd((int*)spec + 2 + n + ((n + 3) >> 2), 1);  // diagnostics: fall-backs
  if (s_spec_ok) {  // the speculative row is the row (block-uniform)
    const int cnt = spec[1];
    for (int k = tid; k <= cnt; k += nt) row[k] = spec[1 + k];
    LSLAM_PHASE_MARK(pck, 3);
    LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 2, (unsigned)(blockIdx.x * (nt >> 6) + (tid >> 6)), (tid & 63) == 0);
    return;
  }
  const double min_sq = ksq(0.1);
  for (int i = tid; i < n; i += nt) {
    first_valid_min(&s_first, !isnan(p[i].x) && !isnan(p[i].y), i);
    next[i] = successor_of(p, n, i, min_sq);
    ja[i] = next[i];
  }
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 1);  // successor of every point
  mark_reachable(n, s_first, ja, jb, reach, tid, nt);
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 2);  // pointer doubling
  anchors_to_row(n, reach, row, s_wc, tid, nt);
  LSLAM_PHASE_MARK(pck, 3);  // ordered compaction of the anchors
  LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 2, (unsigned)(blockIdx.x * (nt >> 6) + (tid >> 6)), (tid & 63) == 0);
}
// The speculative chain of ONE scan at the pose its match starts from: out = [ok, count, anchors (n), classes (n bytes)].
// Same walk as above; a comparison between two FINITE points within kSpecBand (1 + coordinates) of the threshold clears `ok`.
// The walk of successor_of with the band test: lo / hi = the threshold -+ kSpecBand (1 + the scan's largest coordinates), a
// block-wide constant (the search is fp64 VALU work on ONE CU -- 12.8 k cycles of the plain kernel's phase -- so the test is
// two more compares per candidate, not a band evaluated per pair).  inf and NaN distances are never "near".
__device__ __forceinline__ int successor_spec(const double2* p, int n, int i, double min_sq, double lo, double hi, bool& clear) {
  const double fx = p[i].x, fy = p[i].y;
  int j = i + 1;
  for (bool found = false; !found && j < n;) {
    double2 q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) q[u] = p[min(j + u, n - 1)];
    int hit = 4;
#pragma unroll
    for (int u = 3; u >= 0; u--) {
      const double dx = fx - q[u].x, dy = fy - q[u].y;
      const double d2 = ksq(dx) + ksq(dy);
      if (d2 > min_sq) hit = u;
      if ((d2 > lo) != (d2 > hi)) clear = false;
    }
    if (hit < 4 && j + hit < n) { j += hit; found = true; }
    else j = min(j + 4, n);
  }
  return j;
}
// sa.local (the scan-frame points k_scan_prep wrote for the match: a rigid image of the world points) spares this block the
// fp64 sincos of every beam: finite readings take their scan-frame point; a non-finite reading takes a point of the CLASS its
// world point will have -- r * cos(angle) is +-inf by the signs of r and of the cosine, i.e. by the angle's quadrant (a
// misjudged quadrant is caught like everything else: k_anchor_chain compares the classes with the final points').
__device__ __forceinline__ void anchor_spec_block(int n, const SpecArgs& sa, const Geom& g, unsigned char* smem) {
  __shared__ int s_first, s_wc[16], s_clear, s_nonfinite, s_mag_hi;
  double2* p = (double2*)smem;
  int* next = (int*)(p + n);
  int* ja = next + n;
  int* jb = ja + n;
  uint8_t* reach = (uint8_t*)(jb + n);
  uint8_t* cls = (uint8_t*)(sa.out + 2 + n);
  const int tid = threadIdx.x, nt = blockDim.x;
  if (tid == 0) { s_first = n; s_clear = 1; s_nonfinite = 0; s_mag_hi = 0; sa.out[0] = 0; }
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += nt) {  // whole waves iterate together: the wave-level reductions below need every lane
    const int i = i0 + tid;
    const bool in = i < n;
    double2 q = make_double2(0.0, 0.0);
    const double r = in ? sa.ranges[i] : 0.0;
    {
      const unsigned long long nf = __ballot(in && (isnan(r) || isinf(r)));
      if (nf && (tid & 63) == 0) atomicAdd(&s_nonfinite, __popcll(nf));
    }
    if (sa.local && !isnan(r) && !isinf(r)) {
      if (in) q = sa.local[i];
    } else if (sa.local) {
      const double inf = __builtin_inf(), nan = __builtin_nan("");
      if (isnan(r)) {
        q = make_double2(nan, nan);
      } else {
        const double angle = sa.pose[2] + g.min_angle + (uint32_t)i * g.ang_res;
        const long long quad = (long long)floor(angle * 0.63661977236758134308);  // angle / (pi / 2)
        const int k4 = (int)(((quad % 4) + 4) % 4);
        const bool cpos = (k4 == 0 || k4 == 3), spos = (k4 == 0 || k4 == 1);
        q = make_double2((r > 0.0) == cpos ? inf : -inf, (r > 0.0) == spos ? inf : -inf);
      }
    } else if (in) {
      beam_world_point(sa.pose[0], sa.pose[1], sa.pose[2], g.min_angle, g.ang_res, (uint32_t)i, r, q.x, q.y);
    }
    const uint32_t c = point_class(q);
    if (in) {
      p[i] = q;
      reach[i] = 0;
      cls[i] = (uint8_t)c;
    }
    const uint32_t mh = wave_max_u32(in && c == 0u ? (uint32_t)__double2hiint(fabs(q.x) + fabs(q.y)) : 0u);
    if ((tid & 63) == 0 && mh) atomicMax(&s_mag_hi, (int)mh);
  }
  __syncthreads();
  const double min_sq = ksq(0.1);
  // >= |x| + |y| of every finite point (the high word of a positive double orders like the double)
  const double mag = __hiloint2double(s_mag_hi + 1, 0);
  const double band = kSpecBand * (1.0 + 2.0 * mag), lo = min_sq - band, hi = min_sq + band;
  bool clear = band < 0.5 * min_sq;  // (coordinates beyond ~1e10 m: no speculation)
  for (int i = tid; i < n; i += nt) {
    first_valid_min(&s_first, !isnan(p[i].x) && !isnan(p[i].y), i);
    next[i] = successor_spec(p, n, i, min_sq, lo, hi, clear);
    ja[i] = next[i];
  }
  if (!clear) s_clear = 0;
  __syncthreads();
  mark_reachable(n, s_first, ja, jb, reach, tid, nt);
  __syncthreads();
  anchors_to_row(n, reach, sa.out + 1, s_wc, tid, nt);
  __syncthreads();
  // 1: the chain holds whatever the signs of the +-inf points (see k_anchor_chain); 2: only if they stay as predicted -- the
  // scan has enough non-finite readings for a run of them to span a quadrant
  if (tid == 0) sa.out[0] = !s_clear ? 0 : ((double)s_nonfinite * fabs(g.ang_res) < 1.5 ? 1 : 2);
}
__global__ void __launch_bounds__(1024)
k_anchor_chain(int n, double2* __restrict__ world, int* __restrict__ row, const double* __restrict__ ranges, PoseArg pose,
               Geom g, const int* __restrict__ spec) {
  anchor_chain_block(n, world, row, ranges, pose, g, spec);
}
// The same for a LIST of resident scans of a scan cache (block e = entry e): slot and pose come from a small table the
// host wrote into pinned memory (read over the bus: 32 bytes per block), or -- `from_result` -- the pose is the mean a
// match has just written (the caller announced that its scan will take that pose: Mapper.cpp:2040-2044), so the refresh
// needs no host round trip and runs behind the match, off the caller's critical path.
struct CacheRefresh {
  int slot, pad;
  double pose[3];
};
__global__ void __launch_bounds__(1024)
k_anchor_chain_list(int n, double2* __restrict__ world, int* __restrict__ rows, const double* __restrict__ ranges,
                    const CacheRefresh* __restrict__ list, const lslam_match_result* __restrict__ from_result,
                    CacheRefresh one, Geom g) {
  PoseArg pose;
  int slot;
  if (from_result) {
    if (from_result->status != LSLAM_OK) return;
    slot = one.slot;
    for (int i = 0; i < 3; i++) pose.v[i] = from_result->pose[i];
  } else {
    const CacheRefresh e = list ? list[blockIdx.x] : one;  // `one`: a single scan whose slot and pose are kernel arguments
    slot = e.slot;
    for (int i = 0; i < 3; i++) pose.v[i] = e.pose[i];
  }
  anchor_chain_block(n, world + (size_t)slot * n, rows + (size_t)slot * (n + 1), ranges + (size_t)slot * n, pose, g, nullptr);
}

// SmearPoint (Mapper.h:971-1005) of every centre k_find_valid marked, as a GATHER over the cleared-and-marked grid: one
// thread owns 16 consecutive grid bytes and takes, for each of them, the maximum of the kernel values of the centres
// (bytes equal to 100) within the kernel's reach -- on the FLAT index, like the reference's pointer arithmetic.  No
// atomics: a thread writes only its own bytes; the smear kernel's only 100 is its own centre (checked at create time),
// so a neighbour's finished or unfinished bytes are never mistaken for centres, each centre counts exactly once
// whoever marked it, and the maximum is order-independent.  (The scatter form -- a compare-and-swap per footprint word
// of every centre -- serialises at the memory-side atomic unit: 57 us for a 70-scan window.)
// The same pass writes the parity planes F_0 / F_1 of the finished bytes (k_deinterleave): one launch fewer.
// HK = half kernel size known at compile time (1, 2, 6 = resolutions 0.05, 0.025, 0.01 m at the default smear): the
// neighbour-row windows live in registers and every byte test is a static shift; HK = 0 is the generic form.
template <int HK>
__global__ void __launch_bounds__(256)
k_smear_gather(Geom g, const uint8_t* __restrict__ kernel, const uint8_t* __restrict__ marks, uint32_t epoch,
               uint8_t* __restrict__ grid, uint8_t* __restrict__ f0, uint8_t* __restrict__ f1) {
  __shared__ uint8_t s_k[kMaxKernel * kMaxKernel];
  const int tid = threadIdx.x, ks = g.kernel_size, hk = ks / 2;
  LSLAM_PHASE_CLOCK(pck);
  for (int i = tid; i < ks * ks; i += 256) s_k[i] = kernel[i];
  __syncthreads();
  LSLAM_PHASE_MARK(pck, 0);  // smear kernel into LDS
  const long long f = ((long long)blockIdx.x * 256 + tid) * 16;
  if (f >= g.data_size) return;
  const uint32_t ep4 = epoch * 0x01010101u;
  uint32_t out[4];
  {  // my own bytes: a centre is 100, everything else starts from zero
    const uint4 q = *(const uint4*)(marks + f);  // 16-byte aligned; may overhang into the guard (never marked)
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
      uint32_t o = 0u;
#pragma unroll
      for (int bb = 0; bb < 4; bb++)
        if (((w[k] >> (8 * bb)) & 0xFFu) == epoch) o |= (uint32_t)kOccupied << (8 * bb);
      out[k] = o;
    }
  }
  auto raise = [&](int o, uint32_t v) {  // out byte o = max(out byte o, v)
    const int sh = 8 * (o & 3);
    if (((out[o >> 2] >> sh) & 0xFFu) < v) out[o >> 2] = (out[o >> 2] & ~(0xFFu << sh)) | (v << sh);
  };
  if constexpr (HK > 0) {
    // widthStep is a multiple of 8 and f of 16, so every window starts at the same byte phase
    constexpr int NWIN = 16 + 2 * HK, LEAD = (4 - (HK & 3)) & 3, NW = (LEAD + NWIN + 3) / 4;
    for (int j = -HK; j <= HK; j++) {  // centre row = my row + j  ->  kernel row HK - j
      const long long w0 = f + (long long)j * g.stride - HK - LEAD;  // aligned; bytes outside [0, dataSize) hold no centre
      uint32_t w[NW], any = 0u;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        const long long byte = w0 + 4 * k;
        w[k] = (byte >= 0 && byte < g.data_size) ? *(const uint32_t*)(marks + byte) : 0u;
        const uint32_t x = w[k] ^ ep4;  // a byte tagged with this epoch becomes 0
        any |= (x - 0x01010101u) & ~x & 0x80808080u;
      }
      if (!any) continue;
      const uint8_t* krow = s_k + (2 * HK + 1) * (HK - j);
#pragma unroll
      for (int c = 0; c < NWIN; c++) {
        if (((w[(LEAD + c) >> 2] >> (8 * ((LEAD + c) & 3))) & 0xFFu) != epoch) continue;
        // centre at column offset c - HK relative to my first byte: reaches my bytes o with |o - (c - HK)| <= HK
#pragma unroll
        for (int o = (c - 2 * HK > 0 ? c - 2 * HK : 0); o <= (c < 15 ? c : 15); o++) raise(o, krow[o - c + 2 * HK]);
      }
    }
  } else {
    const int nwin = 16 + 2 * hk;
    for (int j = -hk; j <= hk; j++) {
      const long long row = f + (long long)j * g.stride - hk;  // first byte of the window
      const uint8_t* krow = s_k + ks * (hk - j);
      for (int c = 0; c < nwin; c++) {
        const long long cb = row + c;
        if (cb < 0 || cb >= g.data_size || (uint32_t)marks[cb] != epoch) continue;
        const int lo = max(0, c - 2 * hk), hi = min(15, c);
        for (int o = lo; o <= hi; o++) raise(o, krow[o - c + 2 * hk]);
      }
    }
  }
  LSLAM_PHASE_MARK(pck, 1);  // own bytes + the neighbour rows' windows of the mark plane
  // every grid byte is written, so the grid needs no clear; the tail of the last chunk lies in the guard band and gets
  // its zeros back (no centre reaches past dataSize: the ROI keeps a border)
  *(uint4*)(grid + f) = make_uint4(out[0], out[1], out[2], out[3]);
  if (f0) {  // bytes b0..b7 -> even: b0 b2 b4 b6, odd: b1 b3 b5 b7 (dataSize is a multiple of 8)
#pragma unroll
    for (int h = 0; h < 2; h++) {
      if (f + 8 * h + 8 > g.data_size) break;
      const uint32_t vx = out[2 * h], vy = out[2 * h + 1];
      const uint32_t e = (vx & 0xFFu) | ((vx >> 8) & 0xFF00u) | ((vy & 0xFFu) << 16) | ((vy << 8) & 0xFF000000u);
      const uint32_t o = ((vx >> 8) & 0xFFu) | ((vx >> 16) & 0xFF00u) | ((vy << 8) & 0xFF0000u) | (vy & 0xFF000000u);
      ((uint32_t*)f0)[f / 8 + h] = e;
      ((uint32_t*)f1)[f / 8 + h] = o;
    }
  }
  LSLAM_PHASE_MARK(pck, 2);  // grid bytes + parity planes stored
  LSLAM_PHASE_FLUSH(pck, g_sm_stamps, 1, (unsigned)(blockIdx.x * 4 + (tid >> 6)), (tid & 63) == 0);
}

// AddScan (Mapper.cpp:716-748) for every valid point of every base scan in parallel: the thread
// that turns a cell into 100 ("not already occupied", :734-740) lists it, and k_smear_list smears
// every listed cell (SmearPoint, Mapper.h:971-1005) as a max-merge scatter.  Order-independent because the kernel's only 100 is
// its centre (checked at create time; otherwise k_add_scans_serial runs).
__global__ void __launch_bounds__(256)
k_mark_centres(int B, int n, const double2* __restrict__ world, int ring_start, int cap,
               const uint8_t* __restrict__ valid, Geom g, uint8_t* __restrict__ grid,
               uint32_t* __restrict__ list, int* __restrict__ count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  uint32_t idx = 0xFFFFFFFFu;  // no cell
  if (i < n && valid[(size_t)b * n + i]) {
    const double2 p = world[(size_t)((ring_start + b) % cap) * n + i];
    const int gx = world_to_grid(p.x, g.off_x, g.scale);
    const int gy = world_to_grid(p.y, g.off_y, g.scale);
    if (gx >= 0 && gx < g.roi_w && gy >= 0 && gy < g.roi_h)  // IsUpTo on the ROI (:724-729)
      idx = (uint32_t)((gx + g.border) + (gy + g.border) * g.stride);
  }
  // neighbouring beams mostly hit the same cell: only the first lane of such a run goes for the CAS
  const uint32_t left = (uint32_t)__shfl_up((int)idx, 1);
  const bool contender = idx != 0xFFFFFFFFu && ((threadIdx.x & 63) == 0 || left != idx);
  // "value already set -> skip" (:734-738): exactly one thread per cell wins and its point smears
  // (the grid was cleared just before and nothing smears until the next kernel: every byte is 0 or 100,
  // so one atomicOr of 100 both sets the cell and tells whether it was still free)
  bool winner = false;
  if (contender) {
    const int sh = (int)(idx & 3u) * 8;
    const uint32_t old = atomicOr((uint32_t*)(grid + (idx & ~3u)), (uint32_t)kOccupied << sh);
    winner = ((old >> sh) & 0xFFu) == 0u;
  }
  const unsigned long long won = __ballot(winner);
  if (won) {  // one counter update per wave
    const int lane = threadIdx.x & 63, leader = __ffsll((long long)won) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(count, __popcll(won));
    base = __shfl(base, leader);
    if (winner) list[base + __popcll(won & ((1ull << lane) - 1ull))] = idx;
  }
}

// SmearPoint of every listed centre, one thread per (centre, kernel row, aligned word of that row):
// byte-wise max of up to four kernel values into the word by CAS.  All of a centre's words are in
// flight together instead of one thread walking its 49 cells.
__global__ void __launch_bounds__(256)
k_smear_list(const uint32_t* __restrict__ list, const int* __restrict__ count, Geom g,
             const uint8_t* __restrict__ kernel, uint8_t* __restrict__ grid, int words_per_row) {
  const int ks = g.kernel_size, hk = ks / 2;
  const int items = ks * words_per_row;
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long c = t / items;
  if (c >= *count) return;
  const int item = (int)(t - c * items), j = item / words_per_row - hk, w = item % words_per_row;
  const long long centre = (long long)list[c] + (long long)j * g.stride;  // this row's centre byte
  const long long word0 = ((centre - hk) & ~3ll) + 4ll * w;               // aligned word of the row
  uint32_t want = 0;
#pragma unroll
  for (int bb = 0; bb < 4; bb++) {
    const long long k = word0 + bb - centre;
    if (k >= -hk && k <= hk) want |= (uint32_t)kernel[(int)(k + hk) + ks * (j + hk)] << (8 * bb);
  }
  if (!want) return;
  uint32_t* wp = (uint32_t*)(grid + word0);
  uint32_t old = *wp;
  for (;;) {
    uint32_t mx = 0;
#pragma unroll
    for (int bb = 0; bb < 4; bb++) {
      const uint32_t o = (old >> (8 * bb)) & 0xFFu, v = (want >> (8 * bb)) & 0xFFu;
      mx |= (o > v ? o : v) << (8 * bb);
    }
    if (mx == old) break;
    const uint32_t prev = atomicCAS(wp, old, mx);
    if (prev == old) break;
    old = prev;
  }
}

// Exact sequential AddScans for smear kernels that contain 100 off-centre (then "already
// occupied -> skip" makes the result order dependent, Mapper.cpp:734-738).  One thread.
__global__ void k_add_scans_serial(int B, int n, const double2* __restrict__ world, int ring_start, int cap,
                                   const uint8_t* __restrict__ valid, Geom g,
                                   const uint8_t* __restrict__ kernel, uint8_t* __restrict__ grid) {
  if (blockIdx.x || threadIdx.x) return;
  const int hk = g.kernel_size / 2;
  for (int i = 0; i < B * n; i++) {
    if (!valid[i]) continue;
    const double2 wp = world[(size_t)((ring_start + i / n) % cap) * n + (i % n)];
    int gx = world_to_grid(wp.x, g.off_x, g.scale);
    int gy = world_to_grid(wp.y, g.off_y, g.scale);
    if (gx < 0 || gx >= g.roi_w || gy < 0 || gy >= g.roi_h) continue;
    size_t idx = (size_t)(gx + g.border) + (size_t)(gy + g.border) * g.stride;
    if (grid[idx] == kOccupied) continue;
    grid[idx] = kOccupied;
    for (int j = -hk; j <= hk; j++) {
      uint8_t* row = grid + (size_t)(gx + g.border) + (size_t)(gy + j + g.border) * g.stride;
      for (int k = -hk; k <= hk; k++) {
        uint8_t kv = kernel[(k + hk) + g.kernel_size * (j + hk)];
        if (kv > row[k]) row[k] = kv;
      }
    }
  }
}

// lookup tables of one scan for the inspection hook
__global__ void k_debug_table(Geom g, const double2* __restrict__ local, double angle_center,
                              double angle_offset, double angle_res, int na, int32_t* __restrict__ out) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  int a = blockIdx.y;
  if (b >= g.n_beams || a >= na) return;
  double angle = (angle_center - angle_offset) + (uint32_t)a * angle_res;  // Karto.h:6439-6442
  double2 p = local[b];
  int32_t v = kInvalidScan;
  if (!isnan(p.x)) v = lookup_offset(p.x, p.y, cos(angle), sin(angle), g.off_x, g.off_y, g.scale, g.stride);
  out[(size_t)a * g.n_beams + b] = v;
}

}  // namespace

// ==========================================================================================
// host side
// ==========================================================================================
struct lslam_matcher {
  lslam_context* ctx = nullptr;
  lslam_matcher_config cfg;
  lslam_laser laser;
  Geom g;
  std::vector<uint8_t> h_kernel;
  bool kernel_center_only = true;
  // device state
  uint8_t* d_grid_alloc = nullptr;  // kGuard + data_size + kGuard
  uint8_t* d_grid = nullptr;        // d_grid_alloc + kGuard
  uint8_t* d_marks_alloc = nullptr; // mark plane of the atomics-free rebuild (same geometry), allocated on first use
  uint8_t* d_marks = nullptr;
  int mark_epoch = 0;               // 1..255: a cell is a centre of THIS rebuild iff its mark byte equals it
  uint8_t* d_kernel = nullptr;
  uint8_t* d_sub_alloc = nullptr;   // two parity planes F_0, F_1, each kGuard + data_size/2 + kGuard
  uint8_t* d_sub[2] = {nullptr, nullptr};
  bool sub_dirty = true;            // the planes lag behind d_grid
  bool occ_dirty = true;            // so does the row-occupancy bitmap (built on demand: not for tiny batches)
  // One grid, one set of workspaces per instance -- like the reference's ScanMatcher (Mapper.h:1273-1278) this object
  // is NOT re-entrant: a second caller (another host thread, or the same matcher driven on a second stream) would
  // grow / free the shared workspaces (DevBuf::reserve) under kernels still in flight -- the memory fault of round 1's
  // two-stream experiment.  The core entry points take this flag and refuse to overlap instead.
  std::atomic<bool> busy{false};
  bool use_row_occupancy = true;    // lslam_matcher_set_option(LSLAM_OPT_ROW_OCCUPANCY)
  bool collect_stats = false;       // lslam_matcher_set_option(LSLAM_OPT_COLLECT_STATS): instrumented coarse kernel
  bool lds_staged = false;          // lslam_matcher_set_option(LSLAM_OPT_LDS_STAGED): the measured-and-dropped LDS-staged phase B
  int rows_waves = 1;               // lslam_matcher_set_option(LSLAM_OPT_ROWS_WAVES): waves per block of the tiled coarse kernel (1 = k_resp_rows, 2 / 4 / 8 = k_resp_rows_mw)
  // the streaming front-end's speculative anchor chain (anchor_spec_block): asked for before a match, reported after it
  SpecArgs spec_req{};
  bool spec_armed = false, spec_launched = false;
  int lone_waves = 0;               // LSLAM_OPT_LONE_KERNEL: 0 = four launches for the match of ONE scan, 4 / 8 / 16 = k_match_lone with that many waves per block
  LoneSync* d_lone_sync = nullptr;  // its hand-over words: a ring of kLoneRing slots
  unsigned lone_seq = 0;            // launches so far: slot seq % kLoneRing of the ring
  bool lone_one_task_wave = false;  // option value + 100: ONE wave of a block takes response tasks (168 blocks; the A/B reference)
  long long lone_launches = 0;
  int step_waves = 0;               // lslam_matcher_set_option(LSLAM_OPT_STEP_KERNEL): 0 = five launches per step, 3 / 4 = k_match_step with that many waves per scan
  int step_min_scans = 64;          // batches below this keep the five-kernel path (its beam-sliced kernels fill the chip)
  uint64_t step_launches = 0;       // k_match_step launches so far (diagnostics / tests)
  int stats_scans = 0;              // scans the per-(scan, beam) flag words behind the counters are sized for
  DevBuf<unsigned long long> d_stats;
  uint32_t* d_occ_t = nullptr;      // transposed row-occupancy bitmap (k_row_occupancy)
  uint2* d_occ_x = nullptr;         // the same bits as x-major 64-bit word pairs (k_occ_pairs): what k_resp_rows reads
  int occ_wpc = 0;
  int occ_win = 21;                 // grid bytes summarised per bit = row span of the coarse lattice
  uint32_t* d_nz = nullptr;         // flat non-zero bitmap, one bit per grid byte (k_nonzero_bits)
  int nz_words = 0;
  uint8_t* d_ptiles = nullptr;      // tiled parity planes (k_tile_planes), allocated on first batch use
  int ptile_tx = 0, ptile_rows = 0;
  bool ptile_dirty = true, ptile_failed = false;
  uint4* d_tiles = nullptr;         // overlapping 4x4 cell blocks (k_tile4), allocated on first batch use
  int tile_cols = 0, tile_rows = 0;
  bool tile_dirty = true, tile_failed = false;
  // workspaces
  DevBuf<double> d_ranges64;
  DevBuf<double> d_poses;
  DevBuf<double2> d_local, d_world;
  DevBuf<uint8_t> d_valid;
  DevBuf<int> d_fv_scratch;
  DevBuf<uint32_t> d_centres;  // [0] = count, [1..] = flat grid index of every newly occupied cell (AddScans)
  DevBuf<Lattice> d_lat;
  DevBuf<double2> d_cossin;  // [S][kMaxAngles] cos/sin of the pass's candidate angles (k_pass_setup)
  DevBuf<CoarseOut> d_coarse;
  DevBuf<int32_t> d_resp;
  size_t resp_prezeroed = 0;  // words at the start of d_resp the last grid rebuild cleared for the NEXT match
  int fast_div = 0;           // SearchCfg::fast_div / inv_denom: set at creation after the exhaustive check
  double inv_denom = 0.0;
  bool prep_done = false;     // the last grid rebuild also ran k_scan_prep for the ONE scan of the next match
  DevBuf<int32_t> d_tbl;     // large lattices: materialised lookup tables [S][nA][N]
  DevBuf<int32_t> d_part;    // large lattices, few scans: per-beam-slice partial numerators [S][slices][resp_stride]
  DevBuf<double> d_big;      // large lattices: reduce scratch
  DevBuf<lslam_match_result> d_results;
  DevBuf<int32_t> d_dbg;
  bool dbg_all = false;        // debug_coarse_sums_batch: copy the coarse numerators of EVERY scan of the batch, not the first's
  size_t dbg_resp_stride = 0;  // ints per scan in that copy
  int32_t* dbg_fine = nullptr;  // debug_fine_sums_batch: device buffer the fine numerators of every scan are copied to
  // single-scan matches: the last kernel posts a ticket in pinned memory behind its record and the host spins on it
  int* h_done = nullptr;   // pinned
  int done_ticket = 0;
  bool arm_next = false;   // set by arm_done_ticket, consumed by the next match_batch_impl
  bool done_armed = false; // the match just enqueued will post done_ticket
  // lslam_matcher_match_scan: the query scan's staging (pinned host) and its resident copy, the pinned result record
  double* h_query = nullptr;
  lslam_match_result* h_result = nullptr;
  DevBuf<double> d_query, d_qpose;
  // ---- pipelined steps (LSLAM_OPT_PIPELINE_DEPTH > 1) ----------------------------------------------------------------
  // Consecutive batched matches take turns on `pipe_depth` internal streams, each with its OWN set of the workspaces a
  // step writes between its first and its last kernel (StepWork), so that the steps in flight share the chip (in practice
  // they run the same phase side by side: the response kernels fill each other's tails, the latency-bound prep / reduce
  // kernels are paid once per `pipe_depth` steps -- DESIGN.md 6).  The grid and everything derived from it are shared and read-only while
  // steps are in flight: whoever changes them joins the internal streams into the context stream first (pipe_join).
  struct StepWork {
    DevBuf<double2> d_local, d_cossin;
    DevBuf<Lattice> d_lat;
    DevBuf<CoarseOut> d_coarse;
    DevBuf<int32_t> d_resp, d_tbl, d_part;
    DevBuf<double> d_big;
  };
  static constexpr int kMaxPipe = 4;
  int pipe_depth = 1;
  int pipe_next = 0;                 // slot of the next pipelined step
  bool pipe_in_step = false;         // match_batch_impl is running for a pipelined step (on a swapped-in stream)
  bool pipe_registered = false;      // pipe_join is in the context's pre_sync list
  hipStream_t pipe_stream[kMaxPipe] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t pipe_done[kMaxPipe] = {nullptr, nullptr, nullptr, nullptr};
  bool pipe_pending[kMaxPipe] = {false, false, false, false};
  hipEvent_t pipe_in[kMaxPipe] = {nullptr, nullptr, nullptr, nullptr};  // the context stream when slot i's step was enqueued
  hipEvent_t view_ready = nullptr;   // behind the last refresh of a grid view (parity planes, bitmap, tiles) by a step
  int view_owner = -1;               // slot whose stream recorded view_ready (-1: none since the last join)
  StepWork pipe_work[kMaxPipe];      // slot 0 is unused: it works in the members above
  uint64_t pipe_steps = 0;           // pipelined steps enqueued so far (diagnostics)
  // LSLAM_OPT_CHECK_OUTPUT_REUSE (debug): the record buffer [begin, end) of the step each slot has in flight; a pipelined
  // step whose records would land in a buffer an unfinished step is still writing is refused instead of corrupting it
  bool check_out_reuse = false;
  const char* pipe_out_lo[kMaxPipe] = {nullptr, nullptr, nullptr, nullptr};
  const char* pipe_out_hi[kMaxPipe] = {nullptr, nullptr, nullptr, nullptr};
};

namespace {

// Hand back the allocations the workspaces have outgrown.  Only from entry points that have JUST synchronised the
// context stream: nothing in flight can still read them (DevBuf::reserve keeps them until then, see common.hpp).
void trim_workspaces(lslam_matcher* m) {
  m->d_stats.trim(); m->d_ranges64.trim(); m->d_poses.trim(); m->d_local.trim(); m->d_world.trim(); m->d_valid.trim();
  m->d_fv_scratch.trim(); m->d_centres.trim(); m->d_lat.trim(); m->d_cossin.trim(); m->d_coarse.trim(); m->d_resp.trim();
  m->d_tbl.trim(); m->d_part.trim(); m->d_big.trim(); m->d_results.trim(); m->d_dbg.trim(); m->d_query.trim(); m->d_qpose.trim();
  for (auto& w : m->pipe_work) {
    w.d_local.trim(); w.d_cossin.trim(); w.d_lat.trim(); w.d_coarse.trim(); w.d_resp.trim(); w.d_tbl.trim(); w.d_part.trim(); w.d_big.trim();
  }
}

// Order the context stream behind every pipelined step still in flight.  Cheap when there is none.  Called by whatever
// is about to change the grid or a workspace, read a result on the context stream, or synchronise the context.
int pipe_join(lslam_matcher* m) {
  lslam_context* ctx = m->ctx;
  for (int i = 0; i < lslam_matcher::kMaxPipe; i++)
    if (m->pipe_pending[i]) {
      // recorded HERE, once per join, not once per step: an event record is a marker packet in front of the slot's next
      // kernel (~5 us of the GPU's time each, rocprofv3 timeline in profiles/r05/experiments/pipe_lockstep)
      LSLAM_HIP(ctx, hipEventRecord(m->pipe_done[i], m->pipe_stream[i]));
      LSLAM_HIP(ctx, hipStreamWaitEvent(ctx->stream, m->pipe_done[i], 0));
      m->pipe_pending[i] = false;
      m->pipe_out_lo[i] = m->pipe_out_hi[i] = nullptr;
    }
  m->view_owner = -1;  // the context stream is behind every refresh now, and every later step is behind it
  return LSLAM_OK;
}
int pipe_join_cb(void* p) { return pipe_join((lslam_matcher*)p); }

void swap_step_work(lslam_matcher* m, lslam_matcher::StepWork& w) {
  std::swap(m->d_local, w.d_local); std::swap(m->d_cossin, w.d_cossin); std::swap(m->d_lat, w.d_lat);
  std::swap(m->d_coarse, w.d_coarse); std::swap(m->d_resp, w.d_resp); std::swap(m->d_tbl, w.d_tbl);
  std::swap(m->d_part, w.d_part); std::swap(m->d_big, w.d_big);
}

struct BusyGuard {
  lslam_matcher* m;
  bool ok;
  explicit BusyGuard(lslam_matcher* mm) : m(mm), ok(!mm->busy.exchange(true)) {}
  ~BusyGuard() {
    if (ok) m->busy.store(false);
  }
};
#define LSLAM_NOT_REENTRANT(m)                                                                                   \
  BusyGuard busy_guard(m);                                                                                       \
  if (!busy_guard.ok)                                                                                            \
    return (m)->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "matcher used concurrently: a ScanMatcher instance is not " \
                                                      "re-entrant (one grid + one set of workspaces, Mapper.h:1273-1278)")

// Ask the next single-scan match to post a ticket behind its record (see k_reduce_fine) ...
void arm_done_ticket(lslam_matcher* m) {
  if (!m->h_done) {
    if (hipHostMalloc((void**)&m->h_done, sizeof(int), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      m->h_done = nullptr;
      return;  // no ticket: wait_record falls back to the stream
    }
    *m->h_done = 0;
  }
  m->arm_next = true;
}
// ... and wait for it: a bounded spin on the pinned word (acquire), the stream itself when no ticket was armed or it does
// not show up in time (a failed launch, a device fault: hipStreamSynchronize reports those).  On return the record in
// pinned memory is complete; kernels behind the match on the stream (a speculative refresh) may still be running.
int wait_record(lslam_matcher* m) {
  lslam_context* ctx = m->ctx;
  if (m->done_armed) {
    m->done_armed = false;
    if (spin_for_ticket(m->h_done, m->done_ticket)) return LSLAM_OK;
  }
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int n_angles_of(double off, double res) { return lattice_count(off, res); }

// coarse pass geometry (Mapper.cpp:228-240)
PassCfg coarse_pass_cfg(const lslam_matcher* m, const Geom& g) {
  const double res = 1.0 / g.scale;  // GetResolution() (Karto.h:4335-4338)
  PassCfg pc;
  pc.off_x = pc.off_y = 0.5 * ((double)g.probs_side - 1) * res;
  pc.res_x = pc.res_y = 2 * res;
  pc.ang_off = m->cfg.coarse_search_angle_offset;
  pc.ang_res = m->cfg.coarse_angle_resolution;
  pc.nx = lattice_count(pc.off_x, pc.res_x);
  pc.ny = lattice_count(pc.off_y, pc.res_y);
  pc.na = n_angles_of(pc.ang_off, pc.ang_res);
  pc.mode = 0;
  return pc;
}

template <typename RT>
int match_batch_impl(lslam_matcher* m, int S, const RT* d_ranges, int stride, const double* d_poses,
                     int do_penalize, int do_refine, lslam_match_result* d_out,
                     int32_t* dbg_coarse_sums /*device, optional*/, int force_generic) {
  lslam_context* ctx = m->ctx;
  if (!m->pipe_in_step) {  // a plain match shares the workspaces of slot 0 and the context stream with nobody
    int jrc = pipe_join(m);
    if (jrc) return jrc;
  }
  Geom g = m->g;
  // what the last grid rebuild prepared for THIS match (streaming front-end); consumed here whatever path the match
  // takes, error returns included, so that no later match can find stale flags
  const bool prep_was_done = m->prep_done;
  size_t prezeroed = m->resp_prezeroed;
  m->prep_done = false;
  m->resp_prezeroed = 0;
  // a caller that will wait for ONE record in pinned memory arms the done ticket (arm_done_ticket); consumed here
  int* done_flag = nullptr;
  int done_ticket = 0;
  m->done_armed = false;
  if (m->arm_next && S == 1 && S < kReduceNarrowMinScans && m->h_done) {
    done_flag = m->h_done;
    done_ticket = ++m->done_ticket;
    m->done_armed = true;
  }
  m->arm_next = false;
  const SpecArgs spec_req = m->spec_armed ? m->spec_req : SpecArgs{};  // consumed here whatever path the match takes
  m->spec_armed = false;
  m->spec_launched = false;
  if (S <= 0) return LSLAM_OK;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  if (g.n_beams == 0) {
    m->done_armed = false;  // k_result_no_readings posts no ticket: wait_record goes to the stream at once
    launch(ctx, "result_no_readings", k_result_no_readings, dim3((S + 255) / 256), dim3(256), 0, S,
           d_poses, m->cfg.coarse_angle_resolution, d_out);
    return LSLAM_OK;
  }
  const double res = 1.0 / g.scale;  // GetResolution() (Karto.h:4335-4338)
  const PassCfg pc = coarse_pass_cfg(m, g);
  // fine pass geometry (:276-281)
  PassCfg pf;
  pf.off_x = pf.off_y = pc.res_x * 0.5;
  pf.res_x = pf.res_y = res;
  pf.ang_off = 0.5 * m->cfg.coarse_angle_resolution;
  pf.ang_res = m->cfg.fine_search_angle_offset;
  pf.nx = lattice_count(pf.off_x, pf.res_x);
  pf.ny = lattice_count(pf.off_y, pf.res_y);
  pf.na = n_angles_of(pf.ang_off, pf.ang_res);
  pf.mode = 2;
  int na_max = pc.na;
  const int n_exp = m->cfg.use_response_expansion ? 3 : 0;
  PassCfg pe[3];
  for (int e = 0; e < n_exp; e++) {
    pe[e] = pc;
    pe[e].mode = 1;
    double o = m->cfg.coarse_search_angle_offset;
    for (int i = 0; i <= e; i++) o += 20.0 * kPi180;  // math::DegreesToRadians(20) (Mapper.cpp:253)
    pe[e].ang_off = o;
    pe[e].na = n_angles_of(o, pc.ang_res);
    na_max = std::max(na_max, pe[e].na);
  }
  na_max = std::max(na_max, pf.na);
  if (pc.nx > kMaxLattice || pc.ny > kMaxLattice || pf.nx > kMaxLattice || pf.ny > kMaxLattice ||
      na_max > kMaxAngles || pc.nx < 1 || pf.nx < 1 || pc.na < 1 || pf.na < 1)
    return ctx->fail(LSLAM_ERR_UNSUPPORTED, "search lattice %dx%dx%d exceeds the built limits (%d,%d,%d)",
                     pc.nx, pc.ny, na_max, kMaxLattice, kMaxLattice, kMaxAngles);
  const size_t resp_stride = (size_t)std::max(pc.nx * pc.ny, pf.nx * pf.ny) * na_max;
  m->dbg_resp_stride = resp_stride;

  LSLAM_HIP(ctx, m->d_local.reserve((size_t)S * g.n_beams));
  LSLAM_HIP(ctx, m->d_lat.reserve(S));
  LSLAM_HIP(ctx, m->d_cossin.reserve((size_t)S * kMaxAngles));
  LSLAM_HIP(ctx, m->d_coarse.reserve(S));
  {
    const int32_t* resp_before = m->d_resp.p;
    LSLAM_HIP(ctx, m->d_resp.reserve((size_t)S * resp_stride));
    if (m->d_resp.p != resp_before) prezeroed = 0;  // the rebuild cleared the buffer this one has just replaced
  }

  SearchCfg sc{m->cfg.distance_variance_penalty, m->cfg.angle_variance_penalty,
               m->cfg.minimum_distance_penalty, m->cfg.minimum_angle_penalty, do_penalize, m->fast_div, m->inv_denom};

  // scan_prep also lays out the coarse lattice of pass 0 (mode 0 needs only the pose): one launch fewer.  It is
  // launched further down, once it is clear that the step does not go out as ONE kernel (k_match_step preps its own scan).
  bool setup_done = true;  // consumed by the first pass


  if (m->sub_dirty) {  // refresh the parity planes of the grid (coarse pass source)
    launch(ctx, "deinterleave", k_deinterleave, dim3((g.data_size / 8 + 255) / 256), dim3(256), 0,
           (const uint8_t*)m->d_grid, m->d_sub[0], m->d_sub[1], g.data_size / 8);
    m->sub_dirty = false;
  }
  // The exact row-occupancy bitmap costs ~30 us to rebuild; a handful of scans (the streaming front-end
  // matches ONE scan per grid) does not earn that back, so tiny batches match without it unless it is
  // already up to date.  Results do not depend on it either way.
  const bool want_occ = m->use_row_occupancy && (S >= kOccMinScans || !m->occ_dirty);
  if (want_occ && m->occ_dirty) {
    launch(ctx, "nonzero_bits", k_nonzero_bits, dim3((m->nz_words + 255) / 256), dim3(256), 0,
           (const uint8_t*)m->d_grid, g.data_size, m->d_nz, m->nz_words);
    launch(ctx, "row_occupancy", k_row_occupancy, dim3((g.stride + 255) / 256, m->occ_wpc), dim3(256), 0,
           (const uint32_t*)m->d_nz, m->nz_words, g.stride, g.height, m->occ_win, m->d_occ_t, m->occ_wpc / 2);
    launch(ctx, "occ_pairs", k_occ_pairs, dim3((g.stride + 255) / 256, m->occ_wpc), dim3(256), 0,
           (const uint32_t*)m->d_occ_t, g.stride, m->occ_wpc / 2, m->d_occ_x);
    m->occ_dirty = false;
  }
  // the 2-D tiled parity planes (coarse pass of chip-filling batches) and the overlapping 4x4 blocks (their fine pass):
  // allocated on first use, refreshed when the grid changed; false = not available (allocation failed / too big)
  auto ensure_ptiles = [&]() -> bool {
    if (m->ptile_failed) return false;
    if (!m->d_ptiles) {
      m->ptile_tx = (g.stride / 2 + 15) / 16;
      m->ptile_rows = (((g.height - 1 + kTileYOff) / 2 + 1) + 3) & ~3;  // class rows, whole 4-row lines
      const size_t bytes = (size_t)kTilePad + 4 * (size_t)m->ptile_rows * m->ptile_tx * 32;
      if (bytes >= (1ull << 32) || hipMalloc((void**)&m->d_ptiles, bytes) != hipSuccess) {
        (void)hipGetLastError();
        m->d_ptiles = nullptr;
        m->ptile_failed = true;  // keep to the linear planes
        return false;
      }
      if (hipMemsetAsync(m->d_ptiles, 0, kTilePad, ctx->stream) != hipSuccess) return false;
    }
    if (m->ptile_dirty) {
      const size_t dwords = (size_t)m->ptile_rows * m->ptile_tx * 32;  // 4 classes x class_bytes / 4
      launch(ctx, "tile_planes", k_tile_planes, dim3((unsigned)((dwords + 255) / 256)), dim3(256), 0,
             (const uint8_t*)m->d_grid, g.stride, g.data_size, (uint32_t*)m->d_ptiles, m->ptile_tx, m->ptile_rows);
      m->ptile_dirty = false;
    }
    return true;
  };
  auto tiles4_possible = [&]() -> bool {
    return g.n_beams <= 16384 && !m->tile_failed &&
           (unsigned long long)g.data_size * 4ull + (1ull << 24) < (1ull << 32);  // k_resp_tile3 uses 32-bit offsets
  };
  auto ensure_tiles4 = [&]() -> bool {
    if (!tiles4_possible()) return false;
    if (!m->d_tiles) {
      m->tile_cols = g.stride / 2;
      m->tile_rows = (g.height + 1) / 2 + kTileYPad / 2 + 1;
      m->tile_rows += m->tile_rows & 1;  // lines hold block rows in pairs
      if (hipMalloc((void**)&m->d_tiles, (size_t)((m->tile_cols + 3) / 4) * 4 * m->tile_rows * sizeof(uint4)) != hipSuccess) {
        (void)hipGetLastError();
        m->d_tiles = nullptr;
        m->tile_failed = true;  // not enough HBM for the 4x copy: keep to the row kernel
        return false;
      }
    }
    if (m->tile_dirty) {
      launch(ctx, "tile4", k_tile4, dim3((m->tile_cols + 255) / 256, m->tile_rows), dim3(256), 0,
             (const uint8_t*)m->d_grid, g.stride, g.data_size, m->d_tiles, m->tile_cols, m->tile_rows);
      m->tile_dirty = false;
    }
    return true;
  };
  auto reduce_lds = [&](const PassCfg& p, bool cache) -> size_t {
    size_t total = (size_t)p.nx * p.ny * p.na;
    return (cache ? total * 8 : 0) + (size_t)p.nx * p.ny * (8 + 32 + 4) + (size_t)g.probs_side * g.probs_side * 8 +
           ((total + 31) / 32) * 4 + 16;
  };
  // k_reduce_coarse_lds keeps no per-candidate cache: two cell-maximum arrays instead
  auto reduce_parts = [&](const PassCfg& p, int nt) -> int { return std::max(1, std::min(8, nt / (p.nx * p.ny))); };
  auto reduce_lds_nocache = [&](const PassCfg& p, int parts) -> size_t {
    size_t total = (size_t)p.nx * p.ny * p.na;
    return (size_t)p.nx * p.ny * (8 * parts + 32 + 4 + 4) + (size_t)g.probs_side * g.probs_side * 8 + ((total + 31) / 32) * 4 + 16;
  };
  // response numerators of one pass: packed row kernel for uniform lattices (step 2 on the parity
  // planes, step 1 on the grid), generic kernel for everything else
  // lattice step the packed kernel of the last pass required (0 = the generic kernel did the pass):
  // the reduce kernels compute the numerators of scans with a different (non-uniform) lattice themselves
  int fb_step = 0;
  bool fine_prezeroed = false;
  auto run_responses = [&](const PassCfg& p, int step, const char* name) -> int {
    const int variant = force_generic ? 0 : (p.nx <= 4 && p.ny <= 4) ? 1 : (p.nx <= 12) ? 2 : (p.nx <= 16) ? 3 : 0;
    fb_step = variant ? step : 0;
    if (!setup_done)
      launch(ctx, "pass_setup", k_pass_setup, dim3(S), dim3(64), 0, S, g, p, d_poses,
             (const CoarseOut*)m->d_coarse.p, m->d_lat.p, m->d_cossin.p, step);
    setup_done = false;
    const long long waves = (long long)((S + 7) / 8) * 8 * p.na;
    // fine pass of a batch big enough to fill the chip: one 16-byte load per beam from the 4x4 blocks
    bool tiled = variant == 1 && step == 1 && p.nx == 3 && p.ny == 3 && waves >= kTileMinWaves && tiles4_possible();
    if (tiled) tiled = ensure_tiles4();
    if (tiled) {
#define LSLAM_TILE3_ARGS                                                                                                \
  (const uint4*)m->d_tiles, m->tile_cols, g, p, (const Lattice*)m->d_lat.p, (const double2*)m->d_cossin.p,              \
      (const double2*)m->d_local.p, m->d_resp.p, resp_stride, S
      if (S >= kTile3ManyMinScans)
        launch(ctx, "resp_tile_fine", k_resp_tile3<kTile3ManyAngles>,
               dim3((unsigned)((long long)((S + 7) / 8) * 8 * ((p.na + kTile3ManyAngles - 1) / kTile3ManyAngles))), dim3(64), 0,
               LSLAM_TILE3_ARGS);
      else
        launch(ctx, "resp_tile_fine", k_resp_tile3<1>, dim3((unsigned)waves), dim3(64), 0, LSLAM_TILE3_ARGS);
#undef LSLAM_TILE3_ARGS
    } else if (variant) {
      // small batches: split the beams of one (scan, angle) over several waves to fill the chip
      int slices = 1;
      while (slices < 8 && waves * slices < 2048) slices *= 2;
      while ((g.n_beams + 64 * slices - 1) / (64 * slices) > kMaxBeamsPerLane) slices *= 2;  // packed 16-bit sums
      if (slices > 1) {
        if (step == 1 && fine_prezeroed) fine_prezeroed = false;  // k_reduce_coarse_lds cleared the fine numerators
        else if (prezeroed >= (size_t)S * resp_stride) prezeroed = 0;  // k_rebuild_begin cleared it; one use
        else LSLAM_HIP(ctx, hipMemsetAsync(m->d_resp.p, 0, (size_t)S * resp_stride * sizeof(int32_t), ctx->stream));
      }
      dim3 grid((unsigned)(waves * slices));
      const uint8_t* s0 = step == 2 ? m->d_sub[0] : m->d_grid;
      const uint8_t* s1 = step == 2 ? m->d_sub[1] : m->d_grid;
      const int limit = step == 2 ? g.data_size / 2 : g.data_size;
      // the bitmap covers row spans of kOccWin grid bytes: step*(nX-1)+1 must fit
      const uint2* occ = (want_occ && step == 2 && step * (p.nx - 1) + 1 <= m->occ_win) ? m->d_occ_x : (const uint2*)nullptr;
      // coarse pass of a batch that fills the chip on its own: gather from the TILED parity planes
      bool ptiled = step == 2 && slices == 1 && waves >= kTileMinWaves && !m->ptile_failed &&
                    !(m->lds_staged && variant == 2);  // the LDS-staged experiment reads the LINEAR planes
      if (ptiled) ptiled = ensure_ptiles();
      const uint32_t class_bytes = (uint32_t)((size_t)m->ptile_rows * m->ptile_tx * 32);
#define LSLAM_ROWS_ARGS(SRC0, SRC1)                                                                              \
  grid, dim3(64), 0, SRC0, SRC1, step, limit, g, p, (const Lattice*)m->d_lat.p, (const double2*)m->d_cossin.p, \
      (const double2*)m->d_local.p, m->d_resp.p, resp_stride, slices, S, occ, m->occ_wpc, m->ptile_rows, class_bytes, \
      (unsigned long long*)(m->collect_stats ? m->d_stats.p : nullptr)
      const uint8_t* pt = m->d_ptiles;
      if (m->collect_stats && (variant == 2 || variant == 3) && step == 2 && m->stats_scans < S) {
        // (re)size the per-(scan, beam) flag words behind the counters; the counters collected so far are carried over
        unsigned long long keep[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        LSLAM_HIP(ctx, hipMemcpyAsync(keep, m->d_stats.p, sizeof keep, hipMemcpyDeviceToHost, ctx->stream));
        LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        const size_t words64 = 8 + ((size_t)S * g.n_beams + 1) / 2;
        LSLAM_HIP(ctx, m->d_stats.reserve(words64));
        LSLAM_HIP(ctx, hipMemsetAsync(m->d_stats.p, 0, words64 * sizeof(unsigned long long), ctx->stream));
        keep[4] = (unsigned long long)S;
        LSLAM_HIP(ctx, hipMemcpyAsync(m->d_stats.p, keep, sizeof keep, hipMemcpyHostToDevice, ctx->stream));
        LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
        m->stats_scans = S;
      }
      if (m->lds_staged && variant == 2 && step == 2) {  // experiment (DESIGN_HISTORY.md B): phase B through LDS patches
        if (m->collect_stats)
          launch(ctx, name, k_resp_rows<3, 11, false, true, true>, LSLAM_ROWS_ARGS(s0, s1));
        else
          launch(ctx, name, k_resp_rows<3, 11, false, false, true>, LSLAM_ROWS_ARGS(s0, s1));
      } else if (m->collect_stats && variant == 2 && step == 2) {  // instrumented twin (untimed diagnostics only)
        if (ptiled)
          launch(ctx, name, k_resp_rows<3, 11, true, true>, LSLAM_ROWS_ARGS(pt, pt));
        else
          launch(ctx, name, k_resp_rows<3, 11, false, true>, LSLAM_ROWS_ARGS(s0, s1));
      } else if (m->collect_stats && variant == 3 && step == 2) {  // the same for lattice rows of 13..16 positions
        if (ptiled)
          launch(ctx, name, k_resp_rows<4, 8, true, true>, LSLAM_ROWS_ARGS(pt, pt));
        else
          launch(ctx, name, k_resp_rows<4, 8, false, true>, LSLAM_ROWS_ARGS(s0, s1));
      } else if (variant == 1)
        launch(ctx, name, k_resp_rows<1, 4, false>, LSLAM_ROWS_ARGS(s0, s1));
      else if ((variant == 2 || variant == 3) && ptiled && m->rows_waves > 1 && g.n_beams <= 64 * kMaxBeamsPerLane) {
        const int W = m->rows_waves, groups = (p.na + W - 1) / W;
        const dim3 mw_grid((unsigned)((long long)((S + 7) / 8) * 8 * groups));
#define LSLAM_MW(NXD_, NYC_, W_)                                                                                         \
  launch(ctx, name, k_resp_rows_mw<NXD_, NYC_, W_>, mw_grid, dim3(64 * W_), 0, pt, limit, g, p, (const Lattice*)m->d_lat.p, \
         (const double2*)m->d_cossin.p, (const double2*)m->d_local.p, m->d_resp.p, resp_stride, S, occ, m->occ_wpc,      \
         m->ptile_rows, class_bytes)
        if (variant == 2 && W == 2) LSLAM_MW(3, 11, 2);
        else if (variant == 2 && W == 4) LSLAM_MW(3, 11, 4);
        else if (variant == 2 && W == 8) LSLAM_MW(3, 11, 8);
        else if (variant == 3 && W == 2) LSLAM_MW(4, 8, 2);
        else if (variant == 3 && W == 4) LSLAM_MW(4, 8, 4);
        else LSLAM_MW(4, 8, 8);
#undef LSLAM_MW
      } else if (variant == 2 && ptiled)
        launch(ctx, name, k_resp_rows<3, 11, true>, LSLAM_ROWS_ARGS(pt, pt));
      else if (variant == 2)
        launch(ctx, name, k_resp_rows<3, 11, false>, LSLAM_ROWS_ARGS(s0, s1));
      else if (ptiled)
        launch(ctx, name, k_resp_rows<4, 8, true>, LSLAM_ROWS_ARGS(pt, pt));
      else
        launch(ctx, name, k_resp_rows<4, 8, false>, LSLAM_ROWS_ARGS(s0, s1));
#undef LSLAM_ROWS_ARGS
    } else {
      int chunks = (p.nx * p.ny + kPosChunk - 1) / kPosChunk;
      long long items = (long long)S * p.na * chunks;
      launch(ctx, "resp_generic", k_resp_generic, dim3((unsigned)std::min<long long>(items, 1 << 20)), dim3(64), 0,
             (const uint8_t*)m->d_grid, g, p, (const Lattice*)m->d_lat.p, (const double2*)m->d_local.p,
             m->d_resp.p, resp_stride, S);
    }
    return LSLAM_OK;
  };
  // a coarse-only match on a loop-closure-size lattice: k_reduce_coarse_big writes the record (and posts the ticket) itself
  lslam_match_result* const big_final_out = (!do_refine && n_exp == 0) ? d_out : (lslam_match_result*)nullptr;
  bool record_written = false;
  auto run_coarse_big = [&](const PassCfg& p, int pass_index) -> int {
    // dense kernel for uniform lattices; scans with a non-uniform lattice fall to the generic kernel
    fb_step = 2;
    if (!setup_done)
      launch(ctx, "pass_setup", k_pass_setup, dim3(S), dim3(64), 0, S, g, p, d_poses,
             (const CoarseOut*)m->d_coarse.p, m->d_lat.p, m->d_cossin.p, 2);
    setup_done = false;
    LSLAM_HIP(ctx, m->d_tbl.reserve((size_t)S * p.na * g.n_beams));
    const size_t ncand = (size_t)p.nx * p.ny, total = ncand * p.na;
    const size_t stride = ncand + (size_t)g.probs_side * g.probs_side + 4 * ncand + (total + 63) / 64 + 8;  // doubles
    LSLAM_HIP(ctx, m->d_big.reserve((size_t)S * stride + S + 1));
    // per-scan best response (bit pattern of a non-negative double), behind the scratch of the last scan
    unsigned long long* best_bits = (unsigned long long*)(m->d_big.p + (size_t)S * stride);
    launch(ctx, "table_big", k_table_big, dim3((g.n_beams + 255) / 256, p.na, S), dim3(256), 0, S, g, p,
           (const Lattice*)m->d_lat.p, (const double2*)m->d_local.p, m->d_tbl.p, best_bits);
    const int lpr = (p.nx + 15) / 16, rpw = 64 / lpr;
    const bool lone = (long long)S * p.na * ((p.ny + rpw * kDenseT - 1) / (rpw * kDenseT)) * 8 < 4096;  // few waves even with 8 slices
    const int dense_t = lone ? 1 : kDenseT;
    const int n_tiles = (p.ny + rpw * dense_t - 1) / (rpw * dense_t);
    // few scans: split the beams of one (scan, angle, tile) over up to 8 waves; every slice writes its own partial sums
    // (plain stores) and k_big_latmax adds them up -- exact (integers), no atomics.  Measured for ONE 101x101x21 match:
    // 4 / 8 / 16 slices -> dense pass 0.102 / 0.065 / 0.057 ms, summing pass 0.018 / 0.018 / 0.095 ms: 8 it is.
    int slices = 1;
    constexpr int max_slices = 8;  // = kMaxSlices of k_big_latmax
    while (slices < max_slices && (long long)S * p.na * n_tiles * slices < 1024) slices *= 2;
    if (slices > 1) LSLAM_HIP(ctx, m->d_part.reserve((size_t)S * slices * resp_stride));
    if (lone)
      launch(ctx, "resp_dense", k_resp_dense<1>, dim3((unsigned)((long long)S * p.na * n_tiles * slices)), dim3(64), 0,
             (const uint8_t*)m->d_sub[0], (const uint8_t*)m->d_sub[1], g.data_size / 2, g, p, (const Lattice*)m->d_lat.p,
             (const int32_t*)m->d_tbl.p, slices > 1 ? m->d_part.p : m->d_resp.p, resp_stride, n_tiles, slices);
    else
      launch(ctx, "resp_dense", k_resp_dense<kDenseT>, dim3((unsigned)((long long)S * p.na * n_tiles * slices)), dim3(64), 0,
             (const uint8_t*)m->d_sub[0], (const uint8_t*)m->d_sub[1], g.data_size / 2, g, p, (const Lattice*)m->d_lat.p,
             (const int32_t*)m->d_tbl.p, slices > 1 ? m->d_part.p : m->d_resp.p, resp_stride, n_tiles, slices);
    // per-cell maxima + best response of every scan on (ncand / 256) x S blocks; the reduce block then only does the
    // order-dependent parts (best_bits was cleared by k_table_big)
    launch(ctx, "big_latmax", k_big_latmax, dim3((unsigned)((ncand + 63) / 64), S), dim3(256), 0, g, p, sc,
           (const Lattice*)m->d_lat.p, m->d_resp.p, resp_stride, m->d_big.p, stride, best_bits, fb_step,
           (const int32_t*)m->d_part.p, slices);
    if (S <= 16)
      launch(ctx, "reduce_coarse_big", k_reduce_coarse_big<1024>, dim3(S), dim3(1024), 0, g, p, sc, (const Lattice*)m->d_lat.p,
             m->d_resp.p, resp_stride, m->d_coarse.p, (int)m->cfg.use_response_expansion, pass_index,
             m->d_big.p, stride, (const uint8_t*)m->d_grid, (const double2*)m->d_local.p, fb_step,
             (const unsigned long long*)best_bits, big_final_out, big_final_out ? done_flag : (int*)nullptr, done_ticket);
    else
      launch(ctx, "reduce_coarse_big", k_reduce_coarse_big<256>, dim3(S), dim3(256), 0, g, p, sc, (const Lattice*)m->d_lat.p,
             m->d_resp.p, resp_stride, m->d_coarse.p, (int)m->cfg.use_response_expansion, pass_index,
             m->d_big.p, stride, (const uint8_t*)m->d_grid, (const double2*)m->d_local.p, fb_step,
             (const unsigned long long*)best_bits, big_final_out, big_final_out ? done_flag : (int*)nullptr, done_ticket);
    if (big_final_out) record_written = true;
    if (dbg_coarse_sums && pass_index == 0)  // after the reduce: it fills in scans the packed kernel skipped
      LSLAM_HIP(ctx, hipMemcpyAsync(dbg_coarse_sums, m->d_resp.p,
                                    (m->dbg_all ? (size_t)S * resp_stride : (size_t)p.nx * p.ny * p.na) * sizeof(int32_t),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    return LSLAM_OK;
  };
  auto run_coarse = [&](const PassCfg& p, int pass_index) -> int {
    if (!force_generic && p.nx > 16 && p.nx <= kDenseMaxNx) return run_coarse_big(p, pass_index);
    if (p.nx > 32 || p.ny > 32 || g.probs_side > 63)
      return ctx->fail(LSLAM_ERR_UNSUPPORTED, "lattice %dx%d is outside the built kernels", p.nx, p.ny);
    int rc = run_responses(p, 2, "resp_rows_coarse");
    if (rc) return rc;
    const bool cache = reduce_lds(p, true) <= 60 * 1024;
#define LSLAM_REDUCE_ARGS                                                                                     \
  g, p, sc, (const Lattice*)m->d_lat.p, m->d_resp.p, resp_stride, m->d_coarse.p,                              \
      (int)m->cfg.use_response_expansion, pass_index, (const uint8_t*)m->d_grid, (const double2*)m->d_local.p, \
      fb_step
    // the LDS form keeps no per-candidate cache, so it also covers lattices whose cached form would not fit (the reference's
    // shipped 16 x 16 x 21: 62 KB cached, 21 KB here -- k_reduce_coarse<false> was 0.36 ms of that configuration's 1.98 ms step)
    const int nt_sel = S >= kReduceNarrowMinScans ? 128 : (S <= 8 ? 1024 : 256);
    if (((size_t)p.nx * p.ny * p.na + 31) / 32 <= 256 && reduce_lds_nocache(p, reduce_parts(p, nt_sel)) <= 60 * 1024) {
      const bool fuse_fine = n_exp == 0 && do_refine;  // nothing between this pass and the fine pass
      // (the block also clears the fine numerators -- unless a debug caller is about to copy the COARSE ones out of the same words)
      const bool fuse_zero = fuse_fine && !(dbg_coarse_sums && pass_index == 0);
      // the front-end's speculative anchor chain of this scan rides in the lone launch (pass 0 only; see k_anchor_chain)
      SpecArgs spec{};
      const size_t spec_lds = (size_t)g.n_beams * (sizeof(double2) + 13) + 16;
      if (spec_req.out && pass_index == 0 && S == 1 && spec_lds <= 60 * 1024) {
        spec = spec_req;
        spec.local = (const double2*)m->d_local.p;  // of scan 0 of this match: the scan the request is for
        m->spec_launched = true;
      }
#define LSLAM_RC_LDS(NT)                                                                                                 \
  launch(ctx, "reduce_coarse", k_reduce_coarse_lds<NT>, dim3(S + (NT == 1024 && spec.out ? 1 : 0)), dim3(NT),                \
         std::max(reduce_lds_nocache(p, reduce_parts(p, NT)), NT == 1024 && spec.out ? spec_lds : (size_t)0), g, p,           \
         sc, m->d_lat.p, m->d_resp.p, resp_stride, m->d_coarse.p, (int)m->cfg.use_response_expansion, pass_index,            \
         (const uint8_t*)m->d_grid, (const double2*)m->d_local.p, fb_step, pf,                                               \
         fuse_fine ? m->d_cossin.p : (double2*)nullptr, 1, fuse_zero ? pf.nx * pf.ny * pf.na : 0, reduce_parts(p, NT),        \
         NT == 1024 ? spec : SpecArgs{})
      // 128 threads: residency for chip-filling batches; 1024: a lone block (streaming front-end, MatchScan) splits a cell's
      // angles over 8 threads -- its fill phase was 11 fp64 divisions in a row per thread
      if (S >= kReduceNarrowMinScans) LSLAM_RC_LDS(128);
      else if (S <= 8) LSLAM_RC_LDS(1024);
      else LSLAM_RC_LDS(256);
#undef LSLAM_RC_LDS
      setup_done = fuse_fine;
      fine_prezeroed = fuse_zero;
    }
    else if (cache)
      launch(ctx, "reduce_coarse", k_reduce_coarse<true>, dim3(S), dim3(256), reduce_lds(p, true), LSLAM_REDUCE_ARGS);
    else
      launch(ctx, "reduce_coarse", k_reduce_coarse<false>, dim3(S), dim3(256), reduce_lds(p, false), LSLAM_REDUCE_ARGS);
#undef LSLAM_REDUCE_ARGS
    if (dbg_coarse_sums && pass_index == 0)  // after the reduce: it fills in scans the packed kernel skipped
      LSLAM_HIP(ctx, hipMemcpyAsync(dbg_coarse_sums, m->d_resp.p,
                                    (m->dbg_all ? (size_t)S * resp_stride : (size_t)p.nx * p.ny * p.na) * sizeof(int32_t),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    return LSLAM_OK;
  };

  // ONE launch for the whole step (k_match_step) when the batch takes the tiled production kernels anyway and nothing
  // needs the numerators or the intermediate records in HBM: no expansion passes, refinement on, the <3,11> / <4,8>
  // coarse and the 3x3 fine instantiations, the LDS form of the coarse reduce.  Everything else -- and every batch below
  // step_min_scans -- keeps the five-kernel path below, which stays the reference point (LSLAM_OPT_STEP_KERNEL 0).
  if (m->step_waves > 0 && S >= m->step_min_scans) {
    const int cvar = (pc.nx <= 4 && pc.ny <= 4) ? 1 : (pc.nx <= 12) ? 2 : (pc.nx <= 16) ? 3 : 0;
    const size_t ctotal = (size_t)pc.nx * pc.ny * pc.na, ftotal = (size_t)pf.nx * pf.ny * pf.na;
    bool ok = !force_generic && n_exp == 0 && do_refine && !m->collect_stats && !m->lds_staged && !done_flag &&
              (cvar == 2 || cvar == 3) && pc.ny <= 32 && pf.nx == 3 && pf.ny == 3 && g.probs_side <= 63 &&
              g.n_beams <= 64 * kMaxBeamsPerLane && (ctotal + 31) / 32 <= 256 &&
              (ftotal + 31) / 32 <= 256 && !(prep_was_done && S == 1);
    if (ok) ok = ensure_ptiles();
    if (ok) ok = ensure_tiles4();
    if (ok) {
      const int W = m->step_waves, NT = 64 * W;
      const int parts = reduce_parts(pc, NT);
      const int area = cvar == 2 ? step_wave_area<3>() : step_wave_area<4>();
      const size_t num_bytes = ((std::max(ctotal, ftotal) * 4 + 15) / 16) * 16;
      const size_t phase_bytes = std::max<size_t>({(size_t)W * area, reduce_lds_nocache(pc, parts), ((ftotal + 31) / 32) * 4 + 16});
      const size_t lds = num_bytes + ((phase_bytes + 15) / 16) * 16;
      // the bitmap covers row spans of occ_win grid bytes: 2 (nX - 1) + 1 must fit
      const uint2* occ = (want_occ && 2 * (pc.nx - 1) + 1 <= m->occ_win) ? m->d_occ_x : (const uint2*)nullptr;
      StepGrid v{m->d_ptiles, occ, m->d_tiles, m->d_grid, g.data_size / 2, m->occ_wpc, m->ptile_rows, m->tile_cols,
                 (uint32_t)((size_t)m->ptile_rows * m->ptile_tx * 32)};
      int32_t* dbg_c = dbg_coarse_sums && m->dbg_all ? dbg_coarse_sums : (int32_t*)nullptr;
      if (dbg_coarse_sums && !m->dbg_all) ok = false;  // the single-scan debug entry reads d_resp's layout
      if (ok && lds <= 64 * 1024) {
#define LSLAM_STEP(NXD_, NYC_, W_)                                                                                       \
  launch(ctx, "match_step", k_match_step<NXD_, NYC_, W_, RT>, dim3(S), dim3(64 * W_), lds, d_ranges, stride, d_poses, g, pc, \
         pf, sc, v, m->d_lat.p, m->d_cossin.p, m->d_local.p, m->d_coarse.p, d_out, parts, (uint32_t)num_bytes, dbg_c,   \
         m->dbg_fine, resp_stride)
        if (cvar == 2 && W == 3) LSLAM_STEP(3, 11, 3);
        else if (cvar == 2 && W == 4) LSLAM_STEP(3, 11, 4);
        else if (cvar == 3 && W == 3) LSLAM_STEP(4, 8, 3);
        else if (cvar == 3 && W == 4) LSLAM_STEP(4, 8, 4);
        else ok = false;
#undef LSLAM_STEP
        if (ok) {
          m->step_launches++;
          LSLAM_HIP(ctx, hipGetLastError());
          return LSLAM_OK;
        }
      }
    }
  }
  // the five-kernel path starts with the prep (unless the rebuild's extra blocks ran it)
  if (!(prep_was_done && S == 1))
    launch(ctx, "scan_prep", k_scan_prep<RT>, dim3(S >= kReduceNarrowMinScans ? 1 : (g.n_beams + 255) / 256, S), dim3(256), 0, d_ranges,
           stride, d_poses, g, m->d_local.p, (double2*)nullptr, pc, m->d_lat.p, m->d_cossin.p, 2, PoseArg{});

  // ONE launch for the match of ONE scan (k_match_lone): the chain's four kernels behind device-side hand-overs.  The same
  // conditions as the kernels it replaces (<3,11> / <4,8> coarse, <1,4> fine, the LDS form of the coarse reduce, eight beam
  // slices at most), nothing that reads the numerators between the passes.
  if (m->lone_waves > 0 && S == 1 && !m->pipe_in_step) {
    const int cvar = (pc.nx <= 4 && pc.ny <= 4) ? 1 : (pc.nx <= 12) ? 2 : (pc.nx <= 16) ? 3 : 0;
    const size_t ctotal = (size_t)pc.nx * pc.ny * pc.na, ftotal = (size_t)pf.nx * pf.ny * pf.na;
    auto slices_for = [&](int na) {
      int sl = 1;
      while (sl < 8 && 8LL * na * sl < 2048) sl *= 2;
      while ((g.n_beams + 64 * sl - 1) / (64 * sl) > kMaxBeamsPerLane) sl *= 2;
      return sl;
    };
    const int W = m->lone_waves, NT = 64 * W, parts = reduce_parts(pc, NT);
    const int TW = m->lone_one_task_wave ? 1 : W;  // waves of a block that take response tasks
    const int c_slices = slices_for(pc.na), f_slices = slices_for(pf.na);
    const int area = cvar == 2 ? rows_wave_area<3>() : rows_wave_area<4>();
    const size_t lds = std::max<size_t>({(size_t)TW * area, reduce_lds_nocache(pc, parts), ((ftotal + 31) / 32) * 4 + 16});
    const int tasks = std::max(pc.na * c_slices, pf.na * f_slices);
    const bool ok = !force_generic && n_exp == 0 && do_refine && !m->collect_stats && !m->lds_staged && !dbg_coarse_sums &&
                    !m->dbg_fine && (cvar == 2 || cvar == 3) && pc.ny <= 32 && pf.nx <= 4 && pf.ny <= 4 && g.probs_side <= 63 &&
                    (ctotal + 31) / 32 <= 256 && (ftotal + 31) / 32 <= 256 && lds <= 60 * 1024 && c_slices > 1 && f_slices > 1 &&
                    (tasks + TW - 1) / TW <= 256;  // every block resident at once: one per CU
    if (ok) {
      if (!m->d_lone_sync) {
        LSLAM_HIP(ctx, hipMalloc((void**)&m->d_lone_sync, kLoneRing * sizeof(LoneSync)));
        LSLAM_HIP(ctx, hipMemsetAsync(m->d_lone_sync, 0, kLoneRing * sizeof(LoneSync), ctx->stream));
        m->lone_seq = 0;
      }
      if (prezeroed >= resp_stride) prezeroed = 0;  // k_rebuild_begin cleared the numerators; one use
      else LSLAM_HIP(ctx, hipMemsetAsync(m->d_resp.p, 0, resp_stride * sizeof(int32_t), ctx->stream));
      const unsigned nb = (unsigned)((tasks + TW - 1) / TW);
      const uint2* occ = (want_occ && 2 * (pc.nx - 1) + 1 <= m->occ_win) ? m->d_occ_x : (const uint2*)nullptr;
      const unsigned seq = m->lone_seq++;
#define LSLAM_LONE(NXD_, NYC_, W_, TW_)                                                                                     \
  launch(ctx, "match_lone", k_match_lone<NXD_, NYC_, W_, TW_>, dim3(nb), dim3(64 * W_), lds, (const uint8_t*)m->d_sub[0],     \
         (const uint8_t*)m->d_sub[1], (const uint8_t*)m->d_grid, g, pc, pf, sc, m->d_lat.p, m->d_cossin.p, m->d_local.p,      \
         m->d_resp.p, m->d_coarse.p, d_out, occ, m->occ_wpc, c_slices, f_slices, parts, m->d_lone_sync, seq, done_flag,       \
         done_ticket)
#define LSLAM_LONE_W(NXD_, NYC_)                                                                                            \
  do {                                                                                                                       \
    if (W == 4 && TW == 1) LSLAM_LONE(NXD_, NYC_, 4, 1);                                                                     \
    else if (W == 8 && TW == 1) LSLAM_LONE(NXD_, NYC_, 8, 1);                                                                \
    else if (TW == 1) LSLAM_LONE(NXD_, NYC_, 16, 1);                                                                         \
    else if (W == 4) LSLAM_LONE(NXD_, NYC_, 4, 4);                                                                           \
    else if (W == 8) LSLAM_LONE(NXD_, NYC_, 8, 8);                                                                           \
    else LSLAM_LONE(NXD_, NYC_, 16, 16);                                                                                     \
  } while (0)
      if (cvar == 2) LSLAM_LONE_W(3, 11);
      else LSLAM_LONE_W(4, 8);
#undef LSLAM_LONE_W
#undef LSLAM_LONE
      m->lone_launches++;
      LSLAM_HIP(ctx, hipGetLastError());
      return LSLAM_OK;
    }
  }

  int rc = run_coarse(pc, 0);
  if (rc) return rc;
  for (int e = 0; e < n_exp; e++) {
    rc = run_coarse(pe[e], e + 1);
    if (rc) return rc;
  }
  if (do_refine) {
    rc = run_responses(pf, 1, "resp_rows_fine");
    if (rc) return rc;
  }
  if (record_written) {
    LSLAM_HIP(ctx, hipGetLastError());
    return LSLAM_OK;
  }
  if (S >= kReduceNarrowMinScans)
    launch(ctx, "reduce_fine", k_reduce_fine<64>, dim3(S), dim3(64), (size_t)(((size_t)pf.nx * pf.ny * pf.na + 31) / 32) * 4 + 16,
           (const uint8_t*)m->d_grid, g, pf, sc, (const Lattice*)m->d_lat.p, m->d_resp.p, resp_stride,
           (const double2*)m->d_local.p, (const CoarseOut*)m->d_coarse.p, d_out, do_refine, do_refine ? fb_step : 0,
           (int*)nullptr, 0);
  else
    launch(ctx, "reduce_fine", k_reduce_fine<256>, dim3(S), dim3(256), (size_t)(((size_t)pf.nx * pf.ny * pf.na + 31) / 32) * 4 + 16,
           (const uint8_t*)m->d_grid, g, pf, sc, (const Lattice*)m->d_lat.p, m->d_resp.p, resp_stride,
           (const double2*)m->d_local.p, (const CoarseOut*)m->d_coarse.p, d_out, do_refine, do_refine ? fb_step : 0,
           done_flag, done_ticket);
  if (m->dbg_fine)  // behind the reduce: its blocks fill in the scans the packed kernel skipped
    LSLAM_HIP(ctx, hipMemcpyAsync(m->dbg_fine, m->d_resp.p, (size_t)S * resp_stride * sizeof(int32_t), hipMemcpyDeviceToDevice,
                                  ctx->stream));
  LSLAM_HIP(ctx, hipGetLastError());
  return LSLAM_OK;
}

// Streams and events of the pipelined steps, created on first use.
int pipe_init(lslam_matcher* m) {
  lslam_context* ctx = m->ctx;
  if (!m->view_ready) LSLAM_HIP(ctx, hipEventCreateWithFlags(&m->view_ready, hipEventDisableTiming));
  for (int i = 0; i < m->pipe_depth; i++) {
    if (!m->pipe_stream[i]) LSLAM_HIP(ctx, hipStreamCreateWithFlags(&m->pipe_stream[i], hipStreamNonBlocking));
    if (!m->pipe_done[i]) LSLAM_HIP(ctx, hipEventCreateWithFlags(&m->pipe_done[i], hipEventDisableTiming));
    if (!m->pipe_in[i]) LSLAM_HIP(ctx, hipEventCreateWithFlags(&m->pipe_in[i], hipEventDisableTiming));
  }
  if (!m->pipe_registered) {  // lslam_synchronize(ctx) means "every step is done" too
    ctx->pre_sync.emplace_back((void*)m, &pipe_join_cb);
    m->pipe_registered = true;
  }
  return LSLAM_OK;
}

// One batched match as a pipelined step: the same kernels with the same arguments as match_batch_impl on the context
// stream -- only on the next internal stream, in that slot's own workspaces.  Ordering: behind everything the context
// stream held when the call was made (inputs, grid changes) and behind the last refresh of a grid view by another
// slot; NOT behind the previous steps -- that is the point.  The context stream itself falls in behind the steps at
// the next pipe_join (lslam_synchronize, any entry point that touches the grid or reads through the context stream).
template <typename RT>
int pipe_step(lslam_matcher* m, int S, const RT* d_ranges, int stride, const double* d_poses, int do_penalize,
              int do_refine, lslam_match_result* d_out) {
  lslam_context* ctx = m->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  int rc = pipe_init(m);
  if (rc) return rc;
  const int slot = m->pipe_next % m->pipe_depth;
  if (m->check_out_reuse) {
    // Up to pipe_depth steps are in flight: the step that ran on THIS slot before is finished by stream order (this step
    // queues behind it), every other pending slot may still be writing its records.
    const char* lo = (const char*)d_out;
    const char* hi = lo + (size_t)S * sizeof(lslam_match_result);
    for (int i = 0; i < m->pipe_depth; i++)
      if (i != slot && m->pipe_pending[i] && m->pipe_out_lo[i] && lo < m->pipe_out_hi[i] && m->pipe_out_lo[i] < hi)
        return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT,
                         "pipelined step: its result buffer overlaps the one of a step still in flight (slot %d); with "
                         "LSLAM_OPT_PIPELINE_DEPTH %d consecutive calls need %d distinct result buffers",
                         i, m->pipe_depth, m->pipe_depth);
    m->pipe_out_lo[slot] = lo;
    m->pipe_out_hi[slot] = hi;
  }
  m->pipe_next = (slot + 1) % m->pipe_depth;
  hipStream_t s = m->pipe_stream[slot];
  // "behind everything the context stream held": nothing to wait for when that stream has drained (the steady state of
  // back-to-back steps -- the query costs the host a microsecond, the record + wait pair cost the GPU ~10 us per step)
  const hipError_t idle = hipStreamQuery(ctx->stream);
  if (idle != hipSuccess) {
    (void)hipGetLastError();
    if (idle != hipErrorNotReady) LSLAM_HIP(ctx, idle);
    LSLAM_HIP(ctx, hipEventRecord(m->pipe_in[slot], ctx->stream));  // (one event per slot: never re-recorded under a pending wait of another stream)
    LSLAM_HIP(ctx, hipStreamWaitEvent(s, m->pipe_in[slot], 0));
  }
  if (m->view_owner >= 0 && m->view_owner != slot) LSLAM_HIP(ctx, hipStreamWaitEvent(s, m->view_ready, 0));
  const bool v0[4] = {m->sub_dirty, m->occ_dirty, m->tile_dirty, m->ptile_dirty};
  const void* const a0[2] = {m->d_tiles, m->d_ptiles};
  if (slot) swap_step_work(m, m->pipe_work[slot]);
  hipStream_t saved = ctx->stream;
  ctx->stream = s;  // launch(), the memsets and the timer events of match_batch_impl all go through ctx->stream
  // whatever gets enqueued, failed call or not, is fenced by this slot's event: pipe_join records it, once, and it covers
  // everything the stream holds by then
  m->pipe_pending[slot] = true;
  m->pipe_in_step = true;
  rc = match_batch_impl<RT>(m, S, d_ranges, stride, d_poses, do_penalize, do_refine, d_out, nullptr, 0);
  m->pipe_in_step = false;
  ctx->stream = saved;
  if (slot) swap_step_work(m, m->pipe_work[slot]);
  const bool refreshed = v0[0] != m->sub_dirty || v0[1] != m->occ_dirty || v0[2] != m->tile_dirty ||
                         v0[3] != m->ptile_dirty || a0[0] != m->d_tiles || a0[1] != m->d_ptiles;
  if (refreshed) {
    LSLAM_HIP(ctx, hipEventRecord(m->view_ready, s));
    m->view_owner = slot;
  }
  m->pipe_steps++;
  return rc;
}

// The batched device entry points: a pipelined step when the option asks for it and nothing stands in the way (the
// instrumented / experimental kernels keep counters in shared buffers; flags a grid rebuild left for the next plain
// match belong to slot 0's buffers on the context stream).
template <typename RT>
int match_batch_dev_entry(lslam_matcher* m, int S, const RT* d_ranges, int stride, const double* d_poses, int do_penalize,
                          int do_refine, lslam_match_result* d_out) {
  if (S <= 0) return LSLAM_OK;
  const bool piped = m->pipe_depth > 1 && !m->collect_stats && !m->lds_staged && !m->prep_done && !m->resp_prezeroed &&
                     !m->arm_next && m->g.n_beams > 0;
  if (piped) return pipe_step<RT>(m, S, d_ranges, stride, d_poses, do_penalize, do_refine, d_out);
  return match_batch_impl<RT>(m, S, d_ranges, stride, d_poses, do_penalize, do_refine, d_out, nullptr, 0);
}

constexpr int kRebuildNeedsContiguous = 1;  // internal return code of rebuild_grid_dev (never crosses the ABI)
// AddScans on the device (Mapper.cpp:699-748) from world points already resident in HBM:
// recentre, clear, FindValidPoints, mark + smear.  `world` is a ring of `cap` scans of n points.
int rebuild_grid_dev(lslam_matcher* m, const double2* d_world, int ring_start, int B, int cap, const double center[3],
                     const RebuildExtras* extras = nullptr) {
  lslam_context* ctx = m->ctx;
  {  // pipelined steps still in flight read the grid this is about to rewrite
    int jrc = pipe_join(m);
    if (jrc) return jrc;
  }
  Geom& g = m->g;
  // Mapper.cpp:212-220: offset = scanPose - 0.5*(roi-1)*resolution
  g.off_x = center[0] - (0.5 * (g.roi_w - 1) * (1.0 / g.scale));
  g.off_y = center[1] - (0.5 * (g.roi_h - 1) * (1.0 / g.scale));
  const int n = g.n_beams;
  const size_t lds = (size_t)n * (sizeof(double2) + 14) + 16;  // points, next, two jump tables, valid, reach
  const int use_lds = lds <= 60 * 1024;
  // The atomics-free rebuild: k_find_valid tags the centres in a mark plane with this rebuild's epoch and the smear is
  // a gather that writes EVERY grid byte -- no list, no atomics, and no clear of anything (the plane is reset when the
  // 8-bit epoch wraps).
  bool fuse_mark = m->kernel_center_only && use_lds && g.kernel_size <= kMaxKernel && B > 0 && n > 0;
  if (fuse_mark && !m->d_marks_alloc) {
    if (hipMalloc((void**)&m->d_marks_alloc, (size_t)g.data_size + 2 * kGuard) != hipSuccess) {
      (void)hipGetLastError();
      m->d_marks_alloc = nullptr;
      fuse_mark = false;
    } else {
      m->d_marks = m->d_marks_alloc + kGuard;
      m->mark_epoch = 0;
    }
  }
  // a window named by a slot list (scan cache) exists only in the clear-free form: the caller gathers the scans into a
  // contiguous workspace for the configurations it does not cover
  if (extras && extras->slot_list && !fuse_mark && B > 0 && n > 0) return kRebuildNeedsContiguous;
  if (fuse_mark && (m->mark_epoch == 0 || m->mark_epoch >= 255)) {
    LSLAM_HIP(ctx, hipMemsetAsync(m->d_marks_alloc, 0, (size_t)g.data_size + 2 * kGuard, ctx->stream));
    m->mark_epoch = 0;
  }
  if (fuse_mark) m->mark_epoch++;
  // The front-end's extras (query pose, readings, zeros, the query scan's prep) ride in k_find_valid's launch when the
  // rebuild is clear-free; otherwise a first launch clears the grid (Grid::Clear, Mapper.cpp:701) and carries them.
  RebuildExtras x{};
  if (extras) x = *extras;
  m->prep_done = false;
  if (x.prep_ranges && g.n_beams > 0) {
    const PassCfg pc = coarse_pass_cfg(m, g);
    if (pc.nx <= kMaxLattice && pc.ny <= kMaxLattice && pc.na <= kMaxAngles && pc.nx >= 1 && pc.na >= 1) {
      LSLAM_HIP(ctx, m->d_local.reserve((size_t)g.n_beams));
      LSLAM_HIP(ctx, m->d_lat.reserve(1));
      LSLAM_HIP(ctx, m->d_cossin.reserve((size_t)kMaxAngles));
      x.prep_local = m->d_local.p;
      x.prep_lat = m->d_lat.p;
      x.prep_cossin = m->d_cossin.p;
      x.prep_pc = pc;
      x.prep_g = g;  // with the new grid offset
      m->prep_done = true;
    } else {
      x.prep_ranges = nullptr;
    }
  } else {
    x.prep_ranges = nullptr;
  }
  const bool have_extras = x.pose_dst || x.zero || x.ranges_src || x.prep_ranges;
  if (!fuse_mark) {
    const size_t n16 = ((size_t)g.data_size + 15) / 16;  // rounds up inside the zero guard band behind the grid
    const size_t threads = std::max<size_t>(n16, std::max<size_t>(x.zero ? (size_t)x.zero_words : 0,
                                                                  x.ranges_src ? (size_t)x.n_ranges : 0));
    unsigned blocks = (unsigned)((threads + 255) / 256);
    x.clear_blocks = (int)blocks;
    if (x.prep_ranges) blocks += (unsigned)((g.n_beams + 255) / 256);
    launch(ctx, "grid_clear", k_rebuild_begin, dim3(blocks), dim3(256), 0, (uint4*)m->d_grid, n16, x);
  }
  m->resp_prezeroed = x.zero == m->d_resp.p && x.zero ? (size_t)x.zero_words : 0;
  m->sub_dirty = m->occ_dirty = m->tile_dirty = m->ptile_dirty = true;
  if (B <= 0 || n <= 0) return LSLAM_OK;
  LSLAM_HIP(ctx, m->d_valid.reserve((size_t)B * n));
  if (!use_lds) LSLAM_HIP(ctx, m->d_fv_scratch.reserve((size_t)B * 2 * n));
  if (m->kernel_center_only && !fuse_mark) {
    LSLAM_HIP(ctx, m->d_centres.reserve((size_t)B * n + 1));
    LSLAM_HIP(ctx, hipMemsetAsync(m->d_centres.p, 0, sizeof(uint32_t), ctx->stream));
  }
  const int fv_threads = n > 512 ? 1024 : 256;
  const int extra_blocks = (fuse_mark && have_extras) ? (x.prep_ranges ? (n + fv_threads - 1) / fv_threads : 0) + 1 : 0;
  launch(ctx, "find_valid", k_find_valid, dim3(B + extra_blocks), dim3(fv_threads), use_lds ? lds : 0, n, d_world, ring_start,
         cap, center[0], center[1], m->d_valid.p, use_lds, m->d_fv_scratch.p, g, fuse_mark ? m->d_marks : (uint8_t*)nullptr,
         extras ? extras->anchor_ring : (const int*)nullptr, m->mark_epoch, B, x);
  if (fuse_mark) {
    const dim3 sg((unsigned)(((size_t)g.data_size + 4095) / 4096));
#define LSLAM_SMEAR(HK) launch(ctx, "smear", k_smear_gather<HK>, sg, dim3(256), 0, g, (const uint8_t*)m->d_kernel, (const uint8_t*)m->d_marks, (uint32_t)m->mark_epoch, m->d_grid, m->d_sub[0], m->d_sub[1])
    switch (g.kernel_size / 2) {
      case 1: LSLAM_SMEAR(1); break;
      case 2: LSLAM_SMEAR(2); break;
      case 6: LSLAM_SMEAR(6); break;
      default: LSLAM_SMEAR(0);
    }
#undef LSLAM_SMEAR
    m->sub_dirty = false;  // the gather pass wrote the parity planes too
  } else if (m->kernel_center_only) {
    // (1) centres: the first point to reach a cell sets it to 100 and is listed; (2) every listed centre
    // smears, one thread per aligned word of its footprint
    if (!fuse_mark)
      launch(ctx, "mark_centres", k_mark_centres, dim3((n + 255) / 256, B), dim3(256), 0, B, n, d_world, ring_start, cap,
             (const uint8_t*)m->d_valid.p, g, m->d_grid, m->d_centres.p + 1, (int*)m->d_centres.p);
    const int wpr = (g.kernel_size + 3) / 4 + 1;
    const long long threads = (long long)B * n * g.kernel_size * wpr;
    launch(ctx, "smear", k_smear_list, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0,
           (const uint32_t*)(m->d_centres.p + 1), (const int*)m->d_centres.p, g, (const uint8_t*)m->d_kernel, m->d_grid,
           wpr);
  } else
    launch(ctx, "add_scans_serial", k_add_scans_serial, dim3(1), dim3(64), 0, B, n, d_world, ring_start, cap,
           (const uint8_t*)m->d_valid.p, g, (const uint8_t*)m->d_kernel, m->d_grid);
  LSLAM_HIP(ctx, hipGetLastError());
  return LSLAM_OK;
}

int upload_scans(lslam_matcher* m, int S, const double* ranges, int stride, const double* poses) {
  lslam_context* ctx = m->ctx;
  const int n = m->g.n_beams;
  if (stride < n) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "ranges_stride %d < num_beams %d", stride, n);
  LSLAM_HIP(ctx, m->d_ranges64.reserve((size_t)S * std::max(n, 1)));
  LSLAM_HIP(ctx, m->d_poses.reserve((size_t)S * 3));
  if (n > 0)
    LSLAM_HIP(ctx, hipMemcpy2DAsync(m->d_ranges64.p, (size_t)n * sizeof(double), ranges, (size_t)stride * sizeof(double),
                                    (size_t)n * sizeof(double), S, hipMemcpyHostToDevice, ctx->stream));
  LSLAM_HIP(ctx, hipMemcpyAsync(m->d_poses.p, poses, (size_t)S * 3 * sizeof(double), hipMemcpyHostToDevice,
                                ctx->stream));
  return LSLAM_OK;
}

}  // namespace

extern "C" {

void lslam_matcher_config_defaults(lslam_matcher_config* c) {
  // Mapper.cpp:1572-1647
  c->search_size = 0.3;
  c->resolution = 0.01;
  c->smear_deviation = 0.03;
  c->range_threshold = 12.0;  // LaserRangeFinder default RangeThreshold (Karto.h:4137)
  c->coarse_search_angle_offset = 20.0 * kPi180;
  c->coarse_angle_resolution = 2.0 * kPi180;
  c->fine_search_angle_offset = 0.2 * kPi180;
  c->distance_variance_penalty = 0.3 * 0.3;
  c->angle_variance_penalty = (20.0 * kPi180) * (20.0 * kPi180);
  c->minimum_distance_penalty = 0.5;
  c->minimum_angle_penalty = 0.9;
  c->use_response_expansion = 0;
  c->reserved = 0;
}

void lslam_sensor_pose_from_robot(const lslam_laser* l, const double robot[3], double sensor[3]) {
  // GetSensorAt: Transform(rPose).TransformPose(offsetPose) (Karto.h:5310-5313, 2881-2887)
  SensorXform t = sensor_xform(robot[0], robot[1], robot[2]);
  double rx, ry;
  rot_apply(t.rot, l->offset_x, l->offset_y, l->offset_heading, rx, ry);
  sensor[0] = t.tx + rx;
  sensor[1] = t.ty + ry;
  sensor[2] = normalize_angle(l->offset_heading + t.th);
}

void lslam_robot_pose_from_sensor(const lslam_laser* l, const double sensor[3], double robot[3]) {
  // SetSensorPose (Karto.h:5289-5303)
  double len = sqrt(ksq(l->offset_x) + ksq(l->offset_y));
  double angleoffset = atan2(l->offset_y, l->offset_x);
  double ch = normalize_angle(sensor[2]);
  double wx = len * cos(ch + angleoffset - l->offset_heading);
  double wy = len * sin(ch + angleoffset - l->offset_heading);
  robot[0] = sensor[0] - wx;
  robot[1] = sensor[1] - wy;
  robot[2] = normalize_angle(sensor[2] - l->offset_heading);
}

int lslam_matcher_create(lslam_context* ctx, const lslam_matcher_config* cfg, const lslam_laser* laser,
                         lslam_matcher** out) {
  if (!ctx || !cfg || !laser || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  // ScanMatcher::Create returns NULL for these (Mapper.cpp:130-145)
  if (!(cfg->resolution > 0) || !(cfg->search_size > 0) || cfg->smear_deviation < 0 || !(cfg->range_threshold > 0))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "invalid matcher parameters (ScanMatcher::Create -> NULL)");
  // CalculateKernel throws for these (Mapper.h:1041-1053)
  if (!(cfg->smear_deviation >= 0.5 * cfg->resolution && cfg->smear_deviation <= 10 * cfg->resolution))
    return ctx->fail(LSLAM_ERR_SMEAR_DEVIATION, "smear deviation must be within [%g, %g]", 0.5 * cfg->resolution,
                     10 * cfg->resolution);
  if (!(laser->angular_resolution != 0.0))
    return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "laser angular_resolution must be non-zero");
  lslam_matcher* m = new lslam_matcher();
  m->ctx = ctx;
  m->cfg = *cfg;
  m->laser = *laser;
  Geom& g = m->g;
  g.n_beams = (int)(uint32_t)kround((laser->maximum_angle - laser->minimum_angle) / laser->angular_resolution);
  uint32_t side = (uint32_t)(kround(cfg->search_size / cfg->resolution) + 1);  // Mapper.cpp:150
  uint32_t margin = (uint32_t)ceil(cfg->range_threshold / cfg->resolution);    // :154
  int grid_size = (int)(side + 2 * margin);                                     // :156
  int half = (int)kround(2.0 * cfg->smear_deviation / cfg->resolution);         // Mapper.h:1096-1101
  g.border = half + 1;                                                          // Mapper.h:928
  g.roi_w = g.roi_h = grid_size;
  g.width = g.height = grid_size + 2 * g.border;  // Mapper.h:1018
  g.stride = (g.width + 7) & ~7;                   // Karto.h:4442
  if (g.stride > kMaxGridSide || g.height > kMaxGridSide) {  // keeps every flat index inside int32 (k_resp_rows)
    delete m;
    return ctx->fail(LSLAM_ERR_UNSUPPORTED, "correlation grid of %dx%d cells exceeds the built limit %d per side",
                     g.width, g.height, kMaxGridSide);
  }
  g.data_size = g.stride * g.height;
  g.scale = 1.0 / cfg->resolution;  // Mapper.h:1020
  g.off_x = g.off_y = 0.0;
  g.min_angle = laser->minimum_angle;
  g.ang_res = laser->angular_resolution;
  g.probs_side = (int)side;
  if (g.probs_side > kMaxProbsSide) {
    delete m;
    return ctx->fail(LSLAM_ERR_UNSUPPORTED, "search space of %d cells per side exceeds the built limit %d",
                     g.probs_side, kMaxProbsSide);
  }
  // CalculateKernel (Mapper.h:1058-1086), host fp64
  const double resolution = 1.0 / g.scale;
  g.kernel_size = 2 * (int)kround(2.0 * cfg->smear_deviation / resolution) + 1;
  const int hk = g.kernel_size / 2;
  m->h_kernel.resize((size_t)g.kernel_size * g.kernel_size);
  for (int i = -hk; i <= hk; i++)
    for (int j = -hk; j <= hk; j++) {
      double d = hypot(i * resolution, j * resolution);
      double z = exp(-0.5 * pow(d / cfg->smear_deviation, 2));
      uint32_t v = (uint32_t)kround(z * kOccupied);
      m->h_kernel[(i + hk) + g.kernel_size * (j + hk)] = (uint8_t)v;
      if (v >= (uint32_t)kOccupied && (i != 0 || j != 0)) m->kernel_center_only = false;
    }
  if (hipSetDevice(ctx->device) != hipSuccess ||
      hipMalloc((void**)&m->d_grid_alloc, (size_t)g.data_size + 2 * kGuard) != hipSuccess ||
      hipMalloc((void**)&m->d_kernel, m->h_kernel.size()) != hipSuccess) {
    if (m->d_grid_alloc) (void)hipFree(m->d_grid_alloc);
    delete m;
    return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the correlation grid in HBM");
  }
  m->d_grid = m->d_grid_alloc + kGuard;
  const size_t plane = (size_t)g.data_size / 2 + 2 * kGuard;  // data_size is a multiple of 8
  if (hipMalloc((void**)&m->d_sub_alloc, 2 * plane) != hipSuccess) {
    (void)hipFree(m->d_grid_alloc);
    (void)hipFree(m->d_kernel);
    delete m;
    return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the grid parity planes in HBM");
  }
  {  // row span of the coarse lattice: 2-cell steps over nX candidates (Mapper.cpp:228-234)
    const double res = 1.0 / g.scale;
    const int nx = lattice_count(0.5 * ((double)g.probs_side - 1) * res, 2 * res);
    m->occ_win = std::min(kOccWinMax, std::max(3, 2 * (nx - 1) + 1));
  }
  {  // SearchCfg::fast_div: the reciprocal form of the response normalisation, checked against the division for every
     // numerator a response sum can take (0 .. nBeams * 100; a beam adds at most GridStates_Occupied = 100)
    const double d = (double)((uint32_t)g.n_beams * (uint32_t)kOccupied), inv = 1.0 / d;
    bool same = g.n_beams > 0 && (long long)g.n_beams * kOccupied <= (1 << 26);
    for (long long sum = 0; same && sum <= (long long)g.n_beams * kOccupied; sum++) {
      const double a = (double)sum, q = a * inv, fast = std::fma(std::fma(-q, d, a), inv, q), exact = a / d;
      same = memcmp(&fast, &exact, sizeof fast) == 0;
    }
    m->fast_div = same ? 1 : 0;
    m->inv_denom = inv;
  }
  m->occ_wpc = 2 * (((g.height + 2) / 2 + 1 + 31) / 32 + 1);  // two row-parity bitmaps per column, +1 word each for the 64-bit read
  m->nz_words = (g.data_size + 31) / 32;
  if (hipMalloc((void**)&m->d_occ_t, (size_t)g.stride * m->occ_wpc * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc((void**)&m->d_occ_x, (size_t)g.stride * m->occ_wpc * sizeof(uint2)) != hipSuccess ||
      hipMalloc((void**)&m->d_nz, (size_t)m->nz_words * sizeof(uint32_t)) != hipSuccess) {
    (void)hipFree(m->d_grid_alloc);
    (void)hipFree(m->d_kernel);
    (void)hipFree(m->d_sub_alloc);
    (void)hipFree(m->d_occ_t);
    (void)hipFree(m->d_occ_x);
    (void)hipFree(m->d_nz);
    delete m;
    return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the row-occupancy bitmap in HBM");
  }
  m->d_sub[0] = m->d_sub_alloc + kGuard;
  m->d_sub[1] = m->d_sub_alloc + plane + kGuard;
  (void)hipMemsetAsync(m->d_sub_alloc, 0, 2 * plane, ctx->stream);
  (void)hipMemsetAsync(m->d_grid_alloc, 0, (size_t)g.data_size + 2 * kGuard, ctx->stream);
  (void)hipMemcpyAsync(m->d_kernel, m->h_kernel.data(), m->h_kernel.size(), hipMemcpyHostToDevice, ctx->stream);
  (void)hipStreamSynchronize(ctx->stream);
  *out = m;
  return LSLAM_OK;
}

void lslam_matcher_destroy(lslam_matcher* m) {
  if (!m) return;
  (void)hipSetDevice(m->ctx->device);
  (void)pipe_join(m);
  (void)hipStreamSynchronize(m->ctx->stream);
  if (m->pipe_registered) {
    auto& ps = m->ctx->pre_sync;
    for (size_t i = 0; i < ps.size(); i++)
      if (ps[i].first == (void*)m) {
        ps.erase(ps.begin() + i);
        break;
      }
  }
  for (int i = 0; i < lslam_matcher::kMaxPipe; i++) {
    if (m->pipe_stream[i]) {
      (void)hipStreamSynchronize(m->pipe_stream[i]);
      (void)hipStreamDestroy(m->pipe_stream[i]);
    }
    if (m->pipe_done[i]) (void)hipEventDestroy(m->pipe_done[i]);
    if (m->pipe_in[i]) (void)hipEventDestroy(m->pipe_in[i]);
  }
  if (m->view_ready) (void)hipEventDestroy(m->view_ready);
  for (auto& w : m->pipe_work) {
    w.d_local.release(); w.d_cossin.release(); w.d_lat.release(); w.d_coarse.release(); w.d_resp.release();
    w.d_tbl.release(); w.d_part.release(); w.d_big.release();
  }
  (void)hipFree(m->d_grid_alloc);
  if (m->d_marks_alloc) (void)hipFree(m->d_marks_alloc);
  (void)hipFree(m->d_kernel);
  (void)hipFree(m->d_sub_alloc);
  (void)hipFree(m->d_occ_t);
  (void)hipFree(m->d_occ_x);
  (void)hipFree(m->d_nz);
  (void)hipFree(m->d_tiles);
  (void)hipFree(m->d_ptiles);
  m->d_ranges64.release(); m->d_poses.release(); m->d_local.release(); m->d_world.release();
  m->d_valid.release(); m->d_fv_scratch.release(); m->d_centres.release(); m->d_lat.release(); m->d_cossin.release(); m->d_coarse.release(); m->d_resp.release();
  m->d_tbl.release(); m->d_part.release(); m->d_big.release(); m->d_results.release(); m->d_dbg.release();
  m->d_query.release(); m->d_qpose.release();
  if (m->h_done) (void)hipHostFree(m->h_done);
  if (m->d_lone_sync) (void)hipFree(m->d_lone_sync);
  if (m->h_query) (void)hipHostFree(m->h_query);
  if (m->h_result) (void)hipHostFree(m->h_result);
  delete m;
}

int lslam_matcher_num_beams(const lslam_matcher* m) { return m ? m->g.n_beams : LSLAM_ERR_INVALID_ARGUMENT; }

int lslam_matcher_grid_info(const lslam_matcher* m, int32_t out[8], double offset_xy[2]) {
  if (!m) return LSLAM_ERR_INVALID_ARGUMENT;
  const Geom& g = m->g;
  out[0] = g.width; out[1] = g.height; out[2] = g.stride; out[3] = g.border; out[4] = g.border;
  out[5] = g.roi_w; out[6] = g.roi_h; out[7] = g.kernel_size;
  if (offset_xy) { offset_xy[0] = g.off_x; offset_xy[1] = g.off_y; }
  return LSLAM_OK;
}

int lslam_matcher_get_grid_u8(lslam_matcher* m, uint8_t* out) {
  if (!m || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  {
    int jrc = pipe_join(m);
    if (jrc) return jrc;
  }
  LSLAM_HIP(ctx, hipMemcpyAsync(out, m->d_grid, (size_t)m->g.data_size, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_matcher_get_kernel_u8(lslam_matcher* m, uint8_t* out) {
  if (!m || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  LSLAM_HIP(ctx, hipMemcpyAsync(out, m->d_kernel, m->h_kernel.size(), hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_matcher_set_grid_u8(lslam_matcher* m, const uint8_t* grid, const double offset_xy[2]) {
  if (!m || !grid || !offset_xy) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  {
    int jrc = pipe_join(m);  // steps in flight read the grid
    if (jrc) return jrc;
  }
  LSLAM_HIP(ctx, hipMemcpyAsync(m->d_grid, grid, (size_t)m->g.data_size, hipMemcpyHostToDevice, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  m->g.off_x = offset_xy[0];
  m->g.off_y = offset_xy[1];
  m->sub_dirty = m->occ_dirty = m->tile_dirty = m->ptile_dirty = true;
  return LSLAM_OK;
}

int lslam_matcher_set_grid_u8_dev(lslam_matcher* m, const uint8_t* grid_dev, const double offset_xy[2]) {
  if (!m || !grid_dev || !offset_xy) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  {
    int jrc = pipe_join(m);  // steps in flight read the grid and its views
    if (jrc) return jrc;
  }
  if (grid_dev != m->d_grid)
    LSLAM_HIP(ctx, hipMemcpyAsync(m->d_grid, grid_dev, (size_t)m->g.data_size, hipMemcpyDeviceToDevice, ctx->stream));
  m->g.off_x = offset_xy[0];
  m->g.off_y = offset_xy[1];
  m->sub_dirty = m->occ_dirty = m->tile_dirty = m->ptile_dirty = true;
  return LSLAM_OK;
}

void* lslam_matcher_grid_dev_ptr(lslam_matcher* m) { return m ? (void*)m->d_grid : nullptr; }

int lslam_matcher_set_option(lslam_matcher* m, int option, int value) {
  if (!m) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  switch (option) {
    case LSLAM_OPT_ROW_OCCUPANCY:
      m->use_row_occupancy = value != 0;
      return LSLAM_OK;
    case LSLAM_OPT_LDS_STAGED:
      m->lds_staged = value != 0;
      return LSLAM_OK;
    case LSLAM_OPT_PIPELINE_DEPTH: {
      if (value < 1 || value > lslam_matcher::kMaxPipe)
        return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "pipeline depth %d outside [1, %d]", value, lslam_matcher::kMaxPipe);
      LSLAM_HIP(ctx, hipSetDevice(ctx->device));
      int rc = pipe_join(m);  // steps in flight keep their slots; the next step starts a fresh rotation
      if (rc) return rc;
      m->pipe_depth = value;
      m->pipe_next = 0;
      return LSLAM_OK;
    }
    case LSLAM_OPT_STEP_KERNEL: {
      if (value != 0 && value != 3 && value != 4)
        return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "step kernel: 0 (five launches per step), 3 or 4 (waves per scan), not %d", value);
      LSLAM_HIP(ctx, hipSetDevice(ctx->device));
      int rc = pipe_join(m);
      if (rc) return rc;
      m->step_waves = value;
      return LSLAM_OK;
    }
    case LSLAM_OPT_CHECK_OUTPUT_REUSE:
      m->check_out_reuse = value != 0;
      return LSLAM_OK;
    case LSLAM_OPT_LONE_KERNEL: {
      const int wv = value >= 100 ? value - 100 : value;  // + 100: one task wave per block (the A/B reference)
      if (wv != 0 && wv != 4 && wv != 8 && wv != 16)
        return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "lone kernel: 0 (four launches per match), 4, 8 or 16 (waves per block), not %d", value);
      LSLAM_HIP(ctx, hipSetDevice(ctx->device));
      int rc = pipe_join(m);
      if (rc) return rc;
      m->lone_waves = wv;
      m->lone_one_task_wave = value >= 100;
      return LSLAM_OK;
    }
    case LSLAM_OPT_ROWS_WAVES: {
      if (value != 1 && value != 2 && value != 4 && value != 8)
        return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "coarse kernel: 1, 2, 4 or 8 waves per block, not %d", value);
      LSLAM_HIP(ctx, hipSetDevice(ctx->device));
      int rc = pipe_join(m);
      if (rc) return rc;
      m->rows_waves = value;
      return LSLAM_OK;
    }
    case LSLAM_OPT_STEP_MIN_SCANS: {
      if (value < 1) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "step kernel: minimum batch %d < 1", value);
      m->step_min_scans = value;
      return LSLAM_OK;
    }
    case LSLAM_OPT_COLLECT_STATS:
      if (value) {
        LSLAM_HIP(ctx, hipSetDevice(ctx->device));
        LSLAM_HIP(ctx, m->d_stats.reserve(8));
        LSLAM_HIP(ctx, hipMemsetAsync(m->d_stats.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
        m->stats_scans = 0;
      }
      m->collect_stats = value != 0;
      return LSLAM_OK;
    default:
      return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "unknown matcher option %d", option);
  }
}

int lslam_matcher_get_option(const lslam_matcher* m, int option) {
  if (!m) return LSLAM_ERR_INVALID_ARGUMENT;
  switch (option) {
    case LSLAM_OPT_ROW_OCCUPANCY: return m->use_row_occupancy ? 1 : 0;
    case LSLAM_OPT_COLLECT_STATS: return m->collect_stats ? 1 : 0;
    case LSLAM_OPT_LDS_STAGED: return m->lds_staged ? 1 : 0;
    case LSLAM_OPT_PIPELINE_DEPTH: return m->pipe_depth;
    case LSLAM_OPT_STEP_KERNEL: return m->step_waves;
    case LSLAM_OPT_STEP_MIN_SCANS: return m->step_min_scans;
    case LSLAM_OPT_ROWS_WAVES: return m->rows_waves;
    case LSLAM_OPT_CHECK_OUTPUT_REUSE: return m->check_out_reuse ? 1 : 0;
    case LSLAM_OPT_LONE_KERNEL: return m->lone_waves + (m->lone_waves && m->lone_one_task_wave ? 100 : 0);
    default: return LSLAM_ERR_INVALID_ARGUMENT;
  }
}

int lslam_matcher_flush(lslam_matcher* m) {
  if (!m) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(m->ctx, hipSetDevice(m->ctx->device));
  return pipe_join(m);
}

int64_t lslam_matcher_pipelined_steps(const lslam_matcher* m) { return m ? (int64_t)m->pipe_steps : 0; }
int64_t lslam_matcher_step_kernel_launches(const lslam_matcher* m) { return m ? (int64_t)m->step_launches : 0; }
int64_t lslam_matcher_lone_kernel_launches(const lslam_matcher* m) { return m ? (int64_t)m->lone_launches : 0; }
// diagnostics: the hand-over words of k_match_lone (kLoneRing slots of 8 words) after a stream sync
int lslam_debug_lone_sync(lslam_matcher* m, unsigned* out128) {
  if (!m || !out128) return LSLAM_ERR_INVALID_ARGUMENT;
  if (!m->d_lone_sync) return LSLAM_ERR_NO_DATA;
  LSLAM_HIP(m->ctx, hipSetDevice(m->ctx->device));
  LSLAM_HIP(m->ctx, hipStreamSynchronize(m->ctx->stream));
  LSLAM_HIP(m->ctx, hipMemcpy(out128, m->d_lone_sync, kLoneRing * sizeof(LoneSync), hipMemcpyDeviceToHost));
  return LSLAM_OK;
}

#if defined(LSLAM_PHASE_STAMPS)
// diagnostic builds only: out[kernel][8 cycle sums | 8 visit counts] of this translation unit's stamp table (summed over the
// wave slots); reset != 0 clears it
int lslam_debug_matcher_stamps(lslam_context* ctx, unsigned long long* out, int reset) {
  if (!ctx || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(ctx, hipDeviceSynchronize());
  const size_t n = (size_t)lslam::kStampKernels * lslam::kStampSlots * 16;
  std::vector<unsigned long long> h(n);
  LSLAM_HIP(ctx, hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_sm_stamps_slots), n * sizeof(unsigned long long)));
  for (int k = 0; k < lslam::kStampKernels; k++)
    for (int i = 0; i < 16; i++) {
      unsigned long long sum = 0;
      for (int sl = 0; sl < lslam::kStampSlots; sl++) sum += h[((size_t)k * lslam::kStampSlots + sl) * 16 + i];
      out[k * 16 + i] = sum;
    }
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    LSLAM_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(g_sm_stamps_slots), h.data(), n * sizeof(unsigned long long)));
  }
  return LSLAM_OK;
}
#endif

int lslam_matcher_read_stats(lslam_matcher* m, uint64_t out[4]) {
  if (!m || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  if (!m->d_stats.p) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "LSLAM_OPT_COLLECT_STATS was never enabled");
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long host[4];
  LSLAM_HIP(ctx, hipMemcpyAsync(host, m->d_stats.p, sizeof host, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 4; i++) out[i] = host[i];
  return LSLAM_OK;
}

int lslam_matcher_read_beam_stats(lslam_matcher* m, uint64_t out[4]) {
  if (!m || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  out[0] = out[1] = out[2] = out[3] = 0;
  if (!m->d_stats.p || m->stats_scans <= 0)  // zeros here would read like a measurement
    return ctx->fail(LSLAM_ERR_NO_DATA, "no instrumented coarse pass has run (LSLAM_OPT_COLLECT_STATS + a batched match first)");
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> flags((size_t)m->stats_scans * m->g.n_beams);
  LSLAM_HIP(ctx, hipMemcpyAsync(flags.data(), (const uint32_t*)(m->d_stats.p + 8), flags.size() * sizeof(uint32_t), hipMemcpyDeviceToHost,
                                ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (uint32_t f : flags) {
    out[0] += f & 1u;
    out[1] += (f >> 1) & 1u;
  }
  unsigned long long fits[2] = {0, 0};  // LSLAM_OPT_LDS_STAGED: drains whose patches fit LDS / took the global path
  LSLAM_HIP(ctx, hipMemcpyAsync(fits, m->d_stats.p + 5, sizeof fits, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  out[2] = fits[0];
  out[3] = fits[1];
  return LSLAM_OK;
}

int lslam_matcher_set_base_scans(lslam_matcher* m, int B, const double* ranges, int stride,
                                 const double* sensor_poses, const double center[3]) {
  if (!m || !center || B < 0 || (B > 0 && (!ranges || !sensor_poses))) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_NOT_REENTRANT(m);
  lslam_context* ctx = m->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  const Geom& g = m->g;
  const int n = g.n_beams;
  if (B > 0 && n > 0) {
    int rc = upload_scans(m, B, ranges, stride, sensor_poses);
    if (rc) return rc;
    LSLAM_HIP(ctx, m->d_world.reserve((size_t)B * n));
    launch(ctx, "scan_prep_base", k_scan_prep<double>, dim3((n + 255) / 256, B), dim3(256), 0,
           (const double*)m->d_ranges64.p, n, (const double*)m->d_poses.p, g, (double2*)nullptr, m->d_world.p, PassCfg{}, (Lattice*)nullptr, (double2*)nullptr, 0, PoseArg{});
  }
  int rc = rebuild_grid_dev(m, m->d_world.p, 0, B, B > 0 ? B : 1, center);
  if (rc) return rc;
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  trim_workspaces(m);
  return LSLAM_OK;
}

int lslam_matcher_match_batch(lslam_matcher* m, int S, const double* ranges, int stride, const double* poses,
                              int do_penalize, int do_refine, lslam_match_result* out) {
  if (!m || S < 0 || (S > 0 && (!ranges || !poses || !out))) return LSLAM_ERR_INVALID_ARGUMENT;
  if (S == 0) return LSLAM_OK;
  LSLAM_NOT_REENTRANT(m);
  lslam_context* ctx = m->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  int rc = pipe_join(m);
  if (rc) return rc;
  const int n = m->g.n_beams, row = std::max(n, 1);
  // pipelined: the batch goes through as `pipe_depth` sub-batches taking turns on the internal streams -- the upload of
  // one runs under the kernels of the one before, and the sub-batches' kernels share the chip.  Same records: every
  // scan is matched on its own against the same grid, whatever sub-batch it travels in.
  int chunks = 1;
  // the same predicate as match_batch_dev_entry's: flags a grid rebuild left for the NEXT plain match (prep_done,
  // resp_prezeroed) or an armed ticket describe slot 0's buffers on the context stream -- such a call goes out as one plain step
  if (m->pipe_depth > 1 && !m->collect_stats && !m->lds_staged && !m->prep_done && !m->resp_prezeroed && !m->arm_next && n > 0)
    chunks = std::max(1, std::min(m->pipe_depth, S / kPipeMinChunk));
  if (chunks > 1) m->pipe_next = 0;  // sub-batch c always travels on slot c
  LSLAM_HIP(ctx, m->d_results.reserve(S));
  if (chunks == 1) {
    rc = upload_scans(m, S, ranges, stride, poses);
    if (rc) return rc;
    rc = match_batch_impl<double>(m, S, m->d_ranges64.p, row, m->d_poses.p, do_penalize, do_refine, m->d_results.p, nullptr, 0);
    if (rc) return rc;
  } else {
    if (stride < n) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "ranges_stride %d < num_beams %d", stride, n);
    LSLAM_HIP(ctx, m->d_ranges64.reserve((size_t)S * row));
    LSLAM_HIP(ctx, m->d_poses.reserve((size_t)S * 3));
    for (int c = 0; c < chunks; c++) {
      const int lo = (int)((long long)c * S / chunks), hi = (int)((long long)(c + 1) * S / chunks);
      LSLAM_HIP(ctx, hipMemcpy2DAsync(m->d_ranges64.p + (size_t)lo * row, (size_t)n * sizeof(double), ranges + (size_t)lo * stride,
                                      (size_t)stride * sizeof(double), (size_t)n * sizeof(double), hi - lo, hipMemcpyHostToDevice,
                                      ctx->stream));
      LSLAM_HIP(ctx, hipMemcpyAsync(m->d_poses.p + (size_t)lo * 3, poses + (size_t)lo * 3, (size_t)(hi - lo) * 3 * sizeof(double),
                                    hipMemcpyHostToDevice, ctx->stream));
      rc = pipe_step<double>(m, hi - lo, m->d_ranges64.p + (size_t)lo * row, row, m->d_poses.p + (size_t)lo * 3, do_penalize,
                             do_refine, m->d_results.p + lo);
      if (rc) break;
    }
    const int jrc = pipe_join(m);  // also after a failed sub-batch: nothing may stay in flight behind the caller's back
    if (rc) return rc;
    if (jrc) return jrc;
  }
  LSLAM_HIP(ctx, hipMemcpyAsync(out, m->d_results.p, (size_t)S * sizeof(lslam_match_result), hipMemcpyDeviceToHost,
                                ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  trim_workspaces(m);
  return LSLAM_OK;
}

int lslam_matcher_match_batch_dev_f32(lslam_matcher* m, int S, const float* ranges_dev, int stride,
                                      const double* poses_dev, int do_penalize, int do_refine,
                                      lslam_match_result* out_dev) {
  if (!m || S < 0 || (S > 0 && (!ranges_dev || !poses_dev || !out_dev))) return LSLAM_ERR_INVALID_ARGUMENT;
  if (stride < m->g.n_beams) return m->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "ranges_stride < num_beams");
  LSLAM_NOT_REENTRANT(m);
  return match_batch_dev_entry<float>(m, S, ranges_dev, stride, poses_dev, do_penalize, do_refine, out_dev);
}

int lslam_matcher_match_batch_dev_f64(lslam_matcher* m, int S, const double* ranges_dev, int stride,
                                      const double* poses_dev, int do_penalize, int do_refine,
                                      lslam_match_result* out_dev) {
  if (!m || S < 0 || (S > 0 && (!ranges_dev || !poses_dev || !out_dev))) return LSLAM_ERR_INVALID_ARGUMENT;
  if (stride < m->g.n_beams) return m->ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "ranges_stride < num_beams");
  LSLAM_NOT_REENTRANT(m);
  return match_batch_dev_entry<double>(m, S, ranges_dev, stride, poses_dev, do_penalize, do_refine, out_dev);
}

int lslam_matcher_match_scan(lslam_matcher* m, int n_base, const double* base_ranges, int stride,
                             const double* base_poses, const double* q_ranges, const double q_pose[3],
                             int do_penalize, int do_refine, lslam_match_result* out) {
  if (!m || !q_ranges || !q_pose || !out || n_base < 0 || (n_base > 0 && (!base_ranges || !base_poses)))
    return LSLAM_ERR_INVALID_ARGUMENT;
  const int n = m->g.n_beams;
  if (n == 0)  // a scan without readings returns before AddScans (Mapper.cpp:199-209)
    return lslam_matcher_match_batch(m, 1, q_ranges, 1, q_pose, do_penalize, do_refine, out);
  // ONE pass over the stream and one host synchronisation (round 3; it was AddScans + sync, then upload + match + download +
  // sync): the window goes up and is rasterised as before; the query's readings are staged in pinned host memory and ride
  // into HBM -- together with its pose, its scan_prep and the zeros for the first pass -- in the rebuild's first launch
  // (RebuildExtras, what the streaming front-end does per scan); the last kernel of the match writes the 112-byte record
  // straight into pinned host memory.
  LSLAM_NOT_REENTRANT(m);
  lslam_context* ctx = m->ctx;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  if (!m->h_query) {
    if (hipHostMalloc((void**)&m->h_query, sizeof(double) * (size_t)n, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void**)&m->h_result, sizeof(lslam_match_result), hipHostMallocDefault) != hipSuccess) {
      (void)hipGetLastError();
      return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the pinned staging of MatchScan");
    }
  }
  LSLAM_HIP(ctx, m->d_query.reserve((size_t)n));
  LSLAM_HIP(ctx, m->d_qpose.reserve(4));
  if (n_base > 0) {
    int rc = upload_scans(m, n_base, base_ranges, stride, base_poses);
    if (rc) return rc;
    LSLAM_HIP(ctx, m->d_world.reserve((size_t)n_base * n));
    launch(ctx, "scan_prep_base", k_scan_prep<double>, dim3((n + 255) / 256, n_base), dim3(256), 0,
           (const double*)m->d_ranges64.p, n, (const double*)m->d_poses.p, m->g, (double2*)nullptr, m->d_world.p, PassCfg{}, (Lattice*)nullptr, (double2*)nullptr, 0, PoseArg{});
  }
  memcpy(m->h_query, q_ranges, sizeof(double) * (size_t)n);
  RebuildExtras x{};
  for (int i = 0; i < 3; i++) x.pose[i] = q_pose[i];
  x.pose_dst = m->d_qpose.p;
  x.zero = m->d_resp.p;
  x.zero_words = (int)std::min<size_t>(m->d_resp.cap, (size_t)1 << 16);
  x.ranges_src = m->h_query;
  x.ranges_dst = m->d_query.p;
  x.n_ranges = n;
  x.prep_ranges = m->h_query;  // the prep blocks read the staged copy (the device row is being written in the same launch)
  int rc = rebuild_grid_dev(m, m->d_world.p, 0, n_base, n_base > 0 ? n_base : 1, q_pose, &x);
  if (rc) return rc;
  arm_done_ticket(m);
  rc = match_batch_impl<double>(m, 1, m->d_query.p, n, m->d_qpose.p, do_penalize, do_refine, m->h_result, nullptr, 0);
  if (rc) return rc;
  rc = wait_record(m);
  if (rc) return rc;
  *out = *m->h_result;
  return LSLAM_OK;
}

int lslam_matcher_debug_lookup_table(lslam_matcher* m, const double* ranges, const double pose[3],
                                     double angle_center, double angle_offset, double angle_res,
                                     int32_t* out, int* n_angles_out) {
  if (!m || !ranges || !pose || !n_angles_out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  const Geom g = m->g;
  int na = n_angles_of(angle_offset, angle_res);
  *n_angles_out = na;
  if (!out || g.n_beams == 0) return LSLAM_OK;
  int rc = pipe_join(m);  // slot 0's d_local is a step workspace
  if (rc) return rc;
  rc = upload_scans(m, 1, ranges, g.n_beams, pose);
  if (rc) return rc;
  LSLAM_HIP(ctx, m->d_local.reserve((size_t)g.n_beams));
  LSLAM_HIP(ctx, m->d_dbg.reserve((size_t)na * g.n_beams));
  launch(ctx, "scan_prep", k_scan_prep<double>, dim3((g.n_beams + 255) / 256, 1), dim3(256), 0,
         (const double*)m->d_ranges64.p, g.n_beams, (const double*)m->d_poses.p, g, m->d_local.p, (double2*)nullptr, PassCfg{}, (Lattice*)nullptr, (double2*)nullptr, 0, PoseArg{});
  launch(ctx, "debug_table", k_debug_table, dim3((g.n_beams + 255) / 256, na), dim3(256), 0, g,
         (const double2*)m->d_local.p, angle_center, angle_offset, angle_res, na, m->d_dbg.p);
  LSLAM_HIP(ctx, hipMemcpyAsync(out, m->d_dbg.p, (size_t)na * g.n_beams * sizeof(int32_t), hipMemcpyDeviceToHost,
                                ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_matcher_debug_coarse_sums(lslam_matcher* m, const double* ranges, const double pose[3], int32_t* out,
                                    int* nx, int* ny, int* na, int force_generic) {
  if (!m || !ranges || !pose) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  const Geom g = m->g;
  const double res = 1.0 / g.scale;
  double off = 0.5 * ((double)g.probs_side - 1) * res;
  int lx = lattice_count(off, 2 * res);
  int la = n_angles_of(m->cfg.coarse_search_angle_offset, m->cfg.coarse_angle_resolution);
  if (nx) *nx = lx;
  if (ny) *ny = lx;
  if (na) *na = la;
  if (!out || g.n_beams == 0) return LSLAM_OK;
  int rc = upload_scans(m, 1, ranges, g.n_beams, pose);
  if (rc) return rc;
  LSLAM_HIP(ctx, m->d_results.reserve(1));
  LSLAM_HIP(ctx, m->d_dbg.reserve((size_t)lx * lx * la));
  rc = match_batch_impl<double>(m, 1, m->d_ranges64.p, g.n_beams, m->d_poses.p, 1, 0, m->d_results.p, m->d_dbg.p,
                                force_generic);
  if (rc) return rc;
  std::vector<int32_t> tmp((size_t)lx * lx * la);
  LSLAM_HIP(ctx, hipMemcpyAsync(tmp.data(), m->d_dbg.p, tmp.size() * sizeof(int32_t), hipMemcpyDeviceToHost,
                                ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int ncand = lx * lx;  // device layout is angle-major; the reference order is y, x, angle
  for (int a = 0; a < la; a++)
    for (int c = 0; c < ncand; c++) out[(size_t)c * la + a] = tmp[(size_t)a * ncand + c];
  return LSLAM_OK;
}

int lslam_matcher_debug_coarse_sums_batch(lslam_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                                          const double* poses, int32_t* out) {
  if (!m || !ranges || !poses || !out || n_scans <= 0) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  LSLAM_NOT_REENTRANT(m);
  {  // pipelined steps in flight read the upload buffers and slot 0's workspaces this call is about to reuse
    const int jrc = pipe_join(m);
    if (jrc) return jrc;
  }
  const Geom g = m->g;
  if (g.n_beams == 0) return LSLAM_OK;
  const double res = 1.0 / g.scale;
  const double off = 0.5 * ((double)g.probs_side - 1) * res;
  const int lx = lattice_count(off, 2 * res);
  const int la = n_angles_of(m->cfg.coarse_search_angle_offset, m->cfg.coarse_angle_resolution);
  int rc = upload_scans(m, n_scans, ranges, ranges_stride, poses);
  if (rc) return rc;
  LSLAM_HIP(ctx, m->d_results.reserve(n_scans));
  // the batch goes through the same launches a match of n_scans scans takes (coarse pass only): whichever variant of the
  // response kernel that batch size selects is the one whose numerators come back
  const size_t cap = (size_t)n_scans * (size_t)std::max(lx * lx, 16) * (size_t)kMaxAngles;  // >= n_scans * resp_stride
  LSLAM_HIP(ctx, m->d_dbg.reserve(cap));
  m->dbg_all = true;
  // (with the step kernel selected the batch goes through IT -- which always refines; the records are not looked at)
  rc = match_batch_impl<double>(m, n_scans, m->d_ranges64.p, g.n_beams, m->d_poses.p, 1, m->step_waves > 0 ? 1 : 0, m->d_results.p,
                                m->d_dbg.p, 0);
  m->dbg_all = false;
  if (rc) return rc;
  const size_t stride = m->dbg_resp_stride;
  std::vector<int32_t> tmp((size_t)n_scans * stride);
  LSLAM_HIP(ctx, hipMemcpyAsync(tmp.data(), m->d_dbg.p, tmp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int ncand = lx * lx;  // device layout is angle-major per scan; the reference order is y, x, angle
  for (int s = 0; s < n_scans; s++)
    for (int a = 0; a < la; a++)
      for (int c = 0; c < ncand; c++)
        out[((size_t)s * ncand + c) * la + a] = tmp[(size_t)s * stride + (size_t)a * ncand + c];
  return LSLAM_OK;
}

int lslam_matcher_debug_fine_sums_batch(lslam_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                                        const double* poses, double* centers_out, int32_t* out, int* nx, int* ny, int* na) {
  if (!m || !ranges || !poses || n_scans <= 0) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  LSLAM_NOT_REENTRANT(m);
  {  // pipelined steps in flight read the upload buffers and slot 0's workspaces this call is about to reuse
    const int jrc = pipe_join(m);
    if (jrc) return jrc;
  }
  const Geom g = m->g;
  const double res = 1.0 / g.scale;
  const double coarse_res = 2 * res;
  const int fx = lattice_count(coarse_res * 0.5, res);  // Mapper.cpp:276-281
  const int fa = n_angles_of(0.5 * m->cfg.coarse_angle_resolution, m->cfg.fine_search_angle_offset);
  if (nx) *nx = fx;
  if (ny) *ny = fx;
  if (na) *na = fa;
  if (!out || !centers_out || g.n_beams == 0) return LSLAM_OK;
  int rc = upload_scans(m, n_scans, ranges, ranges_stride, poses);
  if (rc) return rc;
  LSLAM_HIP(ctx, m->d_results.reserve(n_scans));
  const double off = 0.5 * ((double)g.probs_side - 1) * res;
  const int lx = lattice_count(off, coarse_res);
  LSLAM_HIP(ctx, m->d_dbg.reserve((size_t)n_scans * (size_t)std::max(lx * lx, 16) * (size_t)kMaxAngles));  // >= n * resp_stride
  m->dbg_fine = m->d_dbg.p;
  rc = match_batch_impl<double>(m, n_scans, m->d_ranges64.p, g.n_beams, m->d_poses.p, 1, 1, m->d_results.p, nullptr, 0);
  m->dbg_fine = nullptr;
  if (rc) return rc;
  const size_t stride = m->dbg_resp_stride;
  std::vector<int32_t> tmp((size_t)n_scans * stride);
  std::vector<CoarseOut> co((size_t)n_scans);
  LSLAM_HIP(ctx, hipMemcpyAsync(tmp.data(), m->d_dbg.p, tmp.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipMemcpyAsync(co.data(), m->d_coarse.p, co.size() * sizeof(CoarseOut), hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int ncand = fx * fx;  // device layout is angle-major per scan; the reference order is y, x, angle
  for (int s = 0; s < n_scans; s++) {
    for (int i = 0; i < 3; i++) centers_out[3 * (size_t)s + i] = co[s].status == 0 ? co[s].mean[i] : __builtin_nan("");
    for (int a = 0; a < fa; a++)
      for (int c = 0; c < ncand; c++)
        out[((size_t)s * ncand + c) * fa + a] = tmp[(size_t)s * stride + (size_t)a * ncand + c];
  }
  return LSLAM_OK;
}

int lslam_matcher_debug_valid_mask(lslam_matcher* m, const double* ranges, const double pose[3],
                                   const double viewpoint[2], uint8_t* out) {
  if (!m || !ranges || !pose || !viewpoint || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  const Geom g = m->g;
  if (g.n_beams == 0) return LSLAM_OK;
  int rc = upload_scans(m, 1, ranges, g.n_beams, pose);
  if (rc) return rc;
  LSLAM_HIP(ctx, m->d_world.reserve((size_t)g.n_beams));
  LSLAM_HIP(ctx, m->d_valid.reserve((size_t)g.n_beams));
  launch(ctx, "scan_prep_base", k_scan_prep<double>, dim3((g.n_beams + 255) / 256, 1), dim3(256), 0,
         (const double*)m->d_ranges64.p, g.n_beams, (const double*)m->d_poses.p, g, (double2*)nullptr, m->d_world.p, PassCfg{}, (Lattice*)nullptr, (double2*)nullptr, 0, PoseArg{});
  {
    const size_t lds = (size_t)g.n_beams * (sizeof(double2) + 14) + 16;
    const int use_lds = lds <= 60 * 1024;
    if (!use_lds) LSLAM_HIP(ctx, m->d_fv_scratch.reserve((size_t)2 * g.n_beams));
    launch(ctx, "find_valid", k_find_valid, dim3(1), dim3(g.n_beams > 512 ? 1024 : 256), use_lds ? lds : 0, g.n_beams,
           (const double2*)m->d_world.p, 0, 1, viewpoint[0], viewpoint[1], m->d_valid.p, use_lds, m->d_fv_scratch.p, g,
           (uint8_t*)nullptr, (const int*)nullptr, 0, 1, RebuildExtras{});
  }
  LSLAM_HIP(ctx, hipMemcpyAsync(out, m->d_valid.p, (size_t)g.n_beams, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

}  // extern "C"

#include "frontend_impl.hpp"
#include "scan_cache_impl.hpp"
