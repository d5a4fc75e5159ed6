/* TEST INFRASTRUCTURE ONLY -- see hector_oracle.h. */
#include "hector_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float v; int idx; } cell_t; /* LogOddsCell (H/map/GridMapLogOdds.h:85-86) */

struct hor_map {
  int sx, sy;
  float cell_length, off_x, off_y;
  float scale_to_map;            /* 1/cellLength (H/map/GridMapBase.h:276) */
  float t_x, t_y;                /* translation of mapTworld = Scale*Translate (:278) */
  float lo_free, lo_occ;         /* H/map/GridMapLogOdds.h:157-158 */
  int curr_update, mark_free, mark_occ; /* H/map/OccGridMapBase.h:334-337 */
  cell_t* cells;
  int64_t visits;
};

static float prob_to_logodds(float prob) { /* H/map/GridMapLogOdds.h:151-155 */
  float odds = prob / (1.0f - prob);
  return (float)log(odds);                  /* log(float) -> double log, stored to float */
}

hor_map* hor_create(int size_x, int size_y, float cell_length, float offset_x, float offset_y) {
  hor_map* m = (hor_map*)calloc(1, sizeof *m);
  m->sx = size_x; m->sy = size_y;
  m->cell_length = cell_length; m->off_x = offset_x; m->off_y = offset_y;
  m->scale_to_map = 1.0f / cell_length;
  m->t_x = m->scale_to_map * offset_x;
  m->t_y = m->scale_to_map * offset_y;
  m->lo_free = prob_to_logodds(0.4f); /* H/map/GridMapLogOdds.h:100-101 */
  m->lo_occ = prob_to_logodds(0.6f);
  m->cells = (cell_t*)malloc(sizeof(cell_t) * (size_t)size_x * size_y);
  hor_reset(m);
  m->curr_update = 0; /* OccGridMapBase ctor: currUpdateIndex(0) */
  return m;
}
void hor_destroy(hor_map* m) { if (m) { free(m->cells); free(m); } }
void hor_reset(hor_map* m) {
  size_t n = (size_t)m->sx * m->sy;
  for (size_t i = 0; i < n; i++) { m->cells[i].v = 0.0f; m->cells[i].idx = -1; } /* resetGridCell :76-80 */
}
void hor_set_update_free_factor(hor_map* m, float p) { m->lo_free = prob_to_logodds(p); }
void hor_set_update_occupied_factor(hor_map* m, float p) { m->lo_occ = prob_to_logodds(p); }
float hor_scale_to_map(const hor_map* m) { return m->scale_to_map; }
float hor_level_factor(int level) { return (float)(1.0 / pow(2.0, (double)level)); }

/* bresenhamCellFree / bresenhamCellOcc (H/map/OccGridMapBase.h:302-330) */
static void cell_free(hor_map* m, unsigned off) {
  cell_t* c = &m->cells[off];
  m->visits++;
  if (c->idx < m->mark_free) { c->v += m->lo_free; c->idx = m->mark_free; }
}
static void cell_occ(hor_map* m, unsigned off) {
  cell_t* c = &m->cells[off];
  m->visits++;
  if (c->idx < m->mark_occ) {
    if (c->idx == m->mark_free) c->v -= m->lo_free;  /* updateUnsetFree */
    if (c->v < 50.0f) c->v += m->lo_occ;              /* updateSetOccupied (:108-114) */
    c->idx = m->mark_occ;
  }
}
/* bresenham2D (H/map/OccGridMapBase.h:270-299) */
static void bresenham2d(hor_map* m, unsigned abs_da, unsigned abs_db, int error_b, int offset_a,
                        int offset_b, unsigned offset) {
  cell_free(m, offset);
  unsigned end = abs_da - 1;
  for (unsigned i = 0; i < end; ++i) {
    offset += offset_a;
    error_b += abs_db;
    if ((unsigned)error_b >= abs_da) { offset += offset_b; error_b -= abs_da; }
    cell_free(m, offset);
  }
}
static int isign(int x) { return x > 0 ? 1 : -1; } /* util::sign (H/util/UtilFunctions.h:55-58) */
/* updateLineBresenhami (H/map/OccGridMapBase.h:220-267) */
static void update_line(hor_map* m, int x0, int y0, int x1, int y1) {
  if (x0 < 0 || x0 >= m->sx || y0 < 0 || y0 >= m->sy) return;
  if (x1 < 0 || x1 >= m->sx || y1 < 0 || y1 >= m->sy) return;
  int dx = x1 - x0, dy = y1 - y0;
  unsigned abs_dx = (unsigned)abs(dx), abs_dy = (unsigned)abs(dy);
  int offset_dx = isign(dx), offset_dy = isign(dy) * m->sx;
  unsigned start = (unsigned)(y0 * m->sx + x0);
  if (abs_dx >= abs_dy) bresenham2d(m, abs_dx, abs_dy, (int)(abs_dx / 2), offset_dx, offset_dy, start);
  else bresenham2d(m, abs_dy, abs_dx, (int)(abs_dy / 2), offset_dy, offset_dx, start);
  cell_occ(m, (unsigned)(y1 * m->sx + x1));
}

void hor_update_by_scan(hor_map* m, const float* p, int n, const float origo[2], const float pose[3]) {
  m->mark_free = m->curr_update + 1; /* :120-121 */
  m->mark_occ = m->curr_update + 2;
  m->visits = 0;
  /* getMapCoordsPose: mapTworld * pose.xy (H/map/GridMapBase.h:238-242); mapTworld =
   * Scale(s)*Translate(off) -> linear diag(s,s), translation s*off (:278) */
  float s = m->scale_to_map;
  float mx = (s * pose[0] + 0.0f * pose[1]) + m->t_x;
  float my = (0.0f * pose[0] + s * pose[1]) + m->t_y;
  float ang = pose[2];
  /* Translation2f(mx,my) * Rotation2Df(ang) (:127-129) */
  float c = cosf(ang), sn = sinf(ang);
  float bxf = (c * origo[0] + (-sn) * origo[1]) + mx; /* :132 */
  float byf = (sn * origo[0] + c * origo[1]) + my;
  int bx = (int)(bxf + 0.5f), by = (int)(byf + 0.5f);  /* :135 */
  for (int i = 0; i < n; i++) {
    float ex = (c * p[2 * i] + (-sn) * p[2 * i + 1]) + mx; /* :145 */
    float ey = (sn * p[2 * i] + c * p[2 * i + 1]) + my;
    ex += 0.5f; ey += 0.5f;                                 /* :149 */
    int ix = (int)ex, iy = (int)ey;                         /* :152 */
    if (bx != ix || by != iy) update_line(m, bx, by, ix, iy); /* :155-158 */
  }
  m->curr_update += 3; /* :167 */
}

void hor_update_just_once(hor_map* m, const float* p, int n, const float origo[2], float begin_x,
                          float begin_y, double metres_per_cell) {
  m->mark_free = m->curr_update + 1;
  m->mark_occ = m->curr_update + 2;
  m->visits = 0;
  /* mapPose(800,800,0) (:182): Translation * Rotation2Df(0) */
  float c = cosf(0.0f), sn = sinf(0.0f);
  float bxf = (c * origo[0] + (-sn) * origo[1]) + begin_x;
  float byf = (sn * origo[0] + c * origo[1]) + begin_y;
  int bx = (int)(bxf + 0.5f), by = (int)(byf + 0.5f);
  for (int i = 0; i < n; i++) {
    int ix = bx + (int)round(p[2 * i] / metres_per_cell);     /* :202-203, float / double */
    int iy = by + (int)round(p[2 * i + 1] / metres_per_cell);
    if (bx != ix || by != iy) update_line(m, bx, by, ix, iy);
  }
  m->curr_update += 3;
}

void hor_read_logodds(const hor_map* m, float* out) {
  size_t n = (size_t)m->sx * m->sy;
  for (size_t i = 0; i < n; i++) out[i] = m->cells[i].v;
}
void hor_read_update_index(const hor_map* m, int32_t* out) {
  size_t n = (size_t)m->sx * m->sy;
  for (size_t i = 0; i < n; i++) out[i] = m->cells[i].idx;
}
void hor_read_occupancy_i8(const hor_map* m, int8_t* out) {
  size_t n = (size_t)m->sx * m->sy;
  for (size_t i = 0; i < n; i++) {
    float v = m->cells[i].v;
    out[i] = v < 0.0f ? 0 : (v > 0.0f ? 100 : -1);
  }
}
int64_t hor_last_cell_visits(const hor_map* m) { return m->visits; }

/* ------------------------------------------------------------------------------------------- */
/* Gauss-Newton scan-to-map matcher (next-row #2) */
static float grid_probability(const hor_map* m, int index) { /* H/map/GridMapLogOdds.h:123-127 */
  float odds = (float)exp((double)m->cells[index].v);
  return odds / (odds + 1.0f);
}

/* interpMapValueWithDerivatives (H/map/OccGridMapUtil.h:139-228); limits = dims - 2 (MapDimensionProperties.h:66-70) */
static void interp_with_derivs(const hor_map* m, float cx, float cy, float out[3]) {
  float lim_x = (float)m->sx - 2.0f, lim_y = (float)m->sy - 2.0f;
  if (cx < 0.0f || cx > lim_x || cy < 0.0f || cy > lim_y) { out[0] = out[1] = out[2] = 0.0f; return; }
  int ix = (int)cx, iy = (int)cy;
  float fx = cx - (float)ix, fy = cy - (float)iy;
  int index = iy * m->sx + ix;
  float i0 = grid_probability(m, index);
  float i1 = grid_probability(m, index + 1);
  float i2 = grid_probability(m, index + m->sx);
  float i3 = grid_probability(m, index + m->sx + 1);
  float dx1 = i0 - i1, dx2 = i2 - i3, dy1 = i0 - i2, dy2 = i1 - i3;
  float xi = 1.0f - fx, yi = 1.0f - fy;
  out[0] = ((i0 * xi + i1 * fx) * (yi)) + ((i2 * xi + i3 * fx) * (fy));
  out[1] = -((dx1 * yi) + (dx2 * fy));
  out[2] = -((dy1 * xi) + (dy2 * fx));
}

static float cof3(const float* m, int i, int j) { /* Eigen cofactor_3x3 */
  int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
  return m[3 * i1 + j1] * m[3 * i2 + j2] - m[3 * i1 + j2] * m[3 * i2 + j1];
}
static void inverse3(const float* m, float* r) { /* Eigen compute_inverse<Matrix3f> */
  float c0 = cof3(m, 0, 0), c1 = cof3(m, 1, 0), c2 = cof3(m, 2, 0);
  float det = c0 * m[0] + (c1 * m[3] + c2 * m[6]);
  float invdet = 1.0f / det;
  r[0] = c0 * invdet; r[1] = c1 * invdet; r[2] = c2 * invdet;
  r[3] = cof3(m, 0, 1) * invdet; r[4] = cof3(m, 1, 1) * invdet; r[5] = cof3(m, 2, 1) * invdet;
  r[6] = cof3(m, 0, 2) * invdet; r[7] = cof3(m, 1, 2) * invdet; r[8] = cof3(m, 2, 2) * invdet;
}

static float normalize_angle_f(float angle) { /* H/util/UtilFunctions.h:36-48 (double arithmetic, M_PI) */
  const double two_pi = 2.0f * 3.14159265358979323846;
  float a = (float)fmod(fmod((double)angle, two_pi) + two_pi, two_pi);
  if ((double)a > 3.14159265358979323846) a = (float)((double)a - two_pi);
  return a;
}

/* getCompleteHessianDerivs (H/map/OccGridMapUtil.h:77-132) */
static void hessian_derivs(const hor_map* m, const float pose[3], const float* pts, float factor, int n, float H[9], float dTr[3]) {
  float c = cosf(pose[2]), s = sinf(pose[2]); /* Rotation2Df of getTransformForState (:437-440) */
  float sinRot = (float)sin((double)pose[2]), cosRot = (float)cos((double)pose[2]); /* :87-88 */
  memset(H, 0, 9 * sizeof(float));
  dTr[0] = dTr[1] = dTr[2] = 0.0f;
  for (int i = 0; i < n; i++) {
    float px = pts[2 * i] * factor, py = pts[2 * i + 1] * factor;
    float tx = (c * px + (-s) * py) + pose[0];
    float ty = (s * px + c * py) + pose[1];
    float t[3];
    interp_with_derivs(m, tx, ty, t);
    float funVal = 1.0f - t[0];
    dTr[0] += t[1] * funVal;
    dTr[1] += t[2] * funVal;
    float rotDeriv = ((-sinRot * px - cosRot * py) * t[1] + (cosRot * px - sinRot * py) * t[2]);
    dTr[2] += rotDeriv * funVal;
    H[0] += t[1] * t[1];
    H[4] += t[2] * t[2];
    H[8] += rotDeriv * rotDeriv;
    H[1] += t[1] * t[2];
    H[2] += t[1] * rotDeriv;
    H[5] += t[2] * rotDeriv;
  }
  H[3] = H[1]; H[6] = H[2]; H[7] = H[5];
}

/* ScanMatcher::matchData on ONE level (H/matcher/ScanMatcher.h:60-99): pts are scaled by `factor` first
 * (DataPointContainer::setFrom); 1 + max_iter Gauss-Newton steps; n == 0 returns begin_world and leaves out_cov. */
void hor_match_level(const hor_map* m, const float* pts, int n, float factor, const float begin_world[3], int max_iter,
                     float out_pose[3], float out_cov[9]) {
  if (n == 0) { /* matchData returns beginEstimateWorld (ScanMatcher.h:96) */
    out_pose[0] = begin_world[0]; out_pose[1] = begin_world[1]; out_pose[2] = begin_world[2];
    return;
  }
  /* getMapCoordsPose */
  float sc = m->scale_to_map;
  float est[3] = {(sc * begin_world[0] + 0.0f * begin_world[1]) + m->t_x,
                  (0.0f * begin_world[0] + sc * begin_world[1]) + m->t_y, begin_world[2]};
  float H[9], dTr[3];
  for (int it = 0; it < 1 + max_iter; it++) { /* ScanMatcher.h:73-80 */
    hessian_derivs(m, est, pts, factor, n, H, dTr);
    if (H[0] != 0.0f && H[4] != 0.0f) {
      float Hi[9], sd[3];
      inverse3(H, Hi);
      /* H.inverse() * dTr: coefficient-based product, inner sum by Eigen's redux_novec_unroller: a0 + (a1 + a2) */
      for (int r = 0; r < 3; r++) sd[r] = Hi[3 * r] * dTr[0] + (Hi[3 * r + 1] * dTr[1] + Hi[3 * r + 2] * dTr[2]);
      if (sd[2] > 0.2f) sd[2] = 0.2f; else if (sd[2] < -0.2f) sd[2] = -0.2f;
      est[0] += sd[0]; est[1] += sd[1]; est[2] += sd[2];
    }
  }
  est[2] = normalize_angle_f(est[2]);
  memcpy(out_cov, H, sizeof H);
  /* getWorldCoordsPose: worldTmap = mapTworld.inverse() (H/map/GridMapBase.h:285) */
  float invdet = 1.0f / (sc * sc - 0.0f * 0.0f);
  float l00 = sc * invdet, l01 = -0.0f * invdet, l10 = -0.0f * invdet, l11 = sc * invdet;
  float wt0 = -(l00 * m->t_x + l01 * m->t_y), wt1 = -(l10 * m->t_x + l11 * m->t_y);
  out_pose[0] = (l00 * est[0] + l01 * est[1]) + wt0;
  out_pose[1] = (l10 * est[0] + l11 * est[1]) + wt1;
  out_pose[2] = est[2];
}

void hor_hessian_derivs(const hor_map* m, const float* pts, int n, const float pose_map[3], float H[9], float dTr[3]) {
  hessian_derivs(m, pose_map, pts, 1.0f, n, H, dTr);
}

void hor_match_data(hor_map* const* levels, int n_levels, const float* pts, int n, const float begin_world[3],
                    float out_pose[3], float out_cov[9]) {
  float tmp[3] = {begin_world[0], begin_world[1], begin_world[2]};
  for (int lv = n_levels - 1; lv >= 0; --lv) { /* MapRepMultiMap.h:151-165 */
    float next[3];
    hor_match_level(levels[lv], pts, n, lv == 0 ? 1.0f : hor_level_factor(lv), tmp, lv == 0 ? 5 : 3, next, out_cov);
    tmp[0] = next[0]; tmp[1] = next[1]; tmp[2] = next[2];
  }
  out_pose[0] = tmp[0]; out_pose[1] = tmp[1]; out_pose[2] = tmp[2];
}
