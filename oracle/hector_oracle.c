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
