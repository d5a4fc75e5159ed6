/* TEST INFRASTRUCTURE ONLY -- see karto_oracle.h.  Plain-C restatement of the reference's Karto
 * correlative scan matcher, written from the reference's behaviour (not copied): flat arrays
 * instead of the Grid/LookupArray/Pose2 class zoo, every step citing the reference lines it
 * follows.  Compiled WITHOUT FMA contraction (oracle/Makefile) like the reference's x86-64 build,
 * because lookup-table rounding depends on separately rounded products.
 *
 * Citations: Mapper.cpp = lesson6/lib/open_karto/src/Mapper.cpp; Mapper.h, Karto.h, Math.h =
 * lesson6/lib/open_karto/include/open_karto/.
 */
#include "karto_oracle.h"

#include <limits.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define KT_PI 3.14159265358979323846  /* Math.h:30 */
#define KT_2PI 6.28318530717958647692 /* Math.h:31 */
#define KT_TOLERANCE 1e-06            /* Math.h:41 */
#define INVALID_SCAN INT_MAX          /* Math.h:47 */
#define MAX_VARIANCE 500.0            /* Mapper.cpp:36 */
#define DISTANCE_PENALTY_GAIN 0.2     /* Mapper.cpp:37 */
#define ANGLE_PENALTY_GAIN 0.2        /* Mapper.cpp:38 */
#define GRID_OCCUPIED 100             /* Karto.h:4196 */

struct kor_matcher {
  kor_config cfg;
  kor_laser laser;
  int n_beams;
  /* correlation grid (Mapper.h:1016-1027, Karto.h:4438-4471) */
  int width, height, stride, border, roi_w, roi_h, data_size;
  int kernel_size;
  uint8_t* kernel;
  uint8_t* grid;
  double scale;     /* 1/resolution (Mapper.h:1020) */
  double off_x, off_y;
  /* search-space probabilities Grid<double> (Mapper.cpp:163-164) */
  int probs_side, probs_stride;
  double* probs;
  double probs_off_x, probs_off_y;
  /* scratch lookup table */
  int32_t* table;
  int table_cap;
  double* scratch_pts;
};

/* math::Round, half away from zero (Math.h:87-90) */
static double kround(double v) { return v >= 0.0 ? floor(v + 0.5) : ceil(v - 0.5); }
static double ksquare(double v) { return v * v; }
/* math::DoubleEqual (Math.h:135-139) */
static int double_equal(double a, double b) {
  double d = a - b;
  return d < 0.0 ? d >= -KT_TOLERANCE : d <= KT_TOLERANCE;
}
/* math::NormalizeAngle (Math.h:182-211) */
static double normalize_angle(double angle) {
  while (angle < -KT_PI) {
    if (angle < -KT_2PI)
      angle += (uint32_t)(angle / -KT_2PI) * KT_2PI;
    else
      angle += KT_2PI;
  }
  while (angle > KT_PI) {
    if (angle > KT_2PI)
      angle -= (uint32_t)(angle / KT_2PI) * KT_2PI;
    else
      angle -= KT_2PI;
  }
  return angle;
}
/* math::NormalizeAngleDifference (Math.h:221-234) */
static double normalize_angle_difference(double minuend, double subtrahend) {
  while (minuend - subtrahend < -KT_PI) minuend += KT_2PI;
  while (minuend - subtrahend > KT_PI) minuend -= KT_2PI;
  return minuend;
}
static int is_up_to(int v, int max) { return v >= 0 && v < max; } /* Math.h:147-150 */

/* Matrix3::FromAxisAngle(0,0,1,radians) rows 0,1 (Karto.h:2392-2417), kept literal so that the
 * zero terms round exactly like the reference's. */
typedef struct { double m[2][3]; } rot2;
static rot2 rot_z(double radians) {
  rot2 r;
  const double x = 0.0, y = 0.0, z = 1.0;
  double c = cos(radians), s = sin(radians), omc = 1.0 - c;
  double xyM = x * y * omc, xzM = x * z * omc, yzM = y * z * omc;
  double xS = x * s, yS = y * s, zS = z * s;
  r.m[0][0] = x * x * omc + c;
  r.m[0][1] = xyM - zS;
  r.m[0][2] = xzM + yS;
  r.m[1][0] = xyM + zS;
  r.m[1][1] = y * y * omc + c;
  r.m[1][2] = yzM - xS;
  return r;
}
/* Matrix3 * Pose2, x and y rows (Karto.h:2574-2583) */
static void rot_apply(const rot2* r, double x, double y, double h, double* ox, double* oy) {
  *ox = r->m[0][0] * x + r->m[0][1] * y + r->m[0][2] * h;
  *oy = r->m[1][0] * x + r->m[1][1] * y + r->m[1][2] * h;
}

/* Transform (Karto.h:2853-2943) */
typedef struct { rot2 rot, inv; double tx, ty, th; } xform;
static xform xform_make(const double p1[3], const double p2[3]) {
  xform t;
  if (p1[0] == p2[0] && p1[1] == p2[1] && p1[2] == p2[2]) { /* Karto.h:2911-2917 */
    memset(&t, 0, sizeof t);
    t.rot.m[0][0] = t.rot.m[1][1] = 1.0;
    t.inv = t.rot;
    return t;
  }
  t.rot = rot_z(p2[2] - p1[2]); /* :2920 */
  t.inv = rot_z(p1[2] - p2[2]); /* :2921 */
  if (p1[0] != 0.0 || p1[1] != 0.0) { /* :2925-2928  rPose2 - m_Rotation * rPose1 */
    double rx, ry;
    rot_apply(&t.rot, p1[0], p1[1], p1[2], &rx, &ry);
    t.tx = p2[0] - rx;
    t.ty = p2[1] - ry;
  } else {
    t.tx = p2[0];
    t.ty = p2[1];
  }
  t.th = p2[2] - p1[2]; /* :2934 */
  return t;
}
/* Transform::TransformPose (Karto.h:2881-2887) */
static void xform_apply(const xform* t, const double src[3], double out[3]) {
  double rx, ry;
  rot_apply(&t->rot, src[0], src[1], src[2], &rx, &ry);
  out[0] = t->tx + rx;
  out[1] = t->ty + ry;
  out[2] = normalize_angle(src[2] + t->th);
}
/* Transform::InverseTransformPose, position part (Karto.h:2894-2901) */
static void xform_inverse_point(const xform* t, double px, double py, double* ox, double* oy) {
  double h = normalize_angle(0.0 - t->th); /* Pose2 operator- normalises the heading (Karto.h:2138-2141) */
  rot_apply(&t->inv, px - t->tx, py - t->ty, h, ox, oy);
}

/* CorrelationGrid::GetHalfKernelSize (Mapper.h:1096-1101) */
static int half_kernel_size(double smear, double res) { return (int)kround(2.0 * smear / res); }

kor_matcher* kor_create(const kor_config* cfg, const kor_laser* laser) {
  /* ScanMatcher::Create parameter checks (Mapper.cpp:130-145) */
  if (cfg->resolution <= 0 || cfg->search_size <= 0 || cfg->smear_deviation < 0 ||
      cfg->range_threshold <= 0)
    return NULL;
  /* CalculateKernel range check (Mapper.h:1041-1053): the reference throws */
  if (!(cfg->smear_deviation >= 0.5 * cfg->resolution && cfg->smear_deviation <= 10 * cfg->resolution))
    return NULL;
  kor_matcher* m = (kor_matcher*)calloc(1, sizeof *m);
  m->cfg = *cfg;
  m->laser = *laser;
  /* LaserRangeFinder::Update (Karto.h:4158-4160) */
  m->n_beams = (int)(uint32_t)kround((laser->maximum_angle - laser->minimum_angle) / laser->angular_resolution);
  uint32_t side = (uint32_t)(kround(cfg->search_size / cfg->resolution) + 1); /* Mapper.cpp:150 */
  uint32_t margin = (uint32_t)ceil(cfg->range_threshold / cfg->resolution);   /* :154 */
  int grid_size = (int)(side + 2 * margin);                                    /* :156 */
  int border = half_kernel_size(cfg->smear_deviation, cfg->resolution) + 1;    /* Mapper.h:928 */
  m->border = border;
  m->roi_w = m->roi_h = grid_size;
  m->width = m->height = grid_size + 2 * border; /* Mapper.h:1018 */
  m->stride = (m->width + 7) & ~7;               /* Karto.h:4442, Math.h:243-247 */
  m->data_size = m->stride * m->height;          /* Karto.h:4628-4631 */
  m->grid = (uint8_t*)calloc((size_t)m->data_size, 1);
  m->scale = 1.0 / cfg->resolution;              /* Mapper.h:1020 */
  /* CalculateKernel (Mapper.h:1058-1086) */
  double resolution = 1.0 / m->scale;            /* GetResolution() = 1/scale (Karto.h:4335-4338) */
  m->kernel_size = 2 * half_kernel_size(cfg->smear_deviation, resolution) + 1;
  m->kernel = (uint8_t*)malloc((size_t)m->kernel_size * m->kernel_size);
  int hk = m->kernel_size / 2;
  for (int i = -hk; i <= hk; i++)
    for (int j = -hk; j <= hk; j++) {
      double d = hypot(i * resolution, j * resolution);
      double z = exp(-0.5 * pow(d / cfg->smear_deviation, 2));
      uint32_t v = (uint32_t)kround(z * GRID_OCCUPIED);
      m->kernel[(i + hk) + m->kernel_size * (j + hk)] = (uint8_t)v;
    }
  m->probs_side = (int)side;
  m->probs_stride = ((int)side + 7) & ~7;
  m->probs = (double*)calloc((size_t)m->probs_stride * side, sizeof(double));
  m->scratch_pts = (double*)malloc(sizeof(double) * 4 * (size_t)(m->n_beams > 0 ? m->n_beams : 1));
  return m;
}

void kor_destroy(kor_matcher* m) {
  if (!m) return;
  free(m->kernel);
  free(m->grid);
  free(m->probs);
  free(m->table);
  free(m->scratch_pts);
  free(m);
}

int kor_num_beams(const kor_matcher* m) { return m->n_beams; }
void kor_grid_info(const kor_matcher* m, int32_t* out, double* offset) {
  out[0] = m->width; out[1] = m->height; out[2] = m->stride; out[3] = m->border; out[4] = m->border;
  out[5] = m->roi_w; out[6] = m->roi_h; out[7] = m->kernel_size;
  offset[0] = m->off_x; offset[1] = m->off_y;
}
const uint8_t* kor_grid_data(const kor_matcher* m) { return m->grid; }
const uint8_t* kor_kernel_data(const kor_matcher* m) { return m->kernel; }

/* LocalizedRangeScan::GetSensorAt (Karto.h:5310-5313) */
void kor_sensor_pose_from_robot(const kor_matcher* m, const double robot[3], double sensor[3]) {
  const double origin[3] = {0, 0, 0};
  const double off[3] = {m->laser.offset_x, m->laser.offset_y, m->laser.offset_heading};
  xform t = xform_make(origin, robot);
  xform_apply(&t, off, sensor);
}
/* LocalizedRangeScan::SetSensorPose (Karto.h:5289-5303) */
void kor_robot_pose_from_sensor(const kor_matcher* m, const double sensor[3], double robot[3]) {
  double ox = m->laser.offset_x, oy = m->laser.offset_y, oh = m->laser.offset_heading;
  double len = sqrt(ksquare(ox) + ksquare(oy));
  double angleoffset = atan2(oy, ox);
  double ch = normalize_angle(sensor[2]);
  double wx = len * cos(ch + angleoffset - oh);
  double wy = len * sin(ch + angleoffset - oh);
  robot[0] = sensor[0] - wx;
  robot[1] = sensor[1] - wy;
  robot[2] = normalize_angle(sensor[2] - oh); /* Pose2 operator- (Karto.h:2138-2141) */
}

/* LocalizedRangeScan::Update, unfiltered list (Karto.h:5379-5404): EVERY beam index < n_beams
 * yields a point, in or out of [minRange, rangeThreshold] alike. */
void kor_point_readings(const kor_matcher* m, const double* ranges, const double sp[3], double* out) {
  for (int i = 0; i < m->n_beams; i++) {
    double r = ranges[i];
    double angle = sp[2] + m->laser.minimum_angle + (uint32_t)i * m->laser.angular_resolution;
    out[2 * i] = sp[0] + (r * cos(angle));
    out[2 * i + 1] = sp[1] + (r * sin(angle));
  }
}

/* ScanMatcher::FindValidPoints (Mapper.cpp:756-811) */
int kor_find_valid_points(const double* p, int n, const double vp[2], double* out) {
  const double min_sq = ksquare(0.1);
  int trailing = 0, n_out = 0, first_time = 1;
  double fx = 0.0, fy = 0.0; /* Vector2 default-constructs to (0,0) (Karto.h:962-966) */
  for (int i = 0; i < n; i++) {
    double cx = p[2 * i], cy = p[2 * i + 1];
    if (first_time && !isnan(cx) && !isnan(cy)) {
      fx = cx; fy = cy; first_time = 0;
    }
    double dx = fx - cx, dy = fy - cy;
    if (ksquare(dx) + ksquare(dy) > min_sq) {
      double a = vp[1] - fy;
      double b = fx - vp[0];
      double c = fy * vp[0] - fx * vp[1];
      double ss = cx * a + cy * b + c;
      fx = cx; fy = cy;
      if (ss < 0.0) {
        trailing = i;
      } else {
        for (; trailing != i; ++trailing) {
          out[2 * n_out] = p[2 * trailing];
          out[2 * n_out + 1] = p[2 * trailing + 1];
          n_out++;
        }
      }
    }
  }
  return n_out;
}

/* CoordinateConverter::WorldToGrid (Karto.h:4237-4252) */
static void world_to_grid(double wx, double wy, double ox, double oy, double scale, int* gx, int* gy) {
  *gx = (int)kround((wx - ox) * scale);
  *gy = (int)kround((wy - oy) * scale);
}

/* CorrelationGrid::SmearPoint (Mapper.h:971-1005); (gx,gy) are ROI coordinates */
static void smear_point(kor_matcher* m, int gx, int gy) {
  int hk = m->kernel_size / 2;
  for (int j = -hk; j <= hk; j++) {
    uint8_t* row = m->grid + (gx + m->border) + (size_t)(gy + j + m->border) * m->stride;
    for (int i = -hk; i <= hk; i++) {
      uint8_t kv = m->kernel[(i + hk) + m->kernel_size * (j + hk)];
      if (kv > row[i]) row[i] = kv;
    }
  }
}

/* ScanMatcher::AddScan (Mapper.cpp:716-748) for an already-validated point list */
static void add_points(kor_matcher* m, const double* pts, int n) {
  for (int i = 0; i < n; i++) {
    int gx, gy;
    world_to_grid(pts[2 * i], pts[2 * i + 1], m->off_x, m->off_y, m->scale, &gx, &gy);
    if (!is_up_to(gx, m->roi_w) || !is_up_to(gy, m->roi_h)) continue;
    size_t idx = (size_t)(gx + m->border) + (size_t)(gy + m->border) * m->stride;
    if (m->grid[idx] == GRID_OCCUPIED) continue;
    m->grid[idx] = GRID_OCCUPIED;
    smear_point(m, gx, gy);
  }
}

static void set_center(kor_matcher* m, const double center[3]) {
  /* Mapper.cpp:212-220 */
  m->off_x = center[0] - (0.5 * (m->roi_w - 1) * (1.0 / m->scale));
  m->off_y = center[1] - (0.5 * (m->roi_h - 1) * (1.0 / m->scale));
}

void kor_set_base_scans(kor_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                        const double* sensor_poses, const double center[3]) {
  set_center(m, center);
  memset(m->grid, 0, (size_t)m->data_size); /* AddScans -> Clear (Mapper.cpp:701) */
  double* pts = m->scratch_pts;
  double* valid = m->scratch_pts + 2 * (size_t)m->n_beams;
  for (int s = 0; s < n_scans; s++) {
    kor_point_readings(m, ranges + (size_t)s * ranges_stride, sensor_poses + 3 * s, pts);
    int nv = kor_find_valid_points(pts, m->n_beams, center, valid); /* viewpoint = scanPose position (:225) */
    add_points(m, valid, nv);
  }
}

void kor_set_grid(kor_matcher* m, const uint8_t* grid, const double offset[2]) {
  memcpy(m->grid, grid, (size_t)m->data_size);
  m->off_x = offset[0];
  m->off_y = offset[1];
}

static int n_angles_of(double angle_offset, double angle_resolution) {
  return (int)(uint32_t)(kround(angle_offset * 2.0 / angle_resolution) + 1); /* Karto.h:6417 */
}

int kor_compute_offsets(kor_matcher* m, const double* ranges, const double sp[3], double angle_center,
                        double angle_offset, double angle_resolution, int32_t* out) {
  int n_angles = n_angles_of(angle_offset, angle_resolution);
  if (!out) return n_angles;
  int n = m->n_beams;
  double* world = m->scratch_pts;
  double* local = m->scratch_pts + 2 * (size_t)n;
  kor_point_readings(m, ranges, sp, world); /* GetPointReadings() unfiltered (Karto.h:6423) */
  const double origin[3] = {0, 0, 0};
  xform t = xform_make(origin, sp);          /* Transform(pScan->GetSensorPose()) (:6426) */
  for (int i = 0; i < n; i++)                /* :6428-6434 */
    xform_inverse_point(&t, world[2 * i], world[2 * i + 1], &local[2 * i], &local[2 * i + 1]);
  double start = angle_center - angle_offset; /* :6439 */
  for (int a = 0; a < n_angles; a++) {
    double angle = start + (uint32_t)a * angle_resolution; /* :6442 */
    double cosine = cos(angle), sine = sin(angle);         /* :6465-6466 */
    int32_t* row = out + (size_t)a * n;
    for (int i = 0; i < n; i++) {
      if (isnan(ranges[i]) || isinf(ranges[i])) { /* :6478-6483 */
        row[i] = INVALID_SCAN;
        continue;
      }
      double lx = local[2 * i], ly = local[2 * i + 1];
      double ox = cosine * lx - sine * ly; /* :6487-6488 */
      double oy = sine * lx + cosine * ly;
      int gx, gy;                           /* :6491  WorldToGrid(offset + rGridOffset) */
      world_to_grid(ox + m->off_x, oy + m->off_y, m->off_x, m->off_y, m->scale, &gx, &gy);
      row[i] = gx + gy * m->stride;         /* :6494 base-class GridIndex, no ROI, no check */
    }
  }
  return n_angles;
}

int64_t kor_response_sum(const kor_matcher* m, const int32_t* row, int32_t pos) {
  /* Mapper.cpp:838-849; the sum of uint8 values is exact in the reference's double too */
  int64_t sum = 0;
  for (int i = 0; i < m->n_beams; i++) {
    if (row[i] == INVALID_SCAN) continue;
    int64_t idx = (int64_t)pos + row[i];
    if (idx < 0 || idx >= m->data_size) continue; /* 1-D check on the flat index only (:842) */
    sum += m->grid[idx];
  }
  return sum;
}

static double response_of(const kor_matcher* m, int64_t sum) {
  if (m->n_beams == 0) return 0.0;                         /* :831-834 */
  return (double)sum / (double)((uint32_t)m->n_beams * GRID_OCCUPIED); /* :852 */
}

/* ComputePositionalCovariance (Mapper.cpp:535-630) */
static int positional_covariance(kor_matcher* m, const double best_pose[3], double best,
                                 const double center[3], double off_x, double off_y, double res_x,
                                 double res_y, double ang_res, double cov[9]) {
  memset(cov, 0, 9 * sizeof(double));
  cov[0] = cov[4] = cov[8] = 1.0; /* SetToIdentity (:542) */
  if (best < KT_TOLERANCE) {
    cov[0] = MAX_VARIANCE; cov[4] = MAX_VARIANCE; cov[8] = 4 * ksquare(ang_res);
    return 0;
  }
  double axx = 0, axy = 0, ayy = 0, norm = 0;
  double dx = best_pose[0] - center[0], dy = best_pose[1] - center[1];
  uint32_t nx = (uint32_t)(kround(off_x * 2.0 / res_x) + 1);
  uint32_t ny = (uint32_t)(kround(off_y * 2.0 / res_y) + 1);
  double sx = -off_x, sy = -off_y;
  for (uint32_t yi = 0; yi < ny; yi++) {
    double y = sy + yi * res_y;
    for (uint32_t xi = 0; xi < nx; xi++) {
      double x = sx + xi * res_x;
      int gx, gy;
      world_to_grid(center[0] + x, center[1] + y, m->probs_off_x, m->probs_off_y, m->scale, &gx, &gy);
      if (!is_up_to(gx, m->probs_side) || !is_up_to(gy, m->probs_side)) return -3; /* Grid::GridIndex throws */
      double r = m->probs[gx + gy * m->probs_stride];
      if (r >= (best - 0.1)) {
        norm += r;
        axx += (ksquare(x - dx) * r);
        axy += ((x - dx) * (y - dy) * r);
        ayy += (ksquare(y - dy) * r);
      }
    }
  }
  if (norm > KT_TOLERANCE) {
    double vxx = axx / norm, vxy = axy / norm, vyy = ayy / norm;
    double vthth = 4 * ksquare(ang_res);
    double min_xx = 0.1 * ksquare(res_x), min_yy = 0.1 * ksquare(res_y);
    vxx = vxx > min_xx ? vxx : min_xx; /* math::Maximum (Math.h:111-114) */
    vyy = vyy > min_yy ? vyy : min_yy;
    double mult = 1.0 / best;
    cov[0] = vxx * mult; cov[1] = vxy * mult; cov[3] = vxy * mult; cov[4] = vyy * mult;
    cov[8] = vthth;
  }
  if (double_equal(cov[0], 0.0)) cov[0] = MAX_VARIANCE;
  if (double_equal(cov[4], 0.0)) cov[4] = MAX_VARIANCE;
  return 0;
}

/* ComputeAngularCovariance (Mapper.cpp:641-692); the lookup table of the fine pass is in m->table */
static int angular_covariance(kor_matcher* m, const double best_pose[3], double best,
                              const double center[3], double ang_off, double ang_res, double cov[9]) {
  double best_angle = normalize_angle_difference(best_pose[2], center[2]);
  int gx, gy;
  world_to_grid(best_pose[0], best_pose[1], m->off_x, m->off_y, m->scale, &gx, &gy);
  gx += m->border; gy += m->border; /* CorrelationGrid::GridIndex adds the ROI (Mapper.h:941-947) */
  if (!is_up_to(gx, m->width) || !is_up_to(gy, m->height)) return -3;
  int32_t pos = gx + gy * m->stride;
  uint32_t n_angles = (uint32_t)(kround(ang_off * 2 / ang_res) + 1);
  double start = center[2] - ang_off, norm = 0.0, acc = 0.0;
  for (uint32_t a = 0; a < n_angles; a++) {
    double angle = start + a * ang_res;
    double r = response_of(m, kor_response_sum(m, m->table + (size_t)a * m->n_beams, pos));
    if (r >= (best - 0.1)) {
      norm += r;
      acc += (ksquare(angle - best_angle) * r);
    }
  }
  if (norm > KT_TOLERANCE) {
    if (acc < KT_TOLERANCE) acc = ksquare(ang_res);
    acc /= norm;
  } else {
    acc = 1000 * ksquare(ang_res);
  }
  cov[8] = acc;
  return 0;
}

double kor_correlate_scan(kor_matcher* m, const double* ranges, const double sp[3],
                          const double center[3], double off_x, double off_y, double res_x,
                          double res_y, double ang_off, double ang_res, int do_penalize,
                          int doing_fine, double mean[3], double cov[9], int32_t* resp_sums,
                          int* status) {
  *status = 0;
  int n_angles = n_angles_of(ang_off, ang_res);
  size_t need = (size_t)n_angles * (size_t)(m->n_beams > 0 ? m->n_beams : 1);
  if ((size_t)m->table_cap < need) {
    free(m->table);
    m->table = (int32_t*)malloc(need * sizeof(int32_t));
    m->table_cap = (int)need;
  }
  kor_compute_offsets(m, ranges, sp, center[2], ang_off, ang_res, m->table); /* Mapper.cpp:324 */
  if (!doing_fine) { /* :327-334 */
    memset(m->probs, 0, sizeof(double) * (size_t)m->probs_stride * m->probs_side);
    m->probs_off_x = center[0] - off_x;
    m->probs_off_y = center[1] - off_y;
  }
  uint32_t nx = (uint32_t)(kround(off_x * 2.0 / res_x) + 1); /* :339-341 */
  uint32_t ny = (uint32_t)(kround(off_y * 2.0 / res_y) + 1); /* :350-352 */
  double sx = -off_x, sy = -off_y;
  size_t total = (size_t)nx * ny * (size_t)n_angles;
  double* resp = (double*)malloc(total * sizeof(double));
  double* px = (double*)malloc(total * 3 * sizeof(double));
  size_t k = 0;
  for (uint32_t yi = 0; yi < ny; yi++) {
    double y = sy + yi * res_y;
    double npy = center[1] + y;
    double sqy = ksquare(y);
    for (uint32_t xi = 0; xi < nx; xi++) {
      double x = sx + xi * res_x;
      double npx = center[0] + x;
      double sqx = ksquare(x);
      int gx, gy;
      world_to_grid(npx, npy, m->off_x, m->off_y, m->scale, &gx, &gy); /* :385 */
      gx += m->border; gy += m->border;                                   /* :386 */
      if (!is_up_to(gx, m->width) || !is_up_to(gy, m->height)) {          /* karto::Exception */
        *status = -3;
        free(resp); free(px);
        return 0.0;
      }
      int32_t pos = gx + gy * m->stride;
      double start_angle = center[2] - ang_off; /* :390 */
      for (int a = 0; a < n_angles; a++) {
        double angle = start_angle + (uint32_t)a * ang_res; /* :393 */
        int64_t sum = kor_response_sum(m, m->table + (size_t)a * m->n_beams, pos);
        if (resp_sums) resp_sums[k] = (int32_t)sum;
        double r = response_of(m, sum);
        if (do_penalize && !double_equal(r, 0.0)) { /* :399-414 */
          double sd = sqx + sqy;
          double dp = 1.0 - (DISTANCE_PENALTY_GAIN * sd / m->cfg.distance_variance_penalty);
          dp = dp > m->cfg.minimum_distance_penalty ? dp : m->cfg.minimum_distance_penalty;
          double sad = ksquare(angle - center[2]);
          double ap = 1.0 - (ANGLE_PENALTY_GAIN * sad / m->cfg.angle_variance_penalty);
          ap = ap > m->cfg.minimum_angle_penalty ? ap : m->cfg.minimum_angle_penalty;
          r *= (dp * ap);
        }
        resp[k] = r;
        px[3 * k] = npx; px[3 * k + 1] = npy; px[3 * k + 2] = normalize_angle(angle); /* :417-418 */
        k++;
      }
    }
  }
  double best = -1; /* :431-451 */
  for (size_t i = 0; i < total; i++) {
    best = best > resp[i] ? best : resp[i];
    if (!doing_fine) {
      int gx, gy;
      world_to_grid(px[3 * i], px[3 * i + 1], m->probs_off_x, m->probs_off_y, m->scale, &gx, &gy);
      if (!is_up_to(gx, m->probs_side) || !is_up_to(gy, m->probs_side)) {
        *status = -4; /* "Index out of range in probability search" / karto::Exception */
        free(resp); free(px);
        return 0.0;
      }
      double* p = &m->probs[gx + gy * m->probs_stride];
      *p = resp[i] > *p ? resp[i] : *p;
    }
  }
  double ax = 0, ay = 0, tx = 0, ty = 0; /* :456-483 */
  int cnt = 0;
  for (size_t i = 0; i < total; i++) {
    if (double_equal(resp[i], best)) {
      ax += px[3 * i]; ay += px[3 * i + 1];
      tx += cos(px[3 * i + 2]); ty += sin(px[3 * i + 2]);
      cnt++;
    }
  }
  free(resp); free(px);
  if (cnt == 0) { *status = -5; return 0.0; } /* "Unable to find best position" */
  double avg[3];
  ax /= cnt; ay /= cnt; tx /= cnt; ty /= cnt;
  avg[0] = ax; avg[1] = ay; avg[2] = atan2(ty, tx);
  int rc;
  if (!doing_fine)
    rc = positional_covariance(m, avg, best, center, off_x, off_y, res_x, res_y, ang_res, cov);
  else
    rc = angular_covariance(m, avg, best, center, ang_off, ang_res, cov);
  if (rc) { *status = rc; return 0.0; }
  mean[0] = avg[0]; mean[1] = avg[1]; mean[2] = avg[2];
  if (best > 1.0) best = 1.0;
  return best;
}

double kor_match(kor_matcher* m, const double* ranges, const double sp[3], int do_penalize,
                 int do_refine, double mean[3], double cov[9], int* status) {
  *status = 0;
  if (m->n_beams == 0) { /* Mapper.cpp:199-209 (scan without readings) */
    mean[0] = sp[0]; mean[1] = sp[1]; mean[2] = sp[2];
    cov[0] = MAX_VARIANCE; cov[4] = MAX_VARIANCE; cov[8] = 4 * ksquare(m->cfg.coarse_angle_resolution);
    return 0.0;
  }
  double res = 1.0 / m->scale;
  double dim = (double)m->probs_side;                 /* :228 */
  double coarse_off = 0.5 * (dim - 1) * res;          /* :229-230 */
  double coarse_res = 2 * res;                        /* :233-234 */
  double best = kor_correlate_scan(m, ranges, sp, sp, coarse_off, coarse_off, coarse_res, coarse_res,
                                   m->cfg.coarse_search_angle_offset, m->cfg.coarse_angle_resolution,
                                   do_penalize, 0, mean, cov, NULL, status);
  if (*status) return 0.0;
  if (m->cfg.use_response_expansion && double_equal(best, 0.0)) { /* :242-272 */
    double nso = m->cfg.coarse_search_angle_offset;
    for (int i = 0; i < 3; i++) {
      nso += 20.0 * (KT_PI / 180.0); /* math::DegreesToRadians(20) = 20*KT_PI_180 (Math.h:56-59) */
      best = kor_correlate_scan(m, ranges, sp, sp, coarse_off, coarse_off, coarse_res, coarse_res, nso,
                                m->cfg.coarse_angle_resolution, do_penalize, 0, mean, cov, NULL, status);
      if (*status) return 0.0;
      if (!double_equal(best, 0.0)) break;
    }
  }
  if (do_refine) { /* :274-282 */
    double fine_off = coarse_res * 0.5;
    double c[3] = {mean[0], mean[1], mean[2]};
    best = kor_correlate_scan(m, ranges, sp, c, fine_off, fine_off, res, res,
                              0.5 * m->cfg.coarse_angle_resolution, m->cfg.fine_search_angle_offset,
                              do_penalize, 1, mean, cov, NULL, status);
    if (*status) return 0.0;
  }
  return best;
}

double kor_match_scan(kor_matcher* m, int n_base, const double* base_ranges, int ranges_stride,
                      const double* base_sensor_poses, const double* q_ranges, const double qsp[3],
                      int do_penalize, int do_refine, double mean[3], double cov[9], int* status) {
  if (m->n_beams != 0) kor_set_base_scans(m, n_base, base_ranges, ranges_stride, base_sensor_poses, qsp);
  return kor_match(m, q_ranges, qsp, do_penalize, do_refine, mean, cov, status);
}

int kor_probs(const kor_matcher* m, double* out) {
  for (int y = 0; y < m->probs_side; y++)
    for (int x = 0; x < m->probs_side; x++) out[y * m->probs_side + x] = m->probs[x + y * m->probs_stride];
  return m->probs_side;
}

/* ------------------------------------------------------------------------------------------- */
/* karto::OccupancyGrid (Karto.h:5609-6039) */
typedef struct { int w, h, stride; double scale, ox, oy; uint32_t *pass, *hit; } occ_t;

/* Grid<T>::TraceLine (Karto.h:4680-4745) with the pass-count increment */
static void occ_trace_line(occ_t* g, int x0, int y0, int x1, int y1) {
  int steep = abs(y1 - y0) > abs(x1 - x0);
  int t;
  if (steep) { t = x0; x0 = y0; y0 = t; t = x1; x1 = y1; y1 = t; }
  if (x0 > x1) { t = x0; x0 = x1; x1 = t; t = y0; y0 = y1; y1 = t; }
  int delta_x = x1 - x0, delta_y = abs(y1 - y0), error = 0, y = y0;
  int ystep = y0 < y1 ? 1 : -1;
  for (int x = x0; x <= x1; x++) {
    int px = steep ? y : x, py = steep ? x : y;
    error += delta_y;
    if (2 * error >= delta_x) { y += ystep; error -= delta_x; }
    if (is_up_to(px, g->w) && is_up_to(py, g->h)) g->pass[px + py * g->stride]++;
  }
}

/* Scan-box union of ComputeDimensions (Karto.h:5799-5817): a scan's box holds its sensor position and its FILTERED
 * readings (minRange <= r <= rangeThreshold, Karto.h:5382,5418-5424).  box = minx, miny, maxx, maxy; starts from
 * BoundingBox2() (Karto.h:2765) -- min/max are exact, so boxes of disjoint scan subsets merge to the box of the union. */
void kor_occgrid_bounds(const kor_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                        const double* sposes, double box[4]) {
  const int n = m->n_beams;
  const double thr = m->laser.range_threshold, rmin = m->laser.minimum_range;
  double* pts = (double*)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
  double mnx = 999999999999999999.99999, mny = mnx, mxx = -mnx, mxy = -mnx; /* Karto.h:2765 */
  for (int s = 0; s < n_scans; s++) {
    const double* sp = sposes + 3 * s;
    const double* r = ranges + (size_t)s * ranges_stride;
    kor_point_readings(m, r, sp, pts);
    double bnx = 999999999999999999.99999, bny = bnx, bxx = -bnx, bxy = -bnx;
#define ADDPT(X, Y) do { if ((X) < bnx) bnx = (X); if ((Y) < bny) bny = (Y); if ((X) > bxx) bxx = (X); if ((Y) > bxy) bxy = (Y); } while (0)
    ADDPT(sp[0], sp[1]);
    for (int i = 0; i < n; i++)
      if (r[i] >= rmin && r[i] <= thr) ADDPT(pts[2 * i], pts[2 * i + 1]);
#undef ADDPT
    /* boundingBox.Add(scanBox): Add(min), Add(max) (Karto.h:2824-2828) */
    const double bx[2] = {bnx, bxx}, by[2] = {bny, bxy};
    for (int c = 0; c < 2; c++) {
      if (bx[c] < mnx) mnx = bx[c];
      if (by[c] < mny) mny = by[c];
      if (bx[c] > mxx) mxx = bx[c];
      if (by[c] > mxy) mxy = by[c];
    }
  }
  free(pts);
  box[0] = mnx; box[1] = mny; box[2] = mxx; box[3] = mxy;
}

static void occ_dims(const double box[4], double resolution, occ_t* g) {
  g->scale = 1.0 / resolution;
  g->w = (int)kround((box[2] - box[0]) * g->scale);
  g->h = (int)kround((box[3] - box[1]) * g->scale);
  g->ox = box[0]; g->oy = box[1];
  g->stride = (g->w + 7) & ~7; /* Grid<kt_int32u>::Resize (Karto.h:4442) */
}

/* AddScan (Karto.h:5851-5895) of every scan into zeroed counters */
static void occ_add_scans(const kor_matcher* m, occ_t* g, int n_scans, const double* ranges, int ranges_stride,
                          const double* sposes) {
  const int n = m->n_beams;
  const double thr = m->laser.range_threshold, rmin = m->laser.minimum_range, rmax = m->laser.maximum_range;
  double* pts = (double*)malloc(sizeof(double) * 2 * (size_t)(n > 0 ? n : 1));
  for (int s = 0; s < n_scans; s++) {
    const double* sp = sposes + 3 * s;
    const double* r = ranges + (size_t)s * ranges_stride;
    kor_point_readings(m, r, sp, pts);
    for (int i = 0; i < n; i++) {
      double px = pts[2 * i], py = pts[2 * i + 1], rr = r[i];
      int end_valid = rr < (thr - KT_TOLERANCE);
      if (rr <= rmin || rr >= rmax || isnan(rr)) continue;
      if (rr >= thr) {
        double ratio = thr / rr;
        double dx = px - sp[0], dy = py - sp[1];
        px = sp[0] + ratio * dx;
        py = sp[1] + ratio * dy;
      }
      int fx, fy, tx, ty; /* RayTrace (Karto.h:5907-5942) */
      world_to_grid(sp[0], sp[1], g->ox, g->oy, g->scale, &fx, &fy);
      world_to_grid(px, py, g->ox, g->oy, g->scale, &tx, &ty);
      occ_trace_line(g, fx, fy, tx, ty);
      if (end_valid && is_up_to(tx, g->w) && is_up_to(ty, g->h)) {
        g->pass[tx + ty * g->stride]++;
        g->hit[tx + ty * g->stride]++;
      }
    }
  }
  free(pts);
}

/* Update / UpdateCell (Karto.h:5950-5990): MinPassThrough = 2, OccupancyThreshold = 0.1 (:5636-5637) */
static void occ_update(const occ_t* g, uint8_t* out) {
  for (int y = 0; y < g->h; y++)
    for (int x = 0; x < g->w; x++) {
      uint32_t pc = g->pass[x + y * g->stride], hc = g->hit[x + y * g->stride];
      uint8_t v = 0;
      if (pc > 2) v = ((double)hc / (double)pc > 0.1) ? 100 : 255;
      out[(size_t)y * g->w + x] = v;
    }
}

int kor_occgrid_from_scans(const kor_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                           const double* sposes, double resolution, int32_t dims[2], double offset_xy[2],
                           uint8_t* out) {
  if (n_scans <= 0) return -1; /* rScans.empty() -> NULL (Karto.h:5661-5664) */
  double box[4];
  kor_occgrid_bounds(m, n_scans, ranges, ranges_stride, sposes, box);
  occ_t g;
  occ_dims(box, resolution, &g);
  dims[0] = g.w; dims[1] = g.h;
  offset_xy[0] = g.ox; offset_xy[1] = g.oy;
  if (!out) return 0;
  size_t cells = (size_t)g.stride * (g.h > 0 ? g.h : 0);
  g.pass = (uint32_t*)calloc(cells ? cells : 1, sizeof(uint32_t));
  g.hit = (uint32_t*)calloc(cells ? cells : 1, sizeof(uint32_t));
  occ_add_scans(m, &g, n_scans, ranges, ranges_stride, sposes);
  occ_update(&g, out);
  free(g.pass); free(g.hit);
  return 0;
}

/* The sharded build's pieces (test stand-in for the device path): the hit/pass counters of a SUBSET of the scans on
 * the grid of a given box -- counters[0 .. stride*h) = pass plane, then the hit plane (dims = w, h, stride; counters
 * NULL = dims only) -- and Update() over summed counters. */
int kor_occgrid_partial(const kor_matcher* m, int n_scans, const double* ranges, int ranges_stride,
                        const double* sposes, double resolution, const double box[4], int32_t dims[3],
                        uint32_t* counters) {
  occ_t g;
  occ_dims(box, resolution, &g);
  dims[0] = g.w; dims[1] = g.h; dims[2] = g.stride;
  if (!counters) return 0;
  size_t cells = (size_t)g.stride * (g.h > 0 ? g.h : 0);
  memset(counters, 0, 2 * cells * sizeof(uint32_t));
  g.pass = counters;
  g.hit = counters + cells;
  occ_add_scans(m, &g, n_scans, ranges, ranges_stride, sposes);
  return 0;
}

void kor_occgrid_update(const int32_t dims[3], const uint32_t* counters, uint8_t* out) {
  occ_t g;
  g.w = dims[0]; g.h = dims[1]; g.stride = dims[2];
  size_t cells = (size_t)g.stride * (g.h > 0 ? g.h : 0);
  g.pass = (uint32_t*)counters;
  g.hit = (uint32_t*)counters + cells;
  occ_update(&g, out);
}

/* Matrix3::InverseFast by cofactors (Karto.h:2460-2493), tolerance 1e-14 as Inverse() passes it */
static int mat3_inverse(const double m[9], double inv[9]) {
  inv[0] = m[4] * m[8] - m[5] * m[7];
  inv[1] = m[2] * m[7] - m[1] * m[8];
  inv[2] = m[1] * m[5] - m[2] * m[4];
  inv[3] = m[5] * m[6] - m[3] * m[8];
  inv[4] = m[0] * m[8] - m[2] * m[6];
  inv[5] = m[2] * m[3] - m[0] * m[5];
  inv[6] = m[3] * m[7] - m[4] * m[6];
  inv[7] = m[1] * m[6] - m[0] * m[7];
  inv[8] = m[0] * m[4] - m[1] * m[3];
  double det = m[0] * inv[0] + m[1] * inv[3] + m[2] * inv[6];
  if (fabs(det) <= 1e-14) return 0;
  double id = 1.0 / det;
  for (int i = 0; i < 9; i++) inv[i] *= id;
  return 1;
}
/* Matrix3 operator* (Karto.h:2552-2568) */
static void mat3_mul(const double a[9], const double b[9], double out[9]) {
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      out[3 * r + c] = a[3 * r] * b[c] + a[3 * r + 1] * b[3 + c] + a[3 * r + 2] * b[6 + c];
}
/* MapperGraph::ComputeWeightedMean (Mapper.cpp:1288-1330): AddEdges ends with
 * SetSensorPose(ComputeWeightedMean(means, covariances)) (Mapper.cpp:969-972), so even a single
 * (mean, covariance) pair passes through inverse(inverse(C)) * inverse(C) * pose. */
static void weighted_mean(int n, const double* means, const double* covs, double out[3]) {
  double sum[9] = {0};
  double* inv = (double*)malloc(sizeof(double) * 9 * (size_t)n);
  for (int i = 0; i < n; i++) {
    mat3_inverse(covs + 9 * i, inv + 9 * i);
    for (int k = 0; k < 9; k++) sum[k] += inv[9 * i + k];
  }
  double ios[9];
  mat3_inverse(sum, ios);
  double ax = 0.0, ay = 0.0, ah = 0.0, tx = 0.0, ty = 0.0;
  for (int i = 0; i < n; i++) {
    const double* p = means + 3 * i;
    tx += cos(p[2]);
    ty += sin(p[2]);
    double w[9];
    mat3_mul(ios, inv + 9 * i, w);
    double wx = w[0] * p[0] + w[1] * p[1] + w[2] * p[2]; /* Matrix3 * Pose2 (Karto.h:2574-2583) */
    double wy = w[3] * p[0] + w[4] * p[1] + w[5] * p[2];
    double wh = w[6] * p[0] + w[7] * p[1] + w[8] * p[2];
    ax += wx; ay += wy;                                   /* Pose2 operator+= (Karto.h:2117-2121) */
    ah = normalize_angle(ah + wh);
  }
  free(inv);
  tx /= n; ty /= n;
  out[0] = ax; out[1] = ay; out[2] = atan2(ty, tx);
}

/* ------------------------------------------------------------------------------------------- */
struct kor_frontend {
  kor_matcher* m;
  int buf_size;
  double buf_dist, min_travel, min_heading;
  int n_run, cap_run;
  double* run_ranges; /* [cap][n_ranges] */
  double* run_robot;  /* [cap][3] corrected robot poses */
  int have_last;
  double last_odom[3], last_corr[3];
  int n_ranges;
};

kor_frontend* kor_frontend_create(kor_matcher* m, int scan_buffer_size, double scan_buffer_max_distance,
                                  double min_travel_distance, double min_travel_heading) {
  kor_frontend* f = (kor_frontend*)calloc(1, sizeof *f);
  f->m = m;
  f->buf_size = scan_buffer_size;
  f->buf_dist = scan_buffer_max_distance;
  f->min_travel = min_travel_distance;
  f->min_heading = min_travel_heading;
  f->n_ranges = m->n_beams;
  f->cap_run = scan_buffer_size + 2;
  f->run_ranges = (double*)malloc(sizeof(double) * (size_t)f->cap_run * f->n_ranges);
  f->run_robot = (double*)malloc(sizeof(double) * 3 * (size_t)f->cap_run);
  return f;
}
void kor_frontend_destroy(kor_frontend* f) {
  if (!f) return;
  free(f->run_ranges);
  free(f->run_robot);
  free(f);
}
int kor_frontend_running_scans(const kor_frontend* f) { return f->n_run; }

static double sq_dist(const double* a, const double* b) {
  return ksquare(a[0] - b[0]) + ksquare(a[1] - b[1]); /* Vector2::SquaredDistance (Karto.h:1020-1023) */
}

int kor_frontend_process(kor_frontend* f, const double* ranges, const double odom[3], double out_pose[3],
                         double out_cov[9], double* out_response) {
  kor_matcher* m = f->m;
  double corrected[3] = {odom[0], odom[1], odom[2]};
  double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}; /* Mapper.cpp:2033-2034 */
  double response = 0.0;
  if (f->have_last) { /* :2021-2025 */
    xform t = xform_make(f->last_odom, f->last_corr);
    xform_apply(&t, odom, corrected);
    /* HasMovedEnough (:2087-2120) on the ODOMETRIC sensor poses */
    double lsp[3], csp[3];
    kor_sensor_pose_from_robot(m, f->last_odom, lsp);
    kor_sensor_pose_from_robot(m, odom, csp);
    double dh = normalize_angle(csp[2] - lsp[2]);
    int moved = fabs(dh) >= f->min_heading;
    if (!moved) moved = sq_dist(lsp, csp) >= ksquare(f->min_travel) - KT_TOLERANCE;
    if (!moved) {
      out_pose[0] = corrected[0]; out_pose[1] = corrected[1]; out_pose[2] = corrected[2];
      return 0;
    }
    /* MatchScan against the running scans (:2037-2045) */
    double* sposes = (double*)malloc(sizeof(double) * 3 * (size_t)(f->n_run > 0 ? f->n_run : 1));
    for (int i = 0; i < f->n_run; i++) kor_sensor_pose_from_robot(m, f->run_robot + 3 * i, sposes + 3 * i);
    double qsp[3], mean[3];
    int status;
    kor_sensor_pose_from_robot(m, corrected, qsp);
    response = kor_match_scan(m, f->n_run, f->run_ranges, f->n_ranges, sposes, ranges, qsp, 1, 1, mean, cov, &status);
    free(sposes);
    if (status) return status;
    kor_robot_pose_from_sensor(m, mean, corrected); /* SetSensorPose(bestPose) (:2044) */
    /* AddEdges (Mapper.cpp:957-972): means = {GetSensorPose()}, covariances = {covariance}; near
     * chains (graph traversal, LinkNearChains) are host-side back-end work and not restated: the
     * harness trajectories do not revisit, so no near chain exists. */
    double sp[3], wm[3];
    kor_sensor_pose_from_robot(m, corrected, sp);
    weighted_mean(1, sp, cov, wm);
    kor_robot_pose_from_sensor(m, wm, corrected);
  }
  /* AddRunningScan (Mapper.h:1365-1386) */
  memcpy(f->run_ranges + (size_t)f->n_run * f->n_ranges, ranges, sizeof(double) * f->n_ranges);
  memcpy(f->run_robot + 3 * (size_t)f->n_run, corrected, sizeof corrected);
  f->n_run++;
  for (;;) {
    double fs[3], bs[3];
    kor_sensor_pose_from_robot(m, f->run_robot, fs);
    kor_sensor_pose_from_robot(m, f->run_robot + 3 * (size_t)(f->n_run - 1), bs);
    double d2 = sq_dist(fs, bs);
    if (!((uint32_t)f->n_run > (uint32_t)f->buf_size || d2 > ksquare(f->buf_dist) - KT_TOLERANCE)) break;
    memmove(f->run_ranges, f->run_ranges + f->n_ranges, sizeof(double) * (size_t)(f->n_run - 1) * f->n_ranges);
    memmove(f->run_robot, f->run_robot + 3, sizeof(double) * 3 * (size_t)(f->n_run - 1));
    f->n_run--;
  }
  memcpy(f->last_odom, odom, sizeof f->last_odom); /* SetLastScan (:2074) */
  memcpy(f->last_corr, corrected, sizeof f->last_corr);
  f->have_last = 1;
  out_pose[0] = corrected[0]; out_pose[1] = corrected[1]; out_pose[2] = corrected[2];
  if (out_cov) memcpy(out_cov, cov, sizeof cov);
  if (out_response) *out_response = response;
  return 1;
}
