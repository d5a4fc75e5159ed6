// Streaming front-end: the pose-relevant part of karto::Mapper::Process (Mapper.cpp:1999-2079)
// with a DEVICE-RESIDENT running-scan window.  Included at the end of scan_matcher.hip (it drives
// that file's kernels).
//
// Per processed scan (everything between the two host decisions stays on the GPU):
//   host   lastTransform propagation (Mapper.cpp:2021-2025), HasMovedEnough (:2087-2120)
//   device AddScans of the window around the query (find_valid, mark+smear; parity planes and
//          row-occupancy refreshed), coarse+fine search of the query (MatchScan, :2040)
//   host   SetSensorPose(bestPose) (:2044), AddEdges' closing SetSensorPose(ComputeWeightedMean)
//          for the single-mean case (:957-972), AddRunningScan window policy (Mapper.h:1365-1386)
//   device world points of the accepted scan into its ring slot (LocalizedRangeScan::Update,
//          Karto.h:5362-5428) -- computed once per scan, reused by every later grid rebuild
// The pose graph itself (vertices/edges, LinkNearChains, loop closure, solvers) is the reference's
// back-end and stays on the host, out of scope.
#pragma once

struct lslam_frontend {
  lslam_matcher* m = nullptr;
  int buf_size = 0;
  double buf_dist = 0, min_travel = 0, min_heading = 0;
  int cap = 0, start = 0, count = 0;  // ring of window scans
  DevBuf<double2> d_world;             // [cap][n] world points of window scans
  DevBuf<double> d_q;                  // query ranges (n) + pose (3)
  DevBuf<lslam_match_result> d_res;
  std::vector<double> robot;           // [cap][3] corrected robot poses, ring-indexed
  bool have_last = false;
  double last_odom[3] = {0, 0, 0}, last_corr[3] = {0, 0, 0};
};

namespace {

inline double sq_dist2(const double* a, const double* b) { return ksq(a[0] - b[0]) + ksq(a[1] - b[1]); }

}  // namespace

extern "C" {

int lslam_frontend_create(lslam_matcher* m, int scan_buffer_size, double scan_buffer_max_distance,
                          double min_travel_distance, double min_travel_heading, lslam_frontend** out) {
  if (!m || !out || scan_buffer_size < 1) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_context* ctx = m->ctx;
  lslam_frontend* f = new lslam_frontend();
  f->m = m;
  f->buf_size = scan_buffer_size;
  f->buf_dist = scan_buffer_max_distance;
  f->min_travel = min_travel_distance;
  f->min_heading = min_travel_heading;
  f->cap = scan_buffer_size + 1;  // the new scan is pushed before the front is trimmed
  const size_t n = (size_t)std::max(m->g.n_beams, 1);
  if (f->d_world.reserve((size_t)f->cap * n) != hipSuccess || f->d_q.reserve(n + 3) != hipSuccess ||
      f->d_res.reserve(1) != hipSuccess) {
    delete f;
    return ctx->fail(LSLAM_ERR_HIP, "cannot allocate the running-scan window in HBM");
  }
  f->robot.assign((size_t)f->cap * 3, 0.0);
  *out = f;
  return LSLAM_OK;
}

void lslam_frontend_destroy(lslam_frontend* f) {
  if (!f) return;
  (void)hipStreamSynchronize(f->m->ctx->stream);
  f->d_world.release();
  f->d_q.release();
  f->d_res.release();
  delete f;
}

int lslam_frontend_reset(lslam_frontend* f) {
  if (!f) return LSLAM_ERR_INVALID_ARGUMENT;
  f->start = f->count = 0;
  f->have_last = false;
  return LSLAM_OK;
}

int lslam_frontend_running_scans(const lslam_frontend* f) { return f ? f->count : LSLAM_ERR_INVALID_ARGUMENT; }

int lslam_frontend_process(lslam_frontend* f, const double* ranges, int n_ranges, const double odom_pose[3],
                           int* processed, double corrected_pose[3], double covariance[9], double* response) {
  if (!f || !ranges || !odom_pose || !processed || !corrected_pose) return LSLAM_ERR_INVALID_ARGUMENT;
  lslam_matcher* m = f->m;
  lslam_context* ctx = m->ctx;
  const Geom g0 = m->g;
  const int n = g0.n_beams;
  if (n_ranges < n) return ctx->fail(LSLAM_ERR_INVALID_ARGUMENT, "n_ranges %d < num_beams %d", n_ranges, n);
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  const lslam_laser* laser = &m->laser;
  double corrected[3] = {odom_pose[0], odom_pose[1], odom_pose[2]};
  double cov[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};  // Mapper.cpp:2033-2034
  double resp = 0.0;
  *processed = 0;
  if (f->have_last) {
    PoseXform t = pose_xform(f->last_odom, f->last_corr);  // :2021-2025
    pose_xform_apply(t, odom_pose, corrected);
    double lsp[3], csp[3];  // HasMovedEnough on the ODOMETRIC sensor poses (:2087-2120)
    lslam_sensor_pose_from_robot(laser, f->last_odom, lsp);
    lslam_sensor_pose_from_robot(laser, odom_pose, csp);
    double dh = normalize_angle(csp[2] - lsp[2]);
    bool moved = fabs(dh) >= f->min_heading;
    if (!moved) moved = sq_dist2(lsp, csp) >= ksq(f->min_travel) - kTol;
    if (!moved) {
      for (int i = 0; i < 3; i++) corrected_pose[i] = corrected[i];
      return LSLAM_OK;
    }
    // MatchScan(pScan, runningScans) (:2037-2045)
    double qsp[3];
    lslam_sensor_pose_from_robot(laser, corrected, qsp);
    if (n > 0) {
      LSLAM_HIP(ctx, hipMemcpyAsync(f->d_q.p, ranges, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      LSLAM_HIP(ctx, hipMemcpyAsync(f->d_q.p + n, qsp, 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      int rc = rebuild_grid_dev(m, f->d_world.p, f->start, f->count, f->cap, qsp);
      if (rc) return rc;
      rc = match_batch_impl<double>(m, 1, f->d_q.p, n, f->d_q.p + n, 1, 1, f->d_res.p, nullptr, 0);
      if (rc) return rc;
      lslam_match_result r;
      LSLAM_HIP(ctx, hipMemcpyAsync(&r, f->d_res.p, sizeof r, hipMemcpyDeviceToHost, ctx->stream));
      LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (r.status != LSLAM_OK) return ctx->fail(r.status, "scan matcher: the reference throws here");
      resp = r.response;
      for (int i = 0; i < 9; i++) cov[i] = r.covariance[i];
      lslam_robot_pose_from_sensor(laser, r.pose, corrected);  // SetSensorPose(bestPose) (:2044)
    } else {  // scan without readings (Mapper.cpp:199-209): rMean = scanPose
      cov[0] = cov[4] = kMaxVariance;
      cov[8] = 4 * ksq(m->cfg.coarse_angle_resolution);
      lslam_robot_pose_from_sensor(laser, qsp, corrected);
    }
    // AddEdges (:957-972): means = {GetSensorPose()}, covariances = {covariance}
    double sp[3], wm[3];
    lslam_sensor_pose_from_robot(laser, corrected, sp);
    weighted_mean_single(sp, cov, wm);
    lslam_robot_pose_from_sensor(laser, wm, corrected);
  }
  // AddRunningScan (Mapper.h:1365-1386): push, then trim the front
  const int slot = (f->start + f->count) % f->cap;
  if (n > 0) {
    double sp[3];
    lslam_sensor_pose_from_robot(laser, corrected, sp);
    if (!f->have_last)  // first scan: its ranges were not uploaded yet
      LSLAM_HIP(ctx, hipMemcpyAsync(f->d_q.p, ranges, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    LSLAM_HIP(ctx, hipMemcpyAsync(f->d_q.p + n, sp, 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    launch(ctx, "scan_prep_base", k_scan_prep<double>, dim3((n + 255) / 256, 1), dim3(256), 0,
           (const double*)f->d_q.p, n, (const double*)(f->d_q.p + n), m->g, (double2*)nullptr,
           f->d_world.p + (size_t)slot * n, PassCfg{}, (Lattice*)nullptr, (double2*)nullptr, 0);
    LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));  // d_q is reused by the next call
  }
  for (int i = 0; i < 3; i++) f->robot[(size_t)slot * 3 + i] = corrected[i];
  f->count++;
  for (;;) {
    double fs[3], bs[3];
    lslam_sensor_pose_from_robot(laser, &f->robot[(size_t)f->start * 3], fs);
    lslam_sensor_pose_from_robot(laser, &f->robot[(size_t)((f->start + f->count - 1) % f->cap) * 3], bs);
    double d2 = sq_dist2(fs, bs);
    if (!((uint32_t)f->count > (uint32_t)f->buf_size || d2 > ksq(f->buf_dist) - kTol)) break;
    f->start = (f->start + 1) % f->cap;
    f->count--;
  }
  for (int i = 0; i < 3; i++) {
    f->last_odom[i] = odom_pose[i];  // SetLastScan (:2074)
    f->last_corr[i] = corrected[i];
    corrected_pose[i] = corrected[i];
  }
  f->have_last = true;
  if (covariance) memcpy(covariance, cov, sizeof cov);
  if (response) *response = resp;
  *processed = 1;
  return LSLAM_OK;
}

}  // extern "C"
