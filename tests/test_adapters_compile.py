"""The C++ host layer above the C ABI (include/lslam_adapters.hpp) compiles stand-alone with g++
and links against liblslam_gpu.so; on a GPU box the little program also runs a MatchScan and a
map update through the adapters.  CPU part: compile + link only."""
import pathlib
import subprocess

import pytest

from lslam_amd import build

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = r'''
#include <cstdio>
#include <vector>
#include <cmath>
#include "lslam_adapters.hpp"
int main(int argc, char**) {
  lslam_context* ctx = nullptr;
  int rc = lslam_create(0, &ctx);
  if (rc != LSLAM_OK) { std::printf("no device: %s\n", lslam_last_error(nullptr)); return argc > 1 ? 1 : 0; }
  lslam_matcher_config cfg; lslam_matcher_config_defaults(&cfg);
  cfg.search_size = 1.0; cfg.resolution = 0.05; cfg.range_threshold = 20.0;
  lslam_laser laser = {-1.5, 1.5, 3.0 / 360.0, 0.1, 30.0, 20.0, 0, 0, 0};
  lslam::GpuScanMatcher* bad = lslam::GpuScanMatcher::Create(ctx, [&]{ auto c = cfg; c.resolution = 0; return c; }(), laser);
  if (bad) return 2;
  lslam::GpuScanMatcher* m = lslam::GpuScanMatcher::Create(ctx, cfg, laser);
  std::vector<double> a(360), b(360);
  for (int i = 0; i < 360; i++) {  // a wall 5 m ahead seen from x=0 and from x=0.2
    double ang = -1.5 + i * 3.0 / 360.0;
    a[i] = 5.0 / std::cos(ang); b[i] = 4.8 / std::cos(ang);
  }
  lslam::RangeScan base{a.data(), {0, 0, 0}}, q{b.data(), {0.1, 0.0, 0.0}};
  lslam::Pose2 mean; lslam::Matrix3 cov;
  double resp = m->MatchScan(q, {base}, mean, cov);
  std::printf("response %.3f mean %.3f %.3f %.3f\n", resp, mean.x, mean.y, mean.heading);
  if (!(resp > 0.2 && std::fabs(mean.x - 0.2) < 0.06)) return 3;
  int nz = 0;
  {  // handles must be released before their context
    lslam::MapRepGpu map(ctx, 0.05f, 256, 256, 2, 0.5f, 0.5f);
    float pts[4] = {40.f, 0.f, 0.f, 30.f}, origo[2] = {0, 0}, pose[3] = {0, 0, 0};
    map.updateByScan(pts, 2, origo, pose);
    std::vector<float> lo(256 * 256);
    map.readLogOdds(0, lo.data());
    for (float v : lo) nz += v != 0.f;
    std::printf("nonzero cells %d\n", nz);
  }
  // the batched many-scan mode over every GPU of the node (hipGetDeviceCount() devices), and -- so that the sharding
  // is exercised on a 1-GPU box too -- over two contexts on device 0: both must equal the single-matcher results
  {
    const int S = 37;
    std::vector<double> rr((size_t)S * 360), pp((size_t)S * 3);
    for (int i = 0; i < S; i++) {
      for (int k = 0; k < 360; k++) rr[(size_t)i * 360 + k] = b[k];
      pp[3 * i] = 0.05 + 0.004 * i; pp[3 * i + 1] = 0.003 * i - 0.05; pp[3 * i + 2] = 0.002 * i - 0.03;
    }
    const double zero[3] = {0, 0, 0};
    std::vector<lslam_match_result> one(S);
    if (lslam_matcher_set_base_scans(m->handle(), 1, a.data(), 360, zero, zero) != LSLAM_OK) return 5;
    if (lslam_matcher_match_batch(m->handle(), S, rr.data(), 360, pp.data(), 1, 1, one.data()) != LSLAM_OK) return 5;
    {  // the same batch, 16 times over, as two pipelined sub-batches of 296 scans (SetPipelineDepth): same records
      const int R = 16;
      std::vector<double> r2((size_t)S * R * 360), p2((size_t)S * R * 3);
      for (int q = 0; q < R; q++) {
        std::memcpy(&r2[(size_t)q * S * 360], rr.data(), sizeof(double) * (size_t)S * 360);
        std::memcpy(&p2[(size_t)q * S * 3], pp.data(), sizeof(double) * (size_t)S * 3);
      }
      std::vector<lslam_match_result> big((size_t)S * R);
      m->SetPipelineDepth(2);
      if (lslam_matcher_match_batch(m->handle(), S * R, r2.data(), 360, p2.data(), 1, 1, big.data()) != LSLAM_OK) return 7;
      m->SetPipelineDepth(1);
      for (int i = 0; i < S * R; i++)
        if (std::memcmp(&big[i], &one[i % S], sizeof(lslam_match_result)) != 0) { std::printf("pipelined mismatch at %d\n", i); return 7; }
      std::printf("pipelined ok\n");
    }
    lslam::GpuMatcherPool all(0, cfg, laser);
    lslam::GpuMatcherPool two(std::vector<int>{0, 0}, cfg, laser);
    std::printf("pool devices %d / %d\n", all.devices(), two.devices());
    lslam::GpuMatcherPool* pools[2] = {&all, &two};
    for (int rebuild = 0; rebuild < 2; rebuild++)
      for (auto* pool : pools) {
        pool->AddScans({base}, 360, lslam::Pose2{0, 0, 0}, rebuild != 0);
        std::vector<lslam_match_result> got = pool->MatchBatch(rr.data(), 360, pp.data(), S);
        for (int i = 0; i < S; i++)
          if (std::memcmp(&got[i], &one[i], sizeof(lslam_match_result)) != 0) { std::printf("pool mismatch at %d\n", i); return 6; }
      }
    bool threw = false;
    try { lslam::GpuMatcherPool toomany(all.devices() + 1, cfg, laser); } catch (const std::exception&) { threw = true; }
    if (!threw) return 7;
    std::printf("pool ok\n");
  }
  delete m;
  lslam_destroy(ctx);
  return nz == 71 ? 0 : 4;
}
'''


def _build(tmp_path):
    lib = build.build_library()
    src = tmp_path / "adapters_demo.cpp"
    src.write_text(SRC)
    exe = tmp_path / "adapters_demo"
    subprocess.run(["g++", "-std=c++14", "-O1", "-Wall", "-I", str(ROOT / "include"), str(src), "-o", str(exe),
                    str(lib), f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_adapters_compile_and_link(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr  # without a GPU it reports "no device" and exits 0


@pytest.mark.gpu
def test_adapters_run_on_gpu(tmp_path):
    exe = _build(tmp_path)
    r = subprocess.run([str(exe), "need-gpu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "response" in r.stdout and "nonzero cells 71" in r.stdout and "pool ok" in r.stdout and "pipelined ok" in r.stdout
