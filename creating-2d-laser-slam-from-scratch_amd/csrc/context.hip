// Context / device plumbing of the C ABI (include/lslam_gpu.h).
#include "common.hpp"

namespace lslam {
thread_local std::string g_last_error;
}

namespace {
// 256 single-wave blocks: each reports WHERE it ran (XCC_ID and HW_ID: shader engine, array, CU), the shader-clock counter
// (s_memtime: one tick per shader cycle, MI355X_MICROARCH.md) and the constant 100 MHz counter (s_memrealtime).  Two samples
// around a timed region give the clock the region actually ran at -- compared per CU, so that nothing depends on whether the
// counters of different CUs / XCDs agree.  The bench line prices cycles with it instead of assuming the 2.4 GHz peak.
constexpr int kClockBlocks = 256;
__global__ void __launch_bounds__(64) k_clock_sample(unsigned long long* out) {
  unsigned long long t, r;
  uint32_t xcc, hw;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)\n s_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n s_getreg_b32 %3, hwreg(HW_REG_HW_ID)"
               : "=s"(t), "=s"(r), "=s"(xcc), "=s"(hw));
  if (threadIdx.x == 0) {
    unsigned long long* o = out + 3 * (size_t)blockIdx.x;
    o[0] = (((unsigned long long)(xcc & 0xFu)) << 16 | (hw & 0xFF00u)) + 1ull;  // se, sh, cu of this XCD (+1: 0 = no report)
    o[1] = t;
    o[2] = r;
  }
}
}  // namespace

extern "C" {

int lslam_abi_version(void) { return LSLAM_ABI_VERSION; }

int lslam_clock_sample(lslam_context* ctx, uint64_t out[768]) {
  if (!ctx || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  unsigned long long* d = nullptr;
  const size_t bytes = 3 * (size_t)kClockBlocks * sizeof(unsigned long long);
  LSLAM_HIP(ctx, hipMalloc((void**)&d, bytes));
  LSLAM_HIP(ctx, hipMemsetAsync(d, 0, bytes, ctx->stream));
  hipLaunchKernelGGL(k_clock_sample, dim3(kClockBlocks), dim3(64), 0, ctx->stream, d);
  hipError_t e = hipMemcpyAsync(out, d, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d);
  LSLAM_HIP(ctx, e);
  return LSLAM_OK;
}

int lslam_create(int device, lslam_context** out) {
  if (!out) return LSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    lslam::g_last_error =
        "lslam_create: no HIP device visible (this library has no CPU fallback; it needs an MI355X/gfx950 GPU)";
    return LSLAM_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= n) {
    lslam::g_last_error = "lslam_create: device ordinal out of range";
    return LSLAM_ERR_INVALID_ARGUMENT;
  }
  lslam_context* ctx = new lslam_context();
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&ctx->prop, device) != hipSuccess ||
      hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&ctx->d_small, 256) != hipSuccess) {
    lslam::g_last_error = "lslam_create: cannot initialise HIP device";
    delete ctx;
    return LSLAM_ERR_HIP;
  }
  *out = ctx;
  return LSLAM_OK;
}

void lslam_destroy(lslam_context* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  ctx->timer.drain();
  (void)hipStreamDestroy(ctx->stream);
  if (ctx->d_small) (void)hipFree(ctx->d_small);
  delete ctx;
}

const char* lslam_last_error(const lslam_context* ctx) {
  if (ctx) return ctx->last_error.c_str();
  return lslam::g_last_error.c_str();
}

int lslam_synchronize(lslam_context* ctx) {
  if (!ctx) return LSLAM_ERR_INVALID_ARGUMENT;
  for (auto& f : ctx->pre_sync) {
    int rc = f.second(f.first);
    if (rc) return rc;
  }
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& f : ctx->post_sync) {
    int rc = f.second(f.first);
    if (rc) return rc;
  }
  return LSLAM_OK;
}

void* lslam_stream(lslam_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

int lslam_dev_alloc(lslam_context* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(ctx, hipSetDevice(ctx->device));
  LSLAM_HIP(ctx, hipMalloc(out, bytes ? bytes : 1));
  return LSLAM_OK;
}
int lslam_dev_free(lslam_context* ctx, void* p) {
  if (!ctx) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(ctx, hipFree(p));
  return LSLAM_OK;
}
int lslam_dev_upload(lslam_context* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}
int lslam_dev_download(lslam_context* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return LSLAM_ERR_INVALID_ARGUMENT;
  LSLAM_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  LSLAM_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return LSLAM_OK;
}

int lslam_profile_enable(lslam_context* ctx, int on) {
  if (!ctx) return LSLAM_ERR_INVALID_ARGUMENT;
  ctx->timer.enabled = on != 0;
  return LSLAM_OK;
}
int lslam_profile_only(lslam_context* ctx, const char* kernel_name) {
  if (!ctx) return LSLAM_ERR_INVALID_ARGUMENT;
  ctx->timer.only = kernel_name ? kernel_name : "";
  return LSLAM_OK;
}
int lslam_profile_reset(lslam_context* ctx) {
  if (!ctx) return LSLAM_ERR_INVALID_ARGUMENT;
  (void)hipStreamSynchronize(ctx->stream);
  ctx->timer.drain();
  ctx->timer.totals.clear();
  return LSLAM_OK;
}
int lslam_profile_read(lslam_context* ctx, lslam_kernel_time* out, int capacity) {
  if (!ctx) return LSLAM_ERR_INVALID_ARGUMENT;
  (void)hipStreamSynchronize(ctx->stream);
  ctx->timer.drain();
  int n = 0;
  for (auto& kv : ctx->timer.totals) {
    if (n >= capacity) break;
    memset(&out[n], 0, sizeof out[n]);
    strncpy(out[n].name, kv.first.c_str(), sizeof(out[n].name) - 1);
    out[n].launches = kv.second.first;
    out[n].total_ms = kv.second.second;
    n++;
  }
  return n;
}

}  // extern "C"
