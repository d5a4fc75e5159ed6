// Context / device plumbing of the C ABI (include/lslam_gpu.h).
#include "common.hpp"

namespace lslam {
thread_local std::string g_last_error;
}

extern "C" {

int lslam_abi_version(void) { return LSLAM_ABI_VERSION; }

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
