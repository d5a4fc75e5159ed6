// Gather-rate microbenchmark for gfx950: TA/vector-L1 cycles per wave64 load instruction as a
// function of load width, address pattern and EXEC population.  Table of 64 KB (L1/L2 resident).
// hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip && ./ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int W>  // dwords per lane
__global__ void __launch_bounds__(64) k(const uint32_t* __restrict__ tab, const uint32_t* __restrict__ offs,
                                        unsigned long long exec_mask, uint32_t* out, int iters, int n_offs) {
  const int lane = threadIdx.x;
  uint32_t acc = 0;
  if (!((exec_mask >> lane) & 1ull)) { out[blockIdx.x * 64 + lane] = 0; return; }
  const uint32_t* o = offs + (blockIdx.x % 8) * n_offs * 64;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const uint32_t off = o[((i * 8 + u) % n_offs) * 64 + lane];  // byte offset, 16-B aligned
      if (W == 1) acc += *(const uint32_t*)((const char*)tab + off);
      if (W == 2) { uint2 v = *(const uint2*)((const char*)tab + off); acc += v.x + v.y; }
      if (W == 4) { uint4 v = *(const uint4*)((const char*)tab + off); acc += v.x + v.y + v.z + v.w; }
    }
  }
  out[blockIdx.x * 64 + lane] = acc;
}

int main() {
  const int TAB = 64 * 1024, n_offs = 64, iters = 256;
  uint32_t *tab, *offs, *out;
  hipMalloc(&tab, TAB); hipMemset(tab, 1, TAB);
  hipMalloc(&offs, 8 * n_offs * 64 * 4);
  const int blocks = 256 * 16;  // 16 waves per CU
  hipMalloc(&out, blocks * 64 * 4);
  struct Pat { const char* name; int kind; };
  const Pat pats[] = {{"contiguous 16B/lane", 0}, {"random lines", 1}, {"all lanes one address", 2}, {"pairs share a line", 3}};
  for (const Pat& pt : pats) {
    std::vector<uint32_t> h(8 * n_offs * 64);
    uint32_t rng = 12345;
    for (int b = 0; b < 8; b++)
      for (int i = 0; i < n_offs; i++)
        for (int l = 0; l < 64; l++) {
          rng = rng * 1664525u + 1013904223u;
          uint32_t off;
          if (pt.kind == 0) off = ((rng >> 8) % (TAB / 1024)) * 1024 * 0 + ((i * 1024) % TAB) + l * 16;
          else if (pt.kind == 1) off = ((rng >> 8) % (TAB / 16)) * 16;
          else if (pt.kind == 2) off = ((i * 16) % TAB);
          else off = (((i * 64 + (l / 2) * 37) * 128) % TAB) + (l & 1) * 16;
          h[(b * n_offs + i) * 64 + l] = off;
        }
    hipMemcpy(offs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (unsigned long long em : {~0ull, 0x5555555555555555ull, 0x0F0F0F0F0F0F0F0Full, 0x00000000FFFFFFFFull}) {
      for (int W : {1, 2, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto launch = [&]() {
          if (W == 1) k<1><<<blocks, 64>>>(tab, offs, em, out, iters, n_offs);
          if (W == 2) k<2><<<blocks, 64>>>(tab, offs, em, out, iters, n_offs);
          if (W == 4) k<4><<<blocks, 64>>>(tab, offs, em, out, iters, n_offs);
        };
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // per CU: 16 waves x iters x 8 gathers (+ as many coalesced offset loads)
        const double instr_per_cu = 16.0 * iters * 8;
        printf("%-24s exec %016llx dwordx%d : %.3f ms -> %.1f CU-cycles per gather instruction (incl. 1 coalesced dword load)\n",
               pt.name, em, W, ms, ms * 1e-3 * 2.4e9 / instr_per_cu);
      }
    }
  }
  return 0;
}
