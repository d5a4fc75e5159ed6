// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction, one wave per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t* out, long long* cyc, uint32_t seed, int iters) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed ^ 0x01020304u, c = seed | 0x0C020C00u;
  double d0 = a0, d1 = a1, d2 = 1.000001, d3 = 0.5;
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { REP64(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (OP == 1) { REP64(asm volatile("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 2) { REP64(asm volatile("v_alignbyte_b32 %0, %0, %4, %5\n v_alignbyte_b32 %1, %1, %4, %5\n v_alignbyte_b32 %2, %2, %4, %5\n v_alignbyte_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 3) { REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (OP == 4) { REP64(asm volatile("v_add_f64 %0, %0, %2\n v_add_f64 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(d3));) }
    if (OP == 5) { REP64(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(d2));) }
    if (OP == 6) { REP64(asm volatile("v_cvt_i32_f64 %0, %2\n v_cvt_i32_f64 %1, %3" : "=v"(a0), "=v"(a1) : "v"(d0), "v"(d1));) }
    if (OP == 7) { REP64(asm volatile("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 8) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (OP == 9) { REP64(asm volatile("v_bfi_b32 %0, %4, %0, %5\n v_bfi_b32 %1, %4, %1, %5\n v_bfi_b32 %2, %4, %2, %5\n v_bfi_b32 %3, %4, %3, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 10) { REP64(asm volatile("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 11) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2" : "+v"(d0), "+v"(d1) : "v"(d3));) }
    if (OP == 12) { REP64(asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (OP == 13) { REP64(asm volatile("v_max_f64 %0, %0, %2\n v_max_f64 %1, %1, %2" : "+v"(d0), "+v"(d1) : "v"(d3));) }
    if (OP == 14) { REP64(asm volatile("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 15) { REP64(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
    if (OP == 16) { REP64(asm volatile("v_mqsad_pk_u16_u8 %0, %0, %2, %0\n v_mqsad_pk_u16_u8 %1, %1, %2, %1" : "+v"(d0), "+v"(d1) : "v"(b));) }
    if (OP == 17) { REP64(asm volatile("v_qsad_pk_u16_u8 %0, %0, %2, %0\n v_qsad_pk_u16_u8 %1, %1, %2, %1" : "+v"(d0), "+v"(d1) : "v"(b));) }
    if (OP == 18) { REP64(asm volatile("v_sad_u8 %0, %0, %4, %0\n v_sad_u8 %1, %1, %4, %1\n v_sad_u8 %2, %2, %4, %2\n v_sad_u8 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (OP == 19) { REP64(asm volatile("v_msad_u8 %0, %0, %4, %0\n v_msad_u8 %1, %1, %4, %1\n v_msad_u8 %2, %2, %4, %2\n v_msad_u8 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (OP == 20) { REP64(asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 21) { REP64(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");) }
    if (OP == 22) { REP64(asm volatile("v_dot4_u32_u8 %0, %0, %4, %0\n v_dot4_u32_u8 %1, %1, %4, %1\n v_dot4_u32_u8 %2, %2, %4, %2\n v_dot4_u32_u8 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (OP == 23) { REP64(asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    if (OP == 24) { REP64(asm volatile("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    if (OP == 25) { REP64(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c));) }
  }
  long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)d0 + (uint32_t)d1;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, int per_rep, int waves_per_simd) {
  uint32_t* out; long long* cyc;
  const int blocks = 1024 * waves_per_simd;
  hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
  const int iters = 64;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 64>>>(out, cyc, 12345u, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 64>>>(out, cyc, 12345u, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double insts = (double)iters * 64 * per_rep;  // per wave
  // SIMD-cycles per instruction assuming 2.4 GHz and waves_per_simd waves sharing each of 1024 SIMDs
  const double cyc_per_inst = ms * 1e-3 * 2.4e9 / (insts * waves_per_simd);
  printf("%-16s waves/SIMD %d  %.3f ms  -> %.2f cycles per wave64 instruction\n", name, waves_per_simd, ms, cyc_per_inst);
  hipFree(out); hipFree(cyc);
}

__global__ void k_sem(unsigned long long* out) {
  // v_mqsad_pk_u16_u8 D(64), S0(64), S1(32), S2(64): which operand masks, and what the windows are
  unsigned long long s0 = 0x8877665544332211ull, acc = 0x0004000300020001ull, d;
  uint32_t ref = 0x000000FFu;
  asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(s0), "v"(ref), "v"(acc));
  out[0] = d;
  ref = 0x00FF0000u;
  asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(s0), "v"(ref), "v"(acc));
  out[1] = d;
  unsigned long long s0b = 0x00000000000000FFull;  // is the mask on S0's zero bytes instead?
  ref = 0x01010101u;
  asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(s0b), "v"(ref), "v"(acc));
  out[2] = d;
  unsigned long long big = 0xFFF0FFF0FFF0FFF0ull;  // saturation or wrap of the 16-bit fields
  ref = 0x000000FFu;
  unsigned long long zero = 0;
  asm volatile("v_mqsad_pk_u16_u8 %0, %1, %2, %3" : "=&v"(d) : "v"(zero), "v"(ref), "v"(big));
  out[3] = d;
}

int main() {
  {
    unsigned long long* o; hipMalloc(&o, 64); k_sem<<<1, 1>>>(o); unsigned long long h[4]; hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
    for (int i = 0; i < 4; i++) printf("mqsad case %d: %016llx\n", i, h[i]);
  }
  for (int w : {1, 4}) {
    run<0>("v_and_b32", 4, w); run<3>("v_add_u32", 4, w); run<1>("v_perm_b32", 4, w); run<2>("v_alignbyte_b32", 4, w);
    run<9>("v_bfi_b32", 4, w); run<14>("v_and_or_b32", 4, w); run<15>("v_add3_u32", 4, w); run<12>("v_pk_add_u16", 4, w);
    run<7>("v_mad_u32_u24", 4, w); run<8>("v_mul_lo_u32", 4, w); run<10>("v_add_u32_dpp", 4, w);
    run<11>("v_lshl_add_u64", 2, w); run<4>("v_add_f64", 2, w); run<5>("v_mul_f64", 2, w); run<13>("v_max_f64", 2, w);
    run<6>("v_cvt_i32_f64", 2, w);
    run<16>("v_mqsad_pk_u16_u8", 2, w); run<17>("v_qsad_pk_u16_u8", 2, w); run<18>("v_sad_u8", 4, w); run<19>("v_msad_u8", 4, w);
    run<20>("v_permlane32_swap", 2, w); run<23>("v_permlane16_swap", 2, w); run<21>("v_cndmask_b32", 4, w); run<22>("v_dot4_u32_u8", 4, w);
    run<24>("v_mul_f32", 4, w); run<25>("v_fma_f32", 4, w);
  }
  return 0;
}
