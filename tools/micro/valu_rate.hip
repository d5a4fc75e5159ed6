// VALU issue-rate microbenchmark for gfx950 (round 6: cycles come from the hardware, not from an assumed clock).
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate [--json out.json]
//
// Every wave runs a straight-line stream of ONE instruction (4 independent dependency chains, 16 384 instructions) and
// stamps s_memtime (shader-clock ticks: MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units") and s_memrealtime (the
// constant 100 MHz reference) in front of and behind it, plus the SIMD it ran on (HW_ID / XCC_ID).  The host groups the
// waves by SIMD and reports, for SIMDs that held exactly W = 1, 4, 8 waves:
//   cyc/inst (SIMD)  = (last wave's end - first wave's start on that SIMD) / (W x instructions per wave)
//                      -- what one wave64 instruction costs the SIMD's issue port when W waves compete for it;
//   cyc/inst (wave)  = one wave's own elapsed ticks / its instructions (= W x the SIMD figure when the port is the limit);
//   clock            = shader ticks / (realtime ticks / 100 MHz) -- the clock the stream actually ran at.
// Nothing here assumes 2.4 GHz.  Run under `rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES` the kernel
// names (k<OP>) give the counters' ratio per instruction kind (tools/micro/README in profiles/r06/micro_valu_rate.txt).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

struct Stamp {
  unsigned long long t0, t1, r0, r1;
  uint32_t hw_id, xcc_id;
};

constexpr int kIters = 64;  // x 64 repeats x per_rep instructions per wave

#define I4(OPSTR) \
  REP64(asm volatile(OPSTR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c), "s"(sb) : "vcc", "s20", "s21");)
#define D2(OPSTR) REP64(asm volatile(OPSTR : "+v"(d0), "+v"(d1) : "v"(d2), "v"(d3));)

template <int OP>
__global__ void __launch_bounds__(64) k(uint32_t* out, Stamp* st, uint32_t seed) {
  uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, b = seed ^ 0x01020304u, c = seed | 0x0C020C00u;
  uint32_t sb = __builtin_amdgcn_readfirstlane(seed * 9u);
  double d0 = a0, d1 = a1, d2 = 1.000001, d3 = 0.5;
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_mov_b64 vcc, -1\n s_mov_b64 s[20:21], -1" ::: "vcc", "s20", "s21");
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  for (int i = 0; i < kIters; i++) {
    // ---- 32-bit VOP2 / VOP1 ----
    if (OP == 0) { I4("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5") }
    if (OP == 1) { I4("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4") }
    if (OP == 2) { I4("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4") }
    if (OP == 3) { I4("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4") }
    if (OP == 4) { I4("v_add_u32 %0, %6, %0\n v_add_u32 %1, %6, %1\n v_add_u32 %2, %6, %2\n v_add_u32 %3, %6, %3") }  // SGPR operand
    if (OP == 5) { I4("v_lshlrev_b32 %0, 1, %0\n v_lshlrev_b32 %1, 1, %1\n v_lshlrev_b32 %2, 1, %2\n v_lshlrev_b32 %3, 1, %3") }
    if (OP == 6) { I4("v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3") }
    if (OP == 7) { I4("v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3") }
    if (OP == 8) { I4("v_mul_i32_i24 %0, %0, %4\n v_mul_i32_i24 %1, %1, %4\n v_mul_i32_i24 %2, %2, %4\n v_mul_i32_i24 %3, %3, %4") }
    if (OP == 9) { I4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc") }
    if (OP == 10) { I4("v_cmp_lt_f32 vcc, %0, %4\n v_cmp_lt_f32 vcc, %1, %4\n v_cmp_lt_f32 vcc, %2, %4\n v_cmp_lt_f32 vcc, %3, %4") }
    if (OP == 11) { I4("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4") }
    // ---- DPP / SDWA forms ----
    if (OP == 12) { I4("v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
    if (OP == 13) { I4("v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xf") }
    if (OP == 14) { I4("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_mirror row_mask:0xf bank_mask:0xf") }
    if (OP == 15) { I4("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1") }
    // ---- 32-bit VOP3 (three operands / 64-bit encoding) ----
    if (OP == 16) { I4("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5") }
    if (OP == 17) { I4("v_alignbit_b32 %0, %0, %4, %5\n v_alignbit_b32 %1, %1, %4, %5\n v_alignbit_b32 %2, %2, %4, %5\n v_alignbit_b32 %3, %3, %4, %5") }
    if (OP == 18) { I4("v_alignbyte_b32 %0, %0, %4, %5\n v_alignbyte_b32 %1, %1, %4, %5\n v_alignbyte_b32 %2, %2, %4, %5\n v_alignbyte_b32 %3, %3, %4, %5") }
    if (OP == 19) { I4("v_bfe_i32 %0, %0, %4, 1\n v_bfe_i32 %1, %1, %4, 1\n v_bfe_i32 %2, %2, %4, 1\n v_bfe_i32 %3, %3, %4, 1") }
    if (OP == 20) { I4("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5") }
    if (OP == 21) { I4("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5") }
    if (OP == 22) { I4("v_lshl_add_u32 %0, %0, 1, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_lshl_add_u32 %2, %2, 1, %4\n v_lshl_add_u32 %3, %3, 1, %4") }
    if (OP == 23) { I4("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5") }
    if (OP == 24) { I4("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4") }
    if (OP == 25) { I4("v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_lo_u32_b32 %1, %4, %1\n v_mbcnt_lo_u32_b32 %2, %4, %2\n v_mbcnt_lo_u32_b32 %3, %4, %3") }
    if (OP == 26) { I4("v_cmp_lt_u32_e64 s[20:21], %0, %4\n v_cmp_lt_u32_e64 s[20:21], %1, %4\n v_cmp_lt_u32_e64 s[20:21], %2, %4\n v_cmp_lt_u32_e64 s[20:21], %3, %4" ) }
    if (OP == 27) { I4("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s21, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s21, %3") }
    if (OP == 28) { I4("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4") }
    if (OP == 29) { I4("v_fma_f32 %0, %6, %0, %5\n v_fma_f32 %1, %6, %1, %5\n v_fma_f32 %2, %6, %2, %5\n v_fma_f32 %3, %6, %3, %5") }  // SGPR operand (phase A's form)
    // ---- forms the hot kernels use with scalar / constant operands, and the rest of their opcode list ----
    if (OP == 40) { I4("v_mov_b32 %0, 7\n v_mov_b32 %1, 7\n v_mov_b32 %2, 7\n v_mov_b32 %3, 7") }
    if (OP == 41) { I4("v_mov_b32 %0, %6\n v_mov_b32 %1, %6\n v_mov_b32 %2, %6\n v_mov_b32 %3, %6") }
    if (OP == 42) { I4("v_and_b32 %0, 0xffff, %0\n v_and_b32 %1, 0xffff, %1\n v_and_b32 %2, 0xffff, %2\n v_and_b32 %3, 0xffff, %3") }
    if (OP == 43) { I4("v_add_u32 %0, 4, %0\n v_add_u32 %1, 4, %1\n v_add_u32 %2, 4, %2\n v_add_u32 %3, 4, %3") }
    if (OP == 44) { I4("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4") }
    if (OP == 45) { I4("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4") }
    if (OP == 46) { I4("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4") }
    if (OP == 47) { I4("v_lshrrev_b32 %0, 1, %0\n v_lshrrev_b32 %1, 1, %1\n v_lshrrev_b32 %2, 1, %2\n v_lshrrev_b32 %3, 1, %3") }
    if (OP == 48) { I4("v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3") }
    if (OP == 49) { I4("v_lshlrev_b32 %0, %4, %0\n v_lshlrev_b32 %1, %4, %1\n v_lshlrev_b32 %2, %4, %2\n v_lshlrev_b32 %3, %4, %3") }
    if (OP == 50) { I4("v_ashrrev_i32 %0, 1, %0\n v_ashrrev_i32 %1, 1, %1\n v_ashrrev_i32 %2, 1, %2\n v_ashrrev_i32 %3, 1, %3") }
    if (OP == 51) { I4("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc") }
    if (OP == 52) { I4("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]") }
    if (OP == 53) { I4("v_cmp_le_u32 vcc, %0, %4\n v_cmp_le_u32 vcc, %1, %4\n v_cmp_le_u32 vcc, %2, %4\n v_cmp_le_u32 vcc, %3, %4") }
    if (OP == 54) { I4("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4") }
    if (OP == 55) { I4("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4") }
    if (OP == 56) { I4("v_xad_u32 %0, %0, %4, %5\n v_xad_u32 %1, %1, %4, %5\n v_xad_u32 %2, %2, %4, %5\n v_xad_u32 %3, %3, %4, %5") }
    if (OP == 57) { I4("v_or3_b32 %0, %0, %4, %5\n v_or3_b32 %1, %1, %4, %5\n v_or3_b32 %2, %2, %4, %5\n v_or3_b32 %3, %3, %4, %5") }
    if (OP == 58) { I4("v_bfe_u32 %0, %0, %4, 1\n v_bfe_u32 %1, %1, %4, 1\n v_bfe_u32 %2, %2, %4, 1\n v_bfe_u32 %3, %3, %4, 1") }
    if (OP == 59) { I4("v_max_i32 %0, %0, %4\n v_max_i32 %1, %1, %4\n v_max_i32 %2, %2, %4\n v_max_i32 %3, %3, %4") }
    if (OP == 60) { I4("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3") }
    if (OP == 61) { I4("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3") }
    if (OP == 62) { I4("v_sub_f32 %0, %0, %4\n v_sub_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_sub_f32 %3, %3, %4") }
    if (OP == 63) { I4("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4") }
    if (OP == 64) { I4("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5") }
    if (OP == 65) { I4("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s20, %2, 7\n v_readlane_b32 s21, %3, 9") }
    if (OP == 66) { I4("v_writelane_b32 %0, %6, 3\n v_writelane_b32 %1, %6, 5\n v_writelane_b32 %2, %6, 7\n v_writelane_b32 %3, %6, 9") }
    if (OP == 67) { I4("v_lshlrev_b32_sdwa %0, %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_lshlrev_b32_sdwa %1, %4, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_lshlrev_b32_sdwa %2, %4, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_lshlrev_b32_sdwa %3, %4, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD") }
    if (OP == 68) { REP64(asm volatile("v_mad_u64_u32 %0, s[20:21], %2, %3, %0\n v_mad_u64_u32 %1, s[20:21], %2, %3, %1" : "+v"(d0), "+v"(d1) : "v"(a0), "v"(a1));) }
    if (OP == 69) { I4("v_and_b32 %0, 3, %0\n v_and_b32 %1, 3, %1\n v_and_b32 %2, 3, %2\n v_and_b32 %3, 3, %3") }
    if (OP == 70) { I4("v_mul_f32 %0, %6, %0\n v_mul_f32 %1, %6, %1\n v_mul_f32 %2, %6, %2\n v_mul_f32 %3, %6, %3") }
    if (OP == 71) { I4("v_cndmask_b32 %0, %0, %4, vcc\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_cndmask_b32 %2, %2, %4, vcc\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %0, %0, %4\n v_cndmask_b32 %3, %3, %4, vcc\n v_add_u32 %3, %3, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4") }
    if (OP == 72) { I4("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc") }
    if (OP == 73) { I4("v_cndmask_b32 %0, 0, %0, vcc\n v_cndmask_b32 %1, 0, %1, vcc\n v_cndmask_b32 %2, 0, %2, vcc\n v_cndmask_b32 %3, 0, %3, vcc") }
    if (OP == 74) { I4("v_cndmask_b32 %0, %1, %4, vcc\n v_cndmask_b32 %1, %2, %4, vcc\n v_cndmask_b32 %2, %3, %4, vcc\n v_cndmask_b32 %3, %0, %4, vcc") }
    if (OP == 75) { I4("v_addc_co_u32 %0, vcc, %0, %4, vcc\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_addc_co_u32 %2, vcc, %2, %4, vcc\n v_addc_co_u32 %3, vcc, %3, %4, vcc") }
    if (OP == 76) { I4("v_cmp_lt_u32 vcc, %0, %4\n v_cndmask_b32 %0, %0, %5, vcc\n v_cmp_lt_u32 vcc, %1, %4\n v_cndmask_b32 %1, %1, %5, vcc\n v_cmp_lt_u32 vcc, %2, %4\n v_cndmask_b32 %2, %2, %5, vcc\n v_cmp_lt_u32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc") }
    if (OP == 77) { I4("v_cmp_lt_u32_e64 s[20:21], %0, %4\n v_cndmask_b32_e64 %0, %0, %5, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %1, %4\n v_cndmask_b32_e64 %1, %1, %5, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %2, %4\n v_cndmask_b32_e64 %2, %2, %5, s[20:21]\n v_cmp_lt_u32_e64 s[20:21], %3, %4\n v_cndmask_b32_e64 %3, %3, %5, s[20:21]") }
    if (OP == 78) { I4("v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %4\n v_add_co_u32 %2, vcc, %2, %4\n v_add_co_u32 %3, vcc, %3, %4") }
    if (OP == 79) { I4("v_min_u32 %0, %0, %4\n v_min_u32 %1, %1, %4\n v_min_u32 %2, %2, %4\n v_min_u32 %3, %3, %4") }
    if (OP == 80) { I4("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4") }
    // ---- 64-bit ----
    if (OP == 30) { D2("v_add_f64 %0, %0, %3\n v_add_f64 %1, %1, %3") }
    if (OP == 31) { D2("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2") }
    if (OP == 32) { D2("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3") }
    if (OP == 33) { D2("v_max_f64 %0, %0, %3\n v_max_f64 %1, %1, %3") }
    if (OP == 34) { REP64(asm volatile("v_cvt_f32_f64 %0, %2\n v_cvt_f32_f64 %1, %3" : "=v"(a0), "=v"(a1) : "v"(d0), "v"(d1));) }
    if (OP == 35) { REP64(asm volatile("v_cvt_i32_f64 %0, %2\n v_cvt_i32_f64 %1, %3" : "=v"(a0), "=v"(a1) : "v"(d0), "v"(d1));) }
    if (OP == 36) { D2("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1") }
    if (OP == 37) { D2("v_lshl_add_u64 %0, %0, 0, %3\n v_lshl_add_u64 %1, %1, 0, %3") }
    if (OP == 38) { REP64(asm volatile("v_cvt_f64_i32 %0, %2\n v_cvt_f64_i32 %1, %3" : "=v"(d0), "=v"(d1) : "v"(a0), "v"(a1));) }
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)d0 + (uint32_t)d1;
  if (threadIdx.x == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    st[blockIdx.x] = Stamp{t0, t1, r0, r1, hw, xcc};
  }
}

struct Row {
  std::string name;
  int w;
  double simd_cyc, wave_cyc, ghz, wall_ms;
  int simds_used;
};
std::vector<Row> rows;

template <int OP>
void run(const char* name, int per_rep, int waves_per_simd) {
  uint32_t* out;
  Stamp* st;
  const int blocks = 1024 * waves_per_simd;
  hipMalloc(&out, (size_t)blocks * 64 * 4);
  hipMalloc(&st, (size_t)blocks * sizeof(Stamp));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<OP><<<blocks, 64>>>(out, st, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<OP><<<blocks, 64>>>(out, st, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> h(blocks);
  hipMemcpy(h.data(), st, (size_t)blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
  const double insts = (double)kIters * 64 * per_rep;  // per wave
  // group by SIMD: XCC_ID[3:0], HW_ID: wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
  struct G { unsigned long long lo = ~0ull, hi = 0; int n = 0; double wave_sum = 0, clk_sum = 0; };
  std::map<uint32_t, G> groups;
  for (const Stamp& s : h) {
    const uint32_t key = ((s.xcc_id & 0xFu) << 16) | (s.hw_id & 0xFF30u);  // se, sh, cu, simd
    G& g = groups[key];
    g.lo = std::min(g.lo, s.t0);
    g.hi = std::max(g.hi, s.t1);
    g.n++;
    g.wave_sum += (double)(s.t1 - s.t0);
    g.clk_sum += (double)(s.t1 - s.t0) / ((double)(s.r1 - s.r0) / 100e6) * 1e-9;
  }
  std::vector<double> simd_cyc, wave_cyc, ghz;
  for (auto& kv : groups) {
    const G& g = kv.second;
    if (g.n != waves_per_simd) continue;  // only SIMDs that held exactly W waves, all co-resident
    simd_cyc.push_back((double)(g.hi - g.lo) / (insts * g.n));
    wave_cyc.push_back(g.wave_sum / g.n / insts);
    ghz.push_back(g.clk_sum / g.n);
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  Row r{name, waves_per_simd, med(simd_cyc), med(wave_cyc), med(ghz), (double)ms, (int)simd_cyc.size()};
  rows.push_back(r);
  printf("%-22s W=%d  SIMD %.2f cyc/inst   wave %.2f cyc/inst   clock %.3f GHz   wall %.3f ms   (%d of %d SIMDs held exactly W)\n",
         name, waves_per_simd, r.simd_cyc, r.wave_cyc, r.ghz, ms, r.simds_used, (int)groups.size());
  fflush(stdout);
  hipFree(out);
  hipFree(st);
}

int main(int argc, char** argv) {
  const char* json = nullptr;
  for (int i = 1; i + 1 < argc; i++)
    if (!strcmp(argv[i], "--json")) json = argv[i + 1];
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("# device %s  CUs %d  clockRate(attr) %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  for (int w : {1, 4, 8}) {
    run<0>("v_fma_f32", 4, w); run<29>("v_fma_f32(sgpr)", 4, w); run<1>("v_mul_f32", 4, w); run<2>("v_add_u32", 4, w);
    run<4>("v_add_u32(sgpr)", 4, w); run<3>("v_and_b32", 4, w); run<5>("v_lshlrev_b32", 4, w); run<11>("v_mov_b32", 4, w);
    run<6>("v_cvt_i32_f32", 4, w); run<7>("v_rndne_f32", 4, w); run<8>("v_mul_i32_i24", 4, w); run<9>("v_cndmask_b32", 4, w);
    run<10>("v_cmp_lt_f32(vcc)", 4, w); run<26>("v_cmp_lt_u32_e64", 4, w);
    run<12>("v_add_u32_dpp(quad)", 4, w); run<13>("v_add_u32_dpp(rhm)", 4, w); run<14>("v_mov_b32_dpp", 4, w);
    run<15>("v_add_u32_sdwa", 4, w);
    run<16>("v_perm_b32", 4, w); run<17>("v_alignbit_b32", 4, w); run<18>("v_alignbyte_b32", 4, w); run<19>("v_bfe_i32", 4, w);
    run<20>("v_and_or_b32", 4, w); run<21>("v_add3_u32", 4, w); run<22>("v_lshl_add_u32", 4, w); run<23>("v_mad_u32_u24", 4, w);
    run<24>("v_mul_lo_u32", 4, w); run<25>("v_mbcnt_lo_u32_b32", 4, w); run<27>("v_readfirstlane_b32", 4, w);
    run<28>("v_pk_add_u16", 4, w);
    run<30>("v_add_f64", 2, w); run<31>("v_mul_f64", 2, w); run<32>("v_fma_f64", 2, w); run<33>("v_max_f64", 2, w);
    run<34>("v_cvt_f32_f64", 2, w); run<35>("v_cvt_i32_f64", 2, w); run<38>("v_cvt_f64_i32", 2, w); run<36>("v_rcp_f64", 2, w);
    run<37>("v_lshl_add_u64", 2, w);
    run<40>("v_mov_b32(const)", 4, w); run<41>("v_mov_b32(sgpr)", 4, w); run<42>("v_and_b32(literal)", 4, w); run<43>("v_add_u32(const)", 4, w); run<44>("v_sub_u32", 4, w); run<45>("v_or_b32", 4, w); run<46>("v_xor_b32", 4, w); run<47>("v_lshrrev_b32(const)", 4, w); run<48>("v_lshrrev_b32(vgpr)", 4, w); run<49>("v_lshlrev_b32(vgpr)", 4, w); run<50>("v_ashrrev_i32(const)", 4, w); run<51>("v_cndmask_b32(vcc=-1)", 4, w); run<52>("v_cndmask_b32_e64", 4, w); run<53>("v_cmp_le_u32(vcc)", 4, w); run<54>("v_mul_hi_u32", 4, w); run<55>("v_lshl_or_b32", 4, w); run<56>("v_xad_u32", 4, w); run<57>("v_or3_b32", 4, w); run<58>("v_bfe_u32", 4, w); run<59>("v_max_i32", 4, w); run<60>("v_cvt_u32_f32", 4, w); run<61>("v_cvt_f32_u32", 4, w); run<62>("v_sub_f32", 4, w); run<63>("v_add_f32", 4, w); run<64>("v_fmac_f32", 4, w); run<65>("v_readlane_b32", 4, w); run<66>("v_writelane_b32", 4, w); run<67>("v_lshlrev_b32_sdwa", 4, w); run<68>("v_mad_u64_u32", 2, w); run<69>("v_and_b32(const)", 4, w); run<70>("v_mul_f32(sgpr)", 4, w);
    run<71>("cndmask_e32+3add", 16, w); run<72>("v_cndmask_b32_e64(vcc)", 4, w); run<73>("v_cndmask_b32(const0)", 4, w); run<74>("v_cndmask_b32(dst!=src)", 4, w); run<75>("v_addc_co_u32", 4, w); run<76>("v_cmp+v_cndmask(vcc)", 8, w); run<77>("v_cmp_e64+cndmask_e64", 8, w); run<78>("v_add_co_u32(vcc)", 4, w); run<79>("v_min_u32", 4, w); run<80>("v_mul_u32_u24", 4, w);
  }
  if (json) {
    FILE* f = fopen(json, "w");
    fprintf(f, "{\n");
    for (size_t i = 0; i < rows.size(); i++)
      fprintf(f, "  \"%s@%d\": {\"simd_cyc_per_inst\": %.4f, \"wave_cyc_per_inst\": %.4f, \"ghz\": %.4f, \"wall_ms\": %.4f, \"simds\": %d}%s\n",
              rows[i].name.c_str(), rows[i].w, rows[i].simd_cyc, rows[i].wave_cyc, rows[i].ghz, rows[i].wall_ms, rows[i].simds_used,
              i + 1 < rows.size() ? "," : "");
    fprintf(f, "}\n");
    fclose(f);
  }
  return 0;
}
