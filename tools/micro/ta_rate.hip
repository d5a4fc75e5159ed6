// Gather-rate microbenchmark for gfx950 (round 6): what a vector-memory instruction costs the CU's address / L1
// pipe as a function of HOW MANY DISTINCT 128-BYTE LINES its 64 lanes touch, of the load width and of where the lines
// live (a 16 KB table = L1-resident, a 4 MB table = L2-resident like the correlation grid).  Cycles come from
// s_memtime per wave, grouped by CU on the host -- nothing assumes a clock.
//
//   hipcc --offload-arch=gfx950 -O3 -o ta_rate ta_rate.hip && ./ta_rate [--json out.json]
//
// One kernel name per (lines, width, table) so that a `rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum
// GRBM_GUI_ACTIVE` pass gives, per pattern, tag lookups per instruction and lookups per CU per GRBM clock: the ceiling
// `roofline.l1_lookup_frac` of the bench line is priced against (max over the patterns).
// Addresses are computed in registers (a multiplicative hash of the iteration and the lane group): no offset loads
// dilute the instruction stream.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

struct Stamp {
  unsigned long long t0, t1, r0, r1;
  uint32_t hw_id, xcc_id;
};

constexpr int kIters = 512, kUnroll = 8;

// LINES distinct lines per instruction (64/LINES lanes share a line), W dwords per lane, table of (1 << LOG2_LINES_TAB) lines
// MIS = 1: the lane's W dwords start at a dword-aligned pseudo-random offset inside its line (a 16-byte segment may then
// straddle the 64-byte halves of the line, like the row loads of k_resp_rows); MIS = 0: at a multiple of its own size.
template <int LINES, int W, int LOG2_TAB, int MIS = 0>
__global__ void __launch_bounds__(64) k(const uint32_t* __restrict__ tab, uint32_t* out, Stamp* st) {
  const uint32_t lane = threadIdx.x;
  constexpr uint32_t per = 64 / LINES;                   // lanes per line
  const uint32_t grp = lane / per, sub = lane % per;     // line group, position inside the line
  const uint32_t in_line = MIS ? ((lane * 2654435761u) >> 25) % (uint32_t)(33 - W) * 4u  // dword-aligned, inside the line
                               : (sub * (W * 4)) & 127u;  // W-dword slots inside the 128-byte line (wraps: 16 B x 8 = a line)
  uint32_t acc = 0;
  uint32_t h = (blockIdx.x * 2654435761u) ^ (grp * 40503u);
  unsigned long long t0, t1, r0, r1;
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
  for (int i = 0; i < kIters; i++) {
    uint32_t v[kUnroll][4];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      h = h * 1664525u + 1013904223u;
      // distinct groups -> distinct lines: the group index occupies the low bits of the line number
      const uint32_t line = (((h >> 8) << 6) | grp) & ((1u << LOG2_TAB) - 1u);
      const char* p = (const char*)tab + ((size_t)line << 7) + in_line;
      if (W == 1) v[u][0] = *(const uint32_t*)p;
      if (W == 2) { uint2 t = *(const uint2*)p; v[u][0] = t.x; v[u][1] = t.y; }
      if (W == 4) { uint4 t = *(const uint4*)p; v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
    }
#pragma unroll
    for (int u = 0; u < kUnroll; u++)
#pragma unroll
      for (int q = 0; q < W; q++) acc += v[u][q];
  }
  asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) {
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    st[blockIdx.x] = Stamp{t0, t1, r0, r1, hw, xcc};
  }
}

struct Row {
  std::string name;
  int lines, width, log2_tab, waves_per_cu;
  double cu_cyc_per_inst, lines_per_cu_clk, ghz, wall_ms;
};
std::vector<Row> rows;
uint32_t *g_tab, *g_out;
Stamp* g_st;

template <int LINES, int W, int LOG2_TAB, int MIS = 0>
void run(int waves_per_cu) {
  const int blocks = 256 * waves_per_cu;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<LINES, W, LOG2_TAB, MIS><<<blocks, 64>>>(g_tab, g_out, g_st);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  k<LINES, W, LOG2_TAB, MIS><<<blocks, 64>>>(g_tab, g_out, g_st);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<Stamp> h(blocks);
  (void)hipMemcpy(h.data(), g_st, (size_t)blocks * sizeof(Stamp), hipMemcpyDeviceToHost);
  struct G { unsigned long long lo = ~0ull, hi = 0; int n = 0; double clk = 0; };
  std::map<uint32_t, G> cus;
  for (const Stamp& s : h) {
    const uint32_t key = ((s.xcc_id & 0xFu) << 16) | (s.hw_id & 0xFF00u);  // se, sh, cu
    G& g = cus[key];
    g.lo = std::min(g.lo, s.t0);
    g.hi = std::max(g.hi, s.t1);
    g.n++;
    g.clk += (double)(s.t1 - s.t0) / ((double)(s.r1 - s.r0) / 100e6) * 1e-9;
  }
  std::vector<double> cyc, ghz;
  for (auto& kv : cus) {
    const G& g = kv.second;
    if (g.n != waves_per_cu) continue;
    cyc.push_back((double)(g.hi - g.lo) / ((double)g.n * kIters * kUnroll));
    ghz.push_back(g.clk / g.n);
  }
  auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  char name[64];
  snprintf(name, sizeof name, "lines%-2d dwordx%d %s%s", LINES, W, LOG2_TAB <= 7 ? "L1(16KB)" : "L2(4MB)", MIS ? " misaligned" : "");
  Row r{name, LINES, W, LOG2_TAB, waves_per_cu, med(cyc), 0.0, med(ghz), (double)ms};
  r.lines_per_cu_clk = r.cu_cyc_per_inst > 0 ? LINES / r.cu_cyc_per_inst : 0.0;
  rows.push_back(r);
  printf("%-26s waves/CU %2d : %6.2f CU-cycles per gather instruction = %.3f lines per CU per clock   clock %.3f GHz  wall %.3f ms (%d CUs)\n",
         name, waves_per_cu, r.cu_cyc_per_inst, r.lines_per_cu_clk, r.ghz, ms, (int)cyc.size());
  fflush(stdout);
}

template <int W, int LOG2_TAB>
void sweep(int waves_per_cu) {
  run<1, W, LOG2_TAB>(waves_per_cu); run<2, W, LOG2_TAB>(waves_per_cu); run<4, W, LOG2_TAB>(waves_per_cu);
  run<8, W, LOG2_TAB>(waves_per_cu); run<16, W, LOG2_TAB>(waves_per_cu); run<32, W, LOG2_TAB>(waves_per_cu);
  run<64, W, LOG2_TAB>(waves_per_cu);
}

int main(int argc, char** argv) {
  const char* json = nullptr;
  for (int i = 1; i + 1 < argc; i++)
    if (!strcmp(argv[i], "--json")) json = argv[i + 1];
  const size_t tab_bytes = (size_t)128 << 15;  // 4 MB
  (void)hipMalloc(&g_tab, tab_bytes);
  (void)hipMemset(g_tab, 1, tab_bytes);
  (void)hipMalloc(&g_out, (size_t)256 * 32 * 64 * 4);
  (void)hipMalloc(&g_st, (size_t)256 * 32 * sizeof(Stamp));
  for (int wpc : {16, 32}) {
    sweep<1, 7>(wpc); sweep<4, 7>(wpc);    // 128 lines = 16 KB: L1-resident
    sweep<1, 15>(wpc); sweep<2, 15>(wpc); sweep<4, 15>(wpc);  // 32768 lines = 4 MB: L2-resident (the grid's size)
    // 16-byte segments at arbitrary dword phases (the row loads of k_resp_rows), L1- and L2-resident
    run<4, 4, 7, 1>(wpc); run<8, 4, 7, 1>(wpc); run<16, 4, 7, 1>(wpc); run<32, 4, 7, 1>(wpc); run<64, 4, 7, 1>(wpc);
    run<16, 4, 15, 1>(wpc); run<32, 4, 15, 1>(wpc); run<64, 4, 15, 1>(wpc);
    run<16, 2, 7, 1>(wpc); run<32, 2, 7, 1>(wpc); run<32, 1, 7, 1>(wpc); run<64, 1, 7, 1>(wpc);
  }
  if (json) {
    FILE* f = fopen(json, "w");
    fprintf(f, "[\n");
    for (size_t i = 0; i < rows.size(); i++)
      fprintf(f, "  {\"name\": \"%s\", \"lines\": %d, \"dwords\": %d, \"table_lines_log2\": %d, \"waves_per_cu\": %d, \"cu_cyc_per_inst\": %.4f, "
                 "\"lines_per_cu_clk\": %.4f, \"ghz\": %.4f, \"wall_ms\": %.4f}%s\n",
              rows[i].name.c_str(), rows[i].lines, rows[i].width, rows[i].log2_tab, rows[i].waves_per_cu, rows[i].cu_cyc_per_inst,
              rows[i].lines_per_cu_clk, rows[i].ghz, rows[i].wall_ms, i + 1 < rows.size() ? "," : "");
    fprintf(f, "]\n");
    fclose(f);
  }
  return 0;
}
