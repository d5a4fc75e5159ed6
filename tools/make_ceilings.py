#!/usr/bin/env python3
"""profiles/ceilings.json: the per-pipe ceilings bench.py's `roofline` prices the hot kernels against, taken from the
microbenchmark outputs of a round (tools/micro/run_micro.sh -> profiles/rNN/micro_*.{txt,json}).  Nothing here is assumed:

  valu_issue_cycles   cycles one wave64 VALU instruction occupies its SIMD's issue port, by class (fast / slow / fp64 /
                      v_rcp_f64), at 4 and 8 waves per SIMD (s_memtime deltas per SIMD / instructions: micro_valu_rate.json)
  l1_hit_lookups_per_cu_clk_max   the highest TCP_TOTAL_CACHE_ACCESSES / CU / GRBM clock any ta_rate pattern reaches
                      (L1-resident table, 16-byte segments at arbitrary dword phases: micro_ta_rate_pmc.txt)
  l1_miss_lines_per_cu_clk        128-byte lines per CU per clock when every line misses the L1 and hits the L2 (4 MB table)
  ta_busy_max         TA_TA_BUSY / CU / clock of the saturating patterns (what "gather pipe 100 % busy" reads as)

    python tools/make_ceilings.py profiles/r06 > profiles/ceilings.json
"""
import json
import pathlib
import re
import sys


def main(root):
    root = pathlib.Path(root)
    valu = json.loads((root / "micro_valu_rate.json").read_text())
    cls = {"fast": "v_add_u32", "slow": "v_perm_b32", "fp64": "v_add_f64", "rcp64": "v_rcp_f64", "sgpr_operand": "v_add_u32(sgpr)",
           "dpp": "v_add_u32_dpp(rhm)"}
    out = {"valu_issue_cycles": {f"w{w}": {k: valu[f"{v}@{w}"]["simd_cyc_per_inst"] for k, v in cls.items()} for w in (1, 4, 8)},
           "valu_clock_ghz_under_load": {k: valu[f"{v}@4"]["ghz"] for k, v in cls.items()}}
    hit_max, miss_rate, ta_max = 0.0, None, 0.0
    miss = []
    for l in (root / "micro_ta_rate_pmc.txt").read_text().splitlines():
        m = re.match(r"k<(\d+), (\d+), (\d+)(?:, (\d+))?>.*lookups/CU/grbm_clk=([\d.]+).*ta_busy/CU/grbm_clk=([\d.]+)", l)
        if not m:
            continue
        lines, w, tab, mis, look, ta = int(m.group(1)), int(m.group(2)), int(m.group(3)), m.group(4), float(m.group(5)) * 8, float(m.group(6)) * 8
        ta_max = max(ta_max, ta)
        if tab <= 7:
            hit_max = max(hit_max, look)
        elif lines >= 32 and w == 4:  # every lane-quad its own line(s): all L1 misses
            miss.append(look)  # = lines per CU per clock (one lookup per line here)
    out["l1_hit_lookups_per_cu_clk_max"] = round(hit_max, 3)
    out["l1_miss_lines_per_cu_clk"] = round(sum(miss) / len(miss), 4) if miss else None
    out["ta_busy_max"] = round(ta_max, 3)
    out["source"] = f"{root}/micro_valu_rate.json, {root}/micro_ta_rate_pmc.txt (tools/micro/run_micro.sh on one MI355X)"
    out["note"] = ("GRBM_GUI_ACTIVE is reported summed over the 8 XCDs: per-clock figures divide it by 8.  An L1 hit costs the CU's "
                   "gather pipe 1 / l1_hit_lookups_per_cu_clk_max clocks, a miss 1 / l1_miss_lines_per_cu_clk")
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r06")
