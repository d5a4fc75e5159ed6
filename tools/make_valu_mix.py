#!/usr/bin/env python3
"""profiles/valu_mix.json: the mix-weighted VALU-issue floor (tools/valu_mix_floor.py) of the two throughput kernels of a
batched match, from the PMC summary of `python bench.py` (tools/pmc_summary.py: SQ_INSTS_VALU, SQ_WAVES, GRBM_GUI_ACTIVE per
launch) and the measured issue cycles per instruction class (profiles/ceilings.json <- tools/micro/valu_rate.hip).

    python tools/make_valu_mix.py profiles/r06/pmc_per_launch.json 4096 > profiles/valu_mix.json

Phase line ranges and the per-phase dynamic shares of k_resp_rows are those of profiles/r05/inst_budget_resp_rows.md (phase A
9 iterations x 120, phase B 110 + 5.15 x 176, epilogue 283, parked ~50, prologue 51); the total is scaled to the PMC's
instruction count, so only the SHARES come from the static budget.  Line ranges follow resp_rows_wave / resp_tile3_wave in
csrc/scan_matcher.hip and are re-derived here from marker lines, so that edits above them do not shift the phases.
"""
import json
import pathlib
import re
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parent.parent
SRC = ROOT / "creating-2d-laser-slam-from-scratch_amd/csrc/scan_matcher.hip"


def line_of(text, needle, after=0):
    for i, l in enumerate(text, 1):
        if i > after and needle in l:
            return i
    raise SystemExit(f"marker not found: {needle}")


def main(pmc_path, scans):
    pmc = json.loads(pathlib.Path(pmc_path).read_text())
    src = SRC.read_text().split("\n")
    f0 = line_of(src, "__device__ __forceinline__ void resp_rows_wave(")
    m = {k: line_of(src, v, f0) for k, v in {
        "drain": "auto drain_as = [&]", "drain_end": "auto drain = [&](int head, int cnt)", "cell": "struct Cell {",
        "enqueue": "auto enqueue = [&]", "estimate": "auto estimate = [&]", "point": "auto point_f = [&]",
        "exact": "auto exact_cell = [&]", "loop": "if constexpr (EST) {", "epi": "// Reduce over the wave.",
        "end": "// The row kernel proper"}.items()}
    phases = (f"B_drain:{m['drain']}-{m['drain_end']},A_cell:{m['cell']}-{m['enqueue'] - 1},A_emit:{m['enqueue']}-{m['estimate'] - 1},"
              f"A_estimate:{m['estimate']}-{m['point'] - 1},A_point:{m['point']}-{m['exact'] - 1},parked:{m['exact']}-{m['loop'] - 1},"
              f"A_loop:{m['loop']}-{m['epi'] - 1},epilogue:{m['epi']}-{m['end']}")
    out = {}
    jobs = [("resp_rows_coarse", "k_resp_rows<3, 11, true", "k_resp_rowsILi3ELi11ELb1ELb0ELb0E", phases,
             "A_cell+A_emit+A_estimate+A_point+A_loop:1080,B_drain:1016,epilogue:283,parked:50,other:51")]
    t0 = line_of(src, "__device__ __forceinline__ void resp_tile3_wave(")
    t1 = line_of(src, "k_resp_tile3(const uint4* __restrict__ tiles", t0)
    jobs.append(("resp_tile_fine", "k_resp_tile3<3>", "k_resp_tile3ILi3E", f"all:{t0}-{t1}", "all+other:1"))
    for short, pmc_prefix, mangled, ph, dyn in jobs:
        c = next((v for k, v in pmc.items() if k.startswith(pmc_prefix)), None)
        if not c or not c.get("GRBM_GUI_ACTIVE") or not c.get("SQ_WAVES"):
            continue
        cmd = [sys.executable, str(ROOT / "tools/valu_mix_floor.py"), "--kernel", mangled, "--phases", ph, "--dyn", dyn,
               "--waves", str(int(c["SQ_WAVES"])), "--measured-cycles", str(c["GRBM_GUI_ACTIVE"] / 8.0),
               "--valu-per-wave", str(c["SQ_INSTS_VALU"] / c["SQ_WAVES"]), "--micro", str(ROOT / "profiles/r06/micro_valu_rate.json")]
        r = json.loads(subprocess.run(cmd, check=True, capture_output=True, text=True).stdout)
        out[short] = {"scans_per_launch": scans, "floor_cycles_per_simd_w8": r["floor_cycles_per_simd_w8"],
                      "floor_cycles_per_simd_w4": r["floor_cycles_per_simd_w4"], "valu_mix_frac": r["valu_mix_frac_w8"],
                      "valu_mix_frac_w4": r["valu_mix_frac_w4"], "avg_cycles_per_valu_inst_w8": r["avg_cycles_per_valu_inst_w8"],
                      "avg_cycles_per_valu_inst_w4": r["avg_cycles_per_valu_inst_w4"], "measured_cycles_per_launch": r["measured_cycles_per_launch"],
                      "valu_insts_per_wave": r["valu_insts_per_wave_pmc"], "dynamic": r["dynamic"], "static_by_phase": r["static_by_phase"],
                      "unknown_opcodes_priced_fast": r["unknown_opcodes_priced_fast"], "pmc_source": str(pmc_path)}
    out["_meta"] = {"tool": "tools/valu_mix_floor.py", "issue_cycles": "profiles/ceilings.json (tools/micro/valu_rate.hip, cycle counters)",
                    "definition": "valu_mix_frac = sum over phases of dynamic VALU instructions x measured issue cycles of their class "
                                  "(at 8 waves/SIMD, the ports' best rate) x waves / 1024 SIMDs, divided by the launch's measured cycles "
                                  "(GRBM_GUI_ACTIVE / 8 of the PMC pass); _w4 uses the rates measured at the kernel's own occupancy"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4096)
