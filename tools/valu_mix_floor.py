#!/usr/bin/env python3
"""Mix-weighted VALU-issue floor of a gfx950 kernel.

The guide's "2 cycles per wave64 VALU instruction" holds for a few opcodes only.  tools/micro/valu_rate.hip (cycle counter,
no assumed clock; profiles/r06/micro_valu_rate.txt) finds THREE issue classes on MI355X:
  fast  2.22 cycles (8 waves/SIMD; 2.48 at 4): v_fma/fmac/mul/add/sub_f32, v_add/sub/subrev_u32, v_and/or/xor_b32, v_mov_b32,
        v_lshrrev_b32, v_ashrrev_i32, an isolated v_cndmask_b32_e32 -- with VGPR / constant sources only
  slow  4.11 cycles (4.22 at 4): ANY instruction with an SGPR source, every VOP3-only opcode (v_perm_b32, v_bfe, v_alignbit,
        v_mad_*, v_add3, v_lshl_add, v_and_or, v_mul_lo/hi, ...), DPP and SDWA forms, conversions, v_rndne, v_cmp*,
        v_cndmask_b32_e64, v_max/min_i32, v_mul_*24, v_lshlrev_b32, v_mbcnt, v_readlane / v_readfirstlane / v_writelane
  fp64  4.20 cycles (4.37 at 4) for add / mul / fma / max / cvt / 64-bit integer; v_rcp_f64 16.2
This tool compiles the source to assembly with line tables, attributes every VALU instruction of the kernel to a PHASE (a
source-line range), classifies it, and prices the kernel's DYNAMIC instruction count per phase (--dyn: trips x static
count on the common path, profiles/r05/inst_budget_resp_rows.md; the total is held against SQ_INSTS_VALU / wave) with the
average cycles of that phase's instructions:

    floor cycles per SIMD = sum_phase dyn_insts(phase) x avg_cycles(phase) x waves / SIMDs
    valu_mix_frac         = floor cycles per SIMD / the launch's measured cycles (GRBM_GUI_ACTIVE / 8 XCDs of the PMC pass)

No clock is assumed anywhere: cycles on both sides.  An opcode the table does not know is priced FAST (so the result stays a
floor).  Output: JSON on stdout, the table on stderr.

    python tools/valu_mix_floor.py --kernel k_resp_rowsILi3ELi11ELb1ELb0ELb0E \\
        --phases 'B_drain:543-645,A_cell:671-728,A_emit:729-798,A_estimate:799-831,A_point:832-839,parked:840-847,A_loop:848-944,epilogue:945-990' \\
        --dyn 'A_cell+A_emit+A_estimate+A_point+A_loop:1080,B_drain:1016,epilogue:283,parked:50,other:51' \\
        --waves 86016 --measured-cycles 971364 --valu-per-wave 2417.1
"""
import argparse
import collections
import json
import pathlib
import re
import subprocess
import sys
import tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-gline-tables-only", "-S"]
FAST = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32",
        "v_and_b32", "v_or_b32", "v_xor_b32", "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_cndmask_b32", "v_max_f32", "v_min_f32"}


def classify(op, operands):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base == "v_rcp_f64":
        return "rcp64"
    if "_f64" in base or "_u64" in base or "_i64" in base or base in ("v_mad_u64_u32", "v_mad_i64_i32"):
        return "fp64"
    if op.endswith(("_dpp", "_sdwa")) or "quad_perm" in operands or "row_" in operands or "sel:" in operands:
        return "slow"
    srcs = operands.split(",")[1:]  # first operand is the destination
    has_sgpr = any(re.match(r"\s*(-?\|?)?(s\d+|s\[\d+:\d+\]|ttmp|m0|exec)", x) for x in srcs)
    if base == "v_cndmask_b32":
        return "slow" if op.endswith("_e64") or has_sgpr and not all("vcc" in x or not re.match(r"\s*s", x) for x in srcs) else "fast"
    if base in FAST and not has_sgpr:
        return "fast"
    if base in FAST:
        return "slow"
    known_slow = ("v_perm", "v_bfe", "v_alignb", "v_mad", "v_add3", "v_lshl", "v_and_or", "v_mul_lo", "v_mul_hi", "v_mul_i32", "v_mul_u32",
                  "v_xad", "v_or3", "v_cvt", "v_rndne", "v_cmp", "v_max_", "v_min_", "v_mbcnt", "v_read", "v_writelane", "v_pk_", "v_sad",
                  "v_addc", "v_subb", "v_add_co", "v_sub_co", "v_rcp", "v_trunc", "v_floor", "v_fract", "v_bfi", "v_not", "v_ffb", "v_bcnt",
                  "v_ldexp", "v_frexp", "v_sqrt", "v_rsq", "v_div", "v_med3", "v_exp", "v_log")
    if base.startswith(known_slow):
        return "slow"
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--source", default=str(ROOT / "creating-2d-laser-slam-from-scratch_amd/csrc/scan_matcher.hip"))
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--phases", required=True)
    ap.add_argument("--dyn", required=True, help="'phase[+phase...]:dynamic VALU instructions per wave,...'")
    ap.add_argument("--waves", type=int, required=True)
    ap.add_argument("--simds", type=int, default=1024)
    ap.add_argument("--measured-cycles", type=float, required=True, help="cycles of one launch (GRBM_GUI_ACTIVE / 8)")
    ap.add_argument("--valu-per-wave", type=float, default=0.0, help="SQ_INSTS_VALU / SQ_WAVES of the PMC pass (sanity check)")
    ap.add_argument("--micro", default=str(ROOT / "profiles/r06/micro_valu_rate.json"))
    ap.add_argument("--asm", default="")
    a = ap.parse_args()
    micro = json.loads(pathlib.Path(a.micro).read_text())
    cyc = {}
    for w in (4, 8):
        cyc[w] = {"fast": micro[f"v_add_u32@{w}"]["simd_cyc_per_inst"], "slow": micro[f"v_perm_b32@{w}"]["simd_cyc_per_inst"],
                  "fp64": micro[f"v_add_f64@{w}"]["simd_cyc_per_inst"], "rcp64": micro[f"v_rcp_f64@{w}"]["simd_cyc_per_inst"]}
        cyc[w]["unknown"] = cyc[w]["fast"]
    if a.asm:
        text = pathlib.Path(a.asm).read_text()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = pathlib.Path(td) / "k.s"
            subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, "-o", str(out), a.source], check=True, stderr=subprocess.DEVNULL)
            text = out.read_text()
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + re.escape(a.kernel) + r"\S*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".Lfunc_end"))
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = m.group(3) or m.group(2)
    main_ids = {k for k, v in files.items() if v.endswith(pathlib.Path(a.source).name)}
    phases = []
    for part in filter(None, a.phases.split(",")):
        n, r = part.split(":")
        lo, hi = r.split("-")
        phases.append((n, int(lo), int(hi)))

    def phase(loc):
        if loc is None or loc[0] not in main_ids:
            return "other"
        for n, lo, hi in phases:
            if lo <= loc[1] <= hi:
                return n
        return "other"

    static = collections.defaultdict(collections.Counter)
    unknown = collections.Counter()
    loc = None
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (int(m.group(1)), int(m.group(2)))
            continue
        if not t.startswith("v_"):
            continue
        op, _, rest = t.partition(" ")
        rest = rest.split(";")[0]
        c = classify(op, rest)
        static[phase(loc)][c] += 1
        if c == "unknown":
            unknown[op] += 1
    table = {}
    for ph, cnt in static.items():
        n = sum(cnt.values())
        table[ph] = {"static": n, **{k: cnt[k] for k in ("fast", "slow", "fp64", "rcp64", "unknown") if cnt[k]},
                     "avg_cycles_w8": round(sum(cnt[k] * cyc[8][k] for k in cnt) / n, 3),
                     "avg_cycles_w4": round(sum(cnt[k] * cyc[4][k] for k in cnt) / n, 3)}
    dyn_total, per_wave = 0.0, {4: 0.0, 8: 0.0}
    dyn_rows = []
    for part in a.dyn.split(","):
        names, count = part.split(":")
        count = float(count)
        group = collections.Counter()
        for n in names.split("+"):
            group.update(static.get(n, {}))
        n_static = sum(group.values())
        if not n_static:
            raise SystemExit(f"phase group {names} has no instructions")
        row = {"phases": names, "dyn_insts_per_wave": count}
        for w in (4, 8):
            avg = sum(group[k] * cyc[w][k] for k in group) / n_static
            per_wave[w] += count * avg
            row[f"avg_cycles_w{w}"] = round(avg, 3)
        row["slow_share"] = round((group["slow"] + group["fp64"] + group["rcp64"]) / n_static, 3)
        dyn_rows.append(row)
        dyn_total += count
    res = {"kernel": a.kernel, "issue_cycles": cyc, "static_by_phase": table, "dynamic": dyn_rows,
           "dyn_insts_per_wave_priced": dyn_total, "valu_insts_per_wave_pmc": a.valu_per_wave or None,
           "waves": a.waves, "simds": a.simds, "measured_cycles_per_launch": a.measured_cycles,
           "unknown_opcodes_priced_fast": dict(unknown)}
    for w in (4, 8):
        # scaled to the PMC's instruction count when given (the static budget is an estimate of it)
        scale = (a.valu_per_wave / dyn_total) if a.valu_per_wave else 1.0
        floor = per_wave[w] * scale * a.waves / a.simds
        res[f"floor_cycles_per_simd_w{w}"] = round(floor, 1)
        res[f"valu_mix_frac_w{w}"] = round(floor / a.measured_cycles, 4)
        res[f"avg_cycles_per_valu_inst_w{w}"] = round(per_wave[w] / dyn_total, 3)
    res["valu_mix_frac"] = res["valu_mix_frac_w8"]
    res["note"] = ("valu_mix_frac = the time the VALU issue ports NEED for this instruction mix at their best measured rate (8 waves per "
                   "SIMD) / the launch's measured cycles; _w4 prices the same mix at the rates measured with 4 waves per SIMD, the "
                   "occupancy the 128-VGPR kernel runs at")
    print(json.dumps(res, indent=1))
    for ph, r in sorted(table.items()):
        print(ph, r, file=sys.stderr)


if __name__ == "__main__":
    main()
