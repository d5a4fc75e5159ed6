#!/usr/bin/env python3
"""Static instruction budget of a gfx950 kernel: compiles a .hip source to assembly with line tables, cuts the kernel
into basic blocks (at labels AND after every branch), and prints per block the VALU / SALU / VMEM / LDS instruction
counts with the source lines the VALU instructions come from.  With --phases 'name:lo-hi,...' source-line ranges are
named.  The dynamic budget of k_resp_rows (profiles/r05/inst_budget_resp_rows.md) multiplies these blocks by the trip
counts the instrumented twin of the kernel measures (lslam_matcher_read_stats).

    python tools/inst_budget.py --kernel 'k_resp_rowsILi3ELi11ELb1ELb0ELb0E' \
        --phases 'B_drain:511-604,A_cell:635-687,A_emit:688-746,A_estimate:749-783,parked:785-792,A_loop:793-805,parked_loop:806-814,epilogue:846-885'
"""
import argparse, collections, pathlib, re, subprocess, sys, tempfile

ROOT = pathlib.Path(__file__).resolve().parent.parent
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "--cuda-device-only", "-gline-tables-only", "-S"]


def cls(op):
    if op.startswith("v_"): return "valu"
    if op.startswith("s_"): return "salu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--source", default=str(ROOT / "creating-2d-laser-slam-from-scratch_amd/csrc/scan_matcher.hip"))
    ap.add_argument("--kernel", required=True, help="substring of the mangled kernel name")
    ap.add_argument("--phases", default="")
    ap.add_argument("--asm", default="", help="reuse this assembly file instead of compiling")
    ap.add_argument("--dump", action="store_true", help="print every instruction")
    a = ap.parse_args()
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
        if m: files[int(m.group(1))] = m.group(3) or m.group(2)
    src_name = pathlib.Path(a.source).name
    main_ids = {k for k, v in files.items() if v.endswith(src_name)}
    phases = []
    for part in filter(None, a.phases.split(",")):
        n, r = part.split(":"); lo, hi = r.split("-"); phases.append((n, int(lo), int(hi)))

    def phase(loc):
        if loc is None or loc[0] not in main_ids: return "other"
        for n, lo, hi in phases:
            if lo <= loc[1] <= hi: return n
        return "line%d" % loc[1] if not phases else "other"

    blocks, cur, loc, k = [], {"name": "entry", "ins": []}, None, 0
    blocks.append(cur)
    for l in lines[start + 1:end]:
        t = l.strip()
        if not t: continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = {"name": m.group(1), "ins": []}; blocks.append(cur); k = 0; continue
        m = re.match(r"^\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (int(m.group(1)), int(m.group(2))); continue
        if t.startswith(".") or t.startswith(";"): continue
        op = t.split()[0]
        cur["ins"].append((op, t, loc))
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
            k += 1
            cur = {"name": cur["name"].split("+")[0] + "+%d" % k, "ins": [], "after": t}; blocks.append(cur)
    tot = collections.Counter()
    for b in blocks:
        if not b["ins"]: continue
        c = collections.Counter(cls(op) for op, _, _ in b["ins"])
        tot.update(c)
        ph = collections.Counter(phase(lc) for op, _, lc in b["ins"] if cls(op) == "valu")
        last = b["ins"][-1][1].replace("\t", " ")
        tail = last if b["ins"][-1][0].startswith(("s_cbranch", "s_branch", "s_endpgm")) else "(falls through)"
        print(f"{b['name']:14s} valu={c['valu']:4d} salu={c['salu']:3d} vmem={c['vmem']:2d} lds={c['lds']:2d}  {dict(ph)}  -> {tail}")
        if a.dump:
            for op, t, lc in b["ins"]: print("      ", lc, t.replace("\t", " "))
    print("static totals:", dict(tot))


if __name__ == "__main__":
    main()
