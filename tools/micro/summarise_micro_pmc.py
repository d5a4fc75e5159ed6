#!/usr/bin/env python3
"""Per-DISPATCH summary of a rocprofv3 --pmc pass over tools/micro/{valu_rate,ta_rate}: the same kernel name is launched
with different grid sizes (waves per SIMD / per CU), so dispatches are kept apart by (kernel, grid size) and only the
SECOND (timed) launch of each pair is reported.  Prints one line per (kernel, grid) with every counter and the ratios the
bench line's roofline uses:
  SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU        (does the "busy" counter say anything beyond the instruction count?)
  TCP_TOTAL_CACHE_ACCESSES / SQ_INSTS_VMEM_RD (tag lookups per vector-memory instruction)
  TCP_TOTAL_CACHE_ACCESSES / GRBM_GUI_ACTIVE / 256  (lookups per CU per GRBM clock: the ceiling l1_lookup_frac is priced against)

    python tools/micro/summarise_micro_pmc.py gpurun_out/micro/pmc_ta
"""
import collections
import csv
import glob
import re
import sys


def main(root):
    disp = collections.OrderedDict()
    for path in sorted(glob.glob(f"{root}/**/*counter_collection.csv", recursive=True)):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = re.sub(r"^void ", "", row["Kernel_Name"])
                key = (int(row["Dispatch_Id"]), k, int(row["Grid_Size"]))
                disp.setdefault(key, collections.defaultdict(float))[row["Counter_Name"]] += float(row["Counter_Value"])
    seen = collections.Counter()
    for (did, k, grid), c in sorted(disp.items()):
        seen[(k, grid)] += 1
        if seen[(k, grid)] != 2:  # the warm-up launch of each pair
            continue
        parts = [f"{n}={v:.0f}" for n, v in sorted(c.items())]
        extra = []
        if c.get("SQ_INSTS_VALU"):
            extra.append("ACTIVE_INST_VALU/INSTS_VALU=%.4f" % (c["SQ_ACTIVE_INST_VALU"] / c["SQ_INSTS_VALU"]))
        if c.get("SQ_INSTS_VMEM_RD") and c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            extra.append("lookups/vmem_inst=%.2f" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["SQ_INSTS_VMEM_RD"]))
        if c.get("GRBM_GUI_ACTIVE") and c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            extra.append("lookups/CU/grbm_clk=%.4f" % (c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["GRBM_GUI_ACTIVE"] / 256.0))
        if c.get("GRBM_GUI_ACTIVE") and c.get("TA_TA_BUSY_sum"):
            extra.append("ta_busy/CU/grbm_clk=%.4f" % (c["TA_TA_BUSY_sum"] / c["GRBM_GUI_ACTIVE"] / 256.0))
        print(f"{k} grid={grid}  " + " ".join(parts) + "  | " + " ".join(extra))


if __name__ == "__main__":
    main(sys.argv[1])
