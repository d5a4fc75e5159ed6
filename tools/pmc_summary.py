#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of tools/pmc_passes.sh: per kernel, the average counter value
per launch (summed over the dispatch's dimensions/XCDs as rocprofv3 reports them).

  python tools/pmc_summary.py gpurun_out/pmc > profiles/rNN/pmc_per_launch.json
"""
import collections
import csv
import glob
import json
import re
import sys


def main(root):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for path in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                k = re.sub(r"^void ", "", re.sub(r"\(.*", "", k)).strip()
                c = row["Counter_Name"]
                acc[k][c] += float(row["Counter_Value"])
                launches[k][c].add((path, row["Dispatch_Id"]))
    out = {k: {c: v / max(1, len(launches[k][c])) for c, v in sorted(cs.items())} for k, cs in sorted(acc.items())}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc")
