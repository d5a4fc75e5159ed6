#!/usr/bin/env python3
"""profiles/traffic.json from a tools/pmc_summary.py result: the per-launch PMC figures bench.py quotes
(HBM bytes, VALU wave-instructions, busy fractions) for the kernels of `python bench.py`.

  python tools/make_traffic.py profiles/r02/pmc_per_launch.json 4096 [pmc_cfg2_per_launch.json] > profiles/traffic.json

The optional third argument is the same summary for `tools/bench_extra.py --only cfg2 --map-scans 256` (the log-odds
kernels): `logodds_pipe` (one launch = one 1081-beam scan, apply of the previous scan + mark of this one) and
`logodds_batched` (the four kernels of one 64-scan lslam_map_update_batch call, summed).

FETCH_SIZE is doubled per the gfx950 note in MI355X_MICROARCH.md (the counter ticks in 64-B units but is
reported as if 32-B); WRITE_SIZE is taken as reported.  Both are KiB.
"""
import json
import sys

NAMES = {  # rocprofv3 kernel name prefix -> the name bench.py's HIP-event timer uses
    "k_resp_rows<3, 11, true, false, false>": "resp_rows_coarse",
    "k_resp_rows<3, 11, true, false>": "resp_rows_coarse",
    "k_resp_rows<3, 11, true>": "resp_rows_coarse",
    "k_resp_tile3": "resp_tile_fine",
    "k_reduce_coarse_lds": "reduce_coarse",
    "k_reduce_fine": "reduce_fine",
    "k_scan_prep<float>": "scan_prep",
    "k_match_fused": "match_fused",
}


def _rel(path):
    """the summary's path as it will be cited: inside the repository when it lies there"""
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap = os.path.abspath(path)
    return os.path.relpath(ap, root) if ap.startswith(root + os.sep) else path


def hbm_bytes(c):
    return int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024) if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None


def logodds_entries(path):
    pmc = json.load(open(path))
    out = {}
    pipe = next((c for k, c in pmc.items() if k.startswith("k_logodds_pipe")), None)
    if pipe:
        out["logodds_pipe"] = {"scans_per_launch": 1, "source": _rel(path), "hbm_bytes_per_launch": hbm_bytes(pipe),
                               "valu_insts_per_launch": int(pipe.get("SQ_INSTS_VALU", 0)), "waves_per_launch": int(pipe.get("SQ_WAVES", 0))}
    batch = [c for k, c in pmc.items() if k.startswith("k_lo_batch_")]
    if batch and all(hbm_bytes(c) is not None for c in batch):
        out["logodds_batched"] = {"scans_per_launch": 64, "source": _rel(path), "kernels": sorted(k for k in pmc if k.startswith("k_lo_batch_")),
                                  "hbm_bytes_per_launch": sum(hbm_bytes(c) for c in batch),
                                  "note": "sum over the kernels of one 64-scan call (per-launch averages), level 0 of a 1000x1000 map"}
    return out


def main(path, scans, cfg2_path=None):
    pmc = json.load(open(path))
    out = {}
    for kname, c in pmc.items():
        short = next((v for k, v in NAMES.items() if kname.startswith(k)), None)
        if short is None or short in out:
            continue
        rec = {"scans_per_launch": scans, "source": _rel(path)}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["FETCH_SIZE_KiB"], rec["WRITE_SIZE_KiB"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
            rec["hbm_bytes_per_launch"] = int((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        if "SQ_INSTS_VALU" in c:
            rec["valu_insts_per_launch"] = int(c["SQ_INSTS_VALU"])
            if c.get("SQ_WAVES"):
                rec["valu_insts_per_wave"] = round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1)
        if c.get("GRBM_GUI_ACTIVE"):
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: the launch's duration in clocks.  (Round 5's `valu_busy` /
            # `avg_busy_cycles_per_valu_inst` are gone: SQ_ACTIVE_INST_VALU ticks once per instruction whatever the
            # instruction -- ratio 1.0000 for v_fma_f32 and v_perm_b32 alike, profiles/r06/micro_valu_rate_pmc.txt -- so it
            # said nothing the instruction count did not.  The VALU side is priced by tools/valu_mix_floor.py instead.)
            cycles = c["GRBM_GUI_ACTIVE"] / 8.0
            rec["cycles_per_launch"] = round(cycles, 1)
            if c.get("SQ_INSTS_VALU"):
                rec["instruction_mix"] = {
                    "salu_per_valu": round(c.get("SQ_INSTS_SALU", 0.0) / c["SQ_INSTS_VALU"], 3),
                    "vmem_rd_per_valu": round(c.get("SQ_INSTS_VMEM_RD", 0.0) / c["SQ_INSTS_VALU"], 4),
                    "lds_per_valu": round(c.get("SQ_INSTS_LDS", 0.0) / c["SQ_INSTS_VALU"], 4),
                }
        if c.get("TA_TA_BUSY_sum") and c.get("GRBM_GUI_ACTIVE"):
            rec["gather_unit_busy"] = round(c["TA_TA_BUSY_sum"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 256), 3)  # 256 TAs, cycles per XCD
        if c.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
            # L1 (TCP) tag lookups.  What one costs was measured (tools/micro/ta_rate.hip, profiles/ceilings.json): a hit
            # 1 / 2.1 clock of the CU's gather pipe, a miss (L2 hit) 2.3 clocks
            rec["l1_line_lookups_per_launch"] = int(c["TCP_TOTAL_CACHE_ACCESSES_sum"])
            if c.get("TA_FLAT_READ_WAVEFRONTS_sum"):
                rec["l1_line_lookups_per_vmem_read"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / c["TA_FLAT_READ_WAVEFRONTS_sum"], 1)
            if c.get("GRBM_GUI_ACTIVE"):
                rec["l1_lookups_per_cu_clk"] = round(c["TCP_TOTAL_CACHE_ACCESSES_sum"] / (c["GRBM_GUI_ACTIVE"] / 8.0) / 256.0, 4)
            if c.get("TCP_TCC_READ_REQ_sum"):
                rec["l1_miss_ratio"] = round(c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
        if c.get("TCP_TCC_READ_REQ_sum"):
            rec["l2_read_requests_per_launch"] = int(c["TCP_TCC_READ_REQ_sum"])  # L1 -> L2 read requests, 128 B lines
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) > 0:
            rec["l2_hit"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 3)
        rec = {k: v for k, v in rec.items() if v is not None}
        out[short] = rec
    if cfg2_path:
        out.update(logodds_entries(cfg2_path))
    # which kernel source these counters describe: bench.py flags `roofline` as stale when the file has changed since
    import hashlib
    import pathlib

    csrc = pathlib.Path(__file__).resolve().parent.parent / "creating-2d-laser-slam-from-scratch_amd" / "csrc"
    out["_meta"] = {"source_sha256": {f: hashlib.sha256((csrc / f).read_bytes()).hexdigest()[:16]
                                      for f in ("scan_matcher.hip", "logodds_map.hip", "common.hpp", "karto_math.hpp",
                                                "scan_cache_impl.hpp", "frontend_impl.hpp")},
                    "note": "hash of the kernel sources at the time of the PMC passes (tools/pmc_passes.sh)"}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4096, sys.argv[3] if len(sys.argv) > 3 else None)
