"""Turn the rocprofv3 --pmc passes of tools/gpu_pmc.sh (gpurun_out/<tag>/pmc_csv/pmc_pass_*.csv, or profiles/rNN/pmc/) into
profiles/<round>/pmc_summary.json, the static per-launch counters bench.py's roofline block combines with its live times.
usage: python tools/pmc_summarize.py <dir with pmc_pass_*.csv> <device_code_sha16> [out.json] [kernel_stats.csv of `rocprofv3 --stats`]

HBM bytes per launch: FETCH_SIZE [KB] x 1024 x f + WRITE_SIZE [KB] x 1024, where f is the factor CALIBRATED for the kernel's
access shape by tools/microbench/fetch_calib.hip (profiles/r04/microbench_fetch_calib.json): the guide's x 2 holds for wide
coalesced 16 B/lane streams (128-byte requests tallied at 64 B); 32-byte and 64-byte gathers issue 64-byte requests and are
counted at their true size.  Without the calibration file f = 2 for every kernel and `fetch_factor_source` says so."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"k_march": "render_march", "k_shade_mlp": "render_shade", "k_shade_pc": "render_shade"}
# standard counter groups of round 4 (<= 8 SQ counters, <= 4 TCP / TA counters, FETCH_SIZE and WRITE_SIZE apart)
PMC_GROUPS = [
    "FETCH_SIZE GRBM_GUI_ACTIVE",
    "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum",
    "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM",
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC",
    "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAVES",
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum",
    "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum",
    "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TD_BUSY_sum",
]
FETCH_SHAPE = {"render_march": "records_32B", "render_shade": "records_384B"}     # keys of microbench_fetch_calib.json


def main():
    tag = sys.argv[1]
    code = sys.argv[2] if len(sys.argv) > 2 else None
    dst = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles", "r05", "pmc_summary.json")
    stats_csv = sys.argv[4] if len(sys.argv) > 4 else None
    avg_ms = {}
    if stats_csv and os.path.exists(stats_csv):      # rocprofv3 --kernel-trace --stats of the same command: average duration per kernel
        for r in csv.DictReader(open(stats_csv)):
            for k, name in NAMES.items():
                if k in r["Name"]:
                    avg_ms[name] = float(r["AverageNs"]) / 1e6
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    files = sorted(glob.glob(os.path.join(tag, "**", "pmc_pass_*.csv"), recursive=True))
    for f in files:
        for r in csv.DictReader(open(f)):
            for k, name in NAMES.items():
                if k in r["Kernel_Name"]:
                    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    cal = None
    try:
        cal = json.load(open(os.path.join(os.path.dirname(dst), "microbench_fetch_calib.json")))
    except Exception:
        pass
    out = {"_comment": "per-launch means over the profiled launches of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                       "--no-secondary` (S1 frame), one rocprofv3 --pmc run per counter group; source csv files in pmc/",
           "device_code_sha16": code, "groups": PMC_GROUPS,
           "fetch_factor_source": "microbench_fetch_calib.json (per access shape)" if cal else "guide's x2 for every kernel (uncalibrated for gathers)"}
    for name, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        e = {"counters": m, "launches": max(len(v) for v in d.values())}
        if name in avg_ms:
            e["rocprofv3_avg_ms"] = avg_ms[name]
        if "FETCH_SIZE" in m:
            f = 2.0
            if cal and FETCH_SHAPE.get(name) in cal.get("fetch_size_factor", {}):
                f = float(cal["fetch_size_factor"][FETCH_SHAPE[name]])
            e["fetch_size_factor"] = f
            e["hbm_read_bytes"] = f * m["FETCH_SIZE"] * 1024
            e["hbm_read_bytes_if_x2"] = 2.0 * m["FETCH_SIZE"] * 1024
            e["hbm_write_bytes"] = m.get("WRITE_SIZE", 0.0) * 1024
            e["hbm_bytes"] = e["hbm_read_bytes"] + e["hbm_write_bytes"]
        if "SQ_INSTS_VALU" in m:
            e["valu_insts"] = m["SQ_INSTS_VALU"]
        if "SQ_INSTS_MFMA" in m:
            e["mfma_insts"] = m["SQ_INSTS_MFMA"]
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            e["mfma_busy_cycles"] = m["SQ_VALU_MFMA_BUSY_CYCLES"]
        if "GRBM_GUI_ACTIVE" in m:
            e["gui_active_cycles"] = m["GRBM_GUI_ACTIVE"] / 8.0     # the counter sums the 8 XCDs
        if "TCC_HIT_sum" in m and "TCC_REQ_sum" in m:
            e["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_REQ_sum"])
        # vector-L1 (TCP) evidence: accesses are 64-byte tag look-ups; requests that go on to L2 = misses
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in m:
            e["l1_accesses"] = m["TCP_TOTAL_CACHE_ACCESSES_sum"]
            e["l1_bytes_64B_per_access"] = 64.0 * m["TCP_TOTAL_CACHE_ACCESSES_sum"]
            if "TCP_TCC_READ_REQ_sum" in m:
                e["l1_to_l2_read_requests"] = m["TCP_TCC_READ_REQ_sum"]
                e["l1_hit_rate"] = 1.0 - m["TCP_TCC_READ_REQ_sum"] / max(1.0, m["TCP_TOTAL_CACHE_ACCESSES_sum"])
        if "TA_TA_BUSY_sum" in m and "GRBM_GUI_ACTIVE" in m:
            e["ta_busy_frac"] = m["TA_TA_BUSY_sum"] / 256.0 / (m["GRBM_GUI_ACTIVE"] / 8.0)      # one TA per CU
        if "TA_TA_BUSY_sum" in m and m.get("TA_FLAT_READ_WAVEFRONTS_sum"):
            e["ta_clocks_per_wave_instruction"] = m["TA_TA_BUSY_sum"] / m["TA_FLAT_READ_WAVEFRONTS_sum"]       # 16 at the path's peak
        if "TCC_EA0_RDREQ_128B_sum" in m:
            e["hbm_read_bytes_by_request_size"] = (32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0)
                                                   + 128 * m["TCC_EA0_RDREQ_128B_sum"])
        if "SQ_WAVE_CYCLES" in m:
            wc = m["SQ_WAVE_CYCLES"]
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM",
                      "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"):
                if k in m:
                    e[k.lower() + "_of_wave_cycles"] = m[k] / wc
        if "SQ_LDS_IDX_ACTIVE" in m and "GRBM_GUI_ACTIVE" in m:
            e["lds_array_busy_frac"] = m["SQ_LDS_IDX_ACTIVE"] / 256.0 / (m["GRBM_GUI_ACTIVE"] / 8.0)
            e["lds_bank_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(1.0, m["SQ_LDS_IDX_ACTIVE"])
        out[name] = e
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} if isinstance(v, dict) else v for k, v in out.items()}, indent=1)[:6000])


if __name__ == "__main__":
    main()
