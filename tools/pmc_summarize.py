"""Turn the rocprofv3 --pmc passes of tools/gpu_pmc.sh (gpurun_out/<tag>/pmc_*/.../*counter_collection.csv) into
profiles/r03/pmc_summary.json, the static per-launch counters bench.py's roofline block combines with its live times.
usage: python tools/pmc_summarize.py gpurun_out/<tag> [device_code_sha16]      (bench.device_code_sha16: sha256 of .hip_fatbin)

hbm_bytes per launch = 2 x FETCH_SIZE [KB] x 1024 (the gfx950 correction of MI355X_MICROARCH.md, section HBM: FETCH_SIZE
tallies 128-byte requests at 64 B) + WRITE_SIZE [KB] x 1024; Infinity-Cache hits are included in these counters."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {"k_march": "render_march", "k_shade_mlp": "render_shade", "k_shade_pc": "render_shade"}


def main():
    tag = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    files = glob.glob(os.path.join(tag, "pmc_*", "**", "*counter_collection.csv"), recursive=True) + glob.glob(os.path.join(tag, "pmc_*.csv"))
    for f in files:
        for r in csv.DictReader(open(f)):
            for k, name in NAMES.items():
                if k in r["Kernel_Name"]:
                    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {"_comment": "per-launch means over the profiled launches of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                       "--no-secondary` (S1 frame); source csv files next to this file",
           "device_code_sha16": sys.argv[2] if len(sys.argv) > 2 else None}
    for name, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        e = {"counters": m, "launches": max(len(v) for v in d.values())}
        if "FETCH_SIZE" in m:
            e["hbm_read_bytes"] = 2 * m["FETCH_SIZE"] * 1024
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
        out[name] = e
    dst = os.path.join(ROOT, "profiles", "r03", "pmc_summary.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1)[:3000])


if __name__ == "__main__":
    main()
