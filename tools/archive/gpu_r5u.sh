#!/bin/bash
# Round-5 visit U: counters of the native DVGO training step's kernels (two rocprofv3 --pmc passes, kernel trace only)
OUT=gpurun_out/r5u; mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for ctrs in "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU" "FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_u$i
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d /tmp/pmc_u$i -o p -- python $R/tools/bench_voxgo_train.py --model dvgo --steps 12 --warmup 4 > $R/$OUT/pmc_$i.log 2>&1 )
  f=$(find /tmp/pmc_u$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && grep "Kernel_Name\|k_lin\|k_wgrad\|k_l3\|k_grid_query\|k_train\|k_rgbnet\|k_render_loss\|k_adam" "$f" > $OUT/pmc_pass_$i.csv
done
python - $OUT <<'PY'
import csv, collections, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    try:
        for r in csv.DictReader(open("%s/pmc_pass_%d.csv" % (out, i))):
            agg[r["Kernel_Name"].split("(")[0][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    except Exception as e:
        print("pass", i, e)
lines = []
for k, d in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in d.items()}
    n = len(next(iter(d.values())))
    busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(m.get("SQ_BUSY_CYCLES", 0), 1)
    rd = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
    lines.append("%-46s n=%3d  mfma_busy/sq_busy=%.3f  gui_active=%.3g  mops_f32=%.3g  valu=%.3g  hbm_read=%.1f MB  fetch_size=%.0f write_size=%.0f" % (
        k, n, busy, m.get("GRBM_GUI_ACTIVE", 0), m.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0), m.get("SQ_INSTS_VALU", 0), rd / 1e6, m.get("FETCH_SIZE", 0), m.get("WRITE_SIZE", 0)))
open(out + "/voxgo_train_dvgo_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
tail -3 $OUT/pmc_1.log | cut -c1-200
