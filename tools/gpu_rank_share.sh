#!/bin/bash
# HBM request bytes of ONE rank's share of an N-way deal: one rocprofv3 --pmc pass (TCC request sizes) per
# (N, deal) over tools/rank_share.py, whole frame first.  -> gpurun_out/<tag>/rank_share_pmc.json
TAG=${1:-r5share}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
i=0
for spec in "1 tiles 0" "8 tiles 0" "8 rows 0" "8 bands 0" "8 bands 4" "4 tiles 0" "4 rows 0" "2 tiles 0" "2 rows 0"; do
  set -- $spec
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d /tmp/rs_$i -o p -- \
      python $R/tools/rank_share.py --n $1 --deal $2 --ranks $3 --frames 3 > $R/$OUT/share_$i.log 2>&1 )
  f=$(find /tmp/rs_$i -name "*counter_collection.csv" | head -1)
  python - "$f" "$1" "$2" "$3" $OUT/share_$i.log >> $OUT/rank_share_pmc.jsonl <<'PY'
import csv, collections, json, sys
f, n, deal, rank, log = sys.argv[1:6]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
if f:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_march" in k or "k_shade" in k:
            agg["march" if "k_march" in k else "shade"][r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {"n": int(n), "deal": deal, "rank": int(rank)}
for k, d in agg.items():
    m = {c: sum(v[1:]) / max(1, len(v) - 1) for c, v in d.items()}          # skip the warm-up launch
    res[k + "_hbm_read_bytes"] = 32 * m.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * m.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * m.get("TCC_EA0_RDREQ_128B_sum", 0)
    res[k + "_launches"] = len(next(iter(d.values())))
try:
    res["times"] = json.loads([l for l in open(log) if l.startswith("{")][-1])["shares"]
except Exception as e:
    res["times_error"] = str(e)
print(json.dumps(res))
PY
  rm -rf /tmp/rs_$i
  tail -1 $OUT/rank_share_pmc.jsonl | cut -c1-400
done
