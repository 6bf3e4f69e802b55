for t in "shade16=0" "shade16=1" "shade16=1 --tune shade_dbg=1" "shade16=1 --tune shade_dbg=2"; do
  echo "== $t"; python bench.py --steps 6 --warmup 2 --no-cpu-baseline --tune $t 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v['ms'],3) for k,v in d['kernels'].items()})"
done
