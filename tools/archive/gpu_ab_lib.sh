for lib in unboundednerfpytorch_amd/libugrid_hip.so build/ab/lib_noxpass.so; do
  echo "== $lib"; UGRID_LIB=$lib python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v['ms'],3) for k,v in d['kernels'].items()}, round(d['ms_per_step'],3))"
done
