"""Static instruction mix of the fused kernels (device-only assembly)."""
import collections, re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "unboundednerfpytorch_amd", "csrc", "ugrid_march.hip")
out = "/tmp/ugrid_fused.s"
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include"),
                       "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
txt = open(out).read()
pat = sys.argv[1:] or ["k_marchILi3ELb0ELi5"]
for f in re.split(r'\n(?=_Z\w+:)', txt):
    name = f.split(':', 1)[0]
    if any(p in name for p in pat):
        body = f.split('s_endpgm')[0]
        lines = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.split()[0].endswith(':')]
        ops = collections.Counter(l.split()[0] for l in lines)
        g = collections.Counter()
        for o, c in ops.items():
            k = 'mfma' if o.startswith('v_mfma') else 'valu' if o.startswith('v_') else 'salu' if o.startswith('s_') else \
                'vmem' if o.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'lds' if o.startswith('ds_') else 'other'
            g[k] += c
        print(name[:44], 'total', sum(ops.values()), dict(g))
        print('   top:', ops.most_common(16))
