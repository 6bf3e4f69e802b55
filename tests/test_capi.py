"""CPU checks of the drop-in boundary: libugrid_hip.so builds for gfx950, loads, and exports every symbol
that include/ugrid_hip.h declares (no compute calls -- there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__
    __graft_entry__.build()
    return os.path.join(ROOT, "unboundednerfpytorch_amd", "libugrid_hip.so")


def declared_functions():
    text = open(os.path.join(ROOT, "include", "ugrid_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ugrid_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(lib_path):
    names = declared_functions()
    assert len(names) >= 25
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_python_binding_table_matches_header(lib_path):
    from unboundednerfpytorch_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared_functions()
    lib = _lib.load()
    assert lib.ugrid_abi_version() == _lib.ABI_VERSION
    assert lib.ugrid_target_arch() == b"gfx950"
    # size helpers are pure host arithmetic: S1 numbers from DESIGN.md
    assert lib.ugrid_brick_bytes(7, 1, 200, 200, 200, 0) == 7 * 199 ** 3 * 32
    assert lib.ugrid_brick_bytes(7, 12, 200, 200, 200, 0) == 7 * 199 ** 3 * 384
    assert lib.ugrid_brick_bytes(1, 3, 160, 160, 160, 1) == 159 ** 3 * 128
    fp32_img = 20 * 256 + 64 * 256 + 128 + 128 + 512 + 4
    bf16_img = (3 + 8) * 4 * 3 * 64 * 4 + 128 + 128 + 512 + 4
    fp16_img = (3 + 8) * 4 * 2 * 64 * 4 + 128 + 128 + 512 + 4 + 4
    assert lib.ugrid_mlp_packed_bytes(12, 4) == 4 * (fp32_img + bf16_img + fp16_img)
    assert lib.ugrid_render_ws_bytes(64, 256) >= 64 * 256 * 17
    # the work list's layout (csrc/ugrid_render.h: ug_ws_make): tile counter | count | entries | ray slots -- the byte count must
    # cover every region, for ragged ray counts too
    a256 = lambda x: (x + 255) & ~255
    for n_rays, S in ((64, 256), (1, 7), (65, 668), (1920 * 1080, 256), (1000003, 1068)):
        nt = (n_rays + 63) // 64
        cap = 64 * S
        assert lib.ugrid_render_ws_bytes(n_rays, S) == 256 + a256(nt * 4) + a256(nt * cap * 16) + a256(nt * cap)


def test_float64_twin_library_exports_its_header(lib_path):
    """include/ugrid_hip_f64.h (the double instantiations of the ops the reference dispatches on the tensor type) = the exports of
    libugrid_hip_f64.so = the ctypes table; every twin has the argument list of its float entry point (the count half of
    sample_pts_on_rays without the scan workspace); the library carries gfx950 code only and the product library does not depend on it"""
    from unboundednerfpytorch_amd import _lib
    text = open(os.path.join(ROOT, "include", "ugrid_hip_f64.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = sorted(set(re.findall(r"\b(ugrid_[a-z0-9_]+)\s*\(", text)))
    assert names == sorted(_lib.EXPORTED_SYMBOLS_F64) and len(names) == 15
    so = os.path.join(ROOT, "unboundednerfpytorch_amd", "libugrid_hip_f64.so")
    lib = ctypes.CDLL(so)
    assert not [n for n in names if not hasattr(lib, n)]
    base = open(os.path.join(ROOT, "include", "ugrid_hip.h")).read()
    base = re.sub(r"/\*.*?\*/", "", base, flags=re.S)
    def params(src, name):
        m = re.search(r"\b%s\s*\((.*?)\)\s*;" % name, src, flags=re.S)
        return [re.sub(r"\s+", " ", a.strip()) for a in m.group(1).split(",")]
    for n in names:
        a32, a64 = params(base, n[:-4]), params(text, n)
        if n == "ugrid_sample_pts_on_rays_count_f64":
            a32 = [a for a in a32 if "scan_ws" not in a]
        assert len(a32) == len(a64), n
        for x, y in zip(a32, a64):
            assert x.replace("const float *", "const double *").replace("float *", "double *").split()[:-1] == y.split()[:-1] \
                or x.split()[:-1] == y.split()[:-1], (n, x, y)
    assert _lib.load_f64() is _lib.load_f64()
    out = subprocess.run(["ldd", lib_path], capture_output=True, text=True).stdout
    assert "ugrid_hip_f64" not in out
    assert b"gfx950" in open(so, "rb").read() and b"gfx942" not in open(so, "rb").read()


def test_voxgo_step_struct_mirror_matches_the_header(lib_path):
    """_lib.VoxgoStep mirrors `ugrid_voxgo_step` field for field: names and order parsed from the header, C type -> ctypes type,
    and the compiled sizeof; the workspace size helpers are host arithmetic over the struct's counts"""
    from unboundednerfpytorch_amd import _lib
    text = open(os.path.join(ROOT, "include", "ugrid_hip.h")).read()
    body = re.search(r"typedef struct ugrid_voxgo_step \{(.*?)\} ugrid_voxgo_step;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    ctype = {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "double": ctypes.c_double}
    want = []
    for decl in [d.strip() for d in body.split(";") if d.strip()]:
        m = re.match(r"(const\s+)?(\w+)\s+(.*)", decl, flags=re.S)
        base = m.group(2)
        for item in [x.strip() for x in m.group(3).split(",")]:
            ptr = item.startswith("*")
            name = item.lstrip("*").strip()
            arr = re.match(r"(\w+)\[(\d+)\]", name)
            if ptr:
                want.append((name, ctypes.c_void_p))
            elif arr:
                want.append((arr.group(1), ctype[base] * int(arr.group(2))))
            else:
                want.append((name, ctype[base]))
    got = list(_lib.VoxgoStep._fields_)
    assert [n for n, _ in got] == [n for n, _ in want]
    for (n, a), (_, b) in zip(got, want):
        assert ctypes.sizeof(a) == ctypes.sizeof(b) and (a is b or getattr(a, "_length_", None) == getattr(b, "_length_", None)), n
    lib = _lib.load()
    assert lib.ugrid_voxgo_step_sizeof() == ctypes.sizeof(_lib.VoxgoStep)
    s = _lib.VoxgoStep()
    s.C, s.pe, s.width, s.n_rays, s.M1, s.M2 = 12, 4, 128, 8192, 1000, 130
    al = lambda n: (n + 63) & ~63
    K = 12 + 27
    assert lib.ugrid_voxgo_step_ws_floats(ctypes.addressof(s)) == al(3000) + 4 * al(1000) + al(390) + al(130 * 12) + al(130 * K) + 2 * al(130 * 128) \
        + al(8192 * 27)
    assert lib.ugrid_voxgo_step_bwd_ws_floats(ctypes.addressof(s)) == al(390) + 2 * al(130) + al(8192) + al(130 * 12) + al(1000) \
        + al(lib.ugrid_rgbnet_train_scratch_floats(130))
    # entry points refuse what they cannot run, before touching the device
    bad = _lib.VoxgoStep()
    assert lib.ugrid_voxgo_step_forward(ctypes.addressof(bad), None) != 0


def test_code_object_is_gfx950_only(lib_path):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", lib_path], capture_output=True, text=True)
    # the fat binary is embedded; roc-obj-ls style check through strings
    blob = open(lib_path, "rb").read()
    assert b"gfx950" in blob
    for other in (b"gfx942", b"gfx90a", b"sm_90", b"nvptx"):
        assert other not in blob, other


def test_reference_module_names_and_signatures():
    """The four drop-in modules expose exactly the reference's m.def names (render_utils.cpp:170-184,
    total_variation.cpp:23, ub360_utils.cpp:21, adam_upd.cpp:79-86) with the same positional parameters."""
    import inspect
    from unboundednerfpytorch_amd import adam_upd_cuda, compat, render_utils_cuda, total_variation_cuda, ub360_utils_cuda
    want = {
        render_utils_cuda: {
            "infer_t_minmax": 6, "infer_n_samples": 4, "infer_ray_start_dir": 3, "sample_pts_on_rays": 7,
            "sample_ndc_pts_on_rays": 5, "sample_bg_pts_on_rays": 5, "maskcache_lookup": 4, "raw2alpha": 3,
            "raw2alpha_backward": 3, "raw2alpha_nonuni": 3, "raw2alpha_nonuni_backward": 3, "alpha2weight": 3,
            "alpha2weight_backward": 9},
        total_variation_cuda: {"total_variation_add_grad": 6},
        ub360_utils_cuda: {"cumdist_thres": 2},
        adam_upd_cuda: {"adam_upd": 9, "masked_adam_upd": 9, "adam_upd_with_perlr": 10},
    }
    for mod, fns in want.items():
        for name, nargs in fns.items():
            assert len(inspect.signature(getattr(mod, name)).parameters) == nargs, name
    import sys
    names = compat.install_as_reference_extensions()
    for n in names:
        assert n in sys.modules
    import render_utils_cuda as r2  # the name the reference imports
    assert r2 is render_utils_cuda
    # the ops refuse host tensors exactly like the reference's CHECK_INPUT
    import torch
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):
        render_utils_cuda.raw2alpha(torch.zeros(3), 0.0, 0.5)


def test_fp16x2_scale_derivation_is_host_arithmetic(lib_path):
    """ugrid_mlp_fp16x2_scales (the decision behind ugrid_pack_mlp's best_mode) runs on host buffers: powers of two
    that put the largest weight / the propagated activation bound just below 2^15, and a refusal for degenerate or
    out-of-range networks."""
    import numpy as np
    from unboundednerfpytorch_amd import _lib
    lib = _lib.load()
    C, pe = 12, 4
    mlp_in = C + 3 + 6 * pe
    rs = np.random.RandomState(0)
    w0 = rs.uniform(-0.16, 0.16, (128, mlp_in)).astype(np.float32)
    b0 = rs.uniform(-0.16, 0.16, 128).astype(np.float32)
    w1 = rs.uniform(-0.088, 0.088, (128, 128)).astype(np.float32)
    sc = (ctypes.c_float * 4)()

    def call(w0_, b0_, w1_, k0max):
        return lib.ugrid_mlp_fp16x2_scales(w0_.ctypes.data, b0_.ctypes.data, w1_.ctypes.data, C, pe,
                                           ctypes.c_float(k0max), ctypes.cast(sc, ctypes.c_void_p))

    assert call(w0, b0, w1, 5.5) == 1
    sX1, sW1, sX2, sW2 = list(sc)
    for v in (sX1, sW1, sX2, sW2):
        assert v > 0 and np.log2(v) == np.round(np.log2(v))          # exact powers of two
    bound = np.array([5.5] * C + [1.0] * (mlp_in - C))
    B1 = (np.abs(w0.astype(np.float64)) @ bound + np.abs(b0)).max()
    # (the layer-1 bias is stored as one more weight column of the 16x16x32 image, so it shares sW1's range)
    for scale, mag in ((sX1, 5.5), (sW1, max(np.abs(w0).max(), np.abs(b0).max())), (sX2, B1), (sW2, np.abs(w1).max())):
        assert 16384.0 < scale * mag <= 32768.0                       # largest scaled operand in (2^14, 2^15]
    assert call(w0, b0, w1, 0.0) == 0 and list(sc) == [1.0] * 4       # unknown feature bound
    assert call(w0, b0, w1, 1e30) == 0                                # bound far outside fp16's reach
    assert call(w0, b0, np.zeros_like(w1), 5.5) == 0                  # degenerate layer
    w_nan = w0.copy(); w_nan[3, 4] = np.nan
    assert call(w_nan, b0, w1, 5.5) == 0


def test_shade_shape_dispatch_table_and_loss_coefficients():
    """host-only entry points: ugrid_shade_supported mirrors the (F, C, PE) instantiations of ugrid_shade.hip, and
    fourier_render.fused_shape_supported routes a reference checkpoint to the fused or the composed renderer;
    ops.loss_coefficients packs cfg_train for ugrid_render_loss (None when nearclip is on without its threshold)."""
    import os
    import types

    import torch
    from unboundednerfpytorch_amd import _lib, ops
    from unboundednerfpytorch_amd.fourier_render import fused_shape_supported
    L = _lib.load()
    assert [L.ugrid_shade_supported(*t) for t in ((3, 12, 4), (4, 12, 4), (5, 12, 4), (2, 3, 2), (3, 3, 2))] == [1] * 5
    assert [L.ugrid_shade_supported(*t) for t in ((3, 12, 8), (3, 3, 8), (3, 15, 4))] == [1] * 3   # waymo_base / mega / train_single shapes
    assert L.ugrid_shade_supported(3, 9, 4) == 1                                                     # free_dataset shapes (width 64 padded)
    assert [L.ugrid_shade_supported(*t) for t in ((4, 12, 8), (2, 9, 8), (6, 12, 4), (0, 3, 2))] == [0] * 4
    assert L.ugrid_shade_supported(0, 12, 4) == 1           # single-level k0: the fused DirectContractedVoxGO path
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    small = torch.load(os.path.join(gold, "fg_ckpt_small.tar"), map_location="cpu", weights_only=False)
    odd = torch.load(os.path.join(gold, "fg_ckpt_odd.tar"), map_location="cpu", weights_only=False)
    assert fused_shape_supported(small) and not fused_shape_supported(odd)
    cfg = dict(weight_main=1.0, weight_entropy_last=1e-3, weight_distortion=0.01, weight_rgbper=0.02, weight_nearclip=0.5)
    c = ops.loss_coefficients(cfg, 4096, 668, near_thres=0.2, world_size=2)
    assert c == (1.0, 1e-3, 0.01, 0.02, 1.0, 0.2, 1.0 / 668, 4096.0, 0.0)
    assert ops.loss_coefficients(types.SimpleNamespace(**cfg), 4096, 668, near_thres=0.2, world_size=2) == c
    assert ops.loss_coefficients(dict(cfg, weight_freq=5.0), 4096, 668, 0.2)[8] == 5.0         # image-space Fourier loss (bicycle_single.py:57): 9th entry
    assert ops.loss_coefficients(cfg, 4096, 668, near_thres=None) is None                     # nearclip needs its threshold
    assert ops.loss_coefficients(dict(cfg, weight_nearclip=0.0), 4096, 668)[4:6] == (0.0, 0.0)


def _kernel_metadata(so):
    """per-kernel {vgpr_count, vgpr_spill_count, sgpr_spill_count, private_segment_fixed_size} of every gfx950 code object
    embedded in the library (.hip_fatbin = one clang offload bundle per translation unit)"""
    import os
    import re
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    out = {}
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for i, s in enumerate(starts):
            part, co = os.path.join(td, "b%d.bin" % i), os.path.join(td, "b%d.co" % i)
            open(part, "wb").write(blob[s:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
            r = subprocess.run([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + part, "--output=" + co], capture_output=True, text=True)
            if r.returncode or not os.path.exists(co):
                continue
            cur = None
            for line in subprocess.run([llvm + "/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout.splitlines():
                m = re.match(r"\s+\.name:\s+(\S+)", line)
                if m:
                    cur = out.setdefault(m.group(1), {})
                    continue
                m = re.match(r"\s+\.(vgpr_count|vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size):\s+(\d+)", line)
                if m and cur is not None:
                    cur[m.group(1)] = int(m.group(2))
    return out


def test_product_kernels_fit_their_register_budget_without_scratch(lib_path):
    """Build-quality guard (no GPU needed): the kernels on the product path keep everything in registers -- no VGPR / SGPR
    spills, no scratch -- and stay inside the occupancy their design assumes: k_march <= 80 VGPR (6 waves / SIMD), the shade
    kernel <= 256 (2 waves / SIMD with 8 waves per CU), streaming kernels <= 64."""
    meta = _kernel_metadata(lib_path)
    assert len(meta) > 100

    def find(prefix):
        ks = [k for k in meta if k.startswith(prefix)]
        assert ks, prefix
        return ks
    budget = {"_Z7k_marchILi": 80, "_Z11k_shade_mlpILi": 256, "_Z10k_shade_pcILi": 256, "_Z17k_train_march_voxILi": 128, "_Z14k_shade_direct": 128, "_Z13k_train_march": 128,
              "_Z15k_train_compact": 64, "_Z12k_grid_queryILb": 128,      # (round 6: the sin and cos level of a frequency gathered together, 16 loads in flight: 78 / 124)
                  "_Z8k_lin_b3ILi": 256, "_Z14k_lin_b3_denseILi": 256, "_Z10k_wgrad_b3ILi": 256, "_Z21k_grid_query_backwardILb": 64, "_Z12k_tv_cl_vec4ILb": 64, "_Z12k_tv_cl_slabILi": 64, "_Z12k_adam_multiILi": 32,
              "_Z14k_tv_adam_vec4ILb": 64, "_Z9k_tv_vec4ILb": 64, "_Z11k_adam_vec4ILi": 64, "_Z17k_render_loss_fwd": 64,
              "_Z17k_render_loss_bwd": 64, "_Z14k_alpha2weight": 64, "_Z18k_alpha2weight_bwd": 64, "_Z16k_rays_of_a_view": 64,
              "_Z11k_pack_quad": 128, "_Z5k_linILi": 128, "_Z7k_wgradILi": 256, "_Z16k_train_compact2": 64, "_Z18k_train_sample_bwd": 64,
              "_Z17k_adam_vec4_touch": 64, "_Z13k_tv_cl_touch": 64, "_Z12k_lin_smallk": 64, "_Z12k_march_dvgo": 80}
    import re
    for prefix, limit in budget.items():
        for k in find(prefix):
            m = meta[k]
            w = re.match(r"_Z7k_marchILi\dELb[01]ELi(\d)EE", k)
            if w:                                           # k_march<F, L2, W>: W waves per SIMD -> 512 / W registers
                limit = {4: 128, 5: 96, 6: 80}[int(w.group(1))]
            lim = limit
            if re.match(r"_Z10k_shade_pcILi\dELi\dELi6E", k):  # the 12-wave geometry (6 + 6): 3 waves per SIMD -> 168 registers
                lim = 168
            assert m.get("vgpr_spill_count", 0) == 0 and m.get("sgpr_spill_count", 0) == 0 and m.get("private_segment_fixed_size", 0) == 0, (k, m)
            assert m["vgpr_count"] <= lim, (k, m)


def _kernel_disassembly(so, name_prefix):
    """gfx950 assembly text of the first kernel whose symbol starts with `name_prefix`, from the code objects embedded in
    the library (llvm-objdump on the unbundled .hip_fatbin)"""
    import os
    import re
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([llvm + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, so])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(b"__CLANG_OFFLOAD_BUNDLE__"), blob)]
        for i, s in enumerate(starts):
            part, co = os.path.join(td, "b%d.bin" % i), os.path.join(td, "b%d.co" % i)
            open(part, "wb").write(blob[s:(starts[i + 1] if i + 1 < len(starts) else len(blob))])
            r = subprocess.run([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                "--input=" + part, "--output=" + co], capture_output=True, text=True)
            if r.returncode or not os.path.exists(co):
                continue
            syms = subprocess.run([llvm + "/llvm-readelf", "-s", "-W", co], capture_output=True, text=True).stdout
            names = [l.split()[-1] for l in syms.splitlines() if l.split() and l.split()[-1].startswith(name_prefix) and " FUNC " in l]
            if not names:
                continue
            txt = subprocess.run([llvm + "/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + names[0], co],
                                 capture_output=True, text=True).stdout
            return [l.strip() for l in txt.splitlines() if l.startswith("\t") or l.startswith("  ")]
    return None


def test_shade_kernel_instruction_stream_regression(lib_path):
    """ISA-level guard of the hand-scheduled parts of the producer / consumer shade kernel (VERDICT r2 "what's weak" 9: the
    gather is volatile inline asm with hand-counted s_waitcnt vmcnt(N), the MFMA chain relies on a pinned issue order): a
    compiler update that reorders, merges or re-counts any of it fails HERE, on the CPU, instead of as rare wrong pixels.
      * no flat_ / scratch_ instruction (flat operations would count on vmcnt and break the hand-counted waits);
      * the k0 gather of a pass: 84 global_load_dwordx4 = 14 (round, level) items x 6, 36 in flight, each wait preceded by
        exactly one item's 6 loads, waits counting down 30 x9, 24, 18, 12, 6, 0;
      * the fp16x2 rgbnet: exactly 132 v_mfma_f32_32x32x16_f16, and NO MFMA reads as SrcC the accumulator written by the
        MFMA issued right before it (the gfx950 dependent-MFMA observation, tools/microbench/mfma_dep_hazard.hip);
      * the hand-off polls sleep (s_sleep) instead of spinning."""
    import re
    # (geometry, gather pattern): 8 waves = 6 items (36 loads) in flight, 12 waves = 3 items (18 loads); F = 4 (truck_single.py,
    # 18 items per pass) in the 12-wave geometry with the rolling cell set-up (round 5)
    for sym, want in (("_Z10k_shade_pcILi3ELi4ELi4ELi4ELi6ELi0E", "L" * 36 + ("<30>" + "L" * 6) * 8 + "<30><24><18><12><6><0>"),
                      ("_Z10k_shade_pcILi3ELi4ELi6ELi2ELi3ELi1E", "L" * 18 + ("<12>" + "L" * 6) * 11 + "<12><6><0>"),
                      ("_Z10k_shade_pcILi4ELi4ELi6ELi2ELi3ELi1ELb1E", "L" * 18 + ("<12>" + "L" * 6) * 15 + "<12><6><0>")):
        asm = _kernel_disassembly(lib_path, sym)
        assert asm and len(asm) > 2000, sym
        ops = [l.split()[0] for l in asm]
        assert not [o for o in ops if o.startswith(("flat_", "scratch_"))], sym
        seq = [("L", None) if o == "global_load_dwordx4" else ("W", int(re.search(r"vmcnt\((\d+)\)", l).group(1)))
               for o, l in zip(ops, asm) if o == "global_load_dwordx4" or (o == "s_waitcnt" and "vmcnt" in l)]
        flat = "".join("L" if k == "L" else "<%d>" % n for k, n in seq)
        assert want in flat, (sym, flat[-400:])
        mfi = [i for i, l in enumerate(asm) if l.startswith("v_mfma")]
        mf = [asm[i] for i in mfi]
        assert len(mf) == 132 and all(l.startswith("v_mfma_f32_32x32x16_f16") for l in mf), (sym, len(mf))
        dst = [re.match(r"\S+\s+([av]\[\d+:\d+\])", l).group(1) for l in mf]
        # (the same registers may serve another accumulator much later, e.g. the second half of the lean pass after its
        # layer-3 block: only MFMAs within a few instructions of each other count as back to back)
        assert all(a != b or j - i > 16 for a, b, i, j in zip(dst[:-1], dst[1:], mfi[:-1], mfi[1:])), "back-to-back MFMAs on one accumulator: " + sym
        for l in mf:                                      # D = A x B + C with C = D: the accumulator chain the order protects
            regs = re.findall(r"[av]\[\d+:\d+\]", l)
            assert regs[0] == regs[-1], l
        assert ops.count("s_sleep") >= 2, sym
    # the classic kernel keeps the same gather with 24 loads in flight
    asm2 = _kernel_disassembly(lib_path, "_Z11k_shade_mlpILi3ELi12ELi4ELi8ELi2E")
    ops2 = [l.split()[0] for l in asm2]
    assert not [o for o in ops2 if o.startswith(("flat_", "scratch_"))]
    assert len([l for l in asm2 if l.startswith("v_mfma")]) == 132


def test_env_tune_is_parsed_defensively():
    """UGRID_TUNE="key=value,..." is applied at import (fourier_render): a malformed or rejected entry is warned about and skipped,
    it must not make the package unimportable."""
    import warnings
    from unboundednerfpytorch_amd import fourier_render as fr
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert fr._apply_env_tune("tv_xcd, a=b=c ,no_such_knob=1,,tv_xcd=3") == ["tv_xcd"]
    assert len(w) == 3 and all("UGRID_TUNE" in str(x.message) for x in w)
