"""oracle/_ref -- the reference's OWN native extension modules, compiled as TEST INFRASTRUCTURE only.

The product never loads anything built here (tests/, tools/gen_native_golden.py and nothing else do).  Purpose:
pin `oracle/ref_ops.c` (the C restatement) and the HIP library on outputs of the reference's own kernels
(`FourierGrid/cuda/*.cu`), see tests/golden/native_ops.npz and tests/test_gpu_ref_native.py.

Recipe (run in the build container, where /root/reference exists):

    python oracle/build_ref.py            # -> oracle/_ref/{nofma,fma}/<module>.so   (git-ignored, shipped by gpurun)

* the four modules of FourierGrid/cuda/setup.py:15-18 are built from the sources where they lie; torch's
  cpp_extension (ROCm build) hipifies `.cu` next to the file it reads, and /root/reference is read-only, so the
  sources are first copied to a throw-away directory under $TMPDIR (never into the repository);
* ONE mechanical edit is applied to that throw-away copy: `AT_DISPATCH_FLOATING_TYPES(x.type(), ...)` ->
  `x.scalar_type()` -- torch >= 2.x removed the implicit DeprecatedTypeProperties -> ScalarType conversion
  (the first attempt's failing log is profiles/r02/oracle_ref_build_fail.log).  No arithmetic is touched;
* two builds: `nofma` (-ffp-contract=off: every a*b+c rounded twice, the semantics the C restatement follows) and
  `fma` (hipcc's default contraction, the analogue of nvcc's default -fmad=true).  Where the two differ the
  reference itself is ambiguous to 1 ulp; the tests demand bit equality against `nofma` and report the
  distance to `fma`.
"""
import glob
import os
import re
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "oracle", "_ref")
MODULES = [  # FourierGrid/cuda/setup.py:15-18
    ("adam_upd_cuda", ["adam_upd.cpp", "adam_upd_kernel.cu"]),
    ("ub360_utils_cuda", ["ub360_utils.cpp", "ub360_utils_kernel.cu"]),
    ("total_variation_cuda", ["total_variation.cpp", "total_variation_kernel.cu"]),
    ("render_utils_cuda", ["render_utils.cpp", "render_utils_kernel.cu"]),
]
VARIANTS = {"nofma": ["-ffp-contract=off"], "fma": []}


def reference_cuda_dir():
    root = os.environ.get("UNERF_REFERENCE_ROOT", "/root/reference")
    d = os.path.join(root, "FourierGrid", "cuda")
    return d if os.path.isdir(d) else None


def built(variant="nofma"):
    return all(os.path.exists(os.path.join(OUT, variant, name + ".so")) for name, _ in MODULES)


def build(variants=("nofma", "fma"), verbose=False):
    """One subprocess per variant: torch's JIT loader renames a module that the process has already loaded
    (name_v1), which would break the PyInit symbol of the second variant."""
    import subprocess
    for variant in variants:
        if not built(variant):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--variant", variant] + (["-v"] if verbose else []))


def _build_variant(variant, verbose=False):
    src = reference_cuda_dir()
    if src is None:
        raise RuntimeError("reference sources not present (UNERF_REFERENCE_ROOT / /root/reference)")
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    from torch.utils import cpp_extension
    if True:
        work = tempfile.mkdtemp(prefix="unerf_ref_%s_" % variant)
        try:
            for f in glob.glob(os.path.join(src, "*.cpp")) + glob.glob(os.path.join(src, "*.cu")):
                dst = os.path.join(work, os.path.basename(f))
                shutil.copyfile(f, dst)
                if dst.endswith(".cu"):
                    txt = open(dst).read()
                    txt = re.sub(r"(AT_DISPATCH_FLOATING_TYPES\(\s*\w+)\.type\(\)", r"\1.scalar_type()", txt)
                    open(dst, "w").write(txt)
            os.makedirs(os.path.join(OUT, variant), exist_ok=True)
            for name, files in MODULES:
                bdir = os.path.join(work, "build_" + name)
                os.makedirs(bdir)
                cpp_extension.load(name=name, sources=[os.path.join(work, f) for f in files], build_directory=bdir,
                                   extra_cuda_cflags=VARIANTS[variant] + ["-w"], extra_cflags=["-w"],
                                   verbose=verbose, is_python_module=False)
                shutil.copyfile(os.path.join(bdir, name + ".so"), os.path.join(OUT, variant, name + ".so"))
        finally:
            shutil.rmtree(work, ignore_errors=True)


REFERENCE_PY = ["__init__.py", "FourierGrid_model.py", "FourierGrid_grid.py", "grid.py", "dvgo.py", "dcvgo.py", "dmpigo.py",
                "masked_adam.py", "utils.py"]


def stage_reference_python():
    """oracle/_ref/reference_py.tar: the reference's own model files (the callers of the four extension modules), so
    that tests/test_gpu_reference_callers.py can run them UNCHANGED over the HIP modules on the GPU box, where
    /root/reference does not exist.  Like the .so files beside it the archive is git-ignored (never part of the
    repository's history) and shipped by gpurun; the test unpacks it into a temporary directory."""
    import tarfile
    root = os.environ.get("UNERF_REFERENCE_ROOT", "/root/reference")
    src = os.path.join(root, "FourierGrid")
    if not os.path.isdir(src):
        raise RuntimeError("reference sources not present")
    os.makedirs(OUT, exist_ok=True)
    dst = os.path.join(OUT, "reference_py.tar")
    with tarfile.open(dst, "w") as tf:
        for f in REFERENCE_PY:
            tf.add(os.path.join(src, f), arcname=os.path.join("FourierGrid", f))
    return dst


def reference_python_root():
    """Directory to put on sys.path so that `import FourierGrid.<module>` finds the reference's files: the reference tree
    itself when present, else the staged archive unpacked into a temporary directory, else None."""
    root = os.environ.get("UNERF_REFERENCE_ROOT", "/root/reference")
    if os.path.isdir(os.path.join(root, "FourierGrid")):
        return root
    tar = os.path.join(OUT, "reference_py.tar")
    if not os.path.exists(tar):
        return None
    import tarfile
    d = tempfile.mkdtemp(prefix="unerf_refpy_")
    with tarfile.open(tar) as tf:
        tf.extractall(d)
    return d


_LOADED = {}


def load(variant="nofma"):
    """Import the four prebuilt modules (needs a GPU at call time, not at import time).  Returns a dict, cached per variant.

    Two precautions, both about CPython's cache of single-phase-init extension modules (pybind11 modules are such): loading a
    cached .so AGAIN re-populates whatever module object sits in sys.modules under the module's NAME with the cached functions
    (import.c: import_find_extension -> import_add_module(name) + dict update) -- and the product registers ITS drop-in modules
    under exactly these names (compat.install_as_reference_extensions).  So (1) every variant is loaded once per process, and
    (2) sys.modules' entries of these names are set aside while the loader runs and put back afterwards: the checker must
    never be able to overwrite the thing it checks."""
    import importlib.util
    import torch  # noqa: F401  (the modules link against libtorch)
    if variant in _LOADED:
        return dict(_LOADED[variant])
    mods = {}
    for name, _ in MODULES:
        path = os.path.join(OUT, variant, name + ".so")
        saved = sys.modules.pop(name, None)
        try:
            spec = importlib.util.spec_from_file_location(name, path)
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)
        finally:
            sys.modules.pop(name, None)
            if saved is not None:
                sys.modules[name] = saved
        mods[name] = m
    _LOADED[variant] = mods
    return dict(mods)


if __name__ == "__main__":
    if "--variant" in sys.argv:
        _build_variant(sys.argv[sys.argv.index("--variant") + 1], verbose="-v" in sys.argv)
    else:
        build(verbose="-v" in sys.argv)
        print("built:", sorted(glob.glob(os.path.join(OUT, "*", "*.so"))), stage_reference_python())
