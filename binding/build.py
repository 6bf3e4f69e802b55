"""Build the four pybind modules of binding/ (render_utils_cuda, total_variation_cuda, ub360_utils_cuda, adam_upd_cuda) against
the C ABI of libugrid_hip.so -- INTEGRATION.md section B as code.  The reference's FourierGrid/cuda/setup.py:15-19 lists one
CUDAExtension per module; here each module is ONE host-only C++ file (no device code: no hipcc needed for them) linked with
-lugrid_hip, with an rpath to the library's directory.  Output: binding/_build/ugrid_<name>.so (git-ignored, shipped by gpurun).

    python binding/build.py            # builds what is out of date (source hash stamp)
    import binding.build as b; mods = b.load()      # -> {name: module}, importable under the reference's names
"""
import hashlib
import importlib.util
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(HERE, "_build")
LIBDIR = os.path.join(ROOT, "unboundednerfpytorch_amd")
MODULES = {"render_utils_cuda": "render_utils.cpp", "total_variation_cuda": "total_variation.cpp", "ub360_utils_cuda": "ub360_utils.cpp",
           "adam_upd_cuda": "adam_upd.cpp"}
# The extension modules are BUILT under a prefixed name (PyInit_ugrid_render_utils_cuda in ugrid_render_utils_cuda.so) and
# REGISTERED under the reference's names by install().  Inside the reference tree a maintainer would build them under the plain
# names (setup.py); here up to three same-named pybind modules live in one test process (oracle/_ref/{nofma,fma} = the reference's
# own kernels, and these), and CPython's cache of single-phase-init extension modules re-populates whatever module sits in
# sys.modules under a NAME when a cached .so of that name is loaded again -- the prefix keeps the families apart.
PREFIX = "ugrid_"


def _hash(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(open(p, "rb").read())
    return h.hexdigest()


def _flags():
    import torch
    from torch.utils import cpp_extension as ce
    inc = ce.include_paths(device_type="cuda") if "device_type" in ce.include_paths.__code__.co_varnames else ce.include_paths(cuda=True)
    cflags = ["-O2", "-fPIC", "-shared", "-std=c++17", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_API_INCLUDE_EXTENSION_H",
              "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations"]
    incs = ["-I" + os.path.join(ROOT, "include"), "-I" + sysconfig.get_paths()["include"]] + ["-isystem" + p for p in inc]
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    libs = ["-L" + tl, "-Wl,-rpath," + tl, "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-ltorch_python",
            "-L" + LIBDIR, "-Wl,-rpath,$ORIGIN/../../unboundednerfpytorch_amd", "-Wl,-rpath," + LIBDIR, "-lugrid_hip", "-lugrid_hip_f64", "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib", "-lamdhip64"]
    return cflags, incs, libs


def build(force=False, verbose=False):
    os.makedirs(OUT, exist_ok=True)
    cflags, incs, libs = _flags()
    built = []
    procs = []
    for name, src in MODULES.items():
        srcs = [os.path.join(HERE, src), os.path.join(HERE, "ugrid_binding_common.h"), os.path.join(ROOT, "include", "ugrid_hip.h"),
                os.path.join(ROOT, "include", "ugrid_hip_f64.h")]
        so = os.path.join(OUT, PREFIX + name + ".so")
        stamp = so + ".srchash"
        want = _hash(srcs)
        if not force and os.path.exists(so) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            continue
        cmd = ["g++"] + cflags + ["-DTORCH_EXTENSION_NAME=" + PREFIX + name] + incs + [srcs[0], "-o", so] + libs
        if verbose:
            print(" ".join(cmd))
        procs.append((name, stamp, want, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for name, stamp, want, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode != 0:
            raise RuntimeError("binding/%s failed to build:\n%s" % (name, out[-4000:]))
        open(stamp, "w").write(want + "\n")
        built.append(name)
    return built


def available():
    return all(os.path.exists(os.path.join(OUT, PREFIX + n + ".so")) for n in MODULES)


def load(names=None):
    """import the built modules (torch and libugrid_hip.so must be loadable); does NOT touch sys.modules"""
    import torch  # noqa: F401  (libtorch symbols)
    mods = {}
    for name in (names or MODULES):
        if PREFIX + name in sys.modules:
            mods[name] = sys.modules[PREFIX + name]
            continue
        spec = importlib.util.spec_from_file_location(PREFIX + name, os.path.join(OUT, PREFIX + name + ".so"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        sys.modules[PREFIX + name] = m
        mods[name] = m
    return mods


def install(names=None):
    """register the native modules under the names the reference imports (`import render_utils_cuda`, ...)"""
    mods = load(names)
    sys.modules.update(mods)
    return mods


if __name__ == "__main__":
    print("built:", build(force="--force" in sys.argv, verbose="-v" in sys.argv) or "nothing (up to date)")
