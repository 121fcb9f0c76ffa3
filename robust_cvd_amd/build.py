"""Build libcvd_hip.so (hipcc, gfx950 only) in-tree under robust_cvd_amd/lib/."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# translation units of libcvd_hip.so (compiled in parallel; every unit includes cvd_host.h + the kernel headers it launches)
UNITS = ["cvd_api", "cvd_comm", "cvd_setup", "cvd_eval", "cvd_matvec", "cvd_precond", "cvd_temporal", "cvd_solve", "cvd_frontend"]
LIB = os.path.join(_HERE, "lib", "libcvd_hip.so")
OBJ = os.path.join(_HERE, "lib", "obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-cuda-compat"]
LINK = ["-L/opt/rocm/lib", "-lrccl", "-lrocsolver", "-lrocblas"]


def _deps(unit):
    """Prerequisites of a unit's object file from the compiler's own dependency file (the unit alone before the first build)."""
    d = os.path.join(OBJ, unit + ".d")
    src = os.path.join(CSRC, unit + ".hip")
    if not os.path.exists(d):
        return None
    with open(d) as f:
        txt = f.read().replace("\\\n", " ")
    out = [t for t in txt.split(":", 1)[1].split() if not t.startswith("/opt/") and not t.startswith("/usr/")]
    return out or [src]


def _stale(unit):
    o = os.path.join(OBJ, unit + ".o")
    deps = _deps(unit)
    if deps is None or not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any((not os.path.exists(d)) or os.path.getmtime(d) > t for d in deps)


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(_stale(u) or os.path.getmtime(os.path.join(OBJ, u + ".o")) > t for u in UNITS)


def _compile(unit, hipcc, verbose):
    src = os.path.join(CSRC, unit + ".hip")
    cmd = [hipcc] + FLAGS + ["-c", src, "-o", os.path.join(OBJ, unit + ".o"), "-MD", "-MF", os.path.join(OBJ, unit + ".d")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_variant(name, defines, units=None, verbose=False):
    """Development: a second library lib/libcvd_hip_<name>.so with extra -D defines (profile stamps) in the given units; the other
    units' objects are shared with the product build.  tools/ load it with api.load_library(variant=<name>)."""
    from concurrent.futures import ThreadPoolExecutor
    build(verbose=verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    units = units or UNITS
    objs = {u: os.path.join(OBJ, u + ".o") for u in UNITS}

    def comp(u):
        o = os.path.join(OBJ, f"{u}_{name}.o")
        subprocess.check_call([hipcc] + FLAGS + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, u + ".hip"), "-o", o])
        objs[u] = o
    with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
        list(pool.map(comp, units))
    out = os.path.join(_HERE, "lib", f"libcvd_hip_{name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + [objs[u] for u in UNITS] + LINK)
    return out


def build_deterministic(verbose=False):
    """lib/libcvd_hip_det.so: every translation unit with -DCVD_DETERMINISTIC=1 (cvd_kernels.h: accumulations through LDS atomics by
    ONE wave, folds in index order -- bit-reproducible solves, several times slower).  tests/test_gpu_determinism.py and
    tools/det_check.py load it with api.load_library(variant="det").  Rebuilt when a source is newer than the library."""
    out = os.path.join(_HERE, "lib", "libcvd_hip_det.so")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))] + [os.path.join(_HERE, "..", "include", "cvd_hip.h"), os.path.join(_HERE, "..", "include", "cvd_hip_debug.h")]
    if os.path.exists(out) and all(os.path.getmtime(f) <= os.path.getmtime(out) for f in srcs if os.path.exists(f)):
        return out
    return build_variant("det", ["CVD_DETERMINISTIC=1"], verbose=verbose)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    todo = [u for u in UNITS if force or _stale(u)]
    with ThreadPoolExecutor(max_workers=min(len(todo) or 1, os.cpu_count() or 1)) as pool:
        list(pool.map(lambda u: _compile(u, hipcc, verbose), todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + [os.path.join(OBJ, u + ".o") for u in UNITS] + LINK
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


PY_SRC = os.path.join(_HERE, "csrc", "lib_python.cpp")


def lib_python_path():
    import sysconfig
    return os.path.join(_HERE, "lib", "lib_python" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_lib_python(force=False, verbose=False):
    """pybind11 module `lib_python` (drop-in for the reference's module of the same name), linked against
    libcvd_hip.so with an $ORIGIN rpath.  Import it with robust_cvd_amd/lib on sys.path."""
    import pybind11
    import sysconfig
    out = lib_python_path()
    deps = [PY_SRC, LIB, os.path.join(CSRC, "cvd_device.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in deps):
        return out
    build(force=False, verbose=verbose)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-x", "hip", "-O2", "-std=c++17", "-shared", "-fPIC",
           "-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], PY_SRC, "-o", out,
           "-L" + os.path.dirname(LIB), "-lcvd_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_lib_python(force=True, verbose=True))
