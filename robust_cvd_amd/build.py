"""Build libcvd_hip.so (hipcc, gfx950 only) in-tree under robust_cvd_amd/lib/."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "cvd_hip.hip")
DEPS = [SRC] + sorted(os.path.join(_HERE, "csrc", f) for f in os.listdir(os.path.join(_HERE, "csrc")) if f.endswith(".h")) + \
       [os.path.join(_HERE, "..", "include", "cvd_hip.h"), os.path.join(_HERE, "..", "include", "cvd_types.h")]
LIB = os.path.join(_HERE, "lib", "libcvd_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-munsafe-fp-atomics"]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + ["-o", LIB, SRC, "-L/opt/rocm/lib", "-lrccl", "-lrocsolver", "-lrocblas"]
    if verbose:
        print(" ".join(cmd))
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
    deps = [PY_SRC, LIB, os.path.join(_HERE, "csrc", "cvd_device.h")]
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
