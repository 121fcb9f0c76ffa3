"""Build libcvd_hip.so (hipcc, gfx950 only) in-tree under robust_cvd_amd/lib/."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "cvd_hip.hip")
DEPS = [SRC, os.path.join(_HERE, "csrc", "cvd_kernels.h"), os.path.join(_HERE, "csrc", "cvd_device.h"),
        os.path.join(_HERE, "..", "include", "cvd_hip.h"), os.path.join(_HERE, "..", "include", "cvd_types.h")]
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
    cmd = [hipcc] + FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
