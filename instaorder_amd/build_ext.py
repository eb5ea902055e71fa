"""Builds libmsda_hip.so in-tree with hipcc for gfx950 (no hipify, no torch cpp_extension)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "msda_hip.hip")
LIB = os.path.join(HERE, "libmsda_hip.so")
HIPCC_FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC",
               "-munsafe-fp-atomics", "-Wall", "-Wextra",
               "-I" + os.path.join(ROOT, "include")]


def find_hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libmsda_hip.so)")


def build(force=False, verbose=False):
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, "include", "msda_hip.h")]
    if (not force and os.path.exists(LIB)
            and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps)):
        return LIB
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", LIB, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
