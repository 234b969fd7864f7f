"""Builds gorp_amd/libgorp_hip.so in-tree with hipcc for gfx950.

    python -m gorp_amd.build [--force]

hipcc cross-compiles without a GPU present.  The .so is git-ignored but travels
with the tree (it is not listed in .gpurunignore).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgorp_hip.so")
SOURCES = ["gx_regex.cpp", "gx_compile.cpp", "gx_host.cpp", "gx_dsl.cpp", "gx_api.cpp", "gx_kernels.hip", "gx_ingest.hip", "gx_jsonl.hip"]
HEADERS = ["gx_common.hpp", "gx_compile.hpp", "gx_device.hpp", "gx_dsl.hpp", os.path.join("..", "..", "include", "gorp_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    objs = []
    common = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result"]
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc()] + common + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src.endswith(".cpp"):
            cmd[1:1] = ["-x", "hip"] if src == "gx_api.cpp" else []
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    # -no-hip-rt: do not record a NEEDED entry for a particular libamdhip64.  The process
    # must hold exactly ONE HIP runtime; PyTorch wheels bundle their own copy (different
    # SONAME from /opt/rocm's), so the loader (gorp_amd/_native.py, or the JNI shim)
    # loads the runtime that the rest of the process uses before loading this library.
    cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-no-hip-rt", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
