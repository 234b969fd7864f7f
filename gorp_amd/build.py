"""Builds gorp_amd/libgorp_hip.so in-tree with hipcc for gfx950.

    python -m gorp_amd.build [--force] [--dev]

hipcc cross-compiles without a GPU present.  The .so is git-ignored but travels
with the tree (it is not listed in .gpurunignore).

--dev additionally builds libgorp_hip_dev.so (-DGX_DEV): the same library plus
the tile kernel's per-phase cycle counters, for tools/phase_cycles.py.  The
product library carries no developer hooks and reads no environment variables.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgorp_hip.so")
LIB_DEV = os.path.join(HERE, "libgorp_hip_dev.so")
SOURCES = ["gx_tile_lds.hip", "gx_tile_l2.hip", "gx_tile_rec.hip", "gx_tile_recg.hip", "gx_tile_lds_w.hip", "gx_tile_hop.hip", "gx_tile_hop_w.hip", "gx_hop.cpp", "gx_kernels.hip", "gx_lanes.hip", "gx_jsonl.hip", "gx_api.cpp",
           "gx_compile.cpp", "gx_dsl.cpp", "gx_json.cpp", "gx_regex.cpp", "gx_host.cpp", "gx_tile.hip", "gx_ingest.hip", "gx_service.hip"]   # (the long ones first)
HEADERS = ["gx_common.hpp", "gx_compile.hpp", "gx_device.hpp", "gx_dsl.hpp", "gx_walk.hpp", "gx_tile_body.hpp", "gx_hop.hpp", "gx_hop_dev.hpp",
           os.path.join("..", "..", "include", "gorp_hip.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    for f in SOURCES + HEADERS:
        if os.path.getmtime(os.path.join(CSRC, f)) > t:
            return True
    return False


def build(force=False, verbose=False, dev=False, variant=None):
    """variant: (name, [defines]) -- an experiment's build of the product library as libgorp_hip_<name>.so (tools/ab_bench.py compares
    such builds on one device)."""
    lib = LIB_DEV if dev else LIB
    if variant:
        lib = os.path.join(HERE, "libgorp_hip_%s.so" % variant[0])
    if not force and not needs_build(lib):
        return lib
    common = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wall", "-Wno-unused-result"]
    if dev:
        common.append("-DGX_DEV")
        common += os.environ.get("GX_BUILD_DEFINES", "").split()   # (experiments of the developer build: e.g. -DGX_HOP_SLICE_BYTES=256)
    if variant:
        common += list(variant[1])
    jobs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, ("dev_" if dev else "") + (variant[0] + "_" if variant else "") + os.path.splitext(src)[0] + ".o")
        cmd = [_hipcc()] + common + ["-c", os.path.join(CSRC, src), "-o", obj]
        if src == "gx_api.cpp":
            cmd[1:1] = ["-x", "hip"]
        if src.startswith("gx_tile_"):
            # the tile kernel draws its next tile from a counter in global memory a whole walk before it needs the answer; the
            # compiler's atomic optimizer would rewrite that draw into a wave reduction that WAITS for the answer on the spot --
            # i.e. for every load of the prefetch issued just before it
            cmd[1:1] = ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
        jobs.append((cmd, obj))

    def run(job):
        if verbose:
            print(" ".join(job[0]), flush=True)
        subprocess.check_call(job[0])
        return job[1]

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        objs = list(pool.map(run, jobs))
    # -no-hip-rt: do not record a NEEDED entry for a particular libamdhip64.  The process
    # must hold exactly ONE HIP runtime; PyTorch wheels bundle their own copy (different
    # SONAME from /opt/rocm's), so the loader (gorp_amd/_native.py, or the JNI shim)
    # loads the runtime that the rest of the process uses before loading this library.
    cmd = [_hipcc(), "-shared", "-fPIC", "--offload-arch=gfx950", "-no-hip-rt", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--dev" in sys.argv:
        print(build(force="--force" in sys.argv, verbose=True, dev=True))
    if "--variant" in sys.argv:   # --variant NAME -DX -DY=1 ...
        at = sys.argv.index("--variant")
        print(build(force=True, verbose=False, variant=(sys.argv[at + 1], [a for a in sys.argv[at + 2:] if a.startswith("-D")])))
