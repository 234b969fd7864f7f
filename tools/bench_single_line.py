#!/usr/bin/env python3
"""Developer tool: latency of the one-line drop-in call (gx_extract_one_utf16 through ctypes): a launch per call (the default) and the
resident wave of GX_CREATE_RESIDENT_ONE (gx_service.hip), lines of 41 and 193 characters."""
import os, subprocess, sys, tempfile, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gorp_amd import workloads as W, _native as N
from gorp_amd.gorp import Gorp
L = N.lib()
for name, flags in (("a launch per call", 0), ("resident wave (GX_CREATE_RESIDENT_ONE)", N.GX_CREATE_RESIDENT_ONE)):
    g = Gorp.construct(W.readme3_definition(), flags=flags)
    for line in ("[123456789]: GET 12ms /index.html?x=1&y=2", "[123456789]: GET 12ms /" + "a" * 170):
        u = np.frombuffer(line.encode("utf-16-le"), np.uint16)
        mid = C.c_int32(0)
        caps = np.zeros(2 * g.max_groups, np.int32)
        for _ in range(50):
            L.gx_extract_one_utf16(g._h.ptr, u.ctypes.data, len(u), C.byref(mid), caps.ctypes.data)
        reps = 5000
        t0 = time.perf_counter()
        for _ in range(reps):
            L.gx_extract_one_utf16(g._h.ptr, u.ctypes.data, len(u), C.byref(mid), caps.ctypes.data)
        dt = (time.perf_counter() - t0) / reps
        print("%-42s %3d characters: %5.1f us per call (match_id %d, caps %s; the wave was started %d times)" %
              (name, len(line), dt * 1e6, mid.value, caps.tolist(), g.stat(28)))
    t0 = time.perf_counter()
    for _ in range(500):
        r = g.extract("[123456789]: GET 12ms /index.html?x=1&y=2")
    print("%-42s Gorp.extract (Python mirror): %.1f us per call -> %s" % (name, (time.perf_counter() - t0) / 500 * 1e6, r.asMap()))
    del g

# the same calls from C (what a JNI shim pays): tools/micro/one_line_latency.cpp against the same library
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "def.grp"), "w").write(W.README3_DEFINITION_TEXT)
exe = os.path.join(tmp, "one_line_latency")
rt = N._load_hip_runtime()._name
subprocess.check_call(["g++", "-std=c++17", "-O2", "-I", os.path.join(root, "include"), os.path.join(root, "tools", "micro", "one_line_latency.cpp"), "-o", exe,
                       N.LIB_PATH, rt, "-Wl,-rpath," + os.path.dirname(N.LIB_PATH), "-Wl,-rpath," + os.path.dirname(rt), "-Wl,--allow-shlib-undefined"])
for name, flags in (("a launch per call", 0), ("resident wave (GX_CREATE_RESIDENT_ONE)", N.GX_CREATE_RESIDENT_ONE)):
    print(name + ":")
    sys.stdout.flush()
    subprocess.check_call([exe, os.path.join(tmp, "def.grp"), str(flags)])
