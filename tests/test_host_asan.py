"""The host-side compiler (definition language -> regexps -> automata -> table blob -> hop tier tables; no HIP involved) built with
AddressSanitizer + UndefinedBehaviorSanitizer and run over the golden definitions and mutated copies of them.
(GPU AddressSanitizer is not available on the GPU pool: sanitizers run on the CPU build only.)"""
import json
import os
import random
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gorp_amd", "csrc")


def _golden_definitions():
    out = []

    def walk(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k == "def" and isinstance(v, str):
                    out.append(v)
                else:
                    walk(v)
        elif isinstance(o, list):
            for v in o:
                walk(v)

    gdir = os.path.join(ROOT, "tests", "golden")
    for f in sorted(os.listdir(gdir)):
        if f.endswith(".json"):
            walk(json.load(open(os.path.join(gdir, f))))
    return out


def test_host_compiler_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_driver")
    srcs = [os.path.join(ROOT, "tests", "cpp", "asan_driver.cpp")] + [os.path.join(CSRC, f) for f in
                                                                        ("gx_regex.cpp", "gx_compile.cpp", "gx_host.cpp", "gx_dsl.cpp", "gx_json.cpp", "gx_hop.cpp")]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I", CSRC, "-I",
           os.path.join(ROOT, "include")] + srcs + ["-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    defs = _golden_definitions()
    assert len(defs) >= 8
    rng = random.Random(11)
    alphabet = "%@$(){}[]\\\"'#\n\t |*+?.,:=abc019"
    texts = list(defs)
    for _ in range(400):   # mutated copies: the parser's error paths
        t = list(rng.choice(defs))
        for _ in range(rng.randint(1, 6)):
            if not t:
                break
            op, pos = rng.random(), rng.randrange(len(t))
            if op < 0.4:
                del t[pos]
            elif op < 0.8:
                t.insert(pos, rng.choice(alphabet))
            else:
                t[pos] = rng.choice(alphabet)
        texts.append("".join(t))
    # one definition whose extraction is too ambiguous for an automaton built ahead of time (blank-separated fields that may be empty:
    # gx_compile.cpp keeps its program, blob version 3) -- packed, unpacked and damaged like the others
    texts.append("pattern %f \\S*\nextract fields {\n  template " + " ".join("$f%d(%%f)" % k for k in range(13)) + "\n}\n")
    paths = []
    for i, t in enumerate(texts):
        p = tmp_path / ("d%04d.grp" % i)
        p.write_text(t, encoding="utf-8")
        paths.append(str(p))
    r = subprocess.run([exe] + paths, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert r.returncode == 0 and "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
    assert "asan driver: " in out
    compiled = int(out.split("asan driver: ")[1].split()[0])
    assert compiled >= len(defs) + 1   # every golden definition compiles, and the one that is run as a program
    # the hop tier's tables (gx_hop.cpp), walked on the host as the kernel walks them, agree with the dense automaton
    hop_defs, hop_lines = [int(x) for x in out.split("hop tier: ")[1].replace(" definitions,", "").split()[:2]]
    assert hop_defs >= len(defs) // 2 and hop_lines >= 400 * hop_defs
    # ... and some of those lines went through a loop set's second chance (\w fields hold digits and upper case)
    assert int(out.split("lines agree with the dense automaton (")[1].split()[0]) > 0
