#!/usr/bin/env python3
"""Developer tool: differential soak of the hop tier on LARGE definitions (tools/fuzz_kernels.py draws small ones): random
syslog-like definitions of 8-96 extractions (what gives deep literal tries, long chains and many hot states), their lines
damaged in every way the hop walk has a branch for, through the tile kernel, the hop slice kernel and the default choice,
dense / compact / u8 rows, against the oracle.  Usage: fuzz_hop.py [definitions] [seed]"""
import os, sys, random
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import numpy as np
from gorp_amd import _native as N
from gorp_amd import workloads as W
from gorp_amd.gorp import Gorp, lines_to_csr, unpack_rows
from oracle import oracle as O

n_defs = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
for d in range(n_defs):
    n_rules, n_keys = rng.choice([8, 17, 33, 64, 96]), rng.randint(1, 7)
    if d % 4 == 3: n_rules = rng.choice([200, 320])   # more reachable states than records fit LDS: the walk that fetches records from global memory
    rules, meta = W.syslog_definition(n_rules, seed=rng.randrange(1 << 30), n_keys=n_keys)
    gorp = Gorp.construct(rules, flags=N.GX_CREATE_TIER_HOP)
    built = [e.build() for e in rules]
    orc = O.OracleGorp([b[0] for b in built], [b[1] for b in built])
    assert gorp.stat(14) > 0
    lo, hi = rng.choice([(None, None), (30, 300), (50, 2000)])
    data, off, _ = W.syslog_lines(meta, 1500, seed=rng.randrange(1 << 30), corrupt_frac=0.15, min_len=lo, max_len=hi,
                                  line_bytes=rng.choice([120, 200, 254, 400]), mixed_case=rng.random() < 0.5)   # (mixed case: the loop sets' second chance)
    lines = [bytes(data[off[i]:off[i + 1]]).decode("latin-1") for i in range(1500)]
    for i in range(0, 1500, 3):
        ln, r = lines[i], rng.random()
        if r < 0.2: ln = ln.replace(" ", rng.choice(["  ", "\t", " \t "]), rng.randint(1, 4))
        elif r < 0.4: ln = ln[:rng.randint(0, len(ln))]
        elif r < 0.6 and ln:
            k = rng.randrange(len(ln)); ln = ln[:k] + rng.choice("_Z9\x7f\xe9=[]: \x00") + ln[k + 1:]
        elif r < 0.7 and ln:
            k = rng.randrange(len(ln)); ln = ln[:k] + ln[k:k + rng.randint(1, 30)] * rng.randint(2, 9) + ln[k:]
        lines[i] = ln
    dd, oo = lines_to_csr([s.encode("latin-1") for s in lines])
    omid, ocaps = orc.extract_batch(dd, oo, nthreads=8)
    for kernel in (N.GX_KERNEL_HOPS, N.GX_KERNEL_HOP_SLICES, N.GX_KERNEL_AUTO):
        for compact in ((False, True, 2) if n_rules <= 126 else (False, True)):   # (u8 rows hold match ids up to 126)
            if compact:
                rows, over = gorp.extract_batch(dd, oo, kernel=kernel, compact=compact)
                mid, caps = unpack_rows(rows)
                lim = 254 if compact == 2 else 65534
                want = np.where(ocaps > lim, lim, ocaps)
                okay = over == int((ocaps > lim).sum())
            else:
                mid, caps = gorp.extract_batch(dd, oo, kernel=kernel)
                want, okay = ocaps, True
            if not (okay and np.array_equal(mid, omid) and np.array_equal(caps, want)):
                bad += 1
                i = int(np.nonzero((mid != omid) | (caps != want).any(axis=1))[0][0]) if not np.array_equal(mid, omid) or not np.array_equal(caps, want) else -1
                print("MISMATCH definition", d, "rules", n_rules, "keys", n_keys, "kernel", kernel, "rows", compact, "line", i, repr(lines[i]) if i >= 0 else "(overflow count)")
    print("definition %d: %d extractions x %d keys, %d states (%d hot, %d reachable), lines %s: ok so far %s" %
          (d, n_rules, n_keys, gorp.stat(14), gorp.stat(15), gorp.stat(16), (lo, hi), bad == 0), flush=True)
print("hop fuzz: %d definitions, %d mismatches" % (n_defs, bad))
sys.exit(1 if bad else 0)
