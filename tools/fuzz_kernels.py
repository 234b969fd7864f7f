#!/usr/bin/env python3
"""Developer tool: a longer differential run than the test-suite's -- random definitions through every kernel
variant (tile, slice, lane and per-line kernels; dense rows in LDS / L2, range records in LDS / global memory; fused and
two-pass layouts; dense and compact result rows -- the `variants` list below) against the oracle.  Usage: fuzz_kernels.py [defs] [seed]"""
import os, sys, random
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_compiler_vs_oracle as TC
from blob_interp import Blob
from gorp_amd import _native as N
from gorp_amd.gorp import Gorp, FlattenedExtraction, lines_to_csr, unpack_rows
from oracle import oracle as O

n_defs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 2024
rng = random.Random(seed)
# (create flags, kernel, compact result rows)
variants = [(0, 0, False), (N.GX_CREATE_TIER_L2, 0, False), (0, N.GX_KERNEL_SLICES, False), (N.GX_CREATE_TIER_L2, N.GX_KERNEL_SLICES, False),
            (N.GX_CREATE_NO_TILES, 0, False), (N.GX_CREATE_NO_FUSED, 0, False), (N.GX_CREATE_NO_FUSED | N.GX_CREATE_TIER_L2, 0, False),
            (0, 0, True), (N.GX_CREATE_TIER_L2, 0, True), (0, N.GX_KERNEL_SLICES, True), (N.GX_CREATE_NO_FUSED, 0, True),
            (N.GX_CREATE_TIER_RECORDS, 0, False), (N.GX_CREATE_TIER_RECORDS, N.GX_KERNEL_SLICES, False),   # (records in LDS: kernel 0 = the lane kernel)
            (N.GX_CREATE_TIER_RECORDS | N.GX_CREATE_NO_FUSED, 0, False), (N.GX_CREATE_TIER_RECORDS, 0, True),
            (N.GX_CREATE_TIER_RECORDS, N.GX_KERNEL_TILES, False), (N.GX_CREATE_TIER_RECORDS, N.GX_KERNEL_TILES, True),
            (N.GX_CREATE_TIER_L2, N.GX_KERNEL_LANES, False), (N.GX_CREATE_TIER_RECORDS_GLOBAL, N.GX_KERNEL_LANES, True),
            (0, N.GX_KERNEL_LANES, False), (0, N.GX_KERNEL_LANES, True),
            (N.GX_CREATE_TIER_RECORDS_GLOBAL, 0, False), (N.GX_CREATE_TIER_RECORDS_GLOBAL, N.GX_KERNEL_SLICES, False),
            (N.GX_CREATE_TIER_RECORDS_GLOBAL | N.GX_CREATE_NO_FUSED, 0, True),
            # the hop tier (run + chain records over dense rows in global memory; the tile kernel on class ids), where the
            # definition is within its limits (else the handle's other tables answer); 2 = u8 result rows
            (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOPS, False), (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOPS, True), (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOPS, 2),
            (N.GX_CREATE_TIER_HOP | N.GX_CREATE_TIER_L2, N.GX_KERNEL_HOPS, False), (0, 0, 2), (N.GX_CREATE_TIER_RECORDS, 0, 2),
            (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOP_SLICES, False), (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOP_SLICES, True), (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOP_SLICES, 2),
            # round 4: the same lines as UTF-16 code units (a fourth entry: the tile kernel reads the units itself on the dense-rows-in-LDS
            # and hop tiers; a few units above 0xFF are mixed in below, whose lines the per-line walk takes again)
            (0, 0, False, True), (0, 0, True, True), (0, 0, 2, True), (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOPS, False, True), (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOPS, 2, True),
            (N.GX_CREATE_TIER_L2, 0, True, True),
            # ... and by the hop slice kernel
            (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOP_SLICES, False, True), (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOP_SLICES, True, True), (N.GX_CREATE_TIER_HOP, N.GX_KERNEL_HOP_SLICES, 2, True)]
done = bad = 0
while done < n_defs:
    exts = [FlattenedExtraction("e%d" % i, TC.gen_pieces(rng)) for i in range(rng.randint(1, 5))]
    try:
        built = [e.build() for e in exts]
        orc = O.OracleGorp([b[0] for b in built], [b[1] for b in built])
        env = variants[done % len(variants)]
        gorp = Gorp.construct(exts, flags=env[0])
    except Exception:
        continue
    b = Blob(gorp.blob())
    lines = [TC.gen_line(rng) for _ in range(40)] + [TC.sample_from_match_automaton(b, rng) for _ in range(88)]
    lines += [ln * rng.randint(2, 40) for ln in lines[:20]]  # longer lines: several windows / slices
    raw = [ln.encode("latin-1") if isinstance(ln, str) else ln for ln in lines]
    data, offsets = lines_to_csr(raw)
    if len(env) > 3:   # UTF-16: the bytes as code units, and in one line in sixteen a unit above 0xFF (the oracle walks the units too)
        data = data.astype(np.uint16)
        for i in range(0, len(raw), 16):
            if offsets[i + 1] > offsets[i]:
                data[rng.randrange(int(offsets[i]), int(offsets[i + 1]))] = rng.choice([0x100, 0x20AC, 0xFF41, 0x4E2D])
    if env[2] and gorp.max_groups > 0:
        rows, over = gorp.extract_batch(data, offsets, kernel=env[1], compact=env[2])
        mid, caps = unpack_rows(rows)
        if env[2] == 2:  # u8 rows: offsets above 254 are stored as 254 and counted
            assert over >= int((oracle_batch()[1] > 254).sum()) if len(env) > 3 else over == int((oracle_batch()[1] > 254).sum())   # (UTF-16: a line with a wide unit is counted by both walks)
        else:
            assert over == 0
    else:
        mid, caps = gorp.extract_batch(data, offsets, kernel=env[1])
    def oracle_batch():
        if len(env) <= 3:
            return orc.extract_batch(data, offsets, nthreads=4)
        om = np.zeros(len(raw), np.int32)   # (code units: the oracle line by line, on the Strings)
        oc = np.full((len(raw), 2 * gorp.max_groups), -1, np.int32)
        for i in range(len(raw)):
            u = data[int(offsets[i]):int(offsets[i + 1])]
            m, groups = orc.extract(u.tobytes().decode("utf-16-le", "surrogatepass"))
            om[i] = m
            for g, span in enumerate(groups):
                if span is not None:
                    oc[i, 2 * g], oc[i, 2 * g + 1] = span
        return om, oc
    omid, ocaps = oracle_batch()
    if env[2] == 2:
        ocaps = np.where(ocaps > 254, 254, ocaps)
    if not (np.array_equal(mid, omid) and np.array_equal(caps, ocaps)):
        bad += 1
        i = int(np.nonzero((mid != omid) | (caps != ocaps).any(axis=1))[0][0])
        print("MISMATCH variant", env, "line", repr(raw[i]), "gpu", mid[i], caps[i].tolist(), "oracle", omid[i], ocaps[i].tolist())
        print("  definition:", [(e.name, e.pieces) for e in exts])
    m2, _ = gorp.extract_batch(data, offsets, match_only=True, kernel=env[1])
    want_m = np.where(omid <= -2, -2 - omid, omid)  # match-only reports the matcher's choice
    if not np.array_equal(m2, want_m):
        bad += 1
        print("MATCH-ONLY MISMATCH variant", env)
    done += 1
    if done % 500 == 0:
        print("fuzz: %d definitions so far, %d mismatches" % (done, bad), flush=True)
print("fuzz: %d definitions, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
