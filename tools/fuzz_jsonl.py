#!/usr/bin/env python3
"""Developer tool: differential soak of result materialisation (gx_results_to_jsonl: the sizes pass, the scan, the write pass with its
verbatim and its escaping path, rounds, the whole-wave fallback) against oracle.results_to_jsonl -- random definitions (the generator of
tests/test_compiler_vs_oracle.py, extractor names that collide now and then, `append` objects), their lines salted with the characters
JSON escapes (quotes, backslashes, control characters, bytes >= 0x80), batches that mix 64-line tiles with and without escapes and lines
of very different lengths, id_as and utf8_passthrough on and off.  Usage: fuzz_jsonl.py [definitions] [seed]"""
import os, sys, random, json
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_compiler_vs_oracle as TC
from blob_interp import Blob
from gorp_amd.gorp import Gorp, FlattenedExtraction, lines_to_csr
from oracle import oracle as O

n_defs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 99)
SALT = ["\"", "\\", "\x00", "\x07", "\t", "\n", "\r", "\x1f", "\x7f", "\x80", "\xe9", "\xff", "\\\"", "\"\""]


def rename(pieces, names):
    out = []
    for p in pieces:
        if p[0] == "extractor":
            out.append(["extractor", rng.choice(names), rename(p[2], names)])
        else:
            out.append(p)
    return out


done = bad = 0
while done < n_defs:
    names = ["a", "b", "c", "id", "x\"y", "k\\", "é"][:rng.randint(2, 7)]
    exts = []
    for i in range(rng.randint(1, 4)):
        extra = None
        if rng.random() < 0.4:
            extra = {rng.choice(["env", "a", "id", "n"]): rng.choice(["prod", 3, True, None, 2.5, {"k": [1, "q\"", None]}]) for _ in range(rng.randint(1, 3))}
        exts.append(FlattenedExtraction("e%d\"" % i if rng.random() < 0.2 else "e%d" % i, rename(TC.gen_pieces(rng), names), extra))
    try:
        built = [e.build() for e in exts]
        gorp = Gorp.construct(exts)
    except Exception:
        continue
    b = Blob(gorp.blob())
    base = [TC.sample_from_match_automaton(b, rng) for _ in range(64)] + [TC.gen_line(rng) for _ in range(16)]
    base = [ln if isinstance(ln, str) else ln.decode("latin-1") for ln in base]
    lines = []
    for tile in range(rng.randint(3, 12)):
        kind = rng.choice(["clean", "clean", "salted", "long", "mixed"])
        for j in range(64):
            ln = rng.choice(base)
            if kind == "salted" or (kind == "mixed" and rng.random() < 0.05):
                k = rng.randrange(len(ln) + 1)
                ln = ln[:k] + rng.choice(SALT) + ln[k:]
            if kind == "long" and rng.random() < 0.3:
                ln = ln * rng.randint(2, 60)                 # rounds; now and then a line for the whole-wave path
            lines.append(ln)
    lines = lines[:len(lines) - rng.randrange(64)]              # a partial last tile
    raw = [ln.encode("latin-1") for ln in lines]
    data, offsets = lines_to_csr(raw)
    mid, caps = gorp.extract_batch(data, offsets)
    xs = gorp.getExtractions()
    for id_as, pt in ((None, False), ("id", False), ("_k", True)):
        text, loff = gorp.results_to_jsonl(data, offsets, mid, caps, id_as=id_as, utf8_passthrough=pt, want_line_offsets=True)
        want, woff = O.results_to_jsonl(raw, mid, caps, [x.getName() for x in xs], [x._extractorNames for x in xs], [x.getExtra() for x in xs],
                                        id_as=id_as, utf8_passthrough=pt)
        if text != want or not np.array_equal(loff, woff):
            bad += 1
            k = next((q for q in range(min(len(text), len(want))) if text[q] != want[q]), min(len(text), len(want)))
            print("MISMATCH definition", done, "id_as", id_as, "passthrough", pt, "at byte", k, "of", len(want), ":", text[max(0, k - 40):k + 40], "|", want[max(0, k - 40):k + 40])
            print("  definition:", [(e.name, e.pieces) for e in exts])
        elif not pt and done % 20 == 0:
            for t in text.decode("utf-8").split("\n")[:20]:
                if t:
                    json.loads(t)
    # the whole pipeline on the same lines as one text (gx_text_to_jsonl: the sizes from the split pass's escape bits, or -- a control
    # character in the text -- from the text): line feeds and carriage returns inside a line would be line ends there
    tl = [ln.replace("\n", "\t").replace("\r", "\t").encode("latin-1") for ln in lines]
    if done % 3 == 0:
        tl = [ln.replace(b"\x00", b"a").replace(b"\x07", b"b").replace(b"\x1f", b"c") for ln in tl]   # (no six-byte escapes: the bits path)
    rawtext = b"\n".join(tl) + b"\n"
    orc = O.OracleGorp([q[0] for q in built], [q[1] for q in built])
    _, wl, _ = O.read_lines(rawtext)
    omid, ocaps = orc.extract_batch(*lines_to_csr(wl), nthreads=4)
    for id_as, pt in ((None, False), ("_k", True)):
        text2, nl, nm, nx = gorp.text_to_jsonl(rawtext, id_as=id_as, utf8_passthrough=pt)
        want2, _ = O.results_to_jsonl(wl, omid, ocaps, [x.getName() for x in xs], [x._extractorNames for x in xs], [x.getExtra() for x in xs],
                                      id_as=id_as, utf8_passthrough=pt)
        if text2 != want2 or nl != len(wl) or nm != int((omid >= 0).sum()):
            bad += 1
            print("TEXT MISMATCH definition", done, "id_as", id_as, "passthrough", pt, len(text2), len(want2))
            print("  definition:", [(e.name, e.pieces) for e in exts])
    done += 1
    if done % 50 == 0:
        print("jsonl fuzz: %d definitions so far, %d mismatches" % (done, bad), flush=True)
print("jsonl fuzz: %d definitions, %d mismatches" % (done, bad))
sys.exit(1 if bad else 0)
