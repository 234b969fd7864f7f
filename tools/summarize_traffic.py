#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/traffic_target.py into
profiles/<tag>_traffic.json + filtered CSVs.  Units and the gfx950 correction follow
MI355X_MICROARCH.md (HBM section): counters are KiB; FETCH_SIZE reads exactly half of a wide coalesced
stream on gfx950 and is doubled; the copy kernel in the same run is the calibration."""
import csv, json, os, sys

tag, fetch_csv, write_csv = sys.argv[1], sys.argv[2], sys.argv[3]
fmt = sys.argv[4] if len(sys.argv) > 4 else "narrow"   # result format of the profiled launches
row_bytes = {"narrow": 9, "compact": 18, "dense": 36, "match_only": 4}[fmt]
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def collect(path, counter):
    tile, copy, keep = [], [], []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        v = float(r["Counter_Value"])
        if "k_extract_tile" in name and v > 1e4:
            tile.append(v); keep.append(r)
        elif ("copy" in name.lower() or "clone" in name.lower()) and v > 5e5:
            copy.append(v); keep.append(r)
    return tile, copy, keep


ft, fc, fk = collect(fetch_csv, "FETCH_SIZE")
wt, wc, wk = collect(write_csv, "WRITE_SIZE")
for name, rows in (("fetch", fk), ("write", wk)):
    with open(os.path.join(out_dir, "%s_pmc_%s.csv" % (tag, name)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(rows)
avg = lambda v: sum(v) / len(v)
n_lines, line_bytes = 10_000_000, 200
res = {
    "workload": "config 2: README 3-extraction definition, %d x %d B lines" % (n_lines, line_bytes),
    "kernel": "k_extract_tile<unsigned int, 13, TIER_LDS, MODE %d>" % (0 if fmt == "match_only" else 1),
    "results": fmt + " (%d B per line)" % row_bytes,
    "fetch_size_kib_raw": avg(ft), "write_size_kib": avg(wt),
    "calibration_copy": {"bytes_read": n_lines * line_bytes, "bytes_written": n_lines * line_bytes,
                         "fetch_size_kib_raw": avg(fc), "write_size_kib": avg(wc[-3:]),
                         "fetch_ratio_raw": avg(fc) * 1024 / (n_lines * line_bytes),
                         "write_ratio": avg(wc[-3:]) * 1024 / (n_lines * line_bytes)},
    "gfx950_fetch_correction": 2.0,
    "hbm_read_bytes_per_launch": avg(ft) * 1024 * 2.0,
    "hbm_write_bytes_per_launch": avg(wt) * 1024,
    "algorithmic_read_bytes": n_lines * line_bytes + 4 * (n_lines + 1),
    "algorithmic_write_bytes": n_lines * row_bytes,
}
res["traffic_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
with open(os.path.join(out_dir, "%s_traffic.json" % tag), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
