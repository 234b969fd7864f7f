#!/usr/bin/env python3
"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/traffic_target.py into
profiles/<tag>_traffic.json + filtered CSVs.  Units and the gfx950 correction follow
MI355X_MICROARCH.md (HBM section): counters are KiB; FETCH_SIZE reads exactly half of a wide coalesced
stream on gfx950 and is doubled; the copy kernel in the same run is the calibration.

    summarize_traffic.py <tag> <fetch.csv> <write.csv> <meta.json written by traffic_target.py>"""
import csv, json, os, sys

tag, fetch_csv, write_csv, meta_path = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
meta = json.load(open(meta_path))
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def collect(path, counter):
    kern, copy, keep = [], [], []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"]
        v = float(r["Counter_Value"])
        if meta["kernel_filter"] in name and v > 1e4:
            kern.append(v); keep.append(r)
        elif ("copy" in name.lower() or "clone" in name.lower()) and v > 5e5:
            copy.append(v); keep.append(r)
    return kern, copy, keep


ft, fc, fk = collect(fetch_csv, "FETCH_SIZE")
wt, wc, wk = collect(write_csv, "WRITE_SIZE")
for name, rows in (("fetch", fk), ("write", wk)):
    with open(os.path.join(out_dir, "%s_pmc_%s.csv" % (tag, name)), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader(); w.writerows(rows)
avg = lambda v: sum(v) / len(v)
cb = meta["copy_bytes"]
res = {
    "workload": meta["workload"],
    "kernel": meta["kernel_filter"],
    "results": meta["results"],
    "fetch_size_kib_raw": avg(ft), "write_size_kib": avg(wt),
    "calibration_copy": {"bytes_read": cb, "bytes_written": cb,
                         "fetch_size_kib_raw": avg(fc[-3:]), "write_size_kib": avg(wc[-3:]),
                         "fetch_ratio_raw": avg(fc[-3:]) * 1024 / cb,
                         "write_ratio": avg(wc[-3:]) * 1024 / cb},
    "gfx950_fetch_correction": 2.0,
    "hbm_read_bytes_per_launch": avg(ft) * 1024 * 2.0,
    "hbm_write_bytes_per_launch": avg(wt) * 1024,
    "algorithmic_read_bytes": meta["algorithmic_read_bytes"],
    "algorithmic_write_bytes": meta["algorithmic_write_bytes"],
}
res["traffic_bytes_per_launch"] = res["hbm_read_bytes_per_launch"] + res["hbm_write_bytes_per_launch"]
res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / (res["algorithmic_read_bytes"] + res["algorithmic_write_bytes"])
res["note"] = ("FETCH_SIZE x 2 is exact for a wide coalesced stream (the calibration copy: fetch_ratio_raw 0.5); a kernel whose reads are "
               "scattered 16-byte pieces (the hop slice kernel's staging loads, record and row reads out of L2) may be tallied differently -- "
               "the figure is the counter's, corrected as the guide prescribes")
with open(os.path.join(out_dir, "%s_traffic.json" % tag), "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res, indent=1))
