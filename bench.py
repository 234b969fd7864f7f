#!/usr/bin/env python3
"""bench.py -- Gorp match-and-extract throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}] [--results {auto,narrow,compact,dense}]

A "step" is one pass of the hot path (gx_extract_batch: product-DFA match + capture scan) over one batch of
synthetic log lines that is already resident in HBM.  Workloads (BASELINE.json `configs`, 1-based as in BASELINE.md):

  2  README 3-extraction GET/PUT/Other definition, 10 M x 200-byte lines            (the metric's config: `value`)
  3  64 syslog-like extractions, 10 M x 200-byte lines                              (N = 1: also.config3, same line)
  4  config 3's definition, 10 M lines PER GPU, table blob broadcast + final gather  (N > 1: configs3_64_extractions)
  5  512 extractions, lines of 50-2000 bytes, ~2 GB                                  (N = 1: also.config5, same line)

The default run (no --config) times config 2 as the headline and then, at N = 1, configs 3 and 5 with the SAME protocol
(spin-up, W warm-up steps, K timed steps between barrier + synchronize, two events on the launch stream around the K
steps) and reports each as a complete object under `also`: ms_per_step, lines_per_s, roofline (achieved, frac, traffic
from the committed counter passes), cpu_baseline (the oracle on a bounded sample, cores stated) and a bit-exact check of
the GPU rows against the oracle on that sample.  also.config3 carries a second generator variant beside the lower-case
one (mixed-case tokens: what a hop table that skips ONE byte interval per state pays for `JohnDoe42`).

Result formats (gx_batch_opts.compact_results): `dense` int32 match id + int32 offsets (4 + 8 G bytes per line), `compact`
rows of int16 id + uint16 offsets (2 + 4 G), `narrow` rows of int8 id + uint8 offsets (1 + 2 G; for batches whose lines are
shorter than 255 bytes and definitions of at most 126 extractions).  --results auto takes the narrowest format the workload
allows; ALL applicable formats are timed with the full step count and reported in `formats`.

Multi-GPU (one rank per GPU, RCCL): rank 0 compiles the tables and broadcasts the packed blob; every rank builds its
handle from the blob, generates its own shard (seeded by rank) and runs the same steps -- lines are independent, so there
is no collective inside a step (weak scaling).  After the timed region the per-line results are gathered to rank 0 over
xGMI; the gather is reported separately (`gather_*_ms`) and, overlapped with the next batch's kernel on a second stream,
as `value_with_overlapped_gather`.  Any rank whose parity check fails makes the run exit non-zero.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            cores = min(cores, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(cores, 64)


def reference_jvm_baseline(definition_text, lines, cores, budget_s=60.0):
    """the preferred CPU baseline is the reference itself on a JVM -- tools/RefBench.java, T threads over
    Gorp.extract (README.md:63-79).  It needs `java`, `javac` and a classpath holding gorp-core, dk.brics.automaton 1.11-8
    and jackson-jr 2.8.2 (GORP_REFERENCE_CLASSPATH).  None of them exists in this image: then this says so and nothing runs.
    Compiles and runs in child processes; returns a dict (RefBench's JSON line) or a string saying why not."""
    import subprocess
    import tempfile
    java, javac = shutil.which("java"), shutil.which("javac")
    cp = os.environ.get("GORP_REFERENCE_CLASSPATH")
    missing = [w for w, ok in (("java on PATH", java), ("javac on PATH", javac), ("GORP_REFERENCE_CLASSPATH", cp)) if not ok]
    if missing:
        return "unavailable (no %s)" % ", no ".join(missing)
    if definition_text is None:
        return "not run (this workload's definition has no .grp text)"
    try:
        with tempfile.TemporaryDirectory() as d:
            r = subprocess.run([javac, "-cp", cp, "-d", d, os.path.join(ROOT, "tools", "RefBench.java")], capture_output=True, text=True, timeout=120)
            if r.returncode != 0:
                return "javac failed: " + r.stderr[-300:]
            open(os.path.join(d, "def.grp"), "w", encoding="utf-8").write(definition_text)
            with open(os.path.join(d, "lines.txt"), "wb") as f:
                f.write(b"\n".join(lines) + b"\n")
            r = subprocess.run([java, "-cp", d + os.pathsep + cp, "RefBench", os.path.join(d, "def.grp"), os.path.join(d, "lines.txt"), str(cores)],
                               capture_output=True, text=True, timeout=budget_s + 120)
            if r.returncode != 0:
                return "RefBench failed: " + r.stderr[-300:]
            return json.loads(r.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, subprocess.TimeoutExpired) as e:
        return "RefBench did not finish: %s" % e


def cpu_baseline(definition, data_cpu, offsets_cpu, budget_s=20.0):
    """The oracle (a CPU port of the reference path, not the JVM) timed on a bounded
    sample of the same lines, all host cores."""
    from oracle import oracle as O
    built = [e.build() for e in definition]
    orc = O.OracleGorp([b[0] for b in built], [b[1] for b in built])
    cores = host_cores()
    n_all = len(offsets_cpu) - 1
    # calibrate on a small slice, then size the sample for ~budget_s CPU-seconds (>= 3 s of wall time)
    probe = min(n_all, 10000 * cores)
    t0 = time.perf_counter()
    orc.extract_batch(data_cpu, offsets_cpu[:probe + 1], nthreads=cores)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    n = int(min(n_all, max(probe, rate * max(3.0, budget_s / cores))))
    t0 = time.perf_counter()
    mid, caps = orc.extract_batch(data_cpu, offsets_cpu[:n + 1], nthreads=cores)
    dt = time.perf_counter() - t0
    # the same on one thread (SURVEY 8d asks for T = 1 beside T = all cores): ~2 s
    n1 = int(min(n, max(20000, 2.0 * (n / dt) / cores)))
    t1 = time.perf_counter()
    orc.extract_batch(data_cpu, offsets_cpu[:n1 + 1], nthreads=1)
    dt1 = time.perf_counter() - t1
    return {"value": n / dt, "unit": "lines/s", "cores": cores, "kind": "port",
            "sample": "first %d lines of the rank-0 shard, %d threads, %.1f s; C++ restatement of "
                      "PolyMatcher.match + java.util.regex capture (oracle/), not the JVM" % (n, cores, dt),
            "single_thread_value": n1 / dt1, "single_thread_sample": "first %d lines, 1 thread, %.1f s" % (n1, dt1)}, mid, caps, n


def self_launch(n_ranks):
    """Start `n_ranks` ranks of this script (one per GPU) through torch.distributed.run in a CHILD process -- never an
    exec: this parent stays as it is, waits, passes the ranks' output through and returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def build_workload(config, n, rank, dev, variant=None):
    """(definition, data u8[total] on dev, offsets u32[n+1] on dev, n, expected match ids or None, known mask or None,
    line_bytes_hint, description).  variant "mixed_case" (configs 3-5): the generator's \\w values and padding as mixed-case
    alphanumeric tokens (`JohnDoe42`) instead of lower-case words."""
    import numpy as np
    import torch
    from gorp_amd import workloads as W
    if config == 2:
        definition = W.readme3_definition()
        data, offsets, category = W.readme3_lines(n, seed=2 + rank, device=dev)
        return definition, data, offsets, n, category.to(torch.int32), None, W.LINE_BYTES, \
            "README 3-extraction GET/PUT/Other definition (BASELINE.json configs[1]), %d x %d B lines per GPU, seed 2+rank" % (n, W.LINE_BYTES)
    mixed = variant == "mixed_case"
    if config in (3, 4):
        rules, meta = W.syslog_definition(64, seed=3)
        base_n = 100_000
        dh, oh, cats = W.syslog_lines(meta, base_n, seed=3 + rank, mixed_case=mixed)
        desc = "64 syslog-like extractions (BASELINE.json configs[%d]), %%d x 200 B lines per GPU (a %d-line sample tiled), seed 3+rank%s" % (
            config - 1, base_n, ", mixed-case \\w values" if mixed else "")
        hint = 200
    else:
        rules, meta = W.syslog_definition(512, seed=3)
        base_n = 20_000
        dh, oh, cats = W.syslog_lines(meta, base_n, seed=5 + rank, min_len=50, max_len=2000, mixed_case=mixed)
        desc = "512 syslog-like extractions (BASELINE.json configs[4]), %%d lines of 50-2000 B per GPU (a %d-line sample tiled), seed 5+rank%s" % (
            base_n, ", mixed-case \\w values" if mixed else "")
        hint = int(int(oh[-1]) / base_n + 0.999)
    total = int(oh[-1])
    reps = max(1, n // base_n)
    while total * reps >= 2 ** 32:
        reps -= 1
    data = torch.from_numpy(dh.copy()).to(dev).repeat(reps)
    off = (torch.from_numpy(oh[:-1].astype(np.int64)).to(dev)[None, :] +
           torch.arange(reps, device=dev, dtype=torch.int64)[:, None] * total).reshape(-1)
    off = torch.cat([off, torch.tensor([total * reps], device=dev, dtype=torch.int64)]).to(torch.uint32)
    n = base_n * reps
    want = torch.from_numpy(cats).to(dev).repeat(reps)
    known = torch.from_numpy(cats != -9).to(dev).repeat(reps)
    return rules, data, off, n, want, known, hint, desc % n


def box_read_rate(data):
    """What a plain read-only sweep of this very buffer reaches on THIS box, GB/s (a torch int64 sum -- the fastest of torch's
    reductions here, tools/read_bw.py: a reference kernel, not the product's) -- devices of the pool differ by 10 % on the headline, and this says how much of that is the box."""
    import torch
    v = data[: data.numel() & ~15].view(torch.int64)
    for _ in range(3):
        v.sum()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        v.sum()
    e1.record()
    torch.cuda.synchronize()
    return v.numel() * 8 / (e0.elapsed_time(e1) / 10 * 1e-3) / 1e9


def recorded_traffic(names, algo_read, algo_write):
    """HBM bytes per launch from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this kernel on this exact workload
    (tools/collect_r05.sh: separate --pmc passes, FETCH_SIZE doubled per the gfx950 correction, calibrated against a
    same-size copy kernel in the same run).  Counters cannot be read from inside this process: the RECORDED figure of
    the committed summary is reported when its workload and result format match this run, else null."""
    for name in names:
        try:
            path = os.path.join(ROOT, "profiles", name)
            tj = json.load(open(path))
            if tj["algorithmic_read_bytes"] == algo_read and tj["algorithmic_write_bytes"] == algo_write:
                return tj["traffic_bytes_per_launch"], "recorded: profiles/%s (%s)" % (name, time.strftime("%Y-%m-%d", time.gmtime(os.path.getmtime(path))))
        except (OSError, ValueError, KeyError):
            pass
    return None, None


KERNEL_NAMES = {1: "tile kernel", 2: "slice kernel", 3: "per-line kernel", 4: "lane kernel", 5: "tile kernel on the hop tier's tables", 6: "hop slice kernel"}


def table_tier(gorp):
    if int(gorp.stat(14)) > 0:
        return ("hop tier: run + chain records, %d of %d states' records in LDS (%d of them reachable by well-formed lines), dense rows in global memory"
                % (int(gorp.stat(15)), int(gorp.stat(14)), int(gorp.stat(16))))
    return {0: "per-line kernel", 1: "LDS (dense rows)", 2: "L2 (dense rows)", 3: "LDS (range records)", 4: "L2 (range records)"}.get(int(gorp.stat(7)), str(gorp.stat(7)))


class Timer:
    """The timing protocol, one place for every workload of the line: K steps bracketed by barrier + synchronize, max over
    ranks, and the launches' average duration from two events on the launch stream, one before the first step and one
    behind the last (per_step: an event behind every step instead -- the spread of the steps, at the price of a marker
    between the kernels; never used for a headline).
    spin_ms: untimed steps for that long BEFORE the W warm-up steps -- the device's power management needs ~25 ms of
    unbroken load to reach its steady clocks (profiles/r04_clock_ramp.txt: 0.54 -> 0.36 -> 0.334 ms per launch over the
    first 30 ms from idle); W = 5 steps of 0.35 ms are not that.  The throughput of a job that keeps the device busy is
    the steady one; the from-idle figure is reported beside it."""

    def __init__(self, dev, distributed):
        self.dev, self.distributed = dev, distributed

    def __call__(self, one, steps, warmup, per_step=False, spin_ms=0.0):
        import torch
        import torch.distributed as dist
        t_spin = time.perf_counter() + spin_ms * 1e-3
        while time.perf_counter() < t_spin:
            for _ in range(4):   # (short groups: ranks leave the spin-up within a group's time of each other)
                one()
            torch.cuda.current_stream().synchronize()
        for _ in range(warmup):
            one()
        torch.cuda.synchronize()
        if self.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1 if per_step else 2)]
        t_start = time.perf_counter()
        ev[0].record()
        for i in range(steps):
            one()
            if per_step:
                ev[i + 1].record()
        if not per_step:
            ev[1].record()
        torch.cuda.synchronize()
        if self.distributed:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_start
        mine = elapsed
        t = torch.tensor([elapsed], dtype=torch.float64, device=self.dev)
        if self.distributed:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        self.last_own_elapsed = mine
        if per_step:
            return float(t.item()), [ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]
        return float(t.item()), [ev[0].elapsed_time(ev[1]) / steps] * steps


def side_workload(config, args, dev, stream, timer, variant=None, with_cpu=True):
    """One of the line's other workloads (BASELINE.json configs[2] / configs[4]) as a complete object: the same protocol as
    the headline, its narrowest result format, roofline with the recorded counter traffic, the oracle on a bounded sample as
    CPU baseline, and the GPU rows of that sample compared with the oracle's bit for bit."""
    import numpy as np
    import torch
    from gorp_amd.gorp import Gorp, unpack_rows
    n_req = args.lines or (3_800_000 if config == 5 else 10_000_000)
    definition, data, offsets, n, want, known, hint, desc = build_workload(config, n_req, 0, dev, variant)
    t0 = time.perf_counter()
    gorp = Gorp.construct(definition)
    setup_s = time.perf_counter() - t0
    G = gorp.max_groups
    total_bytes = int(data.numel())
    max_line = int((offsets[1:].to(torch.int64) - offsets[:-1].to(torch.int64)).max().item())
    narrow = max_line < 255 and len(definition) <= 126
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    rows = torch.empty((n, 1 + 2 * G), dtype=torch.uint8 if narrow else torch.int16, device=dev)
    overflow = torch.zeros(1, dtype=torch.int64, device=dev)

    def one():
        gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, None, rows.data_ptr(), stream=stream, no_sync=True,
                                  line_bytes_hint=hint, compact=2 if narrow else True, overflow_ptr=overflow.data_ptr(), max_line_bytes=max_line)

    elapsed, kernel_ms = timer(one, args.steps, args.warmup, spin_ms=args.spin_up_ms)
    got = (rows[:, 0].view(torch.int8) if narrow else rows[:, 0]).to(torch.int32)
    ok = bool(torch.equal(got[known], want[known])) and int(overflow.item()) == 0
    if not ok:
        raise SystemExit("bench: config %d%s: match ids differ from the generator's expected categories" % (config, " (%s)" % variant if variant else ""))
    k_avg = sum(kernel_ms) / len(kernel_ms)
    algo_read = total_bytes + 4 * (n + 1)
    algo_write = n * (1 + 2 * G) * (1 if narrow else 2)
    achieved = algo_read / (k_avg * 1e-3) / 1e9
    ms_per_step = elapsed * 1e3 / args.steps
    traffic, traffic_src = recorded_traffic(["r05_config%d%s_traffic.json" % (config, "_" + variant if variant else "")], algo_read, algo_write)
    obj = {"workload": desc, "baseline_config": config, "lines_per_s": n * args.steps / elapsed, "unit": "lines/s", "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": ms_per_step, "kernel_ms_avg": k_avg, "lines": n, "mean_line_bytes": total_bytes / n, "max_line_bytes": max_line,
           "gb_per_s_scanned": total_bytes * args.steps / elapsed / 1e9,
           "results": "%s rows, %d B/line" % ("u8" if narrow else "u16", (1 + 2 * G) * (1 if narrow else 2)),
           "match_dfa_states": int(gorp.stat(0)), "char_classes": int(gorp.stat(1)), "capture_states": int(gorp.stat(2)), "table_blob_bytes": int(gorp.stat(4)),
           "table_tier": table_tier(gorp), "kernel": KERNEL_NAMES.get(int(gorp.stat(25)), "?") + " (gx_stat(h, 25): what the library launched)",
           "setup_s": setup_s,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "frac_by_wall_clock": algo_read / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_source": traffic_src,
                        "wasted_traffic_ratio": (traffic / (algo_read + algo_write)) if traffic else None,
                        "algorithmic_read_bytes": algo_read, "algorithmic_write_bytes": algo_write}}
    if with_cpu:
        sample = min(n, 2_000_000 if config != 5 else 400_000)
        end = int(offsets[sample].item())
        d_cpu = data[:end].cpu().numpy()
        o_cpu = offsets[: sample + 1].cpu().numpy().astype(np.uint32)
        base, omid, ocaps, ns = cpu_baseline(definition, d_cpu, o_cpu, budget_s=args.side_cpu_budget_s)
        one()
        torch.cuda.synchronize()
        r = rows[:ns].cpu().numpy()
        gm, gc = unpack_rows(r if narrow else r.view(np.uint16))
        if not (np.array_equal(gm, omid) and np.array_equal(gc, ocaps)):
            raise SystemExit("bench: config %d%s: GPU results differ from the oracle on the baseline sample" % (config, " (%s)" % variant if variant else ""))
        obj["cpu_baseline"] = base
        obj["parity"] = "GPU rows of the first %d lines bit-identical to the oracle's (ids and capture offsets)" % ns
    del data, offsets, rows, gorp
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    return obj


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="BASELINE.md config (0: config 2 per GPU at any N -- the metric's workload, so that the N = 1, 2, 4, 8 values are one "
                         "weak-scaling curve; at N = 1 configs 3 and 5 are measured beside it as `also`, at N > 1 configs[3] as `configs3_64_extractions`)")
    ap.add_argument("--lines", type=int, default=0, help="lines per GPU (0: the config's size)")
    ap.add_argument("--results", default="auto", choices=["auto", "narrow", "compact", "dense"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="N = 1, no --config: skip configs 3 and 5 (the `also` objects)")
    ap.add_argument("--side-cpu-budget-s", type=float, default=10.0, help="CPU-seconds of the oracle per `also` workload")
    ap.add_argument("--result-buffer-candidates", type=int, default=1,
                    help="u8 result buffers timed before the timed region; `value` uses the MEDIAN one (1, the default: the buffer as allocated)")
    ap.add_argument("--spin-up-ms", type=float, default=150.0,
                    help="untimed steps for this long before the W warm-up steps of a timed region: the device at its steady clocks (0: none)")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` with no launcher: this process has not touched the GPU (torch is not even imported
        # yet), so it starts the N ranks as fresh child processes and relays rank 0's line and the exit code.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: the launcher's world size and --gpus must agree" % (args.gpus, world))
    config = args.config or 2
    also_config4 = args.config == 0 and world > 1   # BASELINE.json configs[3]: its hardware run rides along with the scaling curve
    also_side = args.config == 0 and world == 1 and not args.no_also

    import numpy as np
    import torch
    import torch.distributed as dist
    from gorp_amd.gorp import Gorp, unpack_rows

    backend = os.environ.get("GORP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = 0   # rehearsal: every rank on the one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") always; GORP_BENCH_BACKEND=gloo only exists to rehearse the N > 1 path with several ranks on ONE GPU
        # (RCCL refuses two ranks on one device), which is all a one-GPU box allows
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)
    timer = Timer(dev, distributed)

    def all_ranks_ok(ok, what):
        """A parity failure on ANY rank ends every rank with a non-zero exit code (the launcher's, and so the parent's)."""
        if os.environ.get("GORP_BENCH_FAIL_RANK") == str(rank):   # (tests/: the exit code of a run in which one rank's check fails)
            ok = False
        okt = torch.tensor([1 if ok else 0], device=dev)
        if distributed:
            dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) != 1:
            raise SystemExit("bench: %s%s" % (what, "" if ok else " (on this rank: %d)" % rank))

    # ---- this rank's shard (generated on / copied to the device before anything is timed) ----
    n_req = args.lines or (3_800_000 if config == 5 else 10_000_000)
    definition, data, offsets, n, want, known, hint, desc = build_workload(config, n_req, rank, dev)
    total_bytes = int(data.numel())

    # ---- tables: compile on rank 0, broadcast the blob (RCCL), build everywhere ----
    t0 = time.perf_counter()
    bcast_ms = None
    if distributed:
        from gorp_amd import dist as gdist
        torch.cuda.synchronize()
        tb = time.perf_counter()
        gorp, _ = gdist.broadcast_gorp(definition, dev, src=0)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - tb) * 1e3   # includes rank 0's compile
    else:
        gorp = Gorp.construct(definition)
    setup_s = time.perf_counter() - t0

    G = gorp.max_groups
    # The result buffers are allocations of their own, as a caller's would be (the generator's freed blocks go back to the
    # driver first): the buffer as allocated is THE result buffer.  --result-buffer-candidates N > 1 times N such buffers
    # before the timed region and uses the MEDIAN one; all times are in the line (config.result_buffer_candidates_ms).
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    mid = torch.empty(n, dtype=torch.int32, device=dev)
    caps = torch.empty((n, 2 * G), dtype=torch.int32, device=dev)
    rows = torch.empty((n, 1 + 2 * G), dtype=torch.int16, device=dev)
    rows8 = torch.empty((n, 1 + 2 * G), dtype=torch.uint8, device=dev)
    max_line = int((offsets[1:].to(torch.int64) - offsets[:-1].to(torch.int64)).max().item())
    narrow_ok = max_line < 255 and len(definition) <= 126
    formats = (["narrow"] if narrow_ok else []) + ["compact", "dense"]
    headline = args.results if args.results != "auto" else formats[0]
    if headline not in formats:
        raise SystemExit("--results narrow needs lines shorter than 255 bytes and at most 126 extractions")
    overflow = torch.zeros(1, dtype=torch.int64, device=dev)
    cand_ms, cand_pick = None, 0

    stream = torch.cuda.current_stream().cuda_stream

    # (the caller knows its longest line -- here from the generator, in a pipeline from gx_split_lines_max -- and says so:
    # gx_batch_opts.max_line_bytes, a promise the library checks; it spares every step a second, nearly empty launch)
    def step(fmt, on_stream=None):
        st = on_stream if on_stream is not None else stream
        if fmt == "compact":
            gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, None, rows.data_ptr(), stream=st, no_sync=True,
                                      line_bytes_hint=hint, compact=True, overflow_ptr=overflow.data_ptr(), max_line_bytes=max_line)
        elif fmt == "narrow":
            gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, None, rows8.data_ptr(), stream=st, no_sync=True,
                                      line_bytes_hint=hint, compact=2, overflow_ptr=overflow.data_ptr(), max_line_bytes=max_line)
        else:
            gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, mid.data_ptr(), caps.data_ptr(), stream=st, no_sync=True,
                                      line_bytes_hint=hint, max_line_bytes=max_line)

    if narrow_ok and args.result_buffer_candidates > 1:
        cands = [rows8] + [torch.empty_like(rows8) for _ in range(args.result_buffer_candidates - 1)]
        cand_ms = []
        for cb in cands:
            launch = lambda: gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, None, cb.data_ptr(), stream=stream, no_sync=True,
                                                       line_bytes_hint=hint, compact=2, overflow_ptr=overflow.data_ptr(), max_line_bytes=max_line)
            _, ms = Timer(dev, False)(launch, 10, 2, spin_ms=60.0)
            cand_ms.append(ms[0])
        order = sorted(range(len(cands)), key=lambda q: cand_ms[q])
        cand_pick = order[(len(order) - 1) // 2]   # the median (lower median of an even count), never the best
        rows8 = cands[cand_pick]
        del cands

    runs = {}
    torch.cuda.synchronize()
    time.sleep(0.25)                                                         # (an idle device, as a first batch finds it)
    from_idle = timer(lambda: step(headline), args.steps, args.warmup)       # the same W + K steps from idle: reported beside, never `value`
    for fmt in [f for f in formats if f != headline]:
        runs[fmt] = timer(lambda: step(fmt), args.steps, args.warmup, spin_ms=args.spin_up_ms)   # the other result formats, same step count, reported beside
    elapsed, kernel_ms = runs[headline] = timer(lambda: step(headline), args.steps, args.warmup, spin_ms=args.spin_up_ms)   # THE timed region
    own_elapsed = timer.last_own_elapsed
    _, step_ms = timer(lambda: step(headline), args.steps, 1, per_step=True)                     # (afterwards: the spread of single steps)

    # ---- correctness of what was timed: the generator knows every (uncorrupted) line's answer ----
    got = {"compact": lambda: rows[:, 0].to(torch.int32), "narrow": lambda: rows8[:, 0].view(torch.int8).to(torch.int32), "dense": lambda: mid}[headline]()
    ok = bool(torch.equal(got, want)) if known is None else bool(torch.equal(got[known], want[known]))
    ok = ok and int(overflow.item()) == 0
    all_ranks_ok(ok, "match ids differ from the generator's expected categories")

    # per-rank step times (N > 1: min / max over the ranks), and the world size as the collective library reports it
    rank_ms = torch.tensor([own_elapsed * 1e3 / args.steps], dtype=torch.float64, device=dev)
    rank_ms_all = [rank_ms.clone() for _ in range(world)]
    if distributed:
        dist.all_gather(rank_ms_all, rank_ms)
    rank_ms_all = [float(x.item()) for x in rank_ms_all]

    # ---- final gather of results to rank 0 over xGMI (reported, not in `value`), then the same gather overlapped with the
    # next batch's kernel on a second stream: what a job pays per batch when it gathers every batch ----
    gather_ms = gather_dense_ms = gather_narrow_ms = None
    overlapped = None
    if distributed and not args.no_gather:
        for fmt in formats:
            step(fmt)
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        gr = gdist.gather_rows(rows, dst=0)            # the compact rows as the kernel wrote them
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        gm, gc = gdist.gather_results(mid, caps, dst=0)  # the same in the dense format
        torch.cuda.synchronize()
        gather_dense_ms = (time.perf_counter() - tg) * 1e3
        same = True
        if rank == 0:
            same = gr.shape[0] == gm.shape[0] and torch.equal(gr[:n], rows) and torch.equal(gm[:n], mid) and torch.equal(gc[:n], caps) and \
                torch.equal(gr[:, 0].to(torch.int32), gm)
        if narrow_ok:
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            g8 = gdist.gather_rows(rows8, dst=0)          # the u8 rows: a quarter of the dense bytes on the links
            torch.cuda.synchronize()
            gather_narrow_ms = (time.perf_counter() - tg) * 1e3
            if rank == 0:
                same = same and torch.equal(g8[:n], rows8) and torch.equal(g8[:, 0].view(torch.int8).to(torch.int32), gm)
            del g8
        del gr, gm, gc
        all_ranks_ok(same, "gathered rows differ from the ranks' own")
        # two-deep pipeline: batch k+1's kernel is enqueued on the compute stream before the gather of batch k (its rows
        # in the other of two buffers, on the gather stream behind an event) is waited for
        head_rows = rows8 if headline == "narrow" else rows
        if headline != "dense":
            bufs = [head_rows, torch.empty_like(head_rows)]
            gstream = torch.cuda.Stream(device=dev)
            comp = torch.cuda.current_stream()

            def kernel_into(buf):
                gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, None, buf.data_ptr(), stream=stream, no_sync=True,
                                          line_bytes_hint=hint, compact=2 if headline == "narrow" else True, overflow_ptr=overflow.data_ptr(), max_line_bytes=max_line)

            def pipelined(steps):
                done = [None, None]
                for k in range(steps + 1):
                    if k < steps:
                        kernel_into(bufs[k & 1])               # batch k's kernel: enqueued before batch k-1's gather is waited for
                        done[k & 1] = torch.cuda.Event()
                        done[k & 1].record(comp)
                    if k >= 1:
                        gstream.wait_event(done[(k - 1) & 1])
                        with torch.cuda.stream(gstream):
                            out = gdist.gather_rows(bufs[(k - 1) & 1], dst=0, sizes=[n] * world)
                        del out
                        free = torch.cuda.Event()              # (buffer (k-1)&1 is written again by batch k+1: that kernel waits for this gather)
                        free.record(gstream)
                        comp.wait_event(free)
                torch.cuda.synchronize()

            pipelined(2)
            torch.cuda.synchronize()
            dist.barrier()
            tp = time.perf_counter()
            pipelined(args.steps)
            dist.barrier()
            el = time.perf_counter() - tp
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            overlapped = {"value": n * world * args.steps / float(tt.item()), "ms_per_step": float(tt.item()) * 1e3 / args.steps,
                          "what": "every batch's %s rows gathered to rank 0 on a second stream while the next batch's kernel runs (two row buffers)" % ("u8" if headline == "narrow" else "u16")}
            del bufs

    # ---- N > 1: BASELINE.json configs[3] beside the curve -- 64 extractions, 10 M x 200 B lines per GPU (80 M on 8), the tables
    # compiled on rank 0 and broadcast over RCCL, u8 rows; the same timing protocol, reported as its own object ----
    config4 = None
    if also_config4:
        definition4, data4, offsets4, n4, want4, known4, hint4, desc4 = build_workload(4, args.lines or 10_000_000, rank, dev)
        torch.cuda.synchronize()
        tb = time.perf_counter()
        gorp4, _ = gdist.broadcast_gorp(definition4, dev, src=0)
        torch.cuda.synchronize()
        bcast4_ms = (time.perf_counter() - tb) * 1e3
        G4 = gorp4.max_groups
        max4 = int((offsets4[1:].to(torch.int64) - offsets4[:-1].to(torch.int64)).max().item())
        narrow4 = max4 < 255 and len(definition4) <= 126
        rows4 = torch.empty((n4, 1 + 2 * G4), dtype=torch.uint8 if narrow4 else torch.int16, device=dev)
        over4 = torch.zeros(1, dtype=torch.int64, device=dev)
        step4 = lambda: gorp4.extract_batch_device(data4.data_ptr(), offsets4.data_ptr(), n4, None, rows4.data_ptr(), stream=stream, no_sync=True,
                                                   line_bytes_hint=hint4, compact=2 if narrow4 else True, overflow_ptr=over4.data_ptr(),
                                                   max_line_bytes=max4)
        elapsed4, kernel4_ms = timer(step4, args.steps, args.warmup, spin_ms=args.spin_up_ms)
        got4 = (rows4[:, 0].view(torch.int8) if narrow4 else rows4[:, 0]).to(torch.int32)
        ok4 = (bool(torch.equal(got4, want4)) if known4 is None else bool(torch.equal(got4[known4], want4[known4]))) and int(over4.item()) == 0
        all_ranks_ok(ok4, "configs[3]: match ids differ from the generator's expected categories")
        gather4_ms = None
        if not args.no_gather:
            torch.cuda.synchronize()
            dist.barrier()
            tg = time.perf_counter()
            g4 = gdist.gather_rows(rows4, dst=0)
            torch.cuda.synchronize()
            gather4_ms = (time.perf_counter() - tg) * 1e3
            del g4
        k4 = sum(kernel4_ms) / len(kernel4_ms)
        bytes4 = int(data4.numel()) + 4 * (n4 + 1)
        config4 = {"workload": desc4, "baseline_config": 4, "value": n4 * world * args.steps / elapsed4, "unit": "lines/s",
                   "ms_per_step": elapsed4 * 1e3 / args.steps, "kernel_ms_avg": k4, "lines_per_gpu": n4,
                   "results": "%s rows, %d B/line" % ("u8" if narrow4 else "u16", (1 + 2 * G4) * (1 if narrow4 else 2)),
                   "frac_of_hbm_peak_per_gpu": bytes4 / (k4 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "table_bcast_ms": bcast4_ms, "gather_ms": gather4_ms, "table_blob_bytes": int(gorp4.stat(4))}
        del data4, offsets4, rows4

    out = None
    if rank == 0:
        steps = args.steps
        ms_per_step = elapsed * 1e3 / steps
        value = n * world * steps / elapsed
        k_avg = sum(kernel_ms) / len(kernel_ms)
        k_sorted = sorted(step_ms)
        algo_read = total_bytes + 4 * (n + 1)            # line bytes + u32 offsets
        write_bytes = {"narrow": n * (1 + 2 * G), "compact": n * (2 + 4 * G), "dense": n * (4 + 8 * G)}
        fmt_desc = {"narrow": "narrow rows (int8 id + uint8 offsets, %d B/line)" % (1 + 2 * G),
                    "compact": "compact rows (int16 id + uint16 offsets, %d B/line)" % (2 + 4 * G),
                    "dense": "dense (int32 id + int32 offsets, %d B/line)" % (4 + 8 * G)}
        achieved = algo_read / (k_avg * 1e-3) / 1e9
        box_rate = box_read_rate(data) if world == 1 else None   # (after the timed region; the same buffer, the same clocks)
        if config == 2:
            traffic, traffic_src = recorded_traffic(("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json"), algo_read, write_bytes[headline])
        else:
            traffic, traffic_src = recorded_traffic(("r05_config%d_traffic.json" % config,), algo_read, write_bytes[headline])
        per_format = {}
        for fmt, (f_elapsed, f_ms) in runs.items():
            f_avg = sum(f_ms) / len(f_ms)
            per_format[fmt] = {"results": fmt_desc[fmt], "steps": len(f_ms), "lines_per_s": n * world * len(f_ms) / f_elapsed,
                               "ms_per_step": f_elapsed * 1e3 / len(f_ms), "kernel_ms_avg": f_avg,
                               "algorithmic_write_bytes": write_bytes[fmt], "read_gb_per_s": algo_read / (f_avg * 1e-3) / 1e9,
                               "frac": algo_read / (f_avg * 1e-3) / 1e9 / HBM_PEAK_GBS}
        out = {
            "metric": "lines/sec (Gorp.extract: product-DFA match + capture offsets; results as %s)" % fmt_desc[headline],
            "value": value,
            "value_compact_rows": per_format["compact"]["lines_per_s"],   # the same workload with u16 rows: the format `value` had in round 2 (a stable key for trends)
            "unit": "lines/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": desc, "baseline_config": config,
                       "lines_per_gpu": n, "mean_line_bytes": total_bytes / n, "offsets": "u32",
                       "results": fmt_desc[headline], "max_line_bytes": max_line,
                       "result_buffer": "as allocated" if cand_ms is None else "the median of %d candidates" % len(cand_ms),
                       "result_buffer_candidates_ms": cand_ms, "result_buffer_choice": cand_pick,
                       "match_dfa_states": int(gorp.stat(0)), "char_classes": int(gorp.stat(1)),
                       "capture_states": int(gorp.stat(2)), "table_blob_bytes": int(gorp.stat(4)),
                       "table_tier": table_tier(gorp),
                       "kernel": KERNEL_NAMES.get(int(gorp.stat(25)), "?") + " (gx_stat(h, 25): what the library launched)",
                       "parallelism": "lines sharded by rank (dp%d), no collective in a step" % world},
            "gb_per_s_scanned": total_bytes * world * steps / elapsed / 1e9,
            "kernel_ms": {"avg": k_avg, "clock": "two events on the launch stream around the %d timed steps" % steps,
                          "spin_up_ms": args.spin_up_ms,   # untimed steps before the W warm-up steps: the device at its steady clocks (see Timer)
                          "from_idle": {"avg": sum(from_idle[1]) / len(from_idle[1]), "ms_per_step": from_idle[0] * 1e3 / steps,
                                        "lines_per_s": n * world * steps / from_idle[0],
                                        "protocol": "0.25 s of idling, then the same W + K steps without the spin-up"},
                          "single_steps_after": {"min": k_sorted[0], "median": k_sorted[len(k_sorted) // 2], "max": k_sorted[-1],
                                                 "clock": "an event behind every step, a second run of %d steps" % steps}},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "frac_by_wall_clock": algo_read / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,   # (the same bytes over ms_per_step, the host's clock around barrier + synchronize)
                         "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "algorithmic_read_bytes": algo_read, "algorithmic_write_bytes": write_bytes[headline],
                         "frac_of_measured_copy_ceiling": achieved / 6290.0,
                         "this_box": None if box_rate is None else {
                             "plain_read_gb_per_s": box_rate, "frac_of_it": achieved / box_rate,
                             "what": "a torch int64 sum over the same 2 GB buffer right after the timed steps: what a read-only sweep reaches on this device"}},
            "formats": per_format,
            "setup_s": setup_s,
            "table_bcast_ms": bcast_ms,
            "gather_ms": gather_ms,
            "gather_dense_ms": gather_dense_ms,
            "gather_narrow_ms": gather_narrow_ms,
        }
        if distributed:
            out["collective_world_size"] = dist.get_world_size()   # as the process group (RCCL) reports it
            out["collective_backend"] = dist.get_backend()
            out["ms_per_step_by_rank"] = {"min": min(rank_ms_all), "max": max(rank_ms_all), "all": rank_ms_all}
            if overlapped is not None:
                out["value_with_overlapped_gather"] = overlapped["value"]
                out["overlapped_gather"] = overlapped
        if config4 is not None:
            out["configs3_64_extractions"] = config4
        if not args.no_cpu_baseline and world == 1:
            sample = min(n, 10_000_000 if config == 2 else 2_000_000)
            end = int(offsets[sample].item())
            d_cpu = data[:end].cpu().numpy()
            o_cpu = offsets[: sample + 1].cpu().numpy().astype(np.uint32)
            base, omid, ocaps, ns = cpu_baseline(definition, d_cpu, o_cpu)
            # the baseline run doubles as a parity check of the timed GPU output (every format)
            for fmt in formats:
                step(fmt)
            torch.cuda.synchronize()
            cm, cc = unpack_rows(rows[:ns].cpu().numpy().view(np.uint16))
            same = (np.array_equal(mid[:ns].cpu().numpy(), omid) and np.array_equal(caps[:ns].cpu().numpy(), ocaps) and
                    np.array_equal(cm, omid) and np.array_equal(cc, ocaps))
            if narrow_ok:
                nm, nc = unpack_rows(rows8[:ns].cpu().numpy())
                same = same and np.array_equal(nm, omid) and np.array_equal(nc, ocaps)
            if not same:
                raise SystemExit("bench: GPU results differ from the oracle on the baseline sample")
            # the reference itself, where a JVM and its jars exist (config 2 only: it has a definition text)
            n_ref = min(ns, 2_000_000)
            ref_lines = [bytes(d_cpu[int(o_cpu[i]):int(o_cpu[i + 1])]) for i in range(n_ref)] if shutil.which("java") and config == 2 else []
            from gorp_amd import workloads as W
            base["reference_jvm"] = reference_jvm_baseline(W.README3_DEFINITION_TEXT if config == 2 else None, ref_lines, base["cores"])
            if isinstance(base["reference_jvm"], dict):   # the reference ran: it IS the baseline, the port stays beside it
                port = dict(base)
                port.pop("reference_jvm")
                ref = base["reference_jvm"]
                base = {"value": ref["value"], "unit": "lines/s", "cores": ref["cores"], "kind": "reference",
                        "sample": "first %d lines of the rank-0 shard, %d threads over Gorp.extract on JVM %s (tools/RefBench.java), %.1f s"
                                  % (ref["lines"], ref["cores"], ref["java"], ref["seconds"]), "port": port}
            out["cpu_baseline"] = base
            del d_cpu, o_cpu
        if world == 1 and config == 2:
            # beside the batch number: what ONE Gorp.extract(String) costs through the same library (gx_extract_one_utf16, the
            # drop-in for a caller that does not batch) -- a latency, reported for context, never part of `value`
            try:
                import ctypes as _C
                import time as _time
                from gorp_amd import _native as _N
                one = "[123456789]: GET 12ms /index.html?x=1&y=2"
                u = np.frombuffer(one.encode("utf-16-le"), np.uint16)
                m1, c1 = _C.c_int32(0), np.zeros(2 * gorp.max_groups, np.int32)
                call = lambda: _N.lib().gx_extract_one_utf16(gorp._h.ptr, u.ctypes.data, len(u), _C.byref(m1), c1.ctypes.data)
                for _ in range(50):
                    call()
                t_one = _time.perf_counter()
                for _ in range(1000):
                    call()
                out["one_line_latency_us"] = round((_time.perf_counter() - t_one) / 1000 * 1e6, 1)
            except Exception as e:   # (context only: the bench line does not depend on it)
                out["one_line_latency_us"] = "unavailable (%s)" % type(e).__name__
    if also_side:
        # ---- N = 1: BASELINE.json configs[2] and configs[4] in the same line, the same protocol, complete objects ----
        del data, offsets, mid, caps, rows, rows8, gorp
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        also = {}
        also["config3"] = side_workload(3, args, dev, stream, timer, with_cpu=not args.no_cpu_baseline)
        mixed = side_workload(3, args, dev, stream, timer, variant="mixed_case", with_cpu=not args.no_cpu_baseline)
        also["config3"]["generator_variants"] = {
            "lower_case": {k: also["config3"][k] for k in ("ms_per_step", "kernel_ms_avg", "lines_per_s")} | {"frac": also["config3"]["roofline"]["frac"]},
            "mixed_case": {k: mixed[k] for k in ("workload", "ms_per_step", "kernel_ms_avg", "lines_per_s")} | {"frac": mixed["roofline"]["frac"], "parity": mixed.get("parity")}}
        also["config5"] = side_workload(5, args, dev, stream, timer, with_cpu=not args.no_cpu_baseline)
        out["also"] = also
    if rank == 0:
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
