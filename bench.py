#!/usr/bin/env python3
"""bench.py -- Gorp match-and-extract throughput on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (gx_extract_batch: product-DFA match +
capture scan) over one batch of synthetic log lines that is already resident in
HBM.  Workload at every N: BASELINE.json configs[1] -- the README 3-extraction
GET/PUT/Other definition over 10 M x 200-byte lines PER GPU (weak scaling,
lines sharded by rank, no data-path collective inside a step).

Multi-GPU (launched by torch.distributed.run, one rank per GPU, RCCL): rank 0
compiles the tables and broadcasts the packed blob; every rank builds its handle
from the blob, generates its own shard, and runs the same steps.  After the
timed region the per-line results are gathered to rank 0 once over xGMI and the
gather time is reported separately (`gather_ms`); it is not part of `value`
(Multi-GPU).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
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


def cpu_baseline(definition, data_cpu, offsets_cpu, budget_s=20.0):
    """The oracle (a CPU port of the reference path, not the JVM) timed on a bounded
    sample of the same lines, all host cores."""
    import numpy as np
    from oracle import oracle as O
    built = [e.build() for e in definition]
    orc = O.OracleGorp([b[0] for b in built], [b[1] for b in built])
    cores = host_cores()
    n_all = len(offsets_cpu) - 1
    # calibrate on a small slice, then size the sample for ~20 CPU-seconds (>= 3 s of wall time)
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lines", type=int, default=10_000_000, help="lines per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from gorp_amd import workloads as W
    from gorp_amd.gorp import Gorp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if os.environ.get("GORP_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = 0   # rehearsal: every rank on the one GPU
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    distributed = world > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl") always; GORP_BENCH_BACKEND=gloo only exists to rehearse the N > 1 path with two ranks on ONE GPU
        # (RCCL refuses two ranks on one device), which is all a one-GPU box allows
        backend = os.environ.get("GORP_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    # ---- tables: compile on rank 0, broadcast the blob (RCCL), build everywhere ----
    definition = W.readme3_definition()
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

    # ---- this rank's shard, generated on the device ----
    n = args.lines
    data, offsets, category = W.readme3_lines(n, seed=2 + rank, device=dev)
    total_bytes = int(data.numel())
    G = gorp.max_groups
    mid = torch.empty(n, dtype=torch.int32, device=dev)
    caps = torch.empty((n, 2 * G), dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        gorp.extract_batch_device(data.data_ptr(), offsets.data_ptr(), n, mid.data_ptr(), caps.data_ptr(),
                                  stream=stream, no_sync=True)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t_start = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    kernel_ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)]

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if distributed:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- correctness of what was timed: the generator knows every line's answer ----
    ok = bool(torch.equal(mid, category.to(torch.int32)))
    okt = torch.tensor([1 if ok else 0], device=dev)
    if distributed:
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
    if int(okt.item()) != 1:
        raise SystemExit("bench: match ids differ from the generator's expected categories")

    # ---- final gather of results to rank 0 over xGMI (reported, not in `value`) ----
    gather_ms = None
    gather_compact_ms = None
    if distributed and not args.no_gather:
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        gm, gc = gdist.gather_results(mid, caps, dst=0)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert gm.shape[0] == n * world and torch.equal(gm[:n], mid) and torch.equal(gc[:n], caps)
        # the same with compact transport rows (int16 id + uint16 offsets, HIP pack/unpack kernels)
        torch.cuda.synchronize()
        dist.barrier()
        tg = time.perf_counter()
        cm, cc = gdist.gather_results_compact(mid, caps, dst=0)
        torch.cuda.synchronize()
        gather_compact_ms = (time.perf_counter() - tg) * 1e3
        if rank == 0:
            assert torch.equal(cm, gm) and torch.equal(cc, gc)
        del gm, gc, cm, cc

    if rank == 0:
        steps = args.steps
        ms_per_step = elapsed * 1e3 / steps
        lines_total = n * world * steps
        value = lines_total / elapsed
        k_avg = sum(kernel_ms) / len(kernel_ms)
        k_sorted = sorted(kernel_ms)
        algo_read = total_bytes + 4 * (n + 1)            # line bytes + u32 offsets
        algo_write = n * (4 + 8 * G)                     # match id + dense captures
        achieved = algo_read / (k_avg * 1e-3) / 1e9
        # HBM bytes per launch from the rocprofv3 FETCH_SIZE / WRITE_SIZE passes of this kernel on this exact
        # workload (tools/traffic_target.py + tools/summarize_traffic.py; FETCH_SIZE doubled per the gfx950
        # correction, calibrated against a same-size copy kernel in the same run).  Counters cannot be read from
        # inside this process, so the committed summary is reported when the workload matches, else null.
        traffic, traffic_src = None, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if tj["algorithmic_read_bytes"] == algo_read and tj["algorithmic_write_bytes"] == algo_write:
                traffic, traffic_src = tj["traffic_bytes_per_launch"], "profiles/r01_traffic.json"
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "lines/sec (Gorp.extract: product-DFA match + capture offsets), 200-byte lines",
            "value": value,
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
            "config": {"workload": "README 3-extraction GET/PUT/Other definition (BASELINE.json configs[1]), "
                                   "%d x %d B lines per GPU, seed 2+rank" % (n, W.LINE_BYTES),
                       "lines_per_gpu": n, "line_bytes": W.LINE_BYTES, "offsets": "u32",
                       "match_dfa_states": int(gorp.stat(0)), "char_classes": int(gorp.stat(1)),
                       "capture_states": int(gorp.stat(2)), "table_blob_bytes": int(gorp.stat(4)),
                       "parallelism": "lines sharded by rank (dp%d)" % world},
            "gb_per_s_scanned": total_bytes * world * steps / elapsed / 1e9,
            "kernel_ms": {"avg": k_avg, "min": k_sorted[0], "median": k_sorted[len(k_sorted) // 2]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_unit": "bytes/launch",
                         "traffic_source": traffic_src,
                         "algorithmic_read_bytes": algo_read, "algorithmic_write_bytes": algo_write,
                         "frac_of_measured_copy_ceiling": achieved / 6290.0},
            "setup_s": setup_s,
            "table_bcast_ms": bcast_ms,
            "gather_ms": gather_ms,
            "gather_compact_ms": gather_compact_ms,
        }
        if not args.no_cpu_baseline and world == 1:
            sample = min(n, 10_000_000)
            d_cpu = data[: sample * W.LINE_BYTES].cpu().numpy()
            o_cpu = offsets[: sample + 1].cpu().numpy().astype(np.uint32)
            base, omid, ocaps, ns = cpu_baseline(definition, d_cpu, o_cpu)
            # the baseline run doubles as a parity check of the timed GPU output
            if not (np.array_equal(mid[:ns].cpu().numpy(), omid) and np.array_equal(caps[:ns].cpu().numpy(), ocaps)):
                raise SystemExit("bench: GPU results differ from the oracle on the baseline sample")
            out["cpu_baseline"] = base
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
