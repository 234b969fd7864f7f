"""One-process-per-GPU plumbing around the batch path (torch.distributed; backend "nccl" = RCCL
on ROCm, "gloo" in the CPU tests).

Lines are independent (core/Gorp.java:159-186 keeps no cross-line state), so a batch shards by line
with no collective on the data path.  The only exchanges are
  * once per definition: rank 0 compiles, the packed table blob is broadcast;
  * once per job (optional): the per-line results are gathered to rank 0.
"""
import numpy as np
import torch
import torch.distributed as dist

from .gorp import Gorp


def broadcast_gorp(definition, device, src=0, host_only=False):
    """Rank `src` compiles `definition`; every rank gets a Gorp built from the same blob."""
    rank = dist.get_rank()
    size = torch.zeros(1, dtype=torch.int64, device=device)
    gorp = None
    if rank == src:
        gorp = Gorp.construct(definition, host_only=host_only)
        blob = torch.from_numpy(gorp.blob()).to(device)
        size[0] = blob.numel()
    dist.broadcast(size, src=src)
    if rank != src:
        blob = torch.empty(int(size.item()), dtype=torch.uint8, device=device)
    dist.broadcast(blob, src=src)
    if rank != src:
        # names / extractor lists are host metadata: rebuilt locally without compiling tables for the device
        cooked = [_cook(i, e) for i, e in enumerate(definition)]
        gorp = Gorp.from_blob(blob.cpu().numpy(), cooked, host_only=host_only)
    return gorp, int(size.item())


def _cook(i, ext):
    from .gorp import CookedExtraction
    _, jdk, names = ext.build()
    return CookedExtraction(i, ext.name, jdk, names, ext.append)


def shard_bounds(offsets, rank, world):
    """Contiguous line range [lo, hi) of `rank`, balanced by BYTES (matters for ragged lines)."""
    offsets = np.asarray(offsets)
    n = len(offsets) - 1
    total = int(offsets[-1])
    cuts = [int(np.searchsorted(offsets, total * r // world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, n
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts[rank], cuts[rank + 1]


def shard_csr(data, offsets, rank, world):
    """This rank's lines as a self-contained CSR pair (offsets rebased to 0)."""
    lo, hi = shard_bounds(offsets, rank, world)
    offsets = np.asarray(offsets)
    b0, b1 = int(offsets[lo]), int(offsets[hi])
    return np.asarray(data)[b0:b1], (offsets[lo:hi + 1] - offsets[lo]).astype(offsets.dtype), lo, hi


def gather_results_compact(match_id, caps, dst=0):
    """gather_results with the compact transport rows of gx_pack_results (int16 id + uint16 offsets: 18 instead
    of 36 bytes per line for 4 groups), packed and unpacked by HIP kernels.  Device tensors only.  Falls back to
    the wide rows when any rank holds an offset above 65534 (lines longer than 64 KiB)."""
    from .gorp import pack_results_device, unpack_results_device
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = match_id.device
    n_loc, slots = match_id.shape[0], caps.shape[1]
    stream = torch.cuda.current_stream().cuda_stream
    sizes_t = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    packed = torch.empty((n_loc, slots + 1), dtype=torch.int16, device=dev)
    over = pack_results_device(match_id.data_ptr(), caps.data_ptr(), n_loc, slots, packed.data_ptr(), stream=stream)
    dist.all_gather(sizes_t, torch.tensor([n_loc, over], dtype=torch.int64, device=dev))
    sizes = [int(s[0].item()) for s in sizes_t]
    if any(int(s[1].item()) for s in sizes_t):
        return gather_results(match_id, caps, dst=dst)
    padded = packed
    if n_loc != max(sizes):
        padded = torch.zeros((max(sizes), slots + 1), dtype=torch.int16, device=dev)
        padded[:n_loc] = packed
    wire = padded.view(torch.uint8)  # (RCCL has no 16-bit integer type: the rows travel as bytes)
    bufs = [torch.empty_like(wire) for _ in range(world)] if rank == dst else None
    dist.gather(wire, bufs, dst=dst)
    if rank != dst:
        return None, None
    total = sum(sizes)
    rows = torch.cat([b.view(torch.int16)[:s] for b, s in zip(bufs, sizes)], dim=0).contiguous()
    mid = torch.empty(total, dtype=torch.int32, device=dev)
    cp = torch.empty((total, slots), dtype=torch.int32, device=dev)
    unpack_results_device(rows.data_ptr(), total, slots, mid.data_ptr(), cp.data_ptr(), stream=stream)
    return mid, cp


def gather_rows(rows, dst=0, sizes=None):
    """Gather the compact result rows the kernels write themselves (gx_batch_opts.compact_results: int16 id + uint16
    offsets per line, [n, 1 + slots] int16/uint16 tensors) on `dst` in rank order -- no pack step, half the bytes of
    the dense rows on the xGMI links.  Shards may differ in length.  Returns the concatenated rows (or None).
    sizes: every rank's row count when the caller knows them (no size exchange, no host read-back: the call only
    enqueues -- what a pipeline that gathers batch k under the kernel of batch k + 1 needs)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = rows.device
    if sizes is None:
        n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n)
        sizes = [int(s.item()) for s in sizes]
    assert len(sizes) == world and sizes[rank] == rows.shape[0]
    padded = rows
    if rows.shape[0] != max(sizes):
        padded = torch.zeros((max(sizes), rows.shape[1]), dtype=rows.dtype, device=dev)
        padded[:rows.shape[0]] = rows
    wire = padded.contiguous().view(torch.uint8)  # (RCCL has no 16-bit integer type: the rows travel as bytes)
    bufs = [torch.empty_like(wire) for _ in range(world)] if rank == dst else None
    dist.gather(wire, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b.view(rows.dtype)[:s] for b, s in zip(bufs, sizes)], dim=0).contiguous()


def gather_results(match_id, caps, dst=0):
    """Gather per-line results of all ranks on `dst` in rank order.  Shards may differ in length:
    they are padded to the longest one for the collective and trimmed afterwards."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = match_id.device
    n = torch.tensor([match_id.shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    width = 1 + caps.shape[1]
    packed = torch.full((max(sizes), width), -1, dtype=torch.int32, device=dev)
    packed[:match_id.shape[0], 0] = match_id
    if caps.shape[1]:
        packed[:match_id.shape[0], 1:] = caps
    bufs = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, bufs, dst=dst)
    if rank != dst:
        return None, None
    out = torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
    return out[:, 0].contiguous(), out[:, 1:].contiguous()
