"""Worker for tests/test_gpu_parity.py::test_two_ranks_share_one_gpu: run with torch.distributed.run, 2 ranks, both
on cuda:0, backend gloo (RCCL refuses two ranks on one device; the 8-GPU RCCL run is the driver's).  Exercises
the N > 1 plumbing of bench.py on device tensors: blob broadcast, byte-balanced sharding, extraction on the GPU,
wide and compact gathers with unequal shard sizes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

from gorp_amd import dist as gdist
from gorp_amd import workloads as W


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    definition = W.readme3_definition()
    gorp, blob_bytes = gdist.broadcast_gorp(definition, dev)
    assert blob_bytes > 1000
    data, offsets, cat = W.readme3_lines(30011, seed=5)
    d, o = data.numpy(), offsets.numpy()
    # unequal shards on purpose: rank 0 gets 1/3 of the lines
    cut = len(o) // 3
    lo, hi = (0, cut) if rank == 0 else (cut, len(o) - 1)
    sd = torch.from_numpy(d[o[lo]:o[hi]].copy()).to(dev)
    so = torch.from_numpy((o[lo:hi + 1] - o[lo]).astype(np.int64)).to(dev).to(torch.int32)
    n = hi - lo
    mid = torch.empty(n, dtype=torch.int32, device=dev)
    caps = torch.empty((n, 2 * gorp.max_groups), dtype=torch.int32, device=dev)
    gorp.extract_batch_device(sd.data_ptr(), so.data_ptr(), n, mid.data_ptr(), caps.data_ptr())
    assert torch.equal(mid.cpu(), cat[lo:hi].to(torch.int32))
    gm, gc = gdist.gather_results(mid, caps, dst=0)
    cm, cc = gdist.gather_results_compact(mid, caps, dst=0)
    if rank == 0:
        assert gm.shape[0] == len(o) - 1 and torch.equal(gm.cpu(), cat.to(torch.int32))
        assert torch.equal(cm, gm) and torch.equal(cc, gc)
        print("dist_gpu_worker ok: %d lines over %d ranks" % (gm.shape[0], world))
    else:
        assert gm is None and cm is None
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
