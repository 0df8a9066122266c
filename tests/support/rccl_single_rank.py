"""The collective calls bench.py makes between ranks, on a one-rank RCCL group on GPU 0: a one-GPU box cannot hold
two RCCL ranks, so this walks the same calls (device-bound init, gather of uint8 tiles to rank 0 through the
product's gather code path, float64 MAX all-reduce, all-gather, barrier) with world size 1.  TEST HARNESS ONLY."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)                         # as bench.py: everything on a side stream of torch's
    tiles = torch.arange(24 * 40 * 3, device=dev).reshape(24, 40, 3).to(torch.uint8)
    gathered = [torch.zeros_like(tiles)]
    dist.gather(tiles, gathered, dst=0)
    assert torch.equal(gathered[0], tiles)
    t = torch.tensor([1.25], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t.item()) == 1.25
    mine = torch.tensor([3.0, 4.0, 5.0], dtype=torch.float64, device=dev)
    allv = [torch.zeros_like(mine)]
    dist.all_gather(allv, mine)
    assert torch.equal(allv[0], mine)
    dist.barrier()
    torch.cuda.synchronize(dev)
    dist.destroy_process_group()
    print("rccl single rank ok", flush=True)


if __name__ == "__main__":
    main()
