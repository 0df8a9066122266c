"""One rank of a CPU rehearsal of the multi-GPU path, started by raytrace_amd/launch.py the way bench.py starts
its ranks when no launcher did: rendezvous from the environment, row tiles of one frame rendered by the oracle
(standing in for the GPU), gathered to rank 0 with the product's gather code.  TEST HARNESS ONLY.
argv: out.npy nx ny spp tile_rows"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    out_path, nx, ny, spp, tile_rows = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["LOCAL_RANK"]) == rank
    dist.init_process_group("gloo")                       # env:// rendezvous, as under torch.distributed.run
    from oracle import oracle as O
    from raytrace_amd.sharding import FrameShards, gather_frame
    shards = FrameShards(nx, ny, spp, world, tile_rows, seed=4)
    shards.check()
    orc = O.Oracle()
    sc, cam = orc.scene_small(), orc.camera_default(nx, ny)
    tiles = torch.zeros((shards.max_rows, nx, 3), dtype=torch.uint8)
    rows = shards.rows_of(rank)
    for k, row in enumerate(rows):
        tiles[k] = torch.from_numpy(orc.render_det(cam, sc, nx, ny, spp, seed=4, row0=int(row), rows=1, host_threads=1,
                                                   want_debug=False)["rgb8"][0])
    frame = gather_frame(shards, tiles, rank)
    # per-rank figures to rank 0, as bench.py gathers kernel times
    mine = torch.tensor([float(rank), float(len(rows))], dtype=torch.float64)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    dist.barrier()
    if rank == 0:
        assert [int(v[0]) for v in allv] == list(range(world)) and sum(int(v[1]) for v in allv) == ny
        np.save(out_path, frame.numpy())
        print("rank0 done", flush=True)
    dist.destroy_process_group()
    if os.environ.get("RT_TEST_FAIL_RANK") == str(rank):
        sys.exit(3)


if __name__ == "__main__":
    main()
