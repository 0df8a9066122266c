"""The N > 1 path on the CPU: two gloo ranks shard one frame by row tiles,
each renders its tiles (here with the oracle standing in for the GPU), the
tiles are gathered to rank 0 with the product's gather code, and the frame
must equal the single-rank frame byte for byte."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, nx, ny, spp, tile_rows, out_path):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from raytrace_amd.sharding import FrameShards, gather_frame
    import raytrace_amd as rt

    shards = FrameShards(nx, ny, spp, world, tile_rows, seed=4)
    shards.check()
    orc = O.Oracle()
    sc = orc.scene_small()
    cam = orc.camera_default(nx, ny)
    tiles = torch.zeros((shards.max_rows, nx, 3), dtype=torch.uint8)
    job = shards.job(rank)
    done = 0
    for t in range(job.n_tiles):                      # what rt_render_device does for this rank's job
        first = job.row0 + t * job.tile_stride
        if first >= ny:
            break
        cnt = min(job.tile_rows, ny - first)
        part = orc.render_det(cam, sc, nx, ny, spp, seed=4, row0=first, rows=cnt, host_threads=1, want_debug=False)
        tiles[done:done + cnt] = torch.from_numpy(part["rgb8"])
        done += cnt
    assert done == rt.job_rows(job)
    frame = gather_frame(shards, tiles, rank)
    dist.barrier()
    if rank == 0:
        np.save(out_path, frame.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ny,tile_rows", [(2, 37, 8), (2, 24, 5)])
def test_two_ranks_shard_and_gather(tmp_path, oracle, world, ny, tile_rows):
    nx, spp = 48, 2
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, _free_port(), nx, ny, spp, tile_rows, out), nprocs=world, join=True)
    got = np.load(out)
    want = oracle.render_det(oracle.camera_default(nx, ny), oracle.scene_small(), nx, ny, spp, seed=4, want_debug=False)["rgb8"]
    assert np.array_equal(got, want)


def _launcher():
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_launch", os.path.join(ROOT, "raytrace_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.mark.parametrize("world", [2, 3])
def test_self_launched_ranks_shard_and_gather(tmp_path, oracle, world):
    """bench.py --gpus N without a launcher starts its own ranks (raytrace_amd/launch.py): RANK / WORLD_SIZE /
    MASTER_* in the environment of N fresh processes, rendezvous on 127.0.0.1, exit code of the worst rank."""
    nx, ny, spp, tile_rows = 40, 21, 2, 4
    out = str(tmp_path / "frame.npy")
    script = os.path.join(ROOT, "tests", "support", "gloo_rank.py")
    cmd = [sys.executable, script, out, str(nx), str(ny), str(spp), str(tile_rows)]
    rc = _launcher().launch_ranks(world, cmd, timeout=300)
    assert rc == 0
    want = oracle.render_det(oracle.camera_default(nx, ny), oracle.scene_small(), nx, ny, spp, seed=4, want_debug=False)["rgb8"]
    assert np.array_equal(np.load(out), want)
    # a failing rank is the launcher's exit code
    assert _launcher().launch_ranks(2, cmd, env_extra={"RT_TEST_FAIL_RANK": "1"}, timeout=300) == 3


def test_a_rank_that_ignores_sigterm_is_killed():
    """One rank fails at once, the other ignores SIGTERM and sleeps (a rank stuck inside a GPU or collective call
    behaves like that): the launcher kills it by pid after its grace period and returns the failing rank's code."""
    import time
    code = ("import os, signal, sys, time\n"
            "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
            "sys.exit(7) if os.environ['RANK'] == '1' else time.sleep(120)\n")
    t0 = time.monotonic()
    rc = _launcher().launch_ranks(2, [sys.executable, "-c", code], timeout=60, kill_after=1.0)
    assert rc == 7 and time.monotonic() - t0 < 30


def test_bench_parent_spawns_before_touching_a_gpu():
    """The self-launch branch of bench.py runs before torch or the library is imported."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("sys.exit(self_launch(args))") < main.index("import torch")
    launch_src = open(os.path.join(ROOT, "raytrace_amd", "launch.py")).read()
    imports = [ln for ln in launch_src.splitlines() if ln.startswith(("import ", "from "))]
    assert imports and all(ln.split()[1] in ("os", "socket", "subprocess", "sys", "time") for ln in imports)
