"""How evenly the multi-GPU split of the baseline's 3840x2160 frame loads the ranks, measured on ONE GPU:
the jobs the N ranks would run (8-row tiles dealt round-robin, raytrace_amd.interleaved_job) are rendered one after
the other with RT_FLAG_SERIAL and each one's trace-kernel time is taken from the library's HIP events.  The same for
N contiguous row bands (the reference's partition, c/main.c:510-529) as the control.  max/mean of the per-rank times
is what the slowest rank costs the frame: N * mean / max is the speed-up the split allows before any other loss.

    python3 benchmarks/tile_balance.py [--spp 64] [--ranks 2 4 8] [--repeats 3]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--nx", type=int, default=3840)
ap.add_argument("--ny", type=int, default=2160)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--ranks", type=int, nargs="+", default=[2, 4, 8])
ap.add_argument("--tile-rows", type=int, default=8)
ap.add_argument("--repeats", type=int, default=3)
args = ap.parse_args()

import torch  # noqa: E402  (device buffers only)

dev = torch.device("cuda", 0)
r = rt.Renderer(0)
r.upload(rt.random_scene(11))
cam = rt.camera_default(args.nx, args.ny)
out_buf = torch.zeros((args.ny, args.nx, 3), dtype=torch.uint8, device=dev)


def kernel_ms(job):
    """median over the repeats of the trace-kernel time of one job rendered alone"""
    times = []
    for _ in range(args.repeats + 1):
        r.render_device(cam, job, out_buf.data_ptr())
        r.synchronize()
        times.append(r.trace_ms(0))
    return float(np.median(times[1:]))


def band_job(rank, world):
    """rank's contiguous band of c/main.c:510-529: remaining rows / remaining threads"""
    row0 = 0
    for i in range(world):
        rows = (args.ny - row0) // (world - i)
        if i == rank:
            return rt.Job(args.nx, args.ny, args.spp, row0=row0, tile_rows=rows, n_tiles=1, flags=rt.FLAG_SERIAL)
        row0 += rows
    raise AssertionError


# clocks up
whole = rt.Job(args.nx, args.ny, args.spp, flags=rt.FLAG_SERIAL)
for _ in range(3):
    r.render_device(cam, whole, out_buf.data_ptr())
r.synchronize()
whole_ms = kernel_ms(whole)

report = {"frame": f"final scene, {args.nx}x{args.ny}, {args.spp} spp (the baseline's frame has 1024: the per-rank times scale with spp, "
                   "their ratios do not)", "tile_rows": args.tile_rows,
          "whole_frame_kernel_ms_one_gpu": whole_ms, "splits": []}
for world in args.ranks:
    for kind in ("interleaved 8-row tiles", "contiguous bands"):
        ms, rows = [], []
        for rank in range(world):
            job = (rt.interleaved_job(args.nx, args.ny, args.spp, rank, world, args.tile_rows, flags=rt.FLAG_SERIAL)
                   if kind.startswith("interleaved") else band_job(rank, world))
            ms.append(kernel_ms(job))
            rows.append(rt.job_rows(job))
        mean = float(np.mean(ms))
        report["splits"].append({"ranks": world, "partition": kind, "rows_per_rank": rows, "kernel_ms_per_rank": ms,
                                 "max_over_mean": max(ms) / mean, "sum_over_whole_frame": sum(ms) / whole_ms,
                                 "speedup_ceiling_from_balance": world * mean / max(ms),
                                 "speedup_ceiling_vs_one_gpu_whole_frame": whole_ms / max(ms)})
r.close()
print(json.dumps(report, indent=1))
