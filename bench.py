#!/usr/bin/env python3
"""bench.py -- camera paths/s of the per-pixel path-tracing loop on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME]

A step is one render of the workload's frame.  N = 1: the final scene (488
spheres) at 1920x1080, 64 spp, max depth 50 (BASELINE config 1).  N > 1
(launched by torch.distributed.run, one rank per GPU): the same frame at 64*N
spp, its 8-row tiles dealt round-robin to the ranks, every rank's tiles
gathered to rank 0 each step (the only communication) -- weak scaling, the
same number of paths per GPU.  Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3      # MI355X vector FP32 peak (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz)
FLOP_PER_TEST = 17            # one ray-sphere discriminant, a and r^2 hoisted (3 sub, 5, 6, 3)

WORKLOADS = {
    # name: (scene builder name, half_extent, nx, ny, spp, max_depth)
    "final1080": ("random_scene", 11, 1920, 1080, 64, 50),
    "final4k": ("random_scene", 11, 3840, 2160, 1024, 50),
    "dense1080": ("random_scene", 100, 1920, 1080, 16, 50),
    "glass1080": ("glass_scene", 11, 1920, 1080, 256, 50),
    "small600": ("small_scene", None, 600, 300, 10, 50),
}


def cpu_baseline(nx, ny, scene_name, half_extent, max_depth):
    """The oracle (the build's C restatement of the reference, byte-identical to
    it on the goldens) timed on this host's cores on a bounded sample."""
    from oracle import oracle as O
    orc = O.Oracle()
    sc = {"random_scene": lambda: orc.scene_random(half_extent or 11), "glass_scene": lambda: orc.scene_glass(half_extent or 11),
          "small_scene": orc.scene_small}[scene_name]()
    cores = max(1, min(16, os.cpu_count() or 1))
    cam = orc.camera_default(nx, ny)
    spp = 4 if len(sc) < 2000 else 1
    rows = ny if len(sc) < 2000 else max(8, ny // 60)
    row0 = (ny - rows) // 2
    t0 = time.perf_counter()
    out = orc.render_det(cam, sc, nx, ny, spp, max_depth=max_depth, seed=0, row0=row0, rows=rows, host_threads=cores,
                         want_debug=False)
    dt = time.perf_counter() - t0
    paths = out["stats"]["paths"]
    res = {"value": paths / dt, "unit": "camera paths/s", "cores": cores, "kind": "port",
           "sample": f"{nx}x{ny} rows {row0}..{row0 + rows - 1}, {spp} spp, {paths} paths in {dt:.2f} s, "
                     f"oracle/rt_oracle.c -O2 -ffp-contract=off, {cores} pthreads"}
    # the reference binary itself (600x300 is hard-coded in it), when it was built
    if scene_name == "random_scene" and (half_extent or 11) == 11 and os.path.exists(O.REF_BIN_FINAL):
        try:
            t0 = time.perf_counter()
            subprocess.run([O.REF_BIN_FINAL, "-n", "10", "-t", str(cores), "-o", "-"], stdout=subprocess.DEVNULL,
                           check=True, timeout=300)
            dt = time.perf_counter() - t0
            res["reference_binary"] = {"value": 600 * 300 * 10 / dt, "unit": "camera paths/s", "cores": cores,
                                       "sample": f"reference CLI, revived random_scene, 600x300 -n 10 -t {cores}, {dt:.2f} s"}
        except Exception as e:  # the baseline is a report, never a reason to fail the bench
            res["reference_binary"] = {"error": str(e)}
    return res


def hbm_traffic(kernel_key):
    """HBM bytes per launch of the trace kernel from the last committed rocprofv3 counter passes
    (profiles/pmc_latest.json, written by benchmarks/summarize_pmc.py): WRITE_SIZE is exact for the 16-byte
    stores of the radiance buffer, FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950 and is doubled."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_latest.json")) as f:
            d = json.load(f)[kernel_key]
        extra = {k: d.get(k) for k in ("VALUBusy", "VALUUtilization", "SQ_INSTS_VALU") if d.get(k) is not None}
        return (d["WRITE_SIZE_KB"] + 2.0 * d["FETCH_SIZE_KB"]) * 1024.0, d.get("source"), extra
    except Exception:
        return None, None, {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="final1080", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--kernel", default="mega", choices=["mega", "wavefront"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-scan", action="store_true", help="RT_FLAG_FULL_SCAN: filter every sphere for every ray (no group culling)")
    ap.add_argument("--no-linear-scan-leg", action="store_true", help="skip the extra (untimed for `value`) full-scan measurement")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytrace_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run (one rank per GPU)")
    # RT_BENCH_REHEARSAL=1: every rank on GPU 0 and the gather over gloo on host copies -- a way to walk the
    # N > 1 control flow on a one-GPU box; its numbers mean nothing
    rehearsal = os.environ.get("RT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    scene_name, half_extent, nx, ny, spp, max_depth = WORKLOADS[args.workload]
    spp = args.spp or spp
    spp_total = spp * world                      # weak scaling: same paths per GPU
    sc = getattr(rt, scene_name)(half_extent) if half_extent else getattr(rt, scene_name)()
    cam = rt.camera_default(nx, ny)
    kernel = rt.KERNEL_MEGA if args.kernel == "mega" else rt.KERNEL_WAVEFRONT

    r = rt.Renderer(local_rank)
    # a side stream of torch's: the kernels, the events that time them and the gather all run on it
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    r.set_stream(stream.cuda_stream)
    r.upload(sc)
    from raytrace_amd.sharding import FrameShards, gather_frame
    tile_rows = 8
    shards = FrameShards(nx, ny, spp_total, world, tile_rows, max_depth=max_depth, seed=0, kernel=kernel,
                         flags=2 if args.full_scan else 0)
    shards.check()
    job = shards.job(rank)
    my_rows = rt.job_rows(job)
    tiles = torch.zeros((shards.max_rows, nx, 3), dtype=torch.uint8, device=dev)       # this rank's tiles, padded
    gathered = [torch.zeros_like(tiles) for _ in range(world)] if (world > 1 and rank == 0) else None
    frame = torch.zeros((ny, nx, 3), dtype=torch.uint8, device=dev) if rank == 0 else None

    def step(ev=None):
        if ev:
            ev[0].record(stream)
        r.render_device(cam, job, tiles.data_ptr())
        if ev:
            ev[1].record(stream)
        if rehearsal and world > 1:
            stream.synchronize()
            host = gather_frame(shards, tiles.cpu(), rank)
            if rank == 0:
                frame.copy_(host)
        else:
            gather_frame(shards, tiles, rank, frame, gathered)     # the only communication: tiles -> rank 0

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # bring the clocks up before anything is measured (the first frames after idle run 5-10 % slower);
    # these frames are neither warm-up steps nor timed steps
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.5:
        r.render_device(cam, job, tiles.data_ptr())
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    fence()
    r.counters(reset=True)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    fence()
    dt = time.perf_counter() - t0

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    counters = r.counters()
    launch_ms = [a.elapsed_time(b) for a, b in events]                      # trace + resolve, torch events
    kernel_ms = [r.trace_ms(back) for back in range(args.steps)]            # trace kernel alone, the library's HIP events
    paths_per_step_rank = my_rows * nx * spp_total
    total_paths_per_step = nx * ny * spp_total

    # the same frame with every sphere filtered for every ray (the reference's linear scan, nothing culled):
    # not part of `value`, reported beside it because it is the mode the FP32 roofline argument is about
    linear = None
    if world == 1 and rank == 0 and args.kernel == "mega" and not args.full_scan and not args.no_linear_scan_leg:
        ljob = FrameShards(nx, ny, spp_total, world, tile_rows, max_depth=max_depth, seed=0, kernel=kernel, flags=2).job(rank)
        r.render_device(cam, ljob, tiles.data_ptr())
        torch.cuda.synchronize(dev)
        r.counters(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            r.render_device(cam, ljob, tiles.data_ptr())
        torch.cuda.synchronize(dev)
        ldt = time.perf_counter() - t0
        lc = r.counters()
        l_ms = float(np.mean([r.trace_ms(back) for back in range(args.steps)]))
        l_ach = FLOP_PER_TEST * (lc["sphere_tests"] + lc["box_tests"]) / args.steps / (l_ms * 1e-3) / 1e12
        linear = {"value": total_paths_per_step * args.steps / ldt, "unit": "camera paths/s", "ms_per_step": ldt / args.steps * 1e3,
                  "roofline": {"bound": "fp32_valu", "achieved": l_ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": l_ach / FP32_PEAK_TFLOPS, "kernel": "trace_kernel (RT_FLAG_FULL_SCAN)",
                               "kernel_ms_avg": l_ms, "tests_per_path": (lc["sphere_tests"] + lc["box_tests"]) / max(1, lc["paths"])}}

    if rank == 0:
        k_ms = float(np.mean(kernel_ms))
        tests_per_launch = (counters["sphere_tests"] + counters["box_tests"]) / args.steps
        achieved = FLOP_PER_TEST * tests_per_launch / (k_ms * 1e-3) / 1e12
        # the counter passes were collected on this workload's 64-spp full frame; scaled to this rank's share
        traffic, traffic_source, counters_pmc = (hbm_traffic("full" if args.full_scan else "culled")
                                                 if args.kernel == "mega" and args.workload == "final1080" else (None, None, {}))
        if traffic is not None:
            traffic *= spp_total / 64.0 * my_rows / ny
        out = {
            "metric": "camera paths/sec (Msamples/s) final scene 1920x1080; % FP32 VALU peak",
            "value": total_paths_per_step * args.steps / dt,
            "unit": "camera paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {scene_name} {len(sc)} spheres, {nx}x{ny}, {spp_total} spp "
                                   f"({spp} per GPU), max_depth {max_depth}, deterministic counter-RNG mode, "
                                   f"{tile_rows}-row tiles round-robin over {world} GPU(s), gather to rank 0",
                       "kernel": args.kernel, "scan": "full" if args.full_scan else "group-culled", "paths_per_step": total_paths_per_step,
                       "msamples_per_s": total_paths_per_step * args.steps / dt / 1e6},
            "roofline": {"bound": "fp32_valu", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP32_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": "trace_kernel" if args.kernel == "mega" else "wavefront stages",
                         "kernel_ms_avg": k_ms, "render_ms_avg": float(np.mean(launch_ms)),
                         "flop_per_test": FLOP_PER_TEST,
                         "tests_per_launch": tests_per_launch,
                         "tests_per_path": (counters["sphere_tests"] + counters["box_tests"]) / max(1, counters["paths"]),
                         "segments_per_path": counters["segments"] / max(1, counters["paths"]),
                         "exact_evaluations_per_segment": counters["candidates"] / max(1, counters["segments"]),
                         "rank0_paths_per_launch": paths_per_step_rank},
        }
        if counters_pmc:
            # from the same committed counter passes: how busy the vector ALU was and how many of its lanes were active
            out["roofline"]["pmc"] = {"valu_busy_pct": counters_pmc.get("VALUBusy"), "valu_thread_efficiency_pct": counters_pmc.get("VALUUtilization"),
                                      "valu_wave_instructions_per_launch": counters_pmc.get("SQ_INSTS_VALU")}
        if linear is not None:
            out["linear_scan"] = linear
            # what the culled kernel delivers in units of the linear scan's work
            out["roofline"]["linear_scan_equivalent_frac"] = (FLOP_PER_TEST * linear["roofline"]["tests_per_path"] * out["value"]
                                                              / 1e12 / FP32_PEAK_TFLOPS)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, ny, scene_name, half_extent, max_depth)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
