#!/usr/bin/env python3
"""bench.py -- camera paths/s of the per-pixel path-tracing loop on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--scaling weak|strong]

A step is one render of the workload's frame, inputs resident on the GPU.

N = 1: the final scene (488 spheres) at 1920x1080, 64 spp, max depth 50 -- the configuration the
baseline metric is quoted on.

N > 1, one rank per GPU: by default the baseline's multi-GPU configuration, the FIXED 3840x2160 frame at
1024 spp cut into 8-row tiles dealt round-robin to the ranks (strong scaling: the generalisation of the
row bands of c/main.c:510-529), every rank's tiles gathered to rank 0 each step -- the only
communication.  `--scaling weak` renders the N = 1 frame at 64*N spp instead; a short leg of that kind is
always reported beside a strong run.  The ranks are started by torch.distributed.run (RANK / WORLD_SIZE in
the environment) or, when the script is run plainly with --gpus N, by the script itself: the parent spawns
N fresh rank processes before anything touches a GPU and relays rank 0's line.

Prints ONE JSON line on rank 0.
"""
import argparse
import csv
import glob
import importlib.util
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP32_PEAK_TFLOPS = 157.3      # MI355X vector FP32 peak (256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz, packed FMA)
FLOP_PER_TEST = 17            # one ray-sphere discriminant, a and r^2 hoisted (3 sub, 5, 6, 3)
FLOP_PER_BOX_TEST = 18        # one slab test as executed: 9 fused multiply-adds (min/max/compare not counted)
SIMDS = 256 * 4
TRACE_HISTORY = 32            # renders whose trace-kernel events the library keeps

WORKLOADS = {
    # name: (scene builder name, half_extent, nx, ny, spp, max_depth)
    "final1080": ("random_scene", 11, 1920, 1080, 64, 50),
    "final4k": ("random_scene", 11, 3840, 2160, 1024, 50),
    "dense1080": ("random_scene", 100, 1920, 1080, 16, 50),
    "glass1080": ("glass_scene", 11, 1920, 1080, 256, 50),
    "small600": ("small_scene", None, 600, 300, 10, 50),
}


def host_cores():
    """Cores this process may really use: its affinity mask, cut to the cgroup's CPU quota and to the worker-pool
    size the box announces (OMP_NUM_THREADS) -- a shared machine shows every CPU in the mask."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            lim, per = f.read().split()[:2]
        if lim != "max":
            quota = max(1, int(int(lim) / int(per)))
    except Exception:
        pass
    pool = None
    try:
        pool = int(os.environ.get("OMP_NUM_THREADS", ""))
    except ValueError:
        pass
    cores = min(x for x in (affinity, quota, pool) if x)
    return max(1, cores), f"affinity {affinity}, cgroup quota {quota}, OMP_NUM_THREADS {pool}, online {os.cpu_count()}"


def cpu_baseline(nx, ny, scene_name, half_extent, max_depth):
    """The oracle (the build's C restatement of the reference, byte-identical to it on the goldens) timed on this
    host's cores on a bounded sample: the parity build (-ffp-contract=off) and a -O2 -march=native build made
    here for this host (the reference Makefile's flags)."""
    from oracle import oracle as O
    cores, cores_note = host_cores()
    builds = [("parity", O.ORACLE_SO, "-O2 -ffp-contract=off")]
    try:
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle-native"], check=True, timeout=120,
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
        builds.append(("native", O.ORACLE_NATIVE_SO, "-O2 -march=native (the reference Makefile's flags)"))
        native_error = None
    except Exception as e:          # no compiler on the box: the parity build alone
        native_error = str(e)[:200]
    results = {}
    sample = None
    for tag, path, flags in builds:
        orc = O.Oracle(path)
        sc = {"random_scene": lambda: orc.scene_random(half_extent or 11), "glass_scene": lambda: orc.scene_glass(half_extent or 11),
              "small_scene": orc.scene_small}[scene_name]()
        cam = orc.camera_default(nx, ny)
        # about 10 s of work per build on 16 cores (the dense field is a linear scan of 40,004 spheres per ray)
        spp = 24 if len(sc) < 2000 else 1
        rows = ny if len(sc) < 2000 else max(8, ny // 12)
        row0 = (ny - rows) // 2
        t0 = time.perf_counter()
        out = orc.render_det(cam, sc, nx, ny, spp, max_depth=max_depth, seed=0, row0=row0, rows=rows, host_threads=cores,
                             want_debug=False)
        dt = time.perf_counter() - t0
        paths = out["stats"]["paths"]
        results[tag] = {"value": paths / dt, "seconds": dt, "flags": flags}
        sample = (f"{nx}x{ny} rows {row0}..{row0 + rows - 1}, {spp} spp, {paths} paths; oracle/rt_oracle.c, {cores} pthreads "
                  f"over interleaved rows ({cores_note})")
    cpu_model = None
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in f if ln.startswith("model name")), None)
    except OSError:
        pass
    best = max(results, key=lambda k: results[k]["value"])
    res = {"cpu_model": cpu_model, "value": results[best]["value"], "unit": "camera paths/s", "cores": cores, "kind": "port", "sample": sample,
           "build": results[best]["flags"], "builds": results}
    if native_error:
        res["native_build_error"] = native_error
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


# ---------------------------------------------------------------- live hardware counters

PMC_PASSES = [
    ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_INSTS_SALU",
            "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_SCA", "GRBM_GUI_ACTIVE"]),
    ("fetch", ["FETCH_SIZE"]),
    ("write", ["WRITE_SIZE"]),
]


PMC_CHILD_RENDERS = 2


def pmc_child(args):
    """Run under rocprofv3 by live_pmc(): two renders of the workload through the host-pointer entry, nothing else."""
    import raytrace_amd as rt
    scene_name, half_extent, nx, ny, spp, max_depth = WORKLOADS[args.workload or "final1080"]
    spp = args.spp or spp
    sc = getattr(rt, scene_name)(half_extent) if half_extent else getattr(rt, scene_name)()
    r = rt.Renderer(0)
    r.upload(sc)
    cam = rt.camera_default(nx, ny)
    job = rt.Job(nx, ny, spp, max_depth=max_depth, seed=0, flags=rt.FLAG_FULL_SCAN if args.full_scan else 0)
    for _ in range(PMC_CHILD_RENDERS):
        r.render_host(cam, job, debug=False)
    r.close()


def live_pmc(args, workload):
    """rocprofv3 --pmc around a child of this script, one pass per counter block (FETCH_SIZE and WRITE_SIZE do not
    fit one pass), each under a hard time limit.  Returns the trace kernel's counters per RENDER of the workload
    (summed over the launches of a render: a frame whose radiance does not fit one buffer takes several), or an error."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    out = {}
    tmp = tempfile.mkdtemp(prefix="rt_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for key in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(key, None)
    try:
        for name, counters in PMC_PASSES:
            d = os.path.join(tmp, name)
            cmd = ["timeout", "-s", "KILL", "150", exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", name, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", workload]
            if args.spp:
                cmd += ["--spp", str(args.spp)]
            if args.full_scan:
                cmd += ["--full-scan"]
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=200)
            if p.returncode != 0:
                return {"error": f"pass {name}: rocprofv3 exit {p.returncode}: {p.stdout.decode(errors='replace')[-300:]}"}
            sums, calls = {}, set()
            for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(path) as f:
                    for row in csv.DictReader(f):
                        if "trace_kernel" not in row["Kernel_Name"]:
                            continue
                        sums[row["Counter_Name"]] = sums.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                        calls.add(row["Dispatch_Id"])
            if not calls:
                return {"error": f"pass {name}: no trace_kernel rows in the counter output"}
            for c, v in sums.items():
                out[c] = v / PMC_CHILD_RENDERS
            out.setdefault("launches_profiled", {})[name] = len(calls)
            out["launches_per_render"] = len(calls) / PMC_CHILD_RENDERS
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


# ---------------------------------------------------------------- the benchmark proper

def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: final1080 (N = 1, or weak scaling), final4k (N > 1)")
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"], help="N > 1 only; default strong")
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel")
    ap.add_argument("--kernel", default="mega", choices=["mega", "wavefront"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full-scan", action="store_true", help="RT_FLAG_FULL_SCAN: filter every sphere for every ray (no group culling)")
    ap.add_argument("--no-linear-scan-leg", action="store_true", help="skip the extra (untimed for `value`) full-scan measurement")
    ap.add_argument("--no-live-pmc", action="store_true", help="skip the rocprofv3 counter passes (N = 1 only) that fill roofline.traffic and roofline.issue")
    ap.add_argument("--serial", action="store_true", help="RT_FLAG_SERIAL for the timed steps too: no overlap between renders (the form to "
                                                          "profile: every trace-kernel launch then shows its own duration)")
    ap.add_argument("--no-ramp", action="store_true", help="skip the untimed clock-ramp frames (profiling runs: keeps the kernel list uniform)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a launcher: spawn the N ranks.  Nothing here may touch a GPU."""
    spec = importlib.util.spec_from_file_location("rt_launch", os.path.join(ROOT, "raytrace_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.launch_ranks(args.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env_extra={"RT_BENCH_SELF_LAUNCHED": "1"})


def timed_steps(r, torch, dist, world, rehearsal, dev, step, steps):
    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
    events = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    fence()
    t0 = time.perf_counter()
    for k in range(steps):
        step(events[k])
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearsal else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    covered = min(steps, TRACE_HISTORY)
    # the trace kernel bracketed by the library's HIP events on the stream it runs on.  Renders are pipelined (the
    # next launch is dispatched while this one runs and then waits for compute units), so outside a serial leg
    # this is dispatch-to-completion, about two steps long, and not the kernel's own duration
    kernel_ms = [r.trace_ms(back) for back in range(covered)]
    # a step's completion on this rank's stream to the next step's: the marginal time per render in the pipeline
    spacing_ms = [events[k][1].elapsed_time(events[k + 1][1]) for k in range(steps - 1)]
    return dt, kernel_ms, spacing_ms


def main():
    args = parse_args()
    if args.pmc_child:
        return pmc_child(args)
    launched = "WORLD_SIZE" in os.environ
    if not launched and args.gpus > 1:
        sys.exit(self_launch(args))

    # Standard output carries the one JSON line and nothing else: whatever libraries print there while they start up
    # (the gloo backend announces its peers on stdout) goes to standard error instead.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    import raytrace_amd as rt

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE is {world}: start one rank per GPU")
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

    scaling = (args.scaling or "strong") if world > 1 else "weak"
    workload = args.workload or ("final4k" if (world > 1 and scaling == "strong") else "final1080")
    scene_name, half_extent, nx, ny, spp, max_depth = WORKLOADS[workload]
    spp = args.spp or spp
    spp_total = spp * world if scaling == "weak" else spp            # weak: same paths per GPU; strong: the same frame
    sc = getattr(rt, scene_name)(half_extent) if half_extent else getattr(rt, scene_name)()
    kernel = rt.KERNEL_MEGA if args.kernel == "mega" else rt.KERNEL_WAVEFRONT

    r = rt.Renderer(local_rank)
    # a side stream of torch's: the kernels, the events that time them and the gather all run on it
    stream = torch.cuda.Stream(dev)
    torch.cuda.set_stream(stream)
    r.set_stream(stream.cuda_stream)
    r.upload(sc)
    from raytrace_amd.sharding import FrameShards, gather_frame
    tile_rows = 8

    def make_leg(nx, ny, spp_total, flags):
        """one frame sharded over the ranks: (step function, this rank's rows)"""
        cam = rt.camera_default(nx, ny)
        shards = FrameShards(nx, ny, spp_total, world, tile_rows, max_depth=max_depth, seed=0, kernel=kernel, flags=flags)
        shards.check()
        job = shards.job(rank)
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
        return step, rt.job_rows(job)

    flags = (rt.FLAG_FULL_SCAN if args.full_scan else 0) | (rt.FLAG_SERIAL if args.serial else 0)
    step, my_rows = make_leg(nx, ny, spp_total, flags)

    # bring the clocks up before anything is measured (the first frames after idle run 5-10 % slower);
    # these frames are neither warm-up steps nor timed steps
    t_ramp = time.perf_counter()
    # (with several ranks the workload's step holds a collective, and ranks looping on their own clocks would call
    # it different numbers of times: they ramp on a local frame)
    if world > 1 or nx * ny * spp_total > 4 * 10 ** 8:                    # or a frame of seconds: ramp on a small one
        ramp_cam, ramp_job = rt.camera_default(1920, 1080), rt.Job(1920, 1080, 16, seed=0, kernel=kernel, flags=flags)
        ramp_out = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device=dev)
        while not args.no_ramp and time.perf_counter() - t_ramp < 0.5:
            r.render_device(ramp_cam, ramp_job, ramp_out.data_ptr())
            torch.cuda.synchronize(dev)
    else:                                                                    # the workload's own frame
        while not args.no_ramp and time.perf_counter() - t_ramp < 0.5:
            step()
            torch.cuda.synchronize(dev)
    if world > 1:
        step()      # the first collective builds the communicator: never inside the timed steps, whatever --warmup says
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize(dev)
    r.counters(reset=True)
    dt, bracketed_ms, spacing_ms = timed_steps(r, torch, dist, world, rehearsal, dev, step, args.steps)
    counters = r.counters()
    paths_per_step_rank = my_rows * nx * spp_total
    total_paths_per_step = nx * ny * spp_total
    step_ms_rank = float(np.mean(spacing_ms)) if spacing_ms else dt / args.steps * 1e3

    # The kernel alone: the same frame with RT_FLAG_SERIAL (every kernel on one stream, nothing overlapped), so
    # that the events around the trace kernel measure its duration.  This is the duration the roofline uses; the
    # counters of this leg give the tests per launch.
    serial_steps = min(args.steps, 10)
    sstep, _ = make_leg(nx, ny, spp_total, flags | rt.FLAG_SERIAL)
    sstep()
    torch.cuda.synchronize(dev)
    r.counters(reset=True)
    sdt, kernel_ms, _ = timed_steps(r, torch, dist, world, rehearsal, dev, sstep, serial_steps)
    serial_counters = r.counters()
    k_ms = float(np.mean(kernel_ms))

    # per-rank view: every rank's mean trace-kernel time per step and its share of the rows
    per_rank = None
    if world > 1:
        mine = torch.tensor([k_ms, float(my_rows), step_ms_rank], dtype=torch.float64, device="cpu" if rehearsal else dev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [[float(x) for x in v.cpu()] for v in allv]

    # weak-scaling leg beside a strong run: the N = 1 frame at 64*N spp, three steps
    weak = None
    if world > 1 and scaling == "strong" and scene_name == "random_scene" and (half_extent or 11) == 11:
        _, _, wnx, wny, wspp, _ = WORKLOADS["final1080"]
        wstep, _ = make_leg(wnx, wny, wspp * world, flags)
        wstep()
        wdt, _, _ = timed_steps(r, torch, dist, world, rehearsal, dev, wstep, 3)
        weak = {"value": wnx * wny * wspp * world * 3 / wdt, "unit": "camera paths/s", "ms_per_step": wdt / 3 * 1e3,
                "workload": f"final1080 at {wspp * world} spp ({wspp} per GPU)", "steps": 3}

    # the same frame with every sphere filtered for every ray (the reference's linear scan, nothing culled):
    # not part of `value`, reported beside it because it is the mode the FP32 roofline argument is about
    linear = None
    if world == 1 and rank == 0 and args.kernel == "mega" and not args.full_scan and not args.no_linear_scan_leg:
        lstep, _ = make_leg(nx, ny, spp_total, rt.FLAG_FULL_SCAN | rt.FLAG_SERIAL)
        lstep()
        torch.cuda.synchronize(dev)
        r.counters(reset=True)
        lsteps = min(args.steps, TRACE_HISTORY)
        ldt, l_kernel_ms, _ = timed_steps(r, torch, dist, world, rehearsal, dev, lstep, lsteps)
        lc = r.counters()
        l_ms = float(np.mean(l_kernel_ms))
        l_ach = FLOP_PER_TEST * lc["sphere_tests"] / lsteps / (l_ms * 1e-3) / 1e12
        linear = {"value": total_paths_per_step * lsteps / ldt, "unit": "camera paths/s", "ms_per_step": ldt / lsteps * 1e3,
                  "roofline": {"bound": "fp32_valu", "achieved": l_ach, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": l_ach / FP32_PEAK_TFLOPS, "kernel": "trace_kernel (RT_FLAG_FULL_SCAN)",
                               "kernel_ms_avg": l_ms, "sphere_tests_per_path": lc["sphere_tests"] / max(1, lc["paths"])}}

    if rank == 0:
        sphere_per_launch = serial_counters["sphere_tests"] / serial_steps
        box_per_launch = serial_counters["box_tests"] / serial_steps
        achieved = FLOP_PER_TEST * sphere_per_launch / (k_ms * 1e-3) / 1e12
        with_boxes = (FLOP_PER_TEST * sphere_per_launch + FLOP_PER_BOX_TEST * box_per_launch) / (k_ms * 1e-3) / 1e12
        roof = {"bound": "fp32_valu", "achieved": achieved, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / FP32_PEAK_TFLOPS, "traffic": None,
                "definition": "17 FLOP x ray-sphere tests the kernel executed (device counter) / mean duration of the trace kernel, both "
                              "from a leg of this run that renders the same frame with RT_FLAG_SERIAL (kernels one after the other on one "
                              "stream; HIP events around the kernel on that stream).  The timed steps behind `value` are pipelined: there a "
                              "launch's events span dispatch to completion (kernel_ms_bracketed_in_pipeline), including its wait for "
                              "compute units behind the launch before it",
                "kernel": "trace_kernel" if args.kernel == "mega" else "wavefront stages",
                "kernel_ms_avg": k_ms, "kernel_ms_steps_averaged": len(kernel_ms), "serial_leg_ms_per_step": sdt / serial_steps * 1e3,
                "kernel_ms_bracketed_in_pipeline": float(np.mean(bracketed_ms)), "pipelined_step_ms_rank0": step_ms_rank,
                "pipelined_step_ms_min_median_max": ([float(np.min(spacing_ms)), float(np.median(spacing_ms)), float(np.max(spacing_ms))]
                                                     if spacing_ms else None),
                "flop_per_test": FLOP_PER_TEST,
                "sphere_tests_per_launch": sphere_per_launch, "box_tests_per_launch": box_per_launch,
                "sphere_tests_per_path": serial_counters["sphere_tests"] / max(1, serial_counters["paths"]),
                "box_tests_per_path": serial_counters["box_tests"] / max(1, serial_counters["paths"]),
                "paths_counted_in_timed_steps": counters["paths"],
                "frac_with_box_tests": with_boxes / FP32_PEAK_TFLOPS, "flop_per_box_test": FLOP_PER_BOX_TEST,
                "segments_per_path": serial_counters["segments"] / max(1, serial_counters["paths"]),
                "exact_evaluations_per_segment": serial_counters["candidates"] / max(1, serial_counters["segments"]),
                "wave_trips_per_launch": serial_counters["wave_iterations"] / serial_steps,
                "lane_occupancy": serial_counters["segments"] / max(1, 64 * serial_counters["wave_iterations"]),
                "rank0_paths_per_launch": paths_per_step_rank}
        out = {
            "metric": "camera paths/sec (Msamples/s) final scene 1920x1080; % FP32 VALU peak",
            "value": total_paths_per_step * args.steps / dt,
            "unit": "camera paths/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{workload}: {scene_name} {len(sc)} spheres, {nx}x{ny}, {spp_total} spp"
                                   + (f" ({spp} per GPU)" if scaling == "weak" and world > 1 else "") +
                                   f", max_depth {max_depth}, deterministic counter-RNG mode, "
                                   f"{tile_rows}-row tiles round-robin over {world} GPU(s), gather to rank 0",
                       "kernel": args.kernel, "scan": "full" if args.full_scan else "group-culled", "paths_per_step": total_paths_per_step,
                       "renders": "one after the other (RT_FLAG_SERIAL)" if args.serial else "pipelined on the context",
                       "msamples_per_s": total_paths_per_step * args.steps / dt / 1e6,
                       "ranks_started_by": "bench.py itself" if (not launched or os.environ.get("RT_BENCH_SELF_LAUNCHED") == "1")
                                           else "a launcher (WORLD_SIZE was set)"},
            "roofline": roof,
        }
        if per_rank is not None:
            ks = [v[0] for v in per_rank]
            st = [v[2] for v in per_rank]
            out["per_rank"] = {"kernel_ms_serial_leg": ks, "rows": [int(v[1]) for v in per_rank], "pipelined_step_ms": st,
                               "kernel_ms_max_over_mean": max(ks) / (sum(ks) / len(ks)),
                               "pipelined_step_ms_max_over_mean": max(st) / (sum(st) / len(st))}
        if weak is not None:
            weak["note"] = ("this leg renders the N = 1 line's workload (final1080, 64 spp per GPU): its value over the N = 1 line's value "
                            "is the like-for-like scaling figure; `value` of this line is the fixed 3840x2160x1024 frame (strong scaling), "
                            "a different workload from the N = 1 line's")
            out["weak_scaling"] = weak
        if linear is not None:
            out["linear_scan"] = linear
            # what the culled kernel delivers in units of the linear scan's work
            roof["linear_scan_equivalent_frac"] = (FLOP_PER_TEST * linear["roofline"]["sphere_tests_per_path"] * out["value"]
                                                   / 1e12 / FP32_PEAK_TFLOPS)
        if world == 1 and args.kernel == "mega" and not args.no_live_pmc:
            # hardware counters of THIS build on THIS box, collected now by rocprofv3 around a child of this script
            torch.cuda.synchronize(dev)
            pmc = live_pmc(args, workload)
            roof["pmc"] = pmc
            if "error" not in pmc:
                roof["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, run by this bench.py in this run"
                if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
                    # KB per render; a 128-byte read request is tallied as 64 bytes on gfx950, 16-byte stores exactly
                    roof["traffic"] = (pmc["WRITE_SIZE"] + 2.0 * pmc["FETCH_SIZE"]) * 1024.0
                    roof["algorithmic_bytes_per_launch"] = 3 * nx * ny + 16 * len(sc)
                if pmc.get("SQ_INSTS_VALU") and pmc.get("GRBM_GUI_ACTIVE"):
                    valu = pmc["SQ_INSTS_VALU"]
                    trips = serial_counters["wave_iterations"] / serial_steps
                    # a wave64 vector instruction occupies its SIMD's issue port for 4 cycles
                    roof["issue"] = {"valu_wave_instructions_per_render": valu,
                                     "valu_wave_instructions_per_trip": valu / max(1.0, trips),
                                     "frac_of_issue_slots": valu * 4.0 / (SIMDS / 8 * pmc["GRBM_GUI_ACTIVE"]),
                                     "thread_efficiency": (pmc.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * pmc["SQ_ACTIVE_INST_VALU"])
                                                           if pmc.get("SQ_ACTIVE_INST_VALU") else None),
                                     "salu_per_valu": pmc.get("SQ_INSTS_SALU", 0.0) / valu,
                                     "gpu_cycles_per_render_under_profiler": pmc["GRBM_GUI_ACTIVE"] / 8.0,
                                     "clock_ghz_if_those_cycles_took_the_timed_kernel_ms": pmc["GRBM_GUI_ACTIVE"] / 8.0 / (k_ms * 1e-3) / 1e9,
                                     "definition": "VALU wave-instructions x 4 cycles / (1024 SIMDs x GPU-active cycles of the launch), both from "
                                                   "the same rocprofv3 pass (GRBM_GUI_ACTIVE is summed over the 8 XCDs)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(nx, ny, scene_name, half_extent, max_depth)
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    r.close()


if __name__ == "__main__":
    main()
