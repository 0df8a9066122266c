"""Where the mega kernel's wave cycles go (RT_FLAG_PROFILE): scan, drain,
shading, and how many lanes carry a live path per bounce iteration."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="random_scene")
ap.add_argument("--half-extent", type=int, default=11)
ap.add_argument("--nx", type=int, default=1920)
ap.add_argument("--ny", type=int, default=1080)
ap.add_argument("--spp", type=int, default=16)
args = ap.parse_args()
sc = rt.random_scene(args.half_extent) if args.scene == "random_scene" else getattr(rt, args.scene)()
r = rt.Renderer(0)
r.upload(sc)
cam = rt.camera_default(args.nx, args.ny)
import time  # noqa: E402
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:   # bring the clocks up: the first frames after idle run 5-15 % slower
    r.render_host(cam, rt.Job(args.nx, args.ny, args.spp, seed=0), debug=False)
ap_flags = [(0, "culled"), (1, "culled+profile"), (2, "full"), (3, "full+profile")]
for flags, label in ap_flags:
    job = rt.Job(args.nx, args.ny, args.spp, seed=0, flags=flags)
    for _ in range(3):
        r.render_host(cam, job, debug=False)
    r.counters(reset=True)
    r.render_host(cam, job, debug=False)
    ms = r.last_kernel_ms()
    c = r.counters(reset=True)
    out = {"mode": label, "flags": flags, "kernel_ms": ms, "paths_per_s": c["paths"] / (ms * 1e-3), **c}
    if flags & 1:
        tot = c["cycles_total"]
        out.update(frac_scan=c["cycles_scan"] / tot, frac_drain=c["cycles_drain"] / tot, frac_shade=c["cycles_shade"] / tot,
                   frac_refill_and_loop=1.0 - (c["cycles_scan"] + c["cycles_drain"] + c["cycles_shade"]) / tot,
                   shader_clock_ghz=c['cycles_total'] / max(1, c['realtime_total']) * 0.1,
                   own_groups_per_segment=c['own_groups'] / c['segments'], wave_groups_per_segment=c['wave_groups'] / c['segments'],
                   lane_occupancy=c["live_lane_iterations"] / (64.0 * c["wave_iterations"]),
                   scan_cycles_per_wave_iteration=c["cycles_scan"] / c["wave_iterations"],
                   drain_cycles_per_wave_iteration=c["cycles_drain"] / c["wave_iterations"],
                   shade_cycles_per_wave_iteration=c["cycles_shade"] / c["wave_iterations"])
    print(json.dumps(out))
