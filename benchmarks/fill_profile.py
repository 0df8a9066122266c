"""Thread efficiency by part of the trace kernel's trip: lanes at work / (64 x times the part ran), from the counting
build _out/exp/lib_exp39.so (benchmarks/build_exp.sh 39).  Run with RT_AMD_DEVICE_LIB pointing at that library:
    RT_AMD_DEVICE_LIB=_out/exp/lib_exp39.so python3 benchmarks/fill_profile.py [--workload final1080|glass1080|dense1080]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

PARTS = ["pair-filter passes (sphere filter for listed (lane, group) pairs)", "exact evaluation passes (drain)",
         "always-scanned groups, filtered per lane", "box tests (every lane tests its ray)", "shading: hit point, normal, material fetch",
         "shading: dielectric", "shading: sky", "new path: pixel, key, sample position", "unit-ball rejection loop (per loop trip)",
         "camera ray", "matte scatter", "metal scatter"]
ORDER = [0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 22]   # counter pair of each part, in the order of PARTS

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="final1080", choices=["final1080", "glass1080", "dense1080"])
args = ap.parse_args()
assert "exp39" in os.environ.get("RT_AMD_DEVICE_LIB", ""), "set RT_AMD_DEVICE_LIB to the counting build"
sc, spp = {"final1080": (rt.random_scene(11), 64), "glass1080": (rt.glass_scene(11), 64), "dense1080": (rt.random_scene(100), 16)}[args.workload]
r = rt.Renderer(0)
r.upload(sc)
cam = rt.camera_default(1920, 1080)
job = rt.Job(1920, 1080, spp, seed=0)


def words():
    log = np.zeros(30, np.uint64)
    rt.N.check(rt.N.device.rt_debug_wave_log(r._h, C.c_void_p(log.ctypes.data), len(log)), "wave_log")
    return log.astype(np.float64)


r.render_host(cam, job, debug=False)
before = words()
r.counters(reset=True)
r.render_host(cam, job, debug=False)
d = words() - before
c = r.counters()
out = {"workload": args.workload, "spp": spp, "wave_trips": c["wave_iterations"], "parts": []}
for name, k in zip(PARTS, ORDER):
    lanes, runs = d[k], d[k + 1]
    out["parts"].append({"part": name, "runs_per_trip": runs / c["wave_iterations"], "lanes_per_run": lanes / max(1.0, runs),
                         "thread_efficiency": lanes / max(1.0, 64.0 * runs)})
out["batch_visits_per_trip"] = d[24] / c["wave_iterations"]
out["batches_picked_per_trip"] = d[25] / c["wave_iterations"]
out["lane_batch_pairs_wanted_per_trip"] = d[26] / c["wave_iterations"]
out["batches_wanted_by_some_lane_per_trip"] = d[27] / c["wave_iterations"]
out["candidates_per_trip"] = c["candidates"] / c["wave_iterations"]
out["candidates_with_a_root_in_range_per_trip"] = d[28] / c["wave_iterations"]
out["candidates_at_scan_position_0_per_trip"] = d[29] / c["wave_iterations"]
print(json.dumps(out, indent=1))
