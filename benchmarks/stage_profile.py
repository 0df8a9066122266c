"""Share of the trace kernel's wave cycles spent in one stage, for every experiment library _out/exp/lib_exp<N>.so
built with a stage timer (RT_EXP 31..38): each runs in a process of its own with RT_AMD_DEVICE_LIB set."""
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = {31: "rejection loop", 32: "shading before the rejection loop", 33: "new paths", 34: "shading after the rejection loop",
          35: "ray set-up", 36: "quarter and super-group boxes", 37: "group-box rounds and list appends", 38: "sphere filter passes",
          40: "choosing the batches of a scene scanned in place (region and batch boxes, near-to-far picks)"}
WORKLOAD = os.environ.get("RT_STAGE_WORKLOAD", "final1080")      # final1080 | glass1080 | dense1080

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import raytrace_amd as rt
    r = rt.Renderer(0)
    r.upload({"final1080": lambda: rt.random_scene(11), "glass1080": lambda: rt.glass_scene(11), "dense1080": lambda: rt.random_scene(100)}[WORKLOAD]())
    cam = rt.camera_default(1920, 1080)
    job = rt.Job(1920, 1080, 16 if WORKLOAD != "dense1080" else 8, seed=0, flags=rt.FLAG_PROFILE)
    r.render_host(cam, job, debug=False)
    r.counters(reset=True)
    r.render_host(cam, job, debug=False)
    c = r.counters()
    print(json.dumps({"stage_frac": c["cycles_drain"] / c["cycles_total"], "stage_cycles_per_trip": c["cycles_drain"] / c["wave_iterations"],
                      "total_cycles_per_trip": c["cycles_total"] / c["wave_iterations"], "kernel_ms": r.last_kernel_ms()}))
    sys.exit(0)

for lib in sorted(glob.glob(os.path.join(ROOT, "_out", "exp", "lib_exp*.so"))):
    n = int(re.search(r"lib_exp(\d+)\.so", lib).group(1))
    if n not in STAGES:
        continue
    env = dict(os.environ, RT_AMD_DEVICE_LIB=lib)
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, timeout=300)
    if p.returncode != 0:
        print(json.dumps({"exp": n, "error": p.stderr.decode()[-300:]}))
        continue
    out = json.loads(p.stdout.decode().strip().splitlines()[-1])
    print(json.dumps({"exp": n, "workload": WORKLOAD, "stage": STAGES[n], **out}), flush=True)
