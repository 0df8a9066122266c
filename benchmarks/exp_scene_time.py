"""Kernel time of one 1080p frame of random_scene(half_extent) for the device library RT_AMD_DEVICE_LIB names."""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

half = int(sys.argv[1]) if len(sys.argv) > 1 else 15
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
r = rt.Renderer(0)
sc = rt.random_scene(half)
r.upload(sc)
cam = rt.camera_default(1920, 1080)
job = rt.Job(1920, 1080, spp, seed=0)
ms = []
digest = None
for _ in range(int(os.environ.get("RT_EXP_REPS", "4"))):
    out = r.render_host(cam, job, debug=False)
    ms.append(round(r.last_kernel_ms(), 3))
    digest = hashlib.sha256(out["rgb8"].tobytes()).hexdigest()[:16]
print(json.dumps({"lib": os.environ.get("RT_AMD_DEVICE_LIB", "default"), "spheres": len(sc), "spp": spp, "kernel_ms": ms, "image": digest}))
