"""Kernel time of one frame for whichever device library RT_AMD_DEVICE_LIB names (timing experiments)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
r = rt.Renderer(0)
r.upload(rt.random_scene(11))
cam = rt.camera_default(1920, 1080)
job = rt.Job(1920, 1080, spp, seed=0, flags=flags)
r.render_host(cam, job, debug=False)
ms = []
for _ in range(5):
    r.render_host(cam, job, debug=False)
    ms.append(r.last_kernel_ms())
print(json.dumps({"lib": os.environ.get("RT_AMD_DEVICE_LIB", "default"), "spp": spp, "flags": flags, "kernel_ms": ms}))
