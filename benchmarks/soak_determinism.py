"""Renders the same 1080p x 64 spp frame many times back to back on one context (pipelined, no synchronisation in
between) and checks that every copy has the same bytes: a race between overlapped launches would show as a copy
that differs.  python3 benchmarks/soak_determinism.py [copies] [half extent of the scene: 11 = the final scene,
100 = the 40,004-sphere field scanned in place] [samples per pixel]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytrace_amd as rt  # noqa: E402

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
r = rt.Renderer(0)
half = int(sys.argv[2]) if len(sys.argv) > 2 else 11
r.upload(rt.random_scene(half))
stream = torch.cuda.Stream(dev)
r.set_stream(stream.cuda_stream)
nx, ny, spp = 1920, 1080, int(sys.argv[3]) if len(sys.argv) > 3 else 64
cam = rt.camera_default(nx, ny)
job = rt.Job(nx, ny, spp, seed=0)
ring = [torch.zeros((ny, nx, 3), dtype=torch.uint8, device=dev) for _ in range(8)]
sums = []
with torch.cuda.stream(stream):
    for k in range(copies):
        out = ring[k % len(ring)]
        r.render_device(cam, job, out.data_ptr())
        # a position-weighted checksum, queued behind the render on the same stream
        sums.append((out.view(-1).to(torch.int64)[:: 7] * torch.arange(0, (ny * nx * 3 + 6) // 7, device=dev)).sum())
stream.synchronize()
vals = torch.stack(sums).cpu().tolist()
bad = [k for k, v in enumerate(vals) if v != vals[0]]
print(f"{copies} renders of random_scene({half}) at {spp} spp, checksum {vals[0]}, differing copies: {bad[:10]} ({len(bad)})")
sys.exit(1 if bad else 0)
