"""Regenerates tests/golden/ from the compiled REFERENCE (oracle/_ref, built by
oracle/Makefile from /root/reference/c where that tree exists).  The fixtures
are data only -- hashes, numbers and small arrays; no reference source.

  reference_kat.json   sha256 of the reference CLI's PPM output for several
                       (-n, -t) pairs, small_scene and revived random_scene;
                       first values of the generator; the default camera;
                       probe pixels.
  final_scene.npy      the 488 spheres random_scene() builds (structured array)
  small_scene.npy      the 27 spheres small_scene() builds
  det_<scene>_WxH_sN_seedK.npz
                       deterministic-mode renders through the reference's own
                       camera_ray_at_xy / ray_color / hit_scene driven by the
                       counter stream: rgb8, linear, first_hit, hit_sum

Run:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def ppm_pixels(blob):
    assert blob[:3] == b"P6\n"
    parts = blob.split(b"\n", 3)
    w, h = [int(x) for x in parts[1].split()[:2]]
    return np.frombuffer(blob[len(blob) - 3 * w * h:], np.uint8).reshape(h, w, 3)


def main():
    O.build()
    assert O.have_reference(), "needs /root/reference to build oracle/_ref"
    ref = O.HookedReference()
    kat = {"source": "compiled from /root/reference/c with gcc -O2 -std=gnu11 -ffp-contract=off -fno-strict-aliasing -fwrapv"}

    kat["f_randi_chain"] = []
    state = 123456789
    for _ in range(8):
        r = ref.lib.ref_f_randi(state)
        kat["f_randi_chain"].append(int(r))
        state = (r + 1) & 0xFFFFFFFF
    kat["my_rand_first8_hex"] = [float(x).hex() for x in ref.seq_stream(8)]

    focus = float(np.sqrt(np.float32(14.0 * 14.0 + 2.5 * 2.5 + 7.0 * 7.0)))
    cam = ref.camera_new((10, 2.5, 5), (-4, 0, -2), (0, 1, 0), 20.0, 2.0, np.float32(0.05), np.float32(focus))
    kat["camera_600x300_hex"] = [float(x).hex() for x in cam]
    cam1080 = ref.camera_new((10, 2.5, 5), (-4, 0, -2), (0, 1, 0), 20.0, np.float32(1920) / np.float32(1080),
                             np.float32(0.05), np.float32(focus))
    kat["camera_1920x1080_hex"] = [float(x).hex() for x in cam1080]

    small = ref.scene(0)
    final = ref.scene(1)
    np.save(os.path.join(HERE, "small_scene.npy"), small)
    np.save(os.path.join(HERE, "final_scene.npy"), final)
    kat["final_scene_counts"] = [int(x) for x in np.bincount(final["kind"], minlength=3)]

    probes = [(0, 0), (299, 150), (300, 150), (599, 299), (150, 200), (450, 200)]
    kat["ppm_sha256"] = {}
    kat["probe_pixels"] = {}
    runs = [("small", O.REF_BIN, n, t) for (n, t) in ((1, 1), (10, 1), (10, 2), (10, 4), (10, 8), (3, 7), (100, 1))]
    runs += [("final", O.REF_BIN_FINAL, n, t) for (n, t) in ((10, 1), (1, 8), (2, 3))]
    for scene, binary, n, t in runs:
        blob = subprocess.check_output([binary, "-n", str(n), "-t", str(t), "-o", "-"])
        key = f"{scene}_n{n}_t{t}"
        kat["ppm_sha256"][key] = hashlib.sha256(blob).hexdigest()
        if (n, t) == (10, 1):
            px = ppm_pixels(blob)
            kat["probe_pixels"][key] = {f"{x},{y}": [int(v) for v in px[y, x]] for (x, y) in probes}
        print(key, kat["ppm_sha256"][key])

    # deterministic mode through the reference's own functions
    glass = final.copy()
    for i in range(1, len(glass)):
        if glass["kind"][i] == 0:
            glass["kind"][i] = 2
            glass["param"][i] = 1.5
            glass["albedo"][i] = 0
        elif glass["kind"][i] == 1:
            glass["param"][i] = 0
    for name, sc, nx, ny, spp, seed in (("small", small, 96, 54, 4, 1), ("final", final, 96, 54, 4, 1),
                                        ("final", final, 64, 36, 16, 9), ("glass", glass, 64, 36, 8, 5)):
        aspect = np.float32(nx) / np.float32(ny)
        c = ref.camera_new((10, 2.5, 5), (-4, 0, -2), (0, 1, 0), 20.0, aspect, np.float32(0.05), np.float32(focus))
        out = ref.det_render(c, sc, nx, ny, spp, seed=seed)
        fn = f"det_{name}_{nx}x{ny}_s{spp}_seed{seed}.npz"
        np.savez_compressed(os.path.join(HERE, fn), camera=c, **out)
        print(fn, hashlib.sha256(out["rgb8"].tobytes()).hexdigest()[:16])

    with open(os.path.join(HERE, "reference_kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
