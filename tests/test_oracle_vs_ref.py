"""Function-by-function check of the oracle against the compiled reference
(oracle/_ref/libref_hooked.so: the reference's own static functions, exported).
Skipped where that library has not been built.  CPU only."""
import ctypes as C

import numpy as np
import pytest

from conftest import random_rays


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def test_struct_sizes(hooked_ref):
    assert hooked_ref.lib.ref_sizeof_camera() == 88
    assert hooked_ref.lib.ref_sizeof_sphere() == 36


def test_camera_new_random_views(oracle, hooked_ref):
    rng = np.random.default_rng(0)
    for _ in range(200):
        frm, at = rng.uniform(-20, 20, 3), rng.uniform(-5, 5, 3)
        args = (frm, at, (0, 1, 0), float(rng.uniform(5, 90)), float(rng.uniform(0.5, 3)), float(rng.uniform(0, 0.5)),
                float(rng.uniform(1, 30)))
        assert np.array_equal(oracle.camera_make(*args).view(np.uint32), hooked_ref.camera_new(*args).view(np.uint32))


def test_hit_sphere_bitwise(oracle, hooked_ref):
    rng = np.random.default_rng(1)
    sc = oracle.scene_random()
    rays = random_rays(rng, 4000, sc)
    tmin, tmax = np.float32(0.001), np.finfo(np.float32).max
    hits = 0
    for i, r in enumerate(rays):
        s = sc[(i * 7) % len(sc): (i * 7) % len(sc) + 1]
        if i % 2:  # aim at the sphere, off-centre by up to 1.2 radii, so that many rays hit or graze
            r = r.copy()
            r[3:] = (s["center"][0] + rng.normal(size=3) * 0.7 * s["radius"][0] - r[:3]) * rng.choice([0.1, 1, 5])
        a = oracle.hit_sphere(s, r[:3], r[3:], tmin, tmax)
        b = hooked_ref.hit_sphere(s, r[:3], r[3:], tmin, tmax)
        assert a[0] == b[0]
        if a[0]:
            hits += 1
            assert a[1].tobytes() == b[1].tobytes() and a[2].tobytes() == b[2].tobytes() and a[3].tobytes() == b[3].tobytes()
    assert hits > 500


def test_schlick_and_refract_bitwise(oracle, hooked_ref):
    rng = np.random.default_rng(2)
    for _ in range(3000):
        c, idx = np.float32(rng.uniform(0, 1)), np.float32(rng.choice([1.5, 1.33, 2.4]))
        assert np.float32(oracle.lib.orc_schlick(c, idx)).tobytes() == np.float32(hooked_ref.lib.ref_schlick(c, idx)).tobytes()
        u = rng.normal(size=3) * rng.choice([0.1, 1, 10])
        n = rng.normal(size=3)
        n /= np.linalg.norm(n)
        k = float(rng.choice([1.5, 1 / 1.5]))
        oa, ob = (C.c_float * 3)(), (C.c_float * 3)()
        ra = oracle.lib.orc_refract(f3(u), f3(n), k, oa)
        rb = hooked_ref.lib.ref_refract(f3(u), f3(n), k, ob)
        assert ra == rb and (not ra or bytes(oa) == bytes(ob))


@pytest.mark.parametrize("scene", ["small", "final", "glass"])
def test_deterministic_render_bitwise(oracle, hooked_ref, scene):
    sc = {"small": oracle.scene_small, "final": oracle.scene_random, "glass": oracle.scene_glass}[scene]()
    nx, ny, spp = 72, 40, 3
    cam = oracle.camera_default(nx, ny)
    a = oracle.render_det(cam, sc, nx, ny, spp, seed=21)
    b = hooked_ref.det_render(cam, sc, nx, ny, spp, seed=21)
    assert np.array_equal(a["rgb8"], b["rgb8"])
    assert np.array_equal(a["linear"].view(np.uint32), b["linear"].view(np.uint32))
    assert np.array_equal(a["first_hit"], b["first_hit"]) and np.array_equal(a["hit_sum"], b["hit_sum"])


def test_sequential_one_band_any_size(oracle, hooked_ref):
    """render() itself (c/main.c:446-483) on a size the CLI cannot produce."""
    nx, ny, spp = 50, 30, 2
    cam = oracle.camera_default(nx, ny)
    sc = oracle.scene_random()
    img, _ = oracle.render_sequential(cam, sc, nx, ny, spp, 1)
    assert np.array_equal(img, hooked_ref.render_one_band(cam, sc, nx, ny, spp))
