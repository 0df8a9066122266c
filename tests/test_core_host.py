"""The kernels' per-lane arithmetic (raytrace_amd/csrc/rt_core.h) compiled for
the host by the test harness and checked against the oracle on the CPU: the
filter never drops a sphere the reference would hit, the filtered scan gives
the reference's closest hit, and the kernel's per-pixel walk is bit-identical
to the oracle.  CPU only; the product never runs this path."""
import ctypes as C

import numpy as np
import pytest

from conftest import adversarial_rays, as_oracle_scene, leaving_rays, random_rays


def vp(a):
    return C.c_void_p(a.ctypes.data)


def test_generator_matches_oracle(core, oracle):
    out = np.zeros(4096, np.float64)
    for seed, pixel, sample in ((0, 0, 0), (7, 123456, 63), (0xFFFFFFFF, 8294399, 1023)):
        core.core_uniforms(seed, pixel, sample, C.c_size_t(len(out)), vp(out))
        key = oracle.lib.orc_path_key(seed, pixel, sample)
        want = np.array([oracle.lib.orc_ctr_uniform(key, k) for k in range(len(out))])
        assert np.array_equal(out.view(np.uint64), want.view(np.uint64))


@pytest.mark.parametrize("builder", ["small_scene", "random_scene", "dense"])
def test_filter_has_no_false_negatives(core, rt, builder):
    sc = rt.random_scene(100)[::9][:3000] if builder == "dense" else getattr(rt, builder)()
    rng = np.random.default_rng(3)
    rays = np.concatenate([random_rays(rng, 3000, sc), adversarial_rays(rng, sc, 6000)])
    passed, pos = C.c_uint64(), C.c_uint64()
    core.core_filter_check.restype = C.c_uint64
    bad = core.core_filter_check(vp(sc), C.c_size_t(len(sc)), vp(rays), C.c_size_t(len(rays)), C.byref(passed), C.byref(pos))
    assert bad == 0
    assert pos.value > 5000                       # the reference discriminant was positive often ...
    assert passed.value < 1.35 * pos.value + 3 * len(rays)   # ... and the filter is not much looser


def many_large_spheres(rt):
    """The final scene with 45 of its small spheres blown up: more large spheres than the kernels list untested."""
    sc = rt.random_scene().copy()
    sc["radius"][10:460:10] *= 6.0
    return sc


@pytest.mark.parametrize("builder", ["small_scene", "random_scene", "glass_scene", "dense", "many_large"])
def test_group_culling_never_hides_a_hit(core, rt, builder):
    sc = (rt.random_scene(100)[::7][:4000] if builder == "dense" else many_large_spheres(rt) if builder == "many_large"
          else getattr(rt, builder)())
    n_scan, n_groups = C.c_uint32(), C.c_uint32()
    assert core.core_layout_check(vp(sc), C.c_size_t(len(sc)), C.byref(n_scan), C.byref(n_groups)) == 0
    assert n_scan.value >= len(sc) and n_scan.value < len(sc) + 64 + len(sc) // 4
    rng = np.random.default_rng(8)
    rays = np.concatenate([random_rays(rng, 20000, sc), adversarial_rays(rng, sc, 20000), leaving_rays(rng, sc, 60000)])
    rays[::5, :3] *= 2.5          # origins farther out: the boxes grow with the distance from the world origin
    rays[3::50, :3] *= 40.0
    culled, groups = C.c_uint64(), C.c_uint64()
    core.core_cull_check.restype = C.c_uint64
    bad = core.core_cull_check(vp(sc), C.c_size_t(len(sc)), vp(rays), C.c_size_t(len(rays)), C.byref(culled), C.byref(groups))
    assert bad == 0
    if builder != "small_scene":
        assert culled.value > 0.5 * groups.value          # and a single ray skips most groups


def test_packed_filter_equals_scalar_filter(core, rt):
    sc = rt.random_scene()
    rays = random_rays(np.random.default_rng(5), 2000, sc)
    core.core_filter_pair_mismatches.restype = C.c_uint64
    assert core.core_filter_pair_mismatches(vp(sc), C.c_size_t(len(sc)), vp(rays), C.c_size_t(len(rays))) == 0


def test_filtered_scan_equals_reference_scan(core, rt, oracle):
    rng = np.random.default_rng(4)
    for sc in (rt.small_scene(), rt.random_scene(), rt.glass_scene()):
        rays = np.concatenate([random_rays(rng, 1500, sc), adversarial_rays(rng, sc, 1500)])
        idx = np.zeros(len(rays), np.int32)
        t = np.zeros(len(rays), np.float32)
        core.core_closest_hit(vp(sc), C.c_size_t(len(sc)), vp(rays), C.c_size_t(len(rays)), C.c_float(0.001),
                              C.c_float(np.finfo(np.float32).max), vp(idx), vp(t))
        osc = as_oracle_scene(sc)
        for i, r in enumerate(rays):
            oi, ot, _, _ = oracle.hit_scene(osc, r[:3], r[3:])
            assert oi == idx[i]
            if oi >= 0:
                assert ot.tobytes() == t[i].tobytes()


@pytest.mark.parametrize("scene,nx,ny,spp", [("small_scene", 96, 54, 4), ("random_scene", 96, 54, 4), ("glass_scene", 64, 36, 8)])
def test_kernel_walk_is_bit_identical_to_oracle(core, rt, oracle, scene, nx, ny, spp):
    sc = getattr(rt, scene)()
    cam = rt.camera_default(nx, ny)
    rgb = np.zeros((ny, nx, 3), np.uint8)
    lin = np.zeros((ny, nx, 3), np.float32)
    fh = np.zeros((ny, nx), np.int32)
    hs = np.zeros((ny, nx), np.uint32)
    st = np.zeros(3, np.uint64)
    core.core_render(C.byref(cam), vp(sc), C.c_size_t(len(sc)), nx, ny, 0, ny, spp, 50, C.c_float(np.float32(0.001)), 5,
                     vp(rgb), vp(lin), vp(fh), vp(hs), vp(st))
    o = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, spp, seed=5)
    assert np.array_equal(rgb, o["rgb8"])
    assert np.array_equal(lin.view(np.uint32), o["linear"].view(np.uint32))
    assert np.array_equal(fh, o["first_hit"]) and np.array_equal(hs, o["hit_sum"])
    assert st[0] == nx * ny * spp and st[1] <= o["stats"]["segments"]   # absorbed paths end early here


def test_reciprocal_division_is_correctly_rounded(core):
    """The drain divides by a per-ray reciprocal (three operations) where c/main.c:220, 231 divide;
    the double quotient must be the same for every operand, the near-boundary ones included."""
    core.core_division_mismatches.restype = C.c_uint64
    assert core.core_division_mismatches(C.c_uint64(20_000_000), C.c_uint64(7)) == 0


def test_float_reciprocal_equals_the_narrowed_double_reciprocal(core):
    """normalize and the dielectric's 1/ref_idx use a float divide where c/vec.c:44 and c/main.c:193 narrow a
    double reciprocal; the two agree for every float (all 2^31 magnitudes; the sign is symmetric)."""
    core.core_reciprocal_mismatches.restype = C.c_uint64
    assert core.core_reciprocal_mismatches(C.c_uint32(0), C.c_uint32(0x7F800000)) == 0


def test_draw_conversion_is_the_division_for_every_draw(core):
    """r / (2^31 - 1) as the kernels compute it (one fused multiply-add for the double, one product for the
    narrowed float) against the division of c/rand.c:15, for all 2^31 values of r."""
    core.core_unit_mismatches.restype = C.c_uint64
    assert core.core_unit_mismatches(C.c_uint32(0), C.c_uint32(0x7FFFFFFF)) == 0


def test_division_by_a_reused_float_divisor(core):
    core.core_float_division_mismatches.restype = C.c_uint64
    assert core.core_float_division_mismatches(C.c_uint64(20_000_000), C.c_uint64(3)) == 0


def test_a_23_bit_draw_narrows_to_float_without_double_arithmetic(core):
    """Draws carry 23 random bits (r = m << 8): the reference's (float)(r / (2^31 - 1)) is then m * 2^-23 and its
    2u - 1 is the kernels' one fused operation on the float 1 + m * 2^-23, for all 2^23 values of m."""
    core.core_draw23_mismatches.restype = C.c_uint64
    assert core.core_draw23_mismatches() == 0
