"""The device's hand-scheduled routines against their plain statement, value for value.

The kernels evaluate the conservative sphere filter and the group box test with packed inline
assembly, and a few divisions and the discriminant's square root by routes of their own; on the
host, the test suite checks the plain C++ statement of the same arithmetic (rt_core.h compiled by
gcc).  These tests run the device routines themselves on a million operand pairs and require the
bits of the host statement."""
import ctypes as C

import numpy as np
import pytest

from conftest import adversarial_rays, leaving_rays, random_rays

pytestmark = pytest.mark.gpu


def vp(a):
    return C.c_void_p(a.ctypes.data)


def some_rays(rt, n_each, seed):
    sc = rt.random_scene()
    rng = np.random.default_rng(seed)
    rays = np.concatenate([random_rays(rng, n_each, sc), adversarial_rays(rng, sc, n_each), leaving_rays(rng, sc, n_each)])
    rays[::7, :3] *= 3.0
    return sc, np.ascontiguousarray(rays, np.float32)


def test_device_filter_is_the_plain_discriminant(renderer, rt, core):
    """filter_four_pairs (full scan) and filter_two_pairs (culled scan, coefficients rebuilt from the ray table)
    on 1536 rays x 1024 spheres = 1.5 million pairs, grazing rays included: the bits of filter_discriminant."""
    sc, rays = some_rays(rt, 512, 41)
    big = rt.random_scene(30)
    rng = np.random.default_rng(42)
    pick = np.concatenate([np.arange(len(sc)), rng.choice(len(big), 1024 - len(sc), replace=False)])
    src = np.concatenate([sc, big[pick[len(sc):]]])
    spheres = np.concatenate([src["center"], src["radius"][:, None]], axis=1).astype(np.float32)
    four, two = renderer.debug_filter(rays, spheres)
    want = np.zeros_like(four)
    core.core_filter_values(vp(rays), C.c_size_t(len(rays)), vp(spheres), C.c_size_t(len(spheres)), vp(want))
    assert np.array_equal(four.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(two.view(np.uint32), want.view(np.uint32))
    assert (want > 0).sum() > 2000 and (want <= 0).sum() > 10 ** 6


def test_device_box_test_is_the_plain_slab_test(renderer, rt, core):
    """box_pair (boxes in vector registers) and box_pair_scalar (boxes through scalar loads) against box_pass,
    given the per-ray coefficients the device formed: the scene's own group, super-group and batch boxes
    plus boxes shrunk and grown around the rays' paths."""
    sc, rays = some_rays(rt, 700, 43)
    big = rt.random_scene(30)
    boxes = np.zeros((2000, 6), np.float32)
    s0, s1 = C.c_float(), C.c_float()
    core.core_layout_boxes.restype = C.c_size_t
    n_big = core.core_layout_boxes(vp(big), C.c_size_t(len(big)), vp(boxes), C.c_size_t(len(boxes)), C.byref(s0), C.byref(s1))
    assert 300 < n_big <= 2000
    boxes = boxes[:n_big - n_big % 2]
    vec, sca, box_ray = renderer.debug_box_pass(rays, boxes, s0.value, s1.value)
    want = np.zeros_like(vec)
    core.core_box_pass(vp(box_ray), C.c_size_t(len(rays)), vp(boxes), C.c_size_t(len(boxes)), vp(want))
    assert np.array_equal(vec, want) and np.array_equal(sca, want)
    assert len(rays) * len(boxes) > 10 ** 6 and 0.005 < want.mean() < 0.7
    # the coefficients themselves: the device's reciprocal is within one ulp of the host's
    host = np.zeros_like(box_ray)
    core.core_make_box_ray(vp(rays), C.c_size_t(len(rays)), s0, s1, vp(host))
    ok = np.isfinite(host).all(axis=1) & (np.abs(rays[:, 3:]).min(axis=1) > 1e-30)
    rel = np.abs(box_ray[ok].astype(np.float64) - host[ok]) / np.maximum(np.abs(host[ok]), 1e-300)
    assert rel.max() < 4e-7


def float_sweep(n, lo, hi, seed):
    """n positive floats with every exponent between lo's and hi's: random significands plus the awkward ones"""
    rng = np.random.default_rng(seed)
    bits = rng.integers(np.float32(lo).view(np.uint32), np.float32(hi).view(np.uint32), n, dtype=np.uint32)
    bits[::11] |= 0x7FFFFF          # significands of all ones
    bits[1::11] &= 0xFF800000       # powers of two
    bits[2::11] = (bits[2::11] & 0xFF800000) | 1
    return bits.view(np.float32)


def test_device_square_root_of_a_widened_float(renderer, rt):
    """The trimmed double square root of the exact hit test (c/main.c:222 takes sqrt of the float discriminant
    widened to double): a dense sweep of positive floats over the whole exponent range."""
    x = float_sweep(1 << 22, 1e-37, 3e38, 5).astype(np.float64)
    got = renderer.debug_arith(rt.N.ARITH_SQRT_WIDENED, x)
    assert np.array_equal(got.view(np.uint64), np.sqrt(x).view(np.uint64))


def test_device_divisions_by_reciprocal(renderer, rt):
    """x / a through the per-ray reciprocal (the roots of c/main.c:223, 231), the float quotient by a reused
    divisor (c/main.c:229, 465-466) and the narrowed double reciprocal (c/vec.c:45) against the divisions."""
    n = 1 << 22
    rng = np.random.default_rng(6)
    a = float_sweep(n, 1e-30, 1e30, 7).astype(np.float64)
    x = rng.standard_normal(n) * np.exp2(rng.integers(-60, 60, n))
    # quotients next to a rounding boundary: x = (q + half an ulp of q) * a, rounded
    q = x[::5] / a[::5]
    x[::5] = ((q + np.nextafter(q, np.inf)) / 2 * a[::5])
    got = renderer.debug_arith(rt.N.ARITH_DIV_WIDENED, x, a)
    assert np.array_equal(got.view(np.uint64), (x / a).view(np.uint64))
    xf = (rng.standard_normal(n) * np.exp2(rng.integers(-20, 20, n))).astype(np.float32)
    kf = float_sweep(n, 1e-6, 1e6, 8)
    qf = xf[::5] / kf[::5]
    mid = ((qf.astype(np.float64) + np.nextafter(qf, np.float32(np.inf)).astype(np.float64)) / 2 * kf[::5].astype(np.float64))
    xf[::5] = mid.astype(np.float32)
    got = renderer.debug_arith(rt.N.ARITH_DIV_F32_BY, xf.astype(np.float64), kf.astype(np.float64))
    assert np.array_equal(got.astype(np.float32).view(np.uint32), (xf / kf).view(np.uint32))
    r = float_sweep(n, 1e-37, 3e38, 9)
    got = renderer.debug_arith(rt.N.ARITH_RCP_NARROWED, r.astype(np.float64))
    want = (1.0 / r.astype(np.float64)).astype(np.float32)
    assert np.array_equal(got.astype(np.float32).view(np.uint32), want.view(np.uint32))


def test_device_draw_conversion(renderer, rt):
    """A draw r (31 bits) becomes r / (2^31 - 1) as a double (c/rand.c:15) and, for the unit-ball points,
    that double narrowed to float (c/vec.c:68): the device's division-free forms on 8 million draws."""
    rng = np.random.default_rng(10)
    r = rng.integers(0, 1 << 31, 1 << 23, dtype=np.uint32)
    r[:4] = (0, 1, 0x7FFFFFFF, 0x7FFFFFFE)
    rd = r.astype(np.float64)
    want = rd / 2147483647.0
    got = renderer.debug_arith(rt.N.ARITH_UNIT_DOUBLE, rd)
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    gotf = renderer.debug_arith(rt.N.ARITH_UNIT_FLOAT, rd)
    assert np.array_equal(gotf.astype(np.float32).view(np.uint32), want.astype(np.float32).view(np.uint32))
