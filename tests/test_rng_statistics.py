"""Sample quality of the counter generator (raytrace_amd/csrc/rt_core.h: draw_hash on key + k * 0x9e3779b9, its top 23
bits the draw).  The parity tests only prove that the GPU and the oracle share one stream; these keep a later change to
the hash from degrading it: per-bit balance, bucket chi-square of single draws, chi-square of consecutive pairs and
triples (what a unit-ball point is made of, c/vec.c:64-72), the acceptance rate of the rejection loop against pi/6 and
the moments of the accepted points, on sequential draws of several paths and on the first draw across neighbouring
paths.  The full-avalanche mix32 passes the same tests (control) and a one-multiply hash fails them (so they have power).
The numpy statement of the hash is pinned to the compiled one first.  CPU only."""
import ctypes as C

import numpy as np
import pytest

PHI = np.uint32(0x9e3779b9)


def _u32(x):
    return np.asarray(x, dtype=np.uint32)


def mix32(x):
    x = _u32(x).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15)
    x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def draw_hash(x):
    x = _u32(x).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x846ca68b)
    return x


def one_multiply(x):
    x = _u32(x).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x7feb352d)
    return x


def path_key(seed, pixel, sample):
    a = mix32(_u32(pixel) + _u32(np.uint64(0x9e3779b9) * np.uint64(seed) & np.uint64(0xffffffff)))
    return mix32(a ^ _u32((np.asarray(sample, np.uint64) * np.uint64(0x85ebca6b) + np.uint64(0x165667b1)) & np.uint64(0xffffffff)))


def z_chi2(counts, expected):
    chi = ((counts - expected) ** 2 / expected).sum()
    df = counts.size - 1
    return float((chi - df) / np.sqrt(2.0 * df))


def statistics(h, n_keys=64, n_draws=3 * (1 << 14)):
    keys = path_key(7, np.arange(n_keys, dtype=np.uint32) + np.uint32(1000), 3)
    k = np.arange(n_draws, dtype=np.uint32)
    m = (h(keys[:, None] + k[None, :] * PHI) >> np.uint32(9)).astype(np.int64)        # the 23 bits of every draw
    n = m.size
    out = {}
    ones = ((m[..., None] >> np.arange(23)) & 1).reshape(-1, 23).mean(0)
    out["bit"] = float(np.abs(ones - 0.5).max() / (0.5 / np.sqrt(n)))
    out["bucket"] = z_chi2(np.bincount((m >> 13).ravel(), minlength=1024).astype(float), n / 1024)
    t = m.reshape(n_keys, -1, 3) >> 19                                                  # 4 bits a coordinate: 4096 cells
    cell = (t[..., 0] << 8) | (t[..., 1] << 4) | t[..., 2]
    out["triple"] = z_chi2(np.bincount(cell.ravel(), minlength=4096).astype(float), cell.size / 4096)
    p = m.reshape(n_keys, -1, 2) >> 13                                                  # 10 bits a coordinate
    cell = (p[..., 0] << 10) | p[..., 1]
    out["pair"] = z_chi2(np.bincount(cell.ravel(), minlength=1 << 20).astype(float), cell.size / (1 << 20))
    u = 2.0 * (m.reshape(n_keys, -1, 3) * 2.0 ** -23) - 1.0                              # c/vec.c:68
    inside = (u * u).sum(-1) < 1.0
    p_in = np.pi / 6
    out["accept"] = float((inside.mean() - p_in) / np.sqrt(p_in * (1 - p_in) / inside.size))
    a = u[inside]
    out["mean"] = float(np.abs(a.mean(0)).max() / np.sqrt(0.2 / len(a)))                # uniform ball: E x = 0, E x^2 = 1/5,
    out["second"] = float(np.abs((a * a).mean(0) - 0.2).max() / np.sqrt((3 / 35 - 1 / 25) / len(a)))   # E x^4 = 3/35
    first = (h(path_key(0, np.arange(1 << 20, dtype=np.uint32), 0)) >> np.uint32(9)).astype(np.int64)   # draw 0 of 2^20 pixels
    out["across_paths"] = z_chi2(np.bincount(first >> 13, minlength=1024).astype(float), first.size / 1024)
    return out


def test_numpy_hash_is_the_compiled_one(core):
    out = np.zeros(512, np.float64)
    for seed, pixel, sample in ((0, 0, 0), (7, 123456, 63), (0xFFFFFFFF, 8294399, 1023)):
        core.core_uniforms(seed, pixel, sample, C.c_size_t(len(out)), C.c_void_p(out.ctypes.data))
        key = path_key(seed, pixel, sample)
        r31 = ((draw_hash(key + np.arange(len(out), dtype=np.uint32) * PHI) >> np.uint32(9)) << np.uint32(8)).astype(np.float64)
        assert np.array_equal(out, r31 / 2147483647.0)                                  # c/rand.c:15


def test_draw_hash_passes_and_mix32_agrees():
    for name, h in (("draw_hash", draw_hash), ("mix32", mix32)):
        for stat, z in statistics(h).items():
            assert abs(z) < 5.0, f"{name}: {stat} z = {z:.1f}"


def test_the_tests_have_power():
    z = statistics(one_multiply)
    assert z["triple"] > 50.0 and z["pair"] > 50.0 and z["second"] > 5.0
