"""Shared fixtures.  `gpu`-marked tests run on an MI355X box only; everything
else runs on the CPU (oracle against the goldens, host code, ABI surface)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CORE_SRC = os.path.join(ROOT, "tests", "support", "core_host.cpp")
CORE_SO = os.path.join(ROOT, "tests", "support", "_out", "libcore_host.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    return O.Oracle()


@pytest.fixture(scope="session")
def hooked_ref():
    from oracle import oracle as O
    if not os.path.exists(O.REF_SO):
        pytest.skip("oracle/_ref/libref_hooked.so not built (needs /root/reference)")
    return O.HookedReference()


@pytest.fixture(scope="session")
def kat():
    with open(os.path.join(GOLDEN, "reference_kat.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def rt():
    import raytrace_amd
    return raytrace_amd


def build_core_host():
    deps = [CORE_SRC] + [os.path.join(ROOT, "raytrace_amd", "csrc", h) for h in ("rt_core.h", "rt_cluster.h")]
    if not os.path.exists(CORE_SO) or any(os.path.getmtime(d) > os.path.getmtime(CORE_SO) for d in deps):
        os.makedirs(os.path.dirname(CORE_SO), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall",
                               "-o", CORE_SO, CORE_SRC, "-lm"])
    return CORE_SO


@pytest.fixture(scope="session")
def core():
    """rt_core.h (the kernels' per-lane arithmetic) compiled for the host: test harness only."""
    return C.CDLL(build_core_host())


@pytest.fixture(scope="session")
def renderer(rt):
    r = rt.Renderer(0)
    yield r
    r.close()


def load_golden_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_scene(name):
    """Spheres saved by make_golden.py -> the product's dtype (same layout, field 'kind' -> 'type')."""
    import raytrace_amd as rt
    a = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return a.view(rt.SPHERE_DTYPE) if a.dtype.itemsize == 36 else a


def as_oracle_scene(s):
    from oracle import oracle as O
    return np.ascontiguousarray(s).view(O.SPHERE_DTYPE)


def random_rays(rng, n, spheres=None):
    """Rays typical of the renderer: origins near the camera or on sphere
    surfaces, directions of every length scale the bounce loop produces."""
    A = rng.uniform(-12, 12, (n, 3)).astype(np.float32)
    A[:, 1] = rng.uniform(0, 3, n).astype(np.float32)
    B = rng.normal(size=(n, 3)).astype(np.float32)
    B *= rng.choice([0.05, 1.0, 1.0, 15.0], size=(n, 1)).astype(np.float32)
    if spheres is not None and len(spheres):
        k = n // 2
        idx = rng.integers(0, len(spheres), k)
        d = rng.normal(size=(k, 3))
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        A[:k] = (spheres["center"][idx] + spheres["radius"][idx, None] * d).astype(np.float32)
    return np.concatenate([A, B], axis=1).astype(np.float32)


def adversarial_rays(rng, spheres, n):
    """Tangent and near-tangent rays, rays leaving a surface, huge and tiny directions."""
    idx = rng.integers(0, len(spheres), n)
    c, r = spheres["center"][idx].astype(np.float64), spheres["radius"][idx].astype(np.float64)
    A = rng.uniform(-15, 15, (n, 3))
    A[:, 1] = rng.uniform(0, 4, n)
    to_c = c - A
    dist = np.linalg.norm(to_c, axis=1)
    w = to_c / dist[:, None]
    t1 = np.cross(w, rng.normal(size=(n, 3)))
    t1 /= np.linalg.norm(t1, axis=1, keepdims=True)
    # aim at a point displaced r*(1+eps) sideways from the centre: grazing by construction
    eps = rng.choice([0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3], n)
    target = c + t1 * (r * (1 + eps))[:, None]
    B = (target - A) * rng.choice([0.01, 1, 1, 40], n)[:, None]
    return np.concatenate([A, B], axis=1).astype(np.float32)


def leaving_rays(rng, spheres, n):
    """Rays that start on, just inside or just outside a sphere and head away from it, along it or barely
    towards it, with directions from 1e-6 to 1e3 long and some components exactly zero: the cases in which
    'the group is behind the ray' is a close call."""
    idx = rng.integers(0, len(spheres), n)
    c, r = spheres["center"][idx].astype(np.float64), spheres["radius"][idx].astype(np.float64)
    nrm = rng.normal(size=(n, 3))
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    off = rng.choice([0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-4, -1e-4, 1e-2], n)
    A = c + nrm * (r * (1 + off))[:, None]
    tang = np.cross(nrm, rng.normal(size=(n, 3)))
    tang /= np.linalg.norm(tang, axis=1, keepdims=True)
    lean = rng.choice([0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3, -1e-3, 0.3, -0.3, 1.0], n)   # > 0 leaves the sphere
    B = tang * rng.uniform(0, 1, n)[:, None] + nrm * lean[:, None]
    B *= rng.choice([1e-6, 1e-4, 1e-2, 1, 1, 1e3], n)[:, None]
    B[rng.random(n) < 0.05, rng.integers(0, 3)] = 0.0
    return np.concatenate([A, B], axis=1).astype(np.float32)
