"""The baseline configurations at their real sizes, through the C ABI, against the oracle.

The oracle cannot render a 1920x1080x64 frame in test time, but it renders row bands of one
(`render_det(row0=, rows=)`), and a job of the library is a set of row tiles: a few bands of every
configuration at its true width, height and sample count pin the index arithmetic the full-size
launches use (tile-major path ids, chunk stepping, the multiply-high row division of interleaved
tiles) and the sample summation order at the sample counts of the benchmark."""
import numpy as np
import pytest

from conftest import as_oracle_scene
from test_gpu_parity import assert_parity

pytestmark = pytest.mark.gpu


def oracle_rows(oracle, cam, osc, nx, ny, spp, seed, row_list, max_depth=50):
    """the oracle's outputs for the given image rows (runs of consecutive rows are rendered together)"""
    parts = []
    rows = list(row_list)
    i = 0
    while i < len(rows):
        j = i
        while j + 1 < len(rows) and rows[j + 1] == rows[j] + 1:
            j += 1
        parts.append(oracle.render_det(cam.as_array(), osc, nx, ny, spp, max_depth=max_depth, seed=seed, row0=rows[i],
                                       rows=j - i + 1, host_threads=16))
        i = j + 1
    return {k: np.concatenate([p[k] for p in parts]) for k in ("rgb8", "linear", "first_hit", "hit_sum")}


@pytest.mark.parametrize("flags", [0, 2])
def test_dense_field_render_equals_oracle(renderer, rt, oracle, flags):
    """Baseline config 4: the 200x200 field, 40,004 spheres, scanned in place from global memory in 79 batches of 512
    scan positions (so a wave's batch mask uses both of its 64-bit halves), rendered by the trace kernel proper:
    culled (region and batch boxes voted per wave, batches visited near to far under the distance cut) and with
    every sphere filtered for every ray."""
    sc = rt.random_scene(100)
    assert len(sc) == 40004
    renderer.upload(sc)
    nx, ny, spp = 96, 54, 2
    cam = rt.camera_default(nx, ny)
    renderer.counters(reset=True)
    got = renderer.render_host(cam, rt.Job(nx, ny, spp, seed=17, flags=flags))
    c = renderer.counters(reset=True)
    want = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, spp, seed=17, host_threads=16)
    assert_parity(got, want)
    assert c["paths"] == nx * ny * spp
    if flags == 2:
        assert c["sphere_tests"] == c["segments"] * len(sc) and c["box_tests"] == 0
    else:
        assert 0 < c["sphere_tests"] < c["segments"] * len(sc) // 20 and c["box_tests"] > 0


def test_dense_field_rows_at_full_size(renderer, rt, oracle):
    """Config 4 at 1920x1080x16: two rows of the real frame (ground rows, where rays graze thousands of spheres)."""
    sc = rt.random_scene(100)
    renderer.upload(sc)
    nx, ny, spp = 1920, 1080, 16
    cam = rt.camera_default(nx, ny)
    job = rt.Job(nx, ny, spp, seed=0, row0=700, tile_rows=1, tile_stride=300, n_tiles=2)      # rows 700 and 1000
    got = renderer.render_host(cam, job)
    want = oracle_rows(oracle, cam, as_oracle_scene(sc), nx, ny, spp, 0, [700, 1000])
    assert_parity(got, want)


def test_final_scene_bands_at_1080p_64spp(renderer, rt, oracle):
    """Config 2 (the benchmark's frame): 8-row bands of the 1920x1080x64 frame -- sky, horizon, spheres, the
    bottom edge -- and one interleaved job (three tiles 64 rows apart, the multi-GPU layout)."""
    sc = rt.random_scene()
    renderer.upload(sc)
    nx, ny, spp = 1920, 1080, 64
    cam = rt.camera_default(nx, ny)
    osc = as_oracle_scene(sc)
    for row0 in (0, 536, 1072):
        got = renderer.render_host(cam, rt.Job(nx, ny, spp, seed=0, row0=row0, tile_rows=8))
        want = oracle_rows(oracle, cam, osc, nx, ny, spp, 0, range(row0, row0 + 8))
        assert_parity(got, want)
    import raytrace_amd as R
    job = rt.Job(nx, ny, spp, seed=0, row0=616, tile_rows=8, tile_stride=64, n_tiles=3)
    rows = R.job_row_indices(job)
    assert list(rows[:9]) == list(range(616, 624)) + [680] and len(rows) == 24
    got = renderer.render_host(cam, job)
    want = oracle_rows(oracle, cam, osc, nx, ny, spp, 0, rows)
    assert_parity(got, want)
    # the last rank's job of an 8-GPU run: tiles clipped at the bottom edge of the image (1080 = 135 tiles of 8)
    job = R.interleaved_job(nx, ny, spp, 6, 8, 8, seed=0)
    rows = R.job_row_indices(job)
    tail = rows[-8:]
    part = renderer.render_host(cam, job)
    want = oracle_rows(oracle, cam, osc, nx, ny, spp, 0, tail)
    assert_parity({k: v[-8:] for k, v in part.items()}, want)


def test_glass_scene_rows_at_1080p_256spp(renderer, rt, oracle):
    """Config 5: all glass and mirrors, depth 50, 256 spp -- four rows through the spheres of the real frame."""
    sc = rt.glass_scene()
    renderer.upload(sc)
    nx, ny, spp = 1920, 1080, 256
    cam = rt.camera_default(nx, ny)
    job = rt.Job(nx, ny, spp, seed=0, row0=640, tile_rows=2, tile_stride=200, n_tiles=2)      # rows 640-641, 840-841
    got = renderer.render_host(cam, job)
    want = oracle_rows(oracle, cam, as_oracle_scene(sc), nx, ny, spp, 0, [640, 641, 840, 841])
    assert_parity(got, want)


def test_final_scene_rows_at_4k_1024spp(renderer, rt, oracle):
    """Config 3: 3840x2160 at 1024 spp -- two rows of the real frame as tiles of an interleaved job."""
    sc = rt.random_scene()
    renderer.upload(sc)
    nx, ny, spp = 3840, 2160, 1024
    cam = rt.camera_default(nx, ny)
    job = rt.Job(nx, ny, spp, seed=0, row0=1199, tile_rows=1, tile_stride=960, n_tiles=2)     # rows 1199 and 2159
    got = renderer.render_host(cam, job)
    want = oracle_rows(oracle, cam, as_oracle_scene(sc), nx, ny, spp, 0, [1199, 2159])
    assert_parity(got, want)


def _check_closest_hits(renderer, oracle, sc, rays):
    idx, t = renderer.closest_hit(rays)
    oi, ot = oracle.hit_scene_many(as_oracle_scene(sc), rays, threads=16)
    assert np.array_equal(idx, oi)
    hit = oi >= 0
    assert np.array_equal(t[hit].view(np.uint32), ot[hit].view(np.uint32))
    return hit.mean()


def test_distance_cut_grazing_rays_in_the_dense_field(renderer, rt, oracle):
    """Scenes scanned in place visit their batches near to far and let a ray's hit rule out the boxes that begin
    farther along it.  Rays that skim the 40,004-sphere field just above the ground cross many batches, and their
    first hit decides which of those are still scanned: index and t bits against the oracle's linear scan."""
    sc = rt.random_scene(100)
    renderer.upload(sc)
    rng = np.random.default_rng(77)
    n = 60000
    A = np.stack([rng.uniform(-95, 95, n), rng.uniform(0.005, 0.45, n), rng.uniform(-95, 95, n)], 1)
    ang = rng.uniform(0, 2 * np.pi, n)
    B = np.stack([np.cos(ang), rng.normal(0, 0.01, n), np.sin(ang)], 1) * rng.uniform(0.2, 5.0, (n, 1))
    # a third of them start on a sphere's surface, as scattered rays do
    k = rng.integers(1, len(sc), n // 3)
    d = rng.normal(size=(n // 3, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    A[: n // 3] = sc["center"][k] + sc["radius"][k, None] * d
    rays = np.concatenate([A, B], 1).astype(np.float32)
    frac = _check_closest_hits(renderer, oracle, sc, rays)
    assert frac > 0.8


def test_distance_cut_overlapping_spheres_many_batches(renderer, rt, oracle):
    """6,000 overlapping spheres of mixed size (scanned in place: 12 batches): origins inside spheres, far roots,
    near-equal hit distances from spheres of different batches."""
    rng = np.random.default_rng(5)
    n = 6000
    sc = np.zeros(n, rt.SPHERE_DTYPE)
    sc["center"] = np.stack([rng.uniform(-12, 12, n), rng.uniform(0, 3, n), rng.uniform(-12, 12, n)], 1)
    sc["radius"] = rng.uniform(0.05, 0.7, n)
    # pairs of spheres that meet a ray at almost the same distance: a copy moved by a few ulps
    sc["center"][n // 2:] = sc["center"][: n // 2] + rng.normal(0, 3e-6, (n // 2, 3))
    sc["radius"][n // 2:] = sc["radius"][: n // 2] * (1 + rng.normal(0, 2e-6, n // 2))
    sc["albedo"] = 0.5
    renderer.upload(sc)
    m = 40000
    A = np.stack([rng.uniform(-13, 13, m), rng.uniform(-0.5, 3.5, m), rng.uniform(-13, 13, m)], 1)
    B = rng.normal(size=(m, 3)) * rng.uniform(0.1, 3.0, (m, 1))
    rays = np.concatenate([A, B], 1).astype(np.float32)
    frac = _check_closest_hits(renderer, oracle, sc, rays)
    assert frac > 0.6


@pytest.mark.parametrize("n", [513, 1100, 2040])
def test_rays_outside_the_box_range_in_scenes_with_padded_quarters(renderer, rt, oracle, n):
    """Rays whose direction is too long for the box arithmetic (|B|^2 >= 2^40: a camera focused two million units
    away) are not culled: they list every group.  In a scene whose scan length is not a multiple of 256 the last
    quarter box is padding, and 'every' must not include it."""
    full = rt.random_scene(30)
    rng = np.random.default_rng(n)
    sc = full[np.sort(rng.choice(len(full), size=n, replace=False))].copy()
    renderer.upload(sc)
    nx, ny, spp = 64, 36, 2
    cam = rt.camera_new((10, 2.5, 5), (-4, 0, -2), (0, 1, 0), 20.0, nx / ny, 0.0, 2.0e6)
    got = renderer.render_host(cam, rt.Job(nx, ny, spp, seed=3))
    want = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, spp, seed=3, host_threads=16)
    assert np.array_equal(got["rgb8"], want["rgb8"])
    assert np.array_equal(got["linear"].view(np.uint32), want["linear"].view(np.uint32))
    assert np.array_equal(got["first_hit"], want["first_hit"])


def test_the_largest_scene_and_the_first_too_large(renderer, rt, oracle):
    """65,535 spheres (64 batches: every bit of the batch mask, scan positions up to 65,535 in the 16-bit list
    fields): closest hits and a small render against the oracle.  One sphere more, or a much larger sphere whose
    group of its own pushes the padded length past 65,536, is RT_ERR_LIMIT with a message, not a wrong picture."""
    rng = np.random.default_rng(65535)
    n = 65535
    sc = np.zeros(n, rt.SPHERE_DTYPE)
    g = np.stack(np.meshgrid(np.arange(256), np.arange(256), indexing="ij"), -1).reshape(-1, 2)[:n]
    sc["center"] = np.stack([g[:, 0] * 0.5 - 64 + rng.uniform(-0.1, 0.1, n), rng.uniform(0.1, 0.3, n), g[:, 1] * 0.5 - 64 + rng.uniform(-0.1, 0.1, n)], 1)
    sc["radius"] = rng.uniform(0.08, 0.15, n)
    sc["type"] = rng.integers(0, 3, n)
    sc["albedo"] = rng.uniform(0.2, 0.9, (n, 3))
    sc["param"] = np.where(sc["type"] == 2, 1.5, rng.uniform(0, 0.3, n))
    renderer.upload(sc)
    m = 4000
    A = np.stack([rng.uniform(-60, 60, m), rng.uniform(0.05, 3.0, m), rng.uniform(-60, 60, m)], 1)
    B = rng.normal(size=(m, 3))
    B[:, 1] = -np.abs(B[:, 1]) * 0.2
    rays = np.concatenate([A, B], 1).astype(np.float32)
    assert _check_closest_hits(renderer, oracle, sc, rays) > 0.5
    nx, ny, spp = 48, 27, 1
    cam = rt.camera_new((20, 6, 12), (0, 0, 0), (0, 1, 0), 30.0, nx / ny, 0.0, 20.0)
    got = renderer.render_host(cam, rt.Job(nx, ny, spp, seed=1))
    want = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, spp, seed=1, host_threads=16)
    assert np.array_equal(got["rgb8"], want["rgb8"]) and np.array_equal(got["first_hit"], want["first_hit"])
    too_many = np.concatenate([sc, sc[:1]])
    with pytest.raises(rt.RtError) as e:
        renderer.upload(too_many)
    assert e.value.code == 5 and "65535" in str(e.value)
    with_a_giant = sc.copy()
    with_a_giant["radius"][0] = 50.0
    with pytest.raises(rt.RtError) as e:
        renderer.upload(with_a_giant)
    assert e.value.code == 5 and "padding" in str(e.value)
    renderer.upload(sc[:1000])                                   # the context is still usable
    assert renderer.closest_hit(rays[:10])[0].shape == (10,)


@pytest.mark.parametrize("case", range(12))
def test_random_scenes_and_cameras(renderer, rt, oracle, case):
    """Scenes nobody arranged: sizes on both sides of every structural boundary, radii over five orders of magnitude
    (tiny ones and many large ones go to the always-scanned groups), duplicated spheres (equal hit distances: the
    lowest index must win), spheres far from the origin, cameras inside spheres -- small renders against the oracle."""
    rng = np.random.default_rng(9000 + case)
    n = int(rng.choice([3, 40, 130, 257, 600, 1025, 1500, 2047, 2100, 3000, 5000, 9000]))
    sc = np.zeros(n, rt.SPHERE_DTYPE)
    spread = float(rng.choice([3.0, 12.0, 60.0]))
    sc["center"] = rng.uniform(-spread, spread, (n, 3)) + (rng.choice([0.0, 0.0, 5000.0]) if case % 4 == 3 else 0.0)
    sc["radius"] = np.exp(rng.uniform(np.log(0.02), np.log(0.6), n))
    big = rng.choice(n, size=min(n, int(rng.choice([0, 1, 5, 40]))), replace=False)
    sc["radius"][big] *= rng.uniform(5, 40, len(big))
    tiny = rng.choice(n, size=min(n, 3), replace=False)
    sc["radius"][tiny] = 1e-6
    dup = rng.choice(n, size=n // 10, replace=False)
    src = rng.choice(n, size=n // 10, replace=True)
    sc["center"][dup] = sc["center"][src]
    sc["radius"][dup] = sc["radius"][src]
    sc["type"] = rng.integers(0, 3, n)
    sc["albedo"] = rng.uniform(0.1, 0.95, (n, 3))
    sc["param"] = np.where(sc["type"] == 2, rng.choice([1.3, 1.5, 2.4], n), rng.uniform(0, 0.5, n))
    renderer.upload(sc)
    c0 = sc["center"].mean(0)
    nx, ny, spp = 40, 24, 2
    for view in range(2):
        if view == 0:
            look_from = c0 + rng.normal(size=3) * spread * 1.5
        else:                                                   # inside (or on the skin of) some sphere
            k = int(rng.integers(0, n))
            look_from = sc["center"][k] + rng.normal(size=3) * sc["radius"][k] * 0.5
        cam = rt.camera_new(tuple(look_from), tuple(c0), (0, 1, 0), float(rng.uniform(15, 70)), nx / ny, float(rng.choice([0.0, 0.1])),
                            float(np.linalg.norm(look_from - c0) + 0.1))
        depth = int(rng.choice([4, 50]))
        got = renderer.render_host(cam, rt.Job(nx, ny, spp, seed=case, max_depth=depth))
        full = renderer.render_host(cam, rt.Job(nx, ny, spp, seed=case, max_depth=depth, flags=rt.FLAG_FULL_SCAN))
        want = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, spp, seed=case, max_depth=depth, host_threads=16)
        for k in ("rgb8", "first_hit", "hit_sum"):
            assert np.array_equal(got[k], want[k]), (case, view, k)
            assert np.array_equal(full[k], want[k]), (case, view, k, "full scan")
        assert np.array_equal(got["linear"].view(np.uint32), want["linear"].view(np.uint32)), (case, view)
