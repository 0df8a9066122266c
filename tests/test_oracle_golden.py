"""The oracle against the reference's golden vectors (tests/golden/, made by
make_golden.py from the compiled reference).  CPU only."""
import hashlib
import re

import numpy as np
import pytest

from conftest import load_golden_npz


def ppm_sha(img):
    h, w, _ = img.shape
    return hashlib.sha256(b"P6\n%d %d 255\n" % (w, h) + img.tobytes()).hexdigest()


def test_generator_known_answers(oracle, kat):
    state = 123456789
    chain = []
    for _ in range(8):
        r = oracle.lib.orc_ref_hash(state)
        chain.append(r)
        state = (r + 1) & 0xFFFFFFFF
    assert chain == kat["f_randi_chain"]
    assert chain[:2] == [894613773, 2064146235]
    assert [float(x).hex() for x in oracle.ref_stream(8)] == kat["my_rand_first8_hex"]


def test_default_camera(oracle, kat):
    assert [float(x).hex() for x in oracle.camera_default(600, 300)] == kat["camera_600x300_hex"]
    assert [float(x).hex() for x in oracle.camera_default(1920, 1080)] == kat["camera_1920x1080_hex"]


def test_scene_dumps(oracle, kat):
    small = np.load("tests/golden/small_scene.npy", allow_pickle=False)
    final = np.load("tests/golden/final_scene.npy", allow_pickle=False)
    assert oracle.scene_small().tobytes() == small.tobytes()
    got = oracle.scene_random()
    assert got.tobytes() == final.tobytes()
    assert len(got) == 488
    assert [int(x) for x in np.bincount(got["kind"], minlength=3)] == kat["final_scene_counts"] == [394, 70, 24]


@pytest.mark.parametrize("n,t", [(1, 1), (10, 1), (10, 2), (10, 4), (10, 8), (3, 7)])
def test_sequential_small_scene_matches_reference_bytes(oracle, kat, n, t):
    cam = oracle.camera_default(600, 300)
    img, _ = oracle.render_sequential(cam, oracle.scene_small(), 600, 300, n, t)
    assert ppm_sha(img) == kat["ppm_sha256"][f"small_n{n}_t{t}"]
    if (n, t) == (10, 1):
        for k, v in kat["probe_pixels"]["small_n10_t1"].items():
            x, y = [int(q) for q in k.split(",")]
            assert list(img[y, x]) == v


@pytest.mark.parametrize("n,t", [(1, 8), (2, 3)])
def test_sequential_final_scene_matches_reference_bytes(oracle, kat, n, t):
    cam = oracle.camera_default(600, 300)
    img, st = oracle.render_sequential(cam, oracle.scene_random(), 600, 300, n, t)
    assert ppm_sha(img) == kat["ppm_sha256"][f"final_n{n}_t{t}"]
    assert st["paths"] == 600 * 300 * n and st["tests"] == st["segments"] * 488


@pytest.mark.parametrize("name,scene", [("det_small_96x54_s4_seed1.npz", "small"), ("det_final_96x54_s4_seed1.npz", "final"),
                                        ("det_final_64x36_s16_seed9.npz", "final"), ("det_glass_64x36_s8_seed5.npz", "glass")])
def test_deterministic_mode_matches_reference_functions(oracle, name, scene):
    g = load_golden_npz(name)
    sc = {"small": oracle.scene_small, "final": oracle.scene_random, "glass": oracle.scene_glass}[scene]()
    ny, nx, _ = g["rgb8"].shape
    m = re.match(r"det_[a-z]+_\d+x\d+_s(\d+)_seed(\d+)\.npz", name)
    spp, seed = int(m.group(1)), int(m.group(2))
    out = oracle.render_det(g["camera"], sc, nx, ny, spp, seed=seed)
    assert np.array_equal(out["rgb8"], g["rgb8"])
    assert np.array_equal(out["linear"].view(np.uint32), g["linear"].view(np.uint32))  # bit for bit
    assert np.array_equal(out["first_hit"], g["first_hit"])
    assert np.array_equal(out["hit_sum"], g["hit_sum"])


def test_deterministic_mode_independent_of_partition(oracle):
    cam = oracle.camera_default(80, 45)
    sc = oracle.scene_random()
    full = oracle.render_det(cam, sc, 80, 45, 3, seed=11, host_threads=1)
    part = oracle.render_det(cam, sc, 80, 45, 3, seed=11, row0=17, rows=9, host_threads=4)
    for k in ("rgb8", "linear", "first_hit", "hit_sum"):
        assert np.array_equal(full[k][17:26], part[k])
    empty = oracle.render_det(cam, sc, 80, 45, 3, seed=11, row0=45, rows=0)
    assert empty["rgb8"].shape[0] == 0


def test_counter_stream_is_keyed_per_path(oracle):
    k0 = oracle.lib.orc_path_key(1, 100, 0)
    assert k0 != oracle.lib.orc_path_key(1, 100, 1) != oracle.lib.orc_path_key(1, 101, 0)
    assert k0 != oracle.lib.orc_path_key(2, 100, 0)
    u = [oracle.lib.orc_ctr_uniform(k0, k) for k in range(1000)]
    assert all(0.0 <= x <= 1.0 for x in u) and 0.4 < sum(u) / 1000 < 0.6
