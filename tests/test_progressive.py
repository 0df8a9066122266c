"""Progressive accumulation and the resident render service (the reference's interactive front end keeps a
worker thread that re-renders on request, rs/src/gui.rs:164-172, 340-357)."""
import numpy as np
import pytest

from conftest import as_oracle_scene

pytestmark = pytest.mark.gpu


def test_k_passes_of_m_samples_equal_one_render(renderer, rt, oracle):
    sc = rt.random_scene()
    renderer.upload(sc)
    nx, ny = 160, 90
    cam = rt.camera_default(nx, ny)
    whole = renderer.render_host(cam, rt.Job(nx, ny, 12, seed=7))
    want = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, 12, seed=7, host_threads=16)
    for k in ("rgb8", "first_hit", "hit_sum"):
        assert np.array_equal(whole[k], want[k])
    done = 0
    for m in (4, 4, 3, 1):                      # uneven passes too
        part = renderer.render_host(cam, rt.Job(nx, ny, m, seed=7, flags=rt.FLAG_PROGRESSIVE, sample0=done))
        done += m
        if done == 8:
            eight = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, 8, seed=7, host_threads=16)
            assert np.array_equal(part["rgb8"], eight["rgb8"])
            assert np.array_equal(part["linear"].view(np.uint32), eight["linear"].view(np.uint32))
    assert done == 12
    for k in ("rgb8", "first_hit", "hit_sum"):
        assert np.array_equal(part[k], whole[k]), k
    assert np.array_equal(part["linear"].view(np.uint32), whole["linear"].view(np.uint32))


def test_progressive_calls_must_continue_the_same_render(renderer, rt):
    sc = rt.small_scene()
    renderer.upload(sc)
    nx, ny = 64, 36
    cam = rt.camera_default(nx, ny)
    P = rt.FLAG_PROGRESSIVE
    with pytest.raises(rt.RtError):             # nothing to continue
        renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P, sample0=2))
    with pytest.raises(rt.RtError):             # sample0 without the flag
        renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, sample0=2))
    renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P))
    with pytest.raises(rt.RtError):             # a gap in the samples
        renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P, sample0=3))
    renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P))
    with pytest.raises(rt.RtError):             # another seed
        renderer.render_host(cam, rt.Job(nx, ny, 2, seed=2, flags=P, sample0=2))
    renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P))
    other = rt.camera_new((9, 2, 4), (-4, 0, -2), (0, 1, 0), 20.0, nx / ny, 0.05, 10.0)
    with pytest.raises(rt.RtError):             # another camera
        renderer.render_host(other, rt.Job(nx, ny, 2, seed=1, flags=P, sample0=2))
    renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P))
    renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1))                      # an ordinary render in between
    with pytest.raises(rt.RtError):
        renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P, sample0=2))
    renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P))
    renderer.upload(sc)                                                        # a new scene
    with pytest.raises(rt.RtError):
        renderer.render_host(cam, rt.Job(nx, ny, 2, seed=1, flags=P, sample0=2))


def test_service_rerenders_on_a_new_camera_without_another_upload(rt, oracle):
    sc = rt.random_scene()
    nx, ny = 128, 72
    svc = rt.RenderService(sc, 0)
    try:
        cam_a = rt.camera_default(nx, ny)
        rid = svc.submit(cam_a, nx, ny, pass_samples=4, target_samples=12, seed=3)
        svc.wait(rid, 12)
        got = svc.latest(nx, ny)
        assert got["id"] == rid and got["samples"] == 12
        want = oracle.render_det(cam_a.as_array(), as_oracle_scene(sc), nx, ny, 12, seed=3, host_threads=16, want_debug=False)
        assert np.array_equal(got["rgb8"], want["rgb8"])
        assert svc.stats() == {"uploads": 1, "passes": 3}
        # the camera moves: the picture restarts from the first sample, the scene is not uploaded again
        cam_b = rt.camera_new((8.0, 3.0, 6.0), (-4.0, 0.0, -2.0), (0.0, 1.0, 0.0), 20.0, nx / ny, 0.05, 12.0)
        rid_b = svc.submit(cam_b, nx, ny, pass_samples=5, target_samples=10, seed=3)
        svc.wait(rid_b, 5)
        mid = svc.latest(nx, ny)
        assert mid["id"] == rid_b and mid["samples"] in (5, 10)
        svc.wait(rid_b, 1000)     # (more than the request's target: returns when the target is reached)
        got_b = svc.latest(nx, ny)
        want_b = oracle.render_det(cam_b.as_array(), as_oracle_scene(sc), nx, ny, 10, seed=3, host_threads=16, want_debug=False)
        assert got_b["samples"] == 10 and np.array_equal(got_b["rgb8"], want_b["rgb8"])
        assert not np.array_equal(got_b["rgb8"], got["rgb8"])
        assert svc.stats()["uploads"] == 1
    finally:
        svc.stop()
