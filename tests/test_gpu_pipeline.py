"""Device renders queued back to back on one context: the library runs consecutive trace kernels on two internal
streams in turn with two radiance buffers and two queue heads, and resolves each on the context's stream.  Nothing
here synchronises between the calls; every picture is then compared with the oracle's."""
import numpy as np
import pytest

from conftest import as_oracle_scene

pytestmark = pytest.mark.gpu


def _oracle_rows(oracle, cam, sc, job, rt):
    import raytrace_amd as R
    rows = R.job_row_indices(job)
    out = np.zeros((len(rows), job.nx, 3), np.uint8)
    for k, row in enumerate(rows):
        out[k] = oracle.render_det(cam.as_array(), sc, job.nx, job.ny, job.nsamples, max_depth=job.max_depth, seed=job.seed,
                                   row0=int(row), rows=1, host_threads=8, want_debug=False)["rgb8"][0]
    return out


def test_renders_queued_without_synchronising(rt, oracle):
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    sc = rt.random_scene()
    osc = as_oracle_scene(sc)
    r = rt.Renderer(0)
    try:
        r.upload(sc)
        stream = torch.cuda.Stream(dev)
        r.set_stream(stream.cuda_stream)
        rng = np.random.default_rng(31)
        plan = []
        # sizes go up and down (the workspace is regrown while earlier renders are still in flight), sample counts
        # differ, some renders are row tiles, some ask for no overlap, one scans without culling
        for k, (nx, ny, spp) in enumerate([(96, 54, 4), (200, 120, 2), (64, 36, 9), (320, 180, 1), (96, 54, 4), (33, 17, 5),
                                           (200, 120, 2), (160, 90, 3), (64, 36, 9), (256, 64, 2), (96, 54, 1), (320, 180, 1)]):
            cam = rt.camera_new(tuple(rng.uniform(6, 12, 3) * (1, 0.3, 1)), (-4 + rng.uniform(-1, 1), 0, -2), (0, 1, 0), 20.0 + 5 * (k % 3),
                                nx / ny, 0.05, 12.0)
            flags = (rt.FLAG_SERIAL if k % 5 == 3 else 0) | (rt.FLAG_FULL_SCAN if k == 7 else 0)
            if k % 4 == 2:
                job = rt.interleaved_job(nx, ny, spp, rank=1, world=3, tile_rows=8, seed=100 + k, flags=flags)
            else:
                job = rt.Job(nx, ny, spp, seed=100 + k, flags=flags)
            out = torch.zeros((rt.job_rows(job), nx, 3), dtype=torch.uint8, device=dev)
            plan.append((cam, job, out))
        with torch.cuda.stream(stream):
            for _ in range(2):                              # the same twelve twice: every buffer is reused
                for cam, job, out in plan:
                    r.render_device(cam, job, out.data_ptr())
        stream.synchronize()
        for k, (cam, job, out) in enumerate(plan):
            want = _oracle_rows(oracle, cam, osc, job, rt)
            assert np.array_equal(out.cpu().numpy(), want), k
    finally:
        r.close()


def test_scene_upload_between_queued_renders(rt, oracle):
    """A new scene is uploaded while renders of the old one are still queued: they finish with the old scene."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    scenes = [rt.random_scene(), rt.small_scene(), rt.glass_scene()]
    r = rt.Renderer(0)
    try:
        nx, ny, spp = 160, 90, 3
        cam = rt.camera_default(nx, ny)
        outs = []
        for k, sc in enumerate(scenes * 2):
            r.upload(sc)
            for s in range(3):
                out = torch.zeros((ny, nx, 3), dtype=torch.uint8, device=dev)
                r.render_device(cam, rt.Job(nx, ny, spp, seed=10 * k + s), out.data_ptr())
                outs.append((sc, 10 * k + s, out))
        r.synchronize()
        for sc, seed, out in outs:
            want = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, spp, seed=seed, host_threads=16, want_debug=False)["rgb8"]
            assert np.array_equal(out.cpu().numpy(), want), seed
    finally:
        r.close()


@pytest.mark.parametrize("scene,n", [("random_scene", None), ("random_scene", 30)])
def test_profiling_build_of_the_kernel_renders_the_same_picture(rt, oracle, scene, n):
    """RT_FLAG_PROFILE selects the instrumented instantiation of the trace kernel (cycle counts per phase, wave start
    and end times): the same bits as the plain one, for a scene in LDS and for one scanned in place, and the cycle
    counters it fills add up."""
    sc = getattr(rt, scene)(n) if n else getattr(rt, scene)()
    r = rt.Renderer(0)
    try:
        r.upload(sc)
        nx, ny, spp = 160, 90, 3
        cam = rt.camera_default(nx, ny)
        plain = r.render_host(cam, rt.Job(nx, ny, spp, seed=2))
        r.counters(reset=True)
        prof = r.render_host(cam, rt.Job(nx, ny, spp, seed=2, flags=rt.FLAG_PROFILE))
        c = r.counters()
        for k in ("rgb8", "first_hit", "hit_sum"):
            assert np.array_equal(plain[k], prof[k]), k
        assert np.array_equal(plain["linear"].view(np.uint32), prof["linear"].view(np.uint32))
        want = oracle.render_det(cam.as_array(), as_oracle_scene(sc), nx, ny, spp, seed=2, host_threads=16)
        assert np.array_equal(prof["rgb8"], want["rgb8"])
        assert c["paths"] == nx * ny * spp and c["cycles_total"] > 0
        assert c["cycles_scan"] + c["cycles_drain"] + c["cycles_shade"] <= c["cycles_total"]
        assert c["first_wave_start"] <= c["last_wave_start"] <= c["last_wave_end"] and c["first_wave_end"] <= c["last_wave_end"]
    finally:
        r.close()
