"""rt_render_device_static: the entry point that allocates nothing (its scratch is a static device array).
Through the C ABI, against rt_render_device, bit for bit, on two calls in a row, the second smaller."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def render_both(r, rt, torch, cam, job):
    rows, nx = rt.job_rows(job), job.nx
    outs = []
    for static in (False, True):
        rgb = torch.full((rows, nx, 3), 7, dtype=torch.uint8, device="cuda:0")
        lin = torch.full((rows, nx, 3), -1.0, dtype=torch.float32, device="cuda:0")
        fh = torch.full((rows, nx), -7, dtype=torch.int32, device="cuda:0")
        hs = torch.full((rows, nx), 99, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        r.render_device(cam, job, rgb.data_ptr(), lin.data_ptr(), fh.data_ptr(), hs.data_ptr(), static=static)
        r.synchronize()
        outs.append((rgb.cpu().numpy(), lin.cpu().numpy(), fh.cpu().numpy(), hs.cpu().numpy()))
    return outs


def test_static_entry_point_matches_the_allocating_one(rt):
    import torch
    r = rt.Renderer(0)
    try:
        r.upload(rt.random_scene())
        # first call: 640x360 at 40 spp = 3.7 MB of sums and 147 MB of radiance -> several sample batches in the
        # 32 MiB scratch; second call: smaller frame, fewer samples, interleaved tiles, one batch
        jobs = [(640, 360, rt.Job(640, 360, 40, seed=5)),
                (200, 120, rt.Job(200, 120, 3, seed=6, row0=8, tile_rows=8, tile_stride=24, n_tiles=4))]
        for nx, ny, job in jobs:
            cam = rt.camera_default(nx, ny)
            (rgb_a, lin_a, fh_a, hs_a), (rgb_s, lin_s, fh_s, hs_s) = render_both(r, rt, torch, cam, job)
            assert np.array_equal(rgb_a, rgb_s)
            assert np.array_equal(lin_a.view(np.uint32), lin_s.view(np.uint32))
            assert np.array_equal(fh_a, fh_s) and np.array_equal(hs_a, hs_s)
            assert lin_a.min() >= 0.0 and (fh_a >= -1).all()          # (the fill values are gone)
        # limits are errors, not faults
        big = rt.Job(1920, 1080, 1)
        out = torch.zeros((1080, 1920, 3), dtype=torch.uint8, device="cuda:0")
        with pytest.raises(rt.RtError) as e:
            r.render_device(rt.camera_default(1920, 1080), big, out.data_ptr(), static=True)
        assert e.value.code == 5 and "row bands" in str(e.value)
        with pytest.raises(rt.RtError):
            r.render_device(rt.camera_default(64, 36), rt.Job(64, 36, 1, kernel=rt.KERNEL_WAVEFRONT), out.data_ptr(), static=True)
        with pytest.raises(rt.RtError):
            r.render_device(rt.camera_default(64, 36), rt.Job(64, 36, 1, flags=rt.FLAG_PROGRESSIVE), out.data_ptr(), static=True)
        # (the profiling kernel logs a word per wave behind the event counts: the scratch's header has no room for it)
        with pytest.raises(rt.RtError) as e:
            r.render_device(rt.camera_default(64, 36), rt.Job(64, 36, 1, flags=rt.FLAG_PROFILE), out.data_ptr(), static=True)
        assert e.value.code == 5
        # and a frame too large for one call goes through in row bands
        nx, ny, spp = 1920, 1080, 2
        cam = rt.camera_default(nx, ny)
        whole = torch.zeros((ny, nx, 3), dtype=torch.uint8, device="cuda:0")
        r.render_device(cam, rt.Job(nx, ny, spp, seed=2), whole.data_ptr())
        bands = torch.zeros_like(whole)
        for row0 in range(0, ny, 360):
            r.render_device(cam, rt.Job(nx, ny, spp, seed=2, row0=row0, tile_rows=360), bands[row0:].data_ptr(), static=True)
        r.synchronize()
        assert torch.equal(whole, bands)
    finally:
        r.close()


def test_static_entry_point_allocates_nothing_and_runs_under_graph_capture(rt):
    """The claim itself: device memory in use does not move across static calls, and a static call can be recorded into
    a HIP graph (capture forbids allocation and synchronisation) whose replays give the allocating entry point's bytes.
    Also with RT_FLAG_FULL_SCAN (another kernel instantiation) and with several sample batches."""
    import torch
    dev = torch.device("cuda", 0)
    r = rt.Renderer(0)
    stream = torch.cuda.Stream(dev)
    try:
        r.set_stream(stream.cuda_stream)
        r.upload(rt.random_scene())
        nx, ny = 320, 180
        cam = rt.camera_default(nx, ny)
        for flags, spp in ((0, 6), (rt.FLAG_FULL_SCAN, 3), (0, 160)):      # 160 spp: 147 MB of radiance, five batches
            job = rt.Job(nx, ny, spp, seed=11, flags=flags)
            want = torch.zeros((ny, nx, 3), dtype=torch.uint8, device=dev)
            want_lin = torch.zeros((ny, nx, 3), dtype=torch.float32, device=dev)
            r.render_device(cam, job, want.data_ptr(), want_lin.data_ptr())
            got = torch.zeros_like(want)
            got_lin = torch.zeros_like(want_lin)
            r.render_device(cam, job, got.data_ptr(), got_lin.data_ptr(), static=True)      # warm-up of the static path
            r.synchronize()
            assert torch.equal(got, want) and torch.equal(got_lin.view(torch.int32), want_lin.view(torch.int32))
            free_before, _ = torch.cuda.mem_get_info(dev)
            for _ in range(3):
                got.zero_()
                r.render_device(cam, job, got.data_ptr(), got_lin.data_ptr(), static=True)
            r.synchronize()
            free_after, _ = torch.cuda.mem_get_info(dev)
            assert free_after == free_before, "device memory moved across static calls"
            assert torch.equal(got, want)
            # recorded into a graph and replayed
            got.zero_()
            got_lin.zero_()
            stream.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=stream):
                r.render_device(cam, job, got.data_ptr(), got_lin.data_ptr(), static=True)
            for _ in range(2):
                got.zero_()
                graph.replay()
                torch.cuda.synchronize(dev)
                assert torch.equal(got, want) and torch.equal(got_lin.view(torch.int32), want_lin.view(torch.int32))
            del graph
    finally:
        r.set_stream(0)
        r.close()
