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
