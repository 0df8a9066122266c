"""raytrace_amd -- MI355X path-tracing kernel for the per-pixel loop of the
inrick/raytrace C renderer, behind the C ABI of include/rt_amd.h.

This Python layer is plumbing for tests and benchmarks: it mirrors the
reference's host-side names (camera_new, small_scene, random_scene,
raytrace) and hands numpy arrays or torch device pointers to the C ABI.  The
product is the native code under csrc/ and host/.
"""
import ctypes as C

import os

import numpy as np

from . import _native as N
from ._native import (DIELECTRIC, FLAG_FULL_SCAN, FLAG_PROFILE, FLAG_PROGRESSIVE, FLAG_SERIAL, KERNEL_MEGA, KERNEL_WAVEFRONT, MATTE, METAL,
                      SCENE_DENSE, SCENE_FINAL, SCENE_GLASS, SCENE_SMALL, SCENE_TRIVIAL, SPHERE_DTYPE, STATIC_MAX_PIXELS,
                      RtCamera, RtError, RtJob)

__all__ = ["camera_new", "camera_default", "small_scene", "trivial_scene", "random_scene", "glass_scene",
           "build_scene", "Job", "Renderer", "RenderService", "raytrace", "write_ppm", "device_count", "RtError", "SPHERE_DTYPE"]


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def camera_new(lookfrom, lookat, vup, vfov, aspect, aperture, focus_dist):
    """camera_new of c/structs.c:49-79."""
    cam = RtCamera()
    N.host.rth_camera_new(_f3(lookfrom), _f3(lookat), _f3(vup), vfov, aspect, aperture, focus_dist, C.byref(cam))
    return cam


def camera_default(nx, ny):
    """The view of c/main.c:490-497 for an nx-by-ny image."""
    cam = RtCamera()
    N.host.rth_camera_default(nx, ny, C.byref(cam))
    return cam


def _two_pass(fn, *extra):
    n = fn(None, 0, *extra)
    s = np.zeros(n, SPHERE_DTYPE)
    got = fn(s.ctypes.data, n, *extra)
    if got == 0:
        raise RuntimeError("scene builder failed")
    return s[:got].copy()


def small_scene():
    return _two_pass(N.host.rth_scene_small)


def trivial_scene():
    return _two_pass(N.host.rth_scene_trivial)


def _fresh_rng():
    g = N.RthRng()
    N.host.rth_rng_init(C.byref(g))
    return g


def random_scene(half_extent=11):
    """random_scene of c/main.c:286-344 (488 spheres at half_extent 11; 40,004 at 100)."""
    n = N.host.rth_scene_random(None, 0, half_extent, None)
    s = np.zeros(n, SPHERE_DTYPE)
    g = _fresh_rng()
    got = N.host.rth_scene_random(s.ctypes.data, n, half_extent, C.byref(g))
    return s[:got].copy()


def glass_scene(half_extent=11):
    n = N.host.rth_scene_glass(None, 0, half_extent, None)
    s = np.zeros(n, SPHERE_DTYPE)
    g = _fresh_rng()
    got = N.host.rth_scene_glass(s.ctypes.data, n, half_extent, C.byref(g))
    return s[:got].copy()


def save_scene(path, spheres):
    """Writes a scene dump (rth_scene_save): header + the 36-byte sphere records."""
    s = np.ascontiguousarray(spheres, SPHERE_DTYPE)
    rc = N.host.rth_scene_save(os.fsencode(path), s.ctypes.data, len(s))
    if rc != 0:
        raise ValueError(f"rth_scene_save: error {rc}: {N.host.rth_last_error().decode()}")


def load_scene(path):
    """Reads a scene dump (rth_scene_load) into a structured array."""
    ptr, n = C.c_void_p(), C.c_size_t()
    rc = N.host.rth_scene_load(os.fsencode(path), C.byref(ptr), C.byref(n))
    if rc != 0:
        raise ValueError(f"rth_scene_load: error {rc}: {N.host.rth_last_error().decode()}")
    try:
        return np.frombuffer(C.string_at(ptr.value, n.value * SPHERE_DTYPE.itemsize), SPHERE_DTYPE).copy()
    finally:
        libc = C.CDLL(None)
        libc.free.argtypes = [C.c_void_p]
        libc.free(ptr)


def build_scene(which):
    return {SCENE_SMALL: small_scene, SCENE_FINAL: random_scene, SCENE_DENSE: lambda: random_scene(100),
            SCENE_GLASS: glass_scene, SCENE_TRIVIAL: trivial_scene}[which]()


def device_count():
    n = C.c_int()
    N.check(N.device.rt_device_count(C.byref(n)), "rt_device_count")
    return n.value


def Job(nx, ny, nsamples, max_depth=50, seed=0, row0=0, tile_rows=None, tile_stride=None, n_tiles=1,
        kernel=KERNEL_MEGA, tmin=np.float32(0.001), flags=0, sample0=0):
    tile_rows = (ny - row0) if tile_rows is None else tile_rows
    tile_stride = tile_rows if tile_stride is None else tile_stride
    return RtJob(nx, ny, row0, tile_rows, tile_stride, n_tiles, nsamples, max_depth, float(tmin), seed, kernel, flags, sample0)


def job_rows(job):
    return int(N.device.rt_job_rows(C.byref(job)))


def interleaved_job(nx, ny, nsamples, rank, world, tile_rows=8, **kw):
    """The row tiles rank `rank` of `world` renders: tiles rank, rank+world, ..."""
    total = (ny + tile_rows - 1) // tile_rows
    mine = (total - rank + world - 1) // world if total > rank else 0
    return Job(nx, ny, nsamples, row0=rank * tile_rows, tile_rows=tile_rows, tile_stride=world * tile_rows,
               n_tiles=mine, **kw)


def job_row_indices(job):
    """Image rows (top = 0) the job's compact output holds, in order."""
    rows = []
    for t in range(job.n_tiles):
        first = job.row0 + t * job.tile_stride
        if first >= job.ny:
            break
        rows.extend(range(first, min(first + job.tile_rows, job.ny)))
    return np.asarray(rows, dtype=np.int64)


class Renderer:
    """One rt_ctx: the stand-in for one render thread of c/main.c:531-539."""

    def __init__(self, device=0):
        h = C.c_void_p()
        N.check(N.device.rt_ctx_create(device, C.byref(h)), "rt_ctx_create")
        self._h = h
        self.device = device

    def close(self):
        if self._h:
            N.device.rt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        N.check(N.device.rt_ctx_set_stream(self._h, C.c_void_p(hip_stream)), "rt_ctx_set_stream")

    def synchronize(self):
        N.check(N.device.rt_ctx_synchronize(self._h), "rt_ctx_synchronize")

    def upload(self, spheres):
        s = np.ascontiguousarray(spheres)
        assert s.dtype == SPHERE_DTYPE
        N.check(N.device.rt_scene_upload(self._h, s.ctypes.data, len(s)), "rt_scene_upload")

    def render_host(self, cam, job, debug=True):
        """render() of c/main.c:446-483 for the job's rows -> numpy arrays."""
        rows, nx = job_rows(job), job.nx
        rgb = np.zeros((rows, nx, 3), np.uint8)
        lin = np.zeros((rows, nx, 3), np.float32) if debug else None
        fh = np.zeros((rows, nx), np.int32) if debug else None
        hs = np.zeros((rows, nx), np.uint32) if debug else None
        p = lambda a: None if a is None else a.ctypes.data
        N.check(N.device.rt_render_host(self._h, C.byref(cam), C.byref(job), p(rgb), p(lin), p(fh), p(hs)),
                "rt_render_host")
        return {"rgb8": rgb, "linear": lin, "first_hit": fh, "hit_sum": hs}

    def render_device(self, cam, job, d_rgb8, d_linear=0, d_first_hit=0, d_hit_sum=0, static=False):
        """Asynchronous render into device pointers (ints, e.g. tensor.data_ptr()).  static: through
        rt_render_device_static, the entry point that allocates nothing."""
        fn, name = ((N.device.rt_render_device_static, "rt_render_device_static") if static
                    else (N.device.rt_render_device, "rt_render_device"))
        N.check(fn(self._h, C.byref(cam), C.byref(job), C.c_void_p(d_rgb8), C.c_void_p(d_linear or None),
                   C.c_void_p(d_first_hit or None), C.c_void_p(d_hit_sum or None)), name)

    def last_kernel_ms(self):
        ms = C.c_float()
        N.check(N.device.rt_ctx_last_kernel_ms(self._h, C.byref(ms)), "rt_ctx_last_kernel_ms")
        return ms.value

    def trace_ms(self, back=0):
        """Trace-kernel time of the render `back` renders ago (HIP events of the library, on its stream)."""
        ms = C.c_float()
        N.check(N.device.rt_ctx_trace_ms(self._h, back, C.byref(ms)), "rt_ctx_trace_ms")
        return ms.value

    def counters(self, reset=False):
        c = N.RtCounters()
        N.check(N.device.rt_ctx_counters(self._h, C.byref(c), int(reset)), "rt_ctx_counters")
        return c.as_dict()

    def closest_hit(self, rays, tmin=np.float32(0.001), tmax=np.finfo(np.float32).max):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        idx = np.zeros(len(rays), np.int32)
        t = np.zeros(len(rays), np.float32)
        N.check(N.device.rt_debug_closest_hit(self._h, rays.ctypes.data, len(rays), float(tmin), float(tmax),
                                              idx.ctypes.data, t.ctypes.data), "rt_debug_closest_hit")
        return idx, t

    def uniforms(self, seed, pixel, sample, n):
        out = np.zeros(n, np.float64)
        N.check(N.device.rt_debug_uniforms(self._h, seed, pixel, sample, n, out.ctypes.data), "rt_debug_uniforms")
        return out

    def debug_filter(self, rays, spheres4):
        """Filter discriminants of every (ray, sphere) pair from the device's two packed routines."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        sp = np.ascontiguousarray(spheres4, np.float32).reshape(-1, 4)
        four = np.zeros((len(rays), len(sp)), np.float32)
        two = np.zeros_like(four)
        N.check(N.device.rt_debug_filter(self._h, rays.ctypes.data, len(rays), sp.ctypes.data, len(sp), four.ctypes.data,
                                         two.ctypes.data), "rt_debug_filter")
        return four, two

    def debug_box_pass(self, rays, boxes6, sigma0, sigma1):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        bx = np.ascontiguousarray(boxes6, np.float32).reshape(-1, 6)
        vec = np.zeros((len(rays), len(bx)), np.uint8)
        sca = np.zeros_like(vec)
        box_ray = np.zeros((len(rays), 9), np.float32)
        N.check(N.device.rt_debug_box_pass(self._h, rays.ctypes.data, len(rays), bx.ctypes.data, len(bx), float(sigma0),
                                           float(sigma1), vec.ctypes.data, sca.ctypes.data, box_ray.ctypes.data),
                "rt_debug_box_pass")
        return vec, sca, box_ray

    def debug_arith(self, op, a, b=None):
        a = np.ascontiguousarray(a, np.float64)
        b = None if b is None else np.ascontiguousarray(b, np.float64)
        out = np.zeros(len(a), np.float64)
        N.check(N.device.rt_debug_arith(self._h, op, a.ctypes.data, None if b is None else b.ctypes.data, len(a),
                                        out.ctypes.data), "rt_debug_arith")
        return out


def raytrace(cam, spheres, nx, ny, nsamples, n_devices=1, max_depth=50, seed=0, tile_rows=8, kernel=KERNEL_MEGA,
             device_ids=None):
    """raytrace() of c/main.c:485-548 through the C driver: row tiles over
    n_devices GPUs, gathered on the host.  Returns (image, max kernel ms)."""
    s = np.ascontiguousarray(spheres)
    img = np.zeros((ny, nx, 3), np.uint8)
    ids = None
    if device_ids is not None:
        ids = (C.c_int * len(device_ids))(*device_ids)
    opts = N.RthRenderOpts(nx, ny, nsamples, max_depth, seed, tile_rows, kernel, n_devices, ids)
    ms = C.c_float()
    rc = N.host.rth_raytrace(C.byref(cam), s.ctypes.data, len(s), C.byref(opts), img.ctypes.data, C.byref(ms))
    if rc != 0:
        raise RuntimeError(f"rth_raytrace: error {rc}: {N.host.rth_last_error().decode()}")
    return img, ms.value


class RenderService:
    """rth_service_* of host/rt_host.h: a resident worker that re-renders on a new camera and refines
    progressively (the worker of rs/src/gui.rs:164-172 with the scene kept on the GPU)."""

    def __init__(self, spheres, device=0):
        s = np.ascontiguousarray(spheres)
        h = C.c_void_p()
        rc = N.host.rth_service_start(device, s.ctypes.data, len(s), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"rth_service_start: error {rc}: {N.host.rth_last_error().decode()}")
        self._h = h

    def submit(self, cam, nx, ny, pass_samples, target_samples, max_depth=50, seed=0):
        rq = N.RthServiceRequest(nx, ny, pass_samples, target_samples, max_depth, seed)
        rid = N.host.rth_service_submit(self._h, C.byref(cam), C.byref(rq))
        if rid == 0:
            raise ValueError("bad request")
        return rid

    def wait(self, rid, min_samples):
        rc = N.host.rth_service_wait(self._h, rid, min_samples)
        if rc != 0:
            raise RuntimeError(f"rth_service_wait: error {rc}: {N.host.rth_last_error().decode()}")

    def latest(self, nx, ny):
        img = np.zeros((ny, nx, 3), np.uint8)
        rid, samples, gx, gy = C.c_uint64(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        rc = N.host.rth_service_latest(self._h, img.ctypes.data, img.nbytes, C.byref(rid), C.byref(samples), C.byref(gx),
                                       C.byref(gy))
        if rc != 0:
            return None
        return {"rgb8": img[:gy.value, :gx.value] if (gx.value, gy.value) == (nx, ny) else None, "id": rid.value,
                "samples": samples.value, "nx": gx.value, "ny": gy.value}

    def stats(self):
        u, p = C.c_uint64(), C.c_uint64()
        N.host.rth_service_stats(self._h, C.byref(u), C.byref(p))
        return {"uploads": u.value, "passes": p.value}

    def stop(self):
        if self._h:
            N.host.rth_service_stop(self._h)
            self._h = None


def write_ppm(path, img):
    """ppm_write of c/main.c:410-415 (the bytes, written from Python)."""
    h, w, _ = img.shape
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d 255\n" % (w, h))
        f.write(np.ascontiguousarray(img).tobytes())
