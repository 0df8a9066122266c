"""ctypes front end of the CPU oracle (oracle/_out/librt_oracle.so) and, when
it has been built, of the compiled reference (oracle/_ref/libref_hooked.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "_out", "librt_oracle.so")
ORACLE_NATIVE_SO = os.path.join(HERE, "_out", "librt_oracle_native.so")   # -march=native, built where it is timed
REF_SO = os.path.join(HERE, "_ref", "libref_hooked.so")
REF_BIN = os.path.join(HERE, "_ref", "ray_ref")
REF_BIN_FINAL = os.path.join(HERE, "_ref", "ray_ref_final")
REF_BIN_GPU = os.path.join(HERE, "_ref", "ray_ref_gpu")      # the reference CLI with render() bound to the GPU library

SPHERE_DTYPE = np.dtype(
    [("center", "<f4", 3), ("radius", "<f4"), ("kind", "<i4"), ("albedo", "<f4", 3), ("param", "<f4")]
)
assert SPHERE_DTYPE.itemsize == 36
CAMERA_FLOATS = 22  # llc, horiz, vert, origin, u, v, w (3 each) + lens radius


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "segments", "tests", "rng_draws", "absorbed", "capped")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    """Compile the oracle (and the reference, where /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or (
        os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "rt_oracle.c"))
    ):
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if os.path.isdir("/root/reference/c"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
        # the reference CLI bound to the GPU library (proof that the drop-in compiles): needs the product library built
        if os.path.exists(os.path.join(os.path.dirname(HERE), "raytrace_amd", "lib", "libraytrace_amd.so")):
            subprocess.check_call(["make", "-s", "-C", HERE, "integration"])


_fp = C.POINTER(C.c_float)
_u8p = C.POINTER(C.c_uint8)
_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)


def _ptr(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Oracle:
    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        L.orc_ref_hash.restype = C.c_uint32
        L.orc_ref_hash.argtypes = [C.c_uint32]
        L.orc_ref_next.restype = C.c_double
        L.orc_ref_next.argtypes = [_u32p]
        L.orc_mix32.restype = C.c_uint32
        L.orc_mix32.argtypes = [C.c_uint32]
        L.orc_path_key.restype = C.c_uint32
        L.orc_path_key.argtypes = [C.c_uint32] * 3
        L.orc_ctr_bits.restype = C.c_uint32
        L.orc_ctr_bits.argtypes = [C.c_uint32] * 2
        L.orc_ctr_uniform.restype = C.c_double
        L.orc_ctr_uniform.argtypes = [C.c_uint32] * 2
        L.orc_camera_make.restype = None
        L.orc_camera_make.argtypes = [_fp, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float, _fp]
        L.orc_camera_default.restype = None
        L.orc_camera_default.argtypes = [C.c_size_t, C.c_size_t, _fp]
        for name in ("orc_scene_small", "orc_scene_trivial"):
            f = getattr(L, name)
            f.restype = C.c_size_t
            f.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_scene_random.restype = C.c_size_t
        L.orc_scene_random.argtypes = [C.c_void_p, C.c_size_t, C.c_int, _u32p]
        L.orc_scene_glassify.restype = None
        L.orc_scene_glassify.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_hit_sphere.restype = C.c_int
        L.orc_hit_sphere.argtypes = [C.c_void_p, _fp, _fp, C.c_float, C.c_float, _fp, _fp, _fp]
        L.orc_hit_scene.restype = C.c_int
        L.orc_hit_scene.argtypes = [C.c_void_p, C.c_size_t, _fp, _fp, C.c_float, C.c_float, _fp, _fp, _fp]
        L.orc_hit_scene_many.restype = C.c_int
        L.orc_hit_scene_many.argtypes = [C.c_void_p, C.c_size_t, _fp, C.c_size_t, C.c_float, C.c_float, _i32p, _fp, C.c_size_t]
        L.orc_schlick.restype = C.c_float
        L.orc_schlick.argtypes = [C.c_float, C.c_float]
        L.orc_refract.restype = C.c_int
        L.orc_refract.argtypes = [_fp, _fp, C.c_float, _fp]
        L.orc_render_sequential.restype = C.c_int
        L.orc_render_sequential.argtypes = [_fp, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                            C.c_size_t, _u8p, C.POINTER(Stats)]
        L.orc_render_det.restype = C.c_int
        L.orc_render_det.argtypes = [_fp, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                     C.c_size_t, C.c_size_t, C.c_uint32, _u8p, _fp, _i32p, _u32p,
                                     C.POINTER(Stats), C.c_size_t]

    # ---- generator ----
    def ref_stream(self, n, state=123456789):
        st = C.c_uint32(state)
        return [self.lib.orc_ref_next(C.byref(st)) for _ in range(n)]

    # ---- inputs ----
    def camera_default(self, nx, ny):
        cam = np.zeros(CAMERA_FLOATS, np.float32)
        self.lib.orc_camera_default(nx, ny, _ptr(cam, _fp))
        return cam

    def camera_make(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus):
        cam = np.zeros(CAMERA_FLOATS, np.float32)
        self.lib.orc_camera_make(_f3(lookfrom), _f3(lookat), _f3(vup), vfov, aspect, aperture, focus, _ptr(cam, _fp))
        return cam

    def scene_small(self):
        n = self.lib.orc_scene_small(None, 0)
        s = np.zeros(n, SPHERE_DTYPE)
        assert self.lib.orc_scene_small(s.ctypes.data, n) == n
        return s

    def scene_trivial(self):
        s = np.zeros(2, SPHERE_DTYPE)
        assert self.lib.orc_scene_trivial(s.ctypes.data, 2) == 2
        return s

    def scene_random(self, half_extent=11, state=123456789):
        st = C.c_uint32(state)
        n = self.lib.orc_scene_random(None, 0, half_extent, C.byref(st))
        s = np.zeros(n, SPHERE_DTYPE)
        got = self.lib.orc_scene_random(s.ctypes.data, n, half_extent, C.byref(st))
        return s[:got].copy()

    def scene_glass(self, half_extent=11):
        s = self.scene_random(half_extent)
        self.lib.orc_scene_glassify(s.ctypes.data, len(s))
        return s

    # ---- functions ----
    def hit_sphere(self, sphere, A, B, tmin, tmax):
        t = C.c_float()
        p = (C.c_float * 3)()
        n = (C.c_float * 3)()
        s = np.ascontiguousarray(sphere)
        ok = self.lib.orc_hit_sphere(s.ctypes.data, _f3(A), _f3(B), tmin, tmax, C.byref(t), p, n)
        return (ok, np.float32(t.value), np.array(p[:], np.float32), np.array(n[:], np.float32))

    def hit_scene(self, spheres, A, B, tmin=np.float32(0.001), tmax=np.finfo(np.float32).max):
        t = C.c_float()
        p = (C.c_float * 3)()
        n = (C.c_float * 3)()
        idx = self.lib.orc_hit_scene(spheres.ctypes.data, len(spheres), _f3(A), _f3(B), tmin, tmax, C.byref(t), p, n)
        return idx, np.float32(t.value), np.array(p[:], np.float32), np.array(n[:], np.float32)

    def hit_scene_many(self, spheres, rays, tmin=np.float32(0.001), tmax=np.finfo(np.float32).max, threads=16):
        """hit_scene for every ray of rays[m, 6] (A, B): (index[m] int32 with -1 for no hit, t[m] float32)."""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        idx = np.full(len(rays), -1, np.int32)
        t = np.zeros(len(rays), np.float32)
        rc = self.lib.orc_hit_scene_many(spheres.ctypes.data, len(spheres), _ptr(rays, _fp), len(rays), tmin, tmax,
                                         _ptr(idx, _i32p), _ptr(t, _fp), threads)
        assert rc == 0
        return idx, t

    # ---- renders ----
    def render_sequential(self, cam, spheres, nx, ny, nsamples, threads):
        out = np.zeros((ny, nx, 3), np.uint8)
        st = Stats()
        rc = self.lib.orc_render_sequential(_ptr(cam, _fp), spheres.ctypes.data, len(spheres), nx, ny, nsamples,
                                            threads, _ptr(out, _u8p), C.byref(st))
        assert rc == 0
        return out, st.as_dict()

    def render_det(self, cam, spheres, nx, ny_total, nsamples, max_depth=50, seed=0, row0=0, rows=None,
                   host_threads=8, want_debug=True):
        rows = ny_total - row0 if rows is None else rows
        rgb = np.zeros((rows, nx, 3), np.uint8)
        lin = np.zeros((rows, nx, 3), np.float32) if want_debug else None
        fh = np.zeros((rows, nx), np.int32) if want_debug else None
        hs = np.zeros((rows, nx), np.uint32) if want_debug else None
        st = Stats()
        rc = self.lib.orc_render_det(_ptr(cam, _fp), spheres.ctypes.data, len(spheres), nx, ny_total, row0, rows,
                                     nsamples, max_depth, seed, _ptr(rgb, _u8p), _ptr(lin, _fp), _ptr(fh, _i32p),
                                     _ptr(hs, _u32p), C.byref(st), host_threads)
        assert rc == 0, rc
        return {"rgb8": rgb, "linear": lin, "first_hit": fh, "hit_sum": hs, "stats": st.as_dict()}


class HookedReference:
    """The real reference TU, compiled by oracle/Makefile (only where
    /root/reference exists; the .so itself travels)."""

    def __init__(self, path=REF_SO):
        L = self.lib = C.CDLL(path)
        L.ref_f_randi.restype = C.c_uint32
        L.ref_f_randi.argtypes = [C.c_uint32]
        L.ref_seq_stream.restype = None
        L.ref_seq_stream.argtypes = [C.c_size_t, C.POINTER(C.c_double)]
        L.ref_camera_new.restype = None
        L.ref_camera_new.argtypes = [_fp, _fp, _fp, C.c_float, C.c_float, C.c_float, C.c_float, _fp]
        L.ref_sizeof_camera.restype = C.c_size_t
        L.ref_sizeof_sphere.restype = C.c_size_t
        L.ref_scene.restype = C.c_size_t
        L.ref_scene.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
        L.ref_hit_sphere.restype = C.c_int
        L.ref_hit_sphere.argtypes = [C.c_void_p, _fp, _fp, C.c_float, C.c_float, _fp, _fp, _fp]
        L.ref_schlick.restype = C.c_float
        L.ref_schlick.argtypes = [C.c_float, C.c_float]
        L.ref_refract.restype = C.c_int
        L.ref_refract.argtypes = [_fp, _fp, C.c_float, _fp]
        L.ref_scatter.restype = C.c_int
        L.ref_scatter.argtypes = [C.c_void_p, _fp, _fp, _fp, _fp, C.c_uint32, _fp, _fp, _fp, _u32p]
        L.ref_det_render.restype = C.c_int
        L.ref_det_render.argtypes = [_fp, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                     C.c_size_t, C.c_uint32, _u8p, _fp, _i32p, _u32p]
        L.ref_render_one_band.restype = C.c_int
        L.ref_render_one_band.argtypes = [_fp, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _u8p]

    def seq_stream(self, n):
        out = (C.c_double * n)()
        self.lib.ref_seq_stream(n, out)
        return list(out)

    def camera_new(self, lookfrom, lookat, vup, vfov, aspect, aperture, focus):
        cam = np.zeros(CAMERA_FLOATS, np.float32)
        self.lib.ref_camera_new(_f3(lookfrom), _f3(lookat), _f3(vup), vfov, aspect, aperture, focus, _ptr(cam, _fp))
        return cam

    def scene(self, which):
        s = np.zeros(600, SPHERE_DTYPE)
        n = self.lib.ref_scene(which, s.ctypes.data, 600)
        return s[:n].copy()

    def hit_sphere(self, sphere, A, B, tmin, tmax):
        t = C.c_float()
        p = (C.c_float * 3)()
        n = (C.c_float * 3)()
        s = np.ascontiguousarray(sphere)
        ok = self.lib.ref_hit_sphere(s.ctypes.data, _f3(A), _f3(B), tmin, tmax, C.byref(t), p, n)
        return (ok, np.float32(t.value), np.array(p[:], np.float32), np.array(n[:], np.float32))

    def det_render(self, cam, spheres, nx, ny_total, nsamples, seed=0, row0=0, rows=None):
        rows = ny_total - row0 if rows is None else rows
        rgb = np.zeros((rows, nx, 3), np.uint8)
        lin = np.zeros((rows, nx, 3), np.float32)
        fh = np.zeros((rows, nx), np.int32)
        hs = np.zeros((rows, nx), np.uint32)
        rc = self.lib.ref_det_render(_ptr(cam, _fp), spheres.ctypes.data, len(spheres), nx, ny_total, row0, rows,
                                     nsamples, seed, _ptr(rgb, _u8p), _ptr(lin, _fp), _ptr(fh, _i32p), _ptr(hs, _u32p))
        assert rc == 0
        return {"rgb8": rgb, "linear": lin, "first_hit": fh, "hit_sum": hs}

    def render_one_band(self, cam, spheres, nx, ny, nsamples):
        rgb = np.zeros((ny, nx, 3), np.uint8)
        rc = self.lib.ref_render_one_band(_ptr(cam, _fp), spheres.ctypes.data, len(spheres), nx, ny, nsamples,
                                          _ptr(rgb, _u8p))
        assert rc == 0
        return rgb


def have_reference():
    return os.path.exists(REF_SO) and os.path.exists(REF_BIN) and os.path.exists(REF_BIN_FINAL)
