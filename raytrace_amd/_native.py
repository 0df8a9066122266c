"""ctypes bindings of the two native libraries.  No fallback of any kind: if a
library is missing or a symbol is absent, importing fails loudly."""
import ctypes as C
import os

import numpy as np

_LIB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
# RT_AMD_DEVICE_LIB: another build of the same library (kernel experiments); it must exist
DEVICE_LIB = os.environ.get("RT_AMD_DEVICE_LIB") or os.path.join(_LIB_DIR, "libraytrace_amd.so")
HOST_LIB = os.path.join(_LIB_DIR, "libraytrace_host.so")

SPHERE_DTYPE = np.dtype(
    [("center", "<f4", 3), ("radius", "<f4"), ("type", "<i4"), ("albedo", "<f4", 3), ("param", "<f4")]
)
assert SPHERE_DTYPE.itemsize == 36

MATTE, METAL, DIELECTRIC = 0, 1, 2
KERNEL_MEGA, KERNEL_WAVEFRONT = 0, 1
FLAG_PROFILE, FLAG_FULL_SCAN, FLAG_PROGRESSIVE, FLAG_SERIAL = 1, 2, 4, 8
ABI_VERSION = 2
STATIC_SCRATCH_BYTES, STATIC_MAX_PIXELS = 32 << 20, 700000
ARITH_SQRT_WIDENED, ARITH_DIV_WIDENED, ARITH_DIV_F32_BY, ARITH_RCP_NARROWED, ARITH_UNIT_FLOAT, ARITH_UNIT_DOUBLE = range(6)
SCENE_SMALL, SCENE_FINAL, SCENE_DENSE, SCENE_GLASS, SCENE_TRIVIAL = 0, 1, 2, 3, 4


class RtCamera(C.Structure):
    _fields_ = [(n, C.c_float * 3) for n in ("llc", "horiz", "vert", "origin", "u", "v", "w")] + [
        ("lens_radius", C.c_float)
    ]

    def as_array(self):
        return np.frombuffer(bytes(self), dtype=np.float32).copy()

    @classmethod
    def from_array(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        assert a.size == 22
        return cls.from_buffer_copy(a.tobytes())


assert C.sizeof(RtCamera) == 88


class RtJob(C.Structure):
    _fields_ = [
        ("nx", C.c_uint32), ("ny", C.c_uint32), ("row0", C.c_uint32), ("tile_rows", C.c_uint32),
        ("tile_stride", C.c_uint32), ("n_tiles", C.c_uint32), ("nsamples", C.c_uint32), ("max_depth", C.c_uint32),
        ("tmin", C.c_float), ("seed", C.c_uint32), ("kernel", C.c_uint32), ("flags", C.c_uint32),
        ("sample0", C.c_uint32),
    ]


class RtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "segments", "sphere_tests", "box_tests", "candidates", "wave_iterations",
                                          "live_lane_iterations", "cycles_scan", "cycles_drain", "cycles_shade",
                                          "cycles_total", "realtime_total", "own_groups", "wave_groups", "first_wave_start",
                                          "last_wave_start", "first_wave_end", "last_wave_end")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class RthRenderOpts(C.Structure):
    _fields_ = [
        ("nx", C.c_uint32), ("ny", C.c_uint32), ("nsamples", C.c_uint32), ("max_depth", C.c_uint32),
        ("seed", C.c_uint32), ("tile_rows", C.c_uint32), ("kernel", C.c_uint32), ("n_devices", C.c_int),
        ("device_ids", C.POINTER(C.c_int)),
    ]


class RthServiceRequest(C.Structure):
    _fields_ = [("nx", C.c_uint32), ("ny", C.c_uint32), ("pass_samples", C.c_uint32), ("target_samples", C.c_uint32),
                ("max_depth", C.c_uint32), ("seed", C.c_uint32)]


class RthRng(C.Structure):
    _fields_ = [("state", C.c_uint32)]


# every symbol include/rt_amd.h declares
DEVICE_SYMBOLS = {
    "rt_job_rows": (C.c_uint32, [C.POINTER(RtJob)]),
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (C.c_int, []),
    "rt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "rt_ctx_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rt_ctx_destroy": (C.c_int, [C.c_void_p]),
    "rt_ctx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rt_ctx_synchronize": (C.c_int, [C.c_void_p]),
    "rt_scene_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "rt_render_device": (C.c_int, [C.c_void_p, C.POINTER(RtCamera), C.POINTER(RtJob), C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p]),
    "rt_render_host": (C.c_int, [C.c_void_p, C.POINTER(RtCamera), C.POINTER(RtJob), C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "rt_render_device_static": (C.c_int, [C.c_void_p, C.POINTER(RtCamera), C.POINTER(RtJob), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "rt_ctx_last_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "rt_ctx_trace_ms": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "rt_ctx_counters": (C.c_int, [C.c_void_p, C.POINTER(RtCounters), C.c_int]),
    "rt_debug_closest_hit": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float, C.c_void_p,
                                       C.c_void_p]),
    "rt_debug_uniforms": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_size_t, C.c_void_p]),
    "rt_debug_filter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "rt_debug_box_pass": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_debug_arith": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "rt_debug_wave_log": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
}

HOST_SYMBOLS = {
    "rth_rng_init": (None, [C.POINTER(RthRng)]),
    "rth_rng_hash": (C.c_uint32, [C.c_uint32]),
    "rth_rng_next": (C.c_double, [C.POINTER(RthRng)]),
    "rth_camera_new": (None, [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_float,
                              C.c_float, C.c_float, C.POINTER(RtCamera)]),
    "rth_camera_default": (None, [C.c_uint32, C.c_uint32, C.POINTER(RtCamera)]),
    "rth_scene_small": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "rth_scene_trivial": (C.c_size_t, [C.c_void_p, C.c_size_t]),
    "rth_scene_random": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(RthRng)]),
    "rth_scene_glass": (C.c_size_t, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(RthRng)]),
    "rth_scene_build": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rth_scene_save": (C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t]),
    "rth_scene_load": (C.c_int, [C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "rth_write_ppm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "rth_write_png": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "rth_write_jpg": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]),
    "rth_raytrace": (C.c_int, [C.POINTER(RtCamera), C.c_void_p, C.c_size_t, C.POINTER(RthRenderOpts), C.c_void_p,
                               C.POINTER(C.c_float)]),
    "rth_raytrace_ex": (C.c_int, [C.POINTER(RtCamera), C.c_void_p, C.c_size_t, C.POINTER(RthRenderOpts), C.c_void_p,
                                  C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "rth_last_error": (C.c_char_p, []),
    "rth_service_start": (C.c_int, [C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "rth_service_submit": (C.c_uint64, [C.c_void_p, C.POINTER(RtCamera), C.POINTER(RthServiceRequest)]),
    "rth_service_latest": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "rth_service_wait": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    "rth_service_stats": (None, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rth_service_stop": (C.c_int, [C.c_void_p]),
}


def _bind(path, table):
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: build it with `python raytrace_amd/build.py` (there is no fallback path)")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is absent
        fn.restype = res
        fn.argtypes = args
    return lib


device = _bind(DEVICE_LIB, DEVICE_SYMBOLS)
host = _bind(HOST_LIB, HOST_SYMBOLS)
if device.rt_abi_version() != ABI_VERSION:
    raise ImportError(f"{DEVICE_LIB} has ABI version {device.rt_abi_version()}, these bindings are for {ABI_VERSION}: rebuild it")


class RtError(RuntimeError):
    def __init__(self, code, where):
        msg = device.rt_last_error()
        super().__init__(f"{where}: error {code}: {msg.decode() if msg else ''}")
        self.code = code


def check(code, where):
    if code != 0:
        raise RtError(code, where)
