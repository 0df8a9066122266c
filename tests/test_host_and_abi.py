"""Host C code (scene builders, camera, job arithmetic) against the oracle, and
the C-ABI surface: the library loads and exports every symbol include/rt_amd.h
declares.  No GPU compute is called here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported(rt):
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)                      # declarations only, not comments
    declared = set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text))
    assert len(declared) >= 15
    lib = C.CDLL(rt._native.DEVICE_LIB)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared <= set(rt._native.DEVICE_SYMBOLS), declared - set(rt._native.DEVICE_SYMBOLS)
    assert rt._native.device.rt_abi_version() == 2
    m = re.search(r"#define RT_AMD_ABI_VERSION (\d+)", text)
    assert m and int(m.group(1)) == 2
    hdr = open(os.path.join(ROOT, "raytrace_amd", "host", "rt_host.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    host_declared = set(re.findall(r"\b(rth_[a-z_0-9]+)\s*\(", hdr))
    hlib = C.CDLL(rt._native.HOST_LIB)
    for name in host_declared:
        assert hasattr(hlib, name), name


def test_struct_layouts(rt):
    assert C.sizeof(rt._native.RtCamera) == 88 and rt.SPHERE_DTYPE.itemsize == 36
    assert C.sizeof(rt._native.RtJob) == 52 and C.sizeof(rt._native.RtCounters) == 144
    assert rt._native.STATIC_MAX_PIXELS * 36 + 512 <= rt._native.STATIC_SCRATCH_BYTES


def test_host_inputs_match_oracle(rt, oracle):
    for nx, ny in ((600, 300), (1920, 1080), (3840, 2160), (7, 5)):
        assert np.array_equal(rt.camera_default(nx, ny).as_array().view(np.uint32), oracle.camera_default(nx, ny).view(np.uint32))
    assert rt.small_scene().tobytes() == oracle.scene_small().tobytes()
    assert rt.trivial_scene().tobytes() == oracle.scene_trivial().tobytes()
    assert rt.random_scene().tobytes() == oracle.scene_random().tobytes()
    assert rt.glass_scene().tobytes() == oracle.scene_glass().tobytes()
    dense = rt.random_scene(100)
    assert len(dense) == 40004 and dense.tobytes() == oracle.scene_random(100).tobytes()
    cam = rt.camera_new((3, 4, 5), (0, 1, 0), (0, 1, 0), 35.0, 1.5, 0.1, 7.0)
    assert np.array_equal(cam.as_array(), oracle.camera_make((3, 4, 5), (0, 1, 0), (0, 1, 0), 35.0, 1.5, 0.1, 7.0))


def test_host_generator_matches_golden(rt, kat):
    g = rt._native.RthRng()
    rt._native.host.rth_rng_init(C.byref(g))
    assert [float(rt._native.host.rth_rng_next(C.byref(g))).hex() for _ in range(8)] == kat["my_rand_first8_hex"]


def test_job_rows_and_interleave(rt):
    import raytrace_amd as R
    assert R.job_rows(rt.Job(10, 100, 1)) == 100
    assert R.job_rows(rt.Job(10, 100, 1, row0=90, tile_rows=20)) == 10       # clipped
    for ny, tile, world in ((1080, 8, 8), (300, 8, 3), (37, 5, 4), (5, 8, 8)):
        seen = []
        for r in range(world):
            j = R.interleaved_job(16, ny, 1, r, world, tile)
            idx = R.job_row_indices(j)
            assert len(idx) == (R.job_rows(j) if j.n_tiles else 0)
            seen.extend(idx.tolist())
        assert sorted(seen) == list(range(ny))                                # every row exactly once


def test_bad_arguments_are_errors_not_aborts(rt):
    d = rt._native.device
    assert d.rt_device_count(None) == 1                                       # RT_ERR_ARG
    assert b"null" in d.rt_last_error()
    assert d.rt_scene_upload(None, None, 0) == 1
    assert d.rt_render_host(None, None, None, None, None, None, None) == 1
    assert d.rt_ctx_destroy(None) == 0


def test_scene_dump_round_trip_and_rejections(rt, tmp_path):
    """The scene dump format: what is saved is what is loaded, byte for byte (sphere + material of c/structs.c:16-36);
    truncated, over-long, foreign and bad-material files are errors."""
    import struct
    for sc in (rt.small_scene(), rt.random_scene(), rt.glass_scene()):
        path = str(tmp_path / f"scene_{len(sc)}.bin")
        rt.save_scene(path, sc)
        raw = open(path, "rb").read()
        assert raw[:8] == b"RTSCENE1" and struct.unpack("<II", raw[8:16]) == (36, len(sc)) and len(raw) == 16 + 36 * len(sc)
        assert raw[16:] == sc.tobytes()
        assert rt.load_scene(path).tobytes() == sc.tobytes()
    good = open(path, "rb").read()
    bad_type = bytearray(good)
    bad_type[16 + 16:16 + 20] = struct.pack("<i", 7)                     # first record's material type
    for name, blob in (("short", good[:-5]), ("long", good + b"x"), ("magic", b"P6\n" + good[3:]), ("empty", b""),
                       ("record", good[:8] + struct.pack("<II", 32, 5) + good[16:]), ("type", bytes(bad_type))):
        p = str(tmp_path / f"bad_{name}.bin")
        open(p, "wb").write(blob)
        with pytest.raises(ValueError):
            rt.load_scene(p)
    with pytest.raises(ValueError):
        rt.load_scene(str(tmp_path / "missing.bin"))


def test_png_writer_produces_a_decodable_image(rt, tmp_path):
    """rth_write_png (c/main.c:427-433 writes PNG through a third-party encoder; this one is the build's own):
    signature, IHDR, stored-deflate IDAT that zlib inflates to the filter-0 rows, chunk CRCs."""
    import ctypes as C
    import struct
    import zlib
    rng = np.random.default_rng(2)
    for w, h in ((1, 1), (37, 5), (300, 200)):   # the last one needs more than one 65535-byte stored block
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        path = str(tmp_path / f"t_{w}x{h}.png")
        libc = C.CDLL(None)
        libc.fopen.restype = C.c_void_p
        libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        libc.fclose.argtypes = [C.c_void_p]
        fp = libc.fopen(path.encode(), b"wb")
        assert fp
        assert rt._native.host.rth_write_png(fp, img.ctypes.data, w, h) == 0
        libc.fclose(fp)
        raw = open(path, "rb").read()
        assert raw[:8] == b"\x89PNG\r\n\x1a\n"
        pos, chunks = 8, []
        while pos < len(raw):
            (n,) = struct.unpack(">I", raw[pos:pos + 4])
            kind, body = raw[pos + 4:pos + 8], raw[pos + 8:pos + 8 + n]
            (crc,) = struct.unpack(">I", raw[pos + 8 + n:pos + 12 + n])
            assert crc == zlib.crc32(kind + body) & 0xFFFFFFFF
            chunks.append((kind, body))
            pos += 12 + n
        assert [k for k, _ in chunks] == [b"IHDR", b"IDAT", b"IEND"]
        assert struct.unpack(">IIBBBBB", chunks[0][1]) == (w, h, 8, 2, 0, 0, 0)
        rows = np.frombuffer(zlib.decompress(chunks[1][1]), np.uint8).reshape(h, 3 * w + 1)
        assert (rows[:, 0] == 0).all()
        assert np.array_equal(rows[:, 1:].reshape(h, w, 3), img)


def _fopen(path):
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]
    fp = libc.fopen(path.encode(), b"wb")
    assert fp
    return libc, fp


def test_jpeg_writer_is_decoded_by_an_independent_decoder(rt, tmp_path):
    """rth_write_jpg (c/main.c:430-435 writes JPG at quality 90 through a third-party encoder; this one is the
    build's own): Pillow decodes the file to the source's size and, at quality 90, to within 36 dB of a smooth
    picture with detail in it; a flat picture comes back flat; quality orders the file sizes; the Huffman
    tables written are complete prefix codes."""
    from PIL import Image
    yy, xx = np.mgrid[0:200, 0:300]
    rng = np.random.default_rng(4)
    smooth = np.stack([127 + 120 * np.sin(xx / 23.0) * np.cos(yy / 31.0), xx * 255.0 / 299, 255 - yy * 255.0 / 199], -1)
    smooth[60:140, 100:200] += rng.normal(0, 12, (80, 100, 3))                 # some texture
    pictures = {"smooth": np.clip(smooth, 0, 255).astype(np.uint8), "flat": np.full((5, 37, 3), (200, 30, 90), np.uint8),
                "one": np.array([[[255, 255, 255]]], np.uint8), "edge": rng.integers(0, 256, (9, 17, 3), dtype=np.uint8)}
    sizes = {}
    for name, img in pictures.items():
        h, w = img.shape[:2]
        for q in (90, 50) if name == "smooth" else (90,):
            path = str(tmp_path / f"{name}_{q}.jpg")
            libc, fp = _fopen(path)
            assert rt._native.host.rth_write_jpg(fp, np.ascontiguousarray(img).ctypes.data, w, h, q) == 0
            libc.fclose(fp)
            raw = open(path, "rb").read()
            assert raw[:4] == b"\xff\xd8\xff\xe0" and raw[6:11] == b"JFIF\x00" and raw[-2:] == b"\xff\xd9"
            with Image.open(path) as im:
                assert im.format == "JPEG" and im.size == (w, h) and im.mode == "RGB"
                back = np.asarray(im.convert("RGB"), np.float64)
            err = np.mean((back - img.astype(np.float64)) ** 2)
            psnr = 99.0 if err == 0 else 10 * np.log10(255.0 ** 2 / err)
            sizes[(name, q)] = len(raw)
            if name == "smooth":
                assert psnr > (36.0 if q == 90 else 30.0), (q, psnr)
            elif name in ("flat", "one"):
                assert np.abs(back - img).max() <= 2, name
            else:
                assert psnr > 20.0                                                # noise at 4:4:4, quality 90
    assert sizes[("smooth", 50)] < sizes[("smooth", 90)] < 3 * 200 * 300
    # the tables as written: every DHT segment is a complete code (Kraft sum 1 less the reserved all-ones code)
    raw = open(str(tmp_path / "smooth_90.jpg"), "rb").read()
    pos, seen = 2, 0
    while raw[pos + 1] != 0xDA:
        n = (raw[pos + 2] << 8) | raw[pos + 3]
        if raw[pos + 1] == 0xC4:
            bits = raw[pos + 5:pos + 21]
            vals = raw[pos + 21:pos + 2 + n]
            assert len(vals) == sum(bits) == len(set(vals))
            assert sum(b * 2.0 ** -(i + 1) for i, b in enumerate(bits)) <= 1.0 - 2.0 ** -16
            seen += 1
        pos += 2 + n
    assert seen == 4


def test_jpeg_writer_rejects_bad_arguments(rt):
    assert rt._native.host.rth_write_jpg(None, None, 1, 1, 90) == 1


def test_host_code_under_address_and_undefined_sanitizers(tmp_path):
    """The host library's CPU-only code built with -fsanitize=address,undefined and run through its builders, writers,
    scene dumps and argument errors (tests/support/host_sanitize.c)."""
    import subprocess
    host = os.path.join(ROOT, "raytrace_amd", "host")
    exe = str(tmp_path / "host_sanitize")
    srcs = [os.path.join(ROOT, "tests", "support", "host_sanitize.c")] + [os.path.join(host, f) for f in ("scene.c", "driver.c", "service.c", "jpeg.c")]
    libdir = os.path.join(ROOT, "raytrace_amd", "lib")
    cmd = ["gcc", "-O1", "-g", "-std=gnu11", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
           "-I", host, "-I", os.path.join(ROOT, "include"), "-o", exe, *srcs, "-L", libdir, "-lraytrace_amd", f"-Wl,-rpath,{libdir}", "-lm", "-lpthread"]
    p = subprocess.run(cmd, capture_output=True, timeout=300)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    work = tmp_path / "work"
    work.mkdir()
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0", UBSAN_OPTIONS="print_stacktrace=1")
    p = subprocess.run([exe, str(work)], capture_output=True, timeout=300, env=env)
    assert p.returncode == 0 and b"host sanitize ok" in p.stdout, (p.stdout + p.stderr).decode()[-3000:]
