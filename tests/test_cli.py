"""The command-line renderer (raytrace_amd/host/main.c): option handling mirrors
the reference's CLI (c/main.c:53-138); the GPU test checks the PPM it writes."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAY = os.path.join(ROOT, "raytrace_amd", "lib", "ray")


def run(*args):
    return subprocess.run([RAY, *args], capture_output=True, timeout=120)


def test_no_output_file_prints_usage_and_exits_1():
    p = run()
    assert p.returncode == 1 and b"Usage:" in p.stderr                    # c/main.c:126-128


@pytest.mark.parametrize("flag", ["-n", "-t", "-x", "-y", "-d"])
def test_non_positive_numbers_are_rejected(flag):
    for bad in ("0", "-3", "12abc"):
        p = run(flag, bad, "-o", "-")
        assert p.returncode == 1 and b"must be a positive integer" in p.stderr   # c/main.c:100-106


def test_extension_handling(tmp_path):
    assert b"missing file name extension" in run("-o", str(tmp_path / "noext")).stderr     # c/main.c:73-77
    assert b"unsupported file extension" in run("-o", str(tmp_path / "a.gif")).stderr     # c/main.c:84-89


@pytest.mark.gpu
def test_cli_renders_the_bytes_of_the_oracle(tmp_path, rt, oracle):
    out = tmp_path / "o.ppm"
    p = run("-s", "final", "-x", "200", "-y", "100", "-n", "4", "-t", "2", "-o", str(out))
    assert p.returncode == 0, p.stderr
    blob = out.read_bytes()
    assert blob.startswith(b"P6\n200 100 255\n")                          # c/main.c:412
    from conftest import as_oracle_scene
    want = oracle.render_det(rt.camera_default(200, 100).as_array(), as_oracle_scene(rt.random_scene()), 200, 100, 4, seed=0,
                             host_threads=16, want_debug=False)["rgb8"]
    assert blob[len(b"P6\n200 100 255\n"):] == want.tobytes()
    p2 = run("-s", "final", "-x", "200", "-y", "100", "-n", "4", "-o", "-")           # '-' = stdout, c/main.c:68-70
    assert p2.returncode == 0 and hashlib.sha256(p2.stdout).digest() == hashlib.sha256(blob).digest()
    # the same picture as PNG (c/main.c:78-79 picks the writer by extension)
    import struct
    import zlib
    png = tmp_path / "o.png"
    p3 = run("-s", "final", "-x", "200", "-y", "100", "-n", "4", "-o", str(png))
    assert p3.returncode == 0, p3.stderr
    raw = png.read_bytes()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    (n,) = struct.unpack(">I", raw[33:37])
    assert raw[37:41] == b"IDAT"
    rows = np.frombuffer(zlib.decompress(raw[41:41 + n]), np.uint8).reshape(100, 601)
    assert np.array_equal(rows[:, 1:].reshape(100, 200, 3), want)



def test_scene_file_errors_are_reported(tmp_path):
    p = run("-i", str(tmp_path / "missing.bin"), "-o", "-")
    assert p.returncode == 1 and b"could not load" in p.stderr
    bad = tmp_path / "bad.bin"
    bad.write_bytes(b"P6\n1 1 255\n")
    p = run("-i", str(bad), "-o", "-")
    assert p.returncode == 1 and b"not a scene file" in p.stderr


@pytest.mark.gpu
def test_cli_renders_a_scene_dump(tmp_path, rt, oracle):
    """-D writes the scene that was rendered; -i renders such a file: a caller's own sphere list, here the final
    scene with a few spheres moved, against the oracle's picture of the same list."""
    dump = tmp_path / "final.bin"
    p = run("-s", "final", "-x", "64", "-y", "36", "-n", "1", "-D", str(dump), "-o", "-")
    assert p.returncode == 0, p.stderr
    sc = rt.load_scene(str(dump))
    assert sc.tobytes() == rt.random_scene().tobytes()
    sc["center"][5:9, 1] += 0.5
    sc["type"][7] = 2
    sc["param"][7] = 1.5
    mine = tmp_path / "mine.bin"
    rt.save_scene(str(mine), sc)
    p = run("-i", str(mine), "-x", "160", "-y", "90", "-n", "3", "-t", "2", "-o", "-")
    assert p.returncode == 0, p.stderr
    from conftest import as_oracle_scene
    want = oracle.render_det(rt.camera_default(160, 90).as_array(), as_oracle_scene(sc), 160, 90, 3, seed=0, host_threads=16,
                             want_debug=False)["rgb8"]
    assert p.stdout == b"P6\n160 90 255\n" + want.tobytes()


@pytest.mark.gpu
def test_cli_writes_a_jpeg_of_the_oracles_picture(tmp_path, rt, oracle):
    """.jpg / .jpeg pick the JPEG writer at quality 90 (c/main.c:80-81, 434); decoded by Pillow the picture is the
    oracle's to within what quality 90 loses."""
    from PIL import Image
    out = tmp_path / "o.jpeg"
    p = run("-s", "final", "-x", "200", "-y", "100", "-n", "4", "-o", str(out))
    assert p.returncode == 0, p.stderr
    from conftest import as_oracle_scene
    want = oracle.render_det(rt.camera_default(200, 100).as_array(), as_oracle_scene(rt.random_scene()), 200, 100, 4, seed=0,
                             host_threads=16, want_debug=False)["rgb8"]
    with Image.open(str(out)) as im:
        assert im.format == "JPEG" and im.size == (200, 100)
        back = np.asarray(im.convert("RGB"), np.float64)
    psnr = 10 * np.log10(255.0 ** 2 / np.mean((back - want.astype(np.float64)) ** 2))
    assert psnr > 30.0, psnr       # (4 spp: the picture is noisy, which costs a JPEG more than a smooth one)
