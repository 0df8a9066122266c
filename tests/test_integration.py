"""The drop-in, compiled: the render_gpu() binding that the integration guide shows, spliced into the reference translation unit
(oracle/Makefile, target `integration`) gives oracle/_ref/ray_ref_gpu, the reference's own CLI with its render
threads bound to libraytrace_amd.so."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "oracle", "integration_stub.c")


def test_the_binding_compiles_against_the_reference_translation_unit():
    from oracle import oracle as O
    if not os.path.isdir("/root/reference/c"):
        pytest.skip("needs /root/reference to compile against")
    if os.path.exists(O.REF_BIN_GPU):
        os.remove(O.REF_BIN_GPU)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "integration"])
    assert os.path.exists(O.REF_BIN_GPU)
    p = subprocess.run([O.REF_BIN_GPU], capture_output=True)                 # no -o: the reference's usage text
    assert p.returncode == 1 and b"Usage:" in p.stderr


def test_integration_document_shows_the_compiled_text():
    doc = os.path.join(ROOT, "INTEGRATION.md")
    if not os.path.exists(doc):
        pytest.skip("documentation is not shipped to this machine")
    stub = open(STUB).read()
    body = stub[stub.index('#include "rt_amd.h"'):]
    assert body in open(doc).read()


@pytest.mark.gpu
def test_reference_cli_renders_on_the_gpu(oracle):
    """The reference's main(), option parsing, small_scene(), camera_new() and ppm_write() around the GPU
    library: -t 3 makes three row bands (c/main.c:510-529), each rendered by render_gpu on its own context."""
    from oracle import oracle as O
    if not os.path.exists(O.REF_BIN_GPU):
        pytest.skip("oracle/_ref/ray_ref_gpu was not built (needs /root/reference at build time)")
    for n, t in ((4, 3), (2, 1)):
        p = subprocess.run([O.REF_BIN_GPU, "-n", str(n), "-t", str(t), "-o", "-"], capture_output=True, timeout=120)
        assert p.returncode == 0, p.stderr
        head = b"P6\n600 300 255\n"                                          # c/main.c:412, 486
        assert p.stdout.startswith(head) and len(p.stdout) == len(head) + 3 * 600 * 300
        got = np.frombuffer(p.stdout[len(head):], np.uint8).reshape(300, 600, 3)
        want = oracle.render_det(oracle.camera_default(600, 300), oracle.scene_small(), 600, 300, n, seed=0, host_threads=16,
                                 want_debug=False)["rgb8"]
        assert np.array_equal(got, want)
