"""Builds the native libraries in-tree (they travel to the GPU box as built):

  raytrace_amd/lib/libraytrace_amd.so   HIP kernels + C ABI (hipcc, gfx950)
  raytrace_amd/lib/libraytrace_host.so  host C: scenes, camera, driver (gcc)
  raytrace_amd/lib/ray                  command-line renderer (gcc)

Run as `python raytrace_amd/build.py` (a plain script: the package itself cannot be
imported before the libraries exist).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "lib")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: plain a*b+c must stay two roundings (see csrc/rt_core.h)
# max-ilp: the scheduler variant that favours instruction-level parallelism over register pressure; the trace kernel
# is pinned to its register budget by __launch_bounds__ either way, and measures 0.9 % faster with it
HIP_FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-std=c++17", "-fPIC", "-shared",
             "-Wall", "-Wno-unused-function"]
C_FLAGS = ["-O2", "-std=gnu11", "-ffp-contract=off", "-fPIC", "-Wall", "-Wextra"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_device(force=False):
    src = os.path.join(PKG, "csrc", "rt_device.hip")
    deps = [src, os.path.join(PKG, "csrc", "rt_core.h"), os.path.join(PKG, "csrc", "rt_cluster.h"),
            os.path.join(ROOT, "include", "rt_amd.h")]
    out = os.path.join(LIB, "libraytrace_amd.so")
    if force or _newer(out, deps):
        os.makedirs(LIB, exist_ok=True)
        _run([HIPCC] + HIP_FLAGS + ["-o", out, src])
    return out


def build_host(force=False):
    hdir = os.path.join(PKG, "host")
    srcs = [os.path.join(hdir, f) for f in ("scene.c", "driver.c", "service.c", "jpeg.c")]
    deps = srcs + [os.path.join(hdir, "rt_host.h"), os.path.join(ROOT, "include", "rt_amd.h")]
    out = os.path.join(LIB, "libraytrace_host.so")
    dev = build_device(force)
    if force or _newer(out, deps + [dev]):
        _run(["gcc"] + C_FLAGS + ["-shared", "-o", out] + srcs +
             ["-L" + LIB, "-lraytrace_amd", "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
    cli_src = os.path.join(hdir, "main.c")
    cli = os.path.join(LIB, "ray")
    if os.path.exists(cli_src) and (force or _newer(cli, [cli_src] + deps + [out])):
        _run(["gcc"] + C_FLAGS + ["-o", cli, cli_src, "-L" + LIB, "-lraytrace_host", "-lraytrace_amd",
                                  "-Wl,-rpath,$ORIGIN", "-lm", "-lpthread"])
    return out


def build_all(force=False):
    build_device(force)
    build_host(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
