"""The multi-rank control flow on the one GPU a test box has: bench.py's own launcher with two ranks sharing GPU 0
(tiles gathered over gloo on host copies), and the RCCL calls of the real path on a one-rank group."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_bench_starts_two_ranks_itself_and_reports_per_rank():
    """`python3 bench.py --gpus 2` with no launcher: the parent spawns both ranks, rank 0 prints the one line.
    Strong scaling on the 4K frame (two samples per pixel here), a weak-scaling leg, per-rank kernel times."""
    env = dict(os.environ, RT_BENCH_REHEARSAL="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "2",
                        "--no-cpu-baseline"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=170)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 2 and out["value"] > 0
    assert out["config"]["workload"].startswith("final4k") and out["config"]["paths_per_step"] == 3840 * 2160 * 2
    assert out["config"]["ranks_started_by"] == "bench.py itself"
    assert sum(out["per_rank"]["rows"]) == 2160 and len(out["per_rank"]["kernel_ms_serial_leg"]) == 2
    assert out["weak_scaling"]["value"] > 0
    assert out["roofline"]["sphere_tests_per_launch"] > 0


def test_rccl_calls_of_the_multi_gpu_path_on_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "support", "rccl_single_rank.py"), str(port)], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=170)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-2000:]
    assert "rccl single rank ok" in p.stdout.decode()
