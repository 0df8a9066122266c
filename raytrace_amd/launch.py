"""Starts one process per rank on this node, the way `torch.distributed.run` would, for callers that were not
started by a launcher: every child gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT and
rendezvous on 127.0.0.1.  The parent never touches a GPU (it only spawns and waits), so the children are
fresh processes that initialise HIP themselves.  This is the multi-GPU form of the thread fan-out of
c/main.c:531-545: spawn, join, fail if any rank failed."""
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


KILL_AFTER_S = 5.0     # a rank that has not gone this long after SIGTERM gets SIGKILL


def launch_ranks(world, argv, env_extra=None, timeout=None, kill_after=KILL_AFTER_S):
    """Runs `argv` (a command line, list of str) once per rank.  Rank 0 inherits stdout; every rank inherits
    stderr.  Returns 0, or the exit code of the first rank seen to fail; when a rank fails the others are
    terminated (by pid) and their exit codes ignored.  A rank stuck inside a GPU or collective call does not act on
    SIGTERM: whatever is still alive `kill_after` seconds after the terminate is killed (by pid), so the caller
    always gets its answer."""
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if env_extra:
            env.update(env_extra)
        out = None if rank == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen(list(argv), env=env, stdout=out))
    t0 = time.monotonic()
    failed = 0
    terminated_at = None
    killed = False
    live = set(range(world))
    while live:
        for r in sorted(live):
            rc = procs[r].poll()
            if rc is None:
                continue
            live.discard(r)
            if rc != 0 and failed == 0:
                failed = rc if rc > 0 else 128 - rc
                for q in live:                      # a rank failed: the others would wait for it for ever
                    procs[q].terminate()
                terminated_at = time.monotonic()
        if live and failed == 0 and timeout is not None and time.monotonic() - t0 > timeout:
            failed = 124
            for q in live:
                procs[q].terminate()
            terminated_at = time.monotonic()
        if live and terminated_at is not None and not killed and time.monotonic() - terminated_at > kill_after:
            for q in live:
                procs[q].kill()
            killed = True
        if live:
            time.sleep(0.05)
    return failed


if __name__ == "__main__":   # python raytrace_amd/launch.py N cmd...
    sys.exit(launch_ranks(int(sys.argv[1]), sys.argv[2:]))
