"""torch.distributed.run launches of the test suites' worker scripts on 127.0.0.1 (TEST INFRASTRUCTURE).

A free port found by bind(0) can be taken again before the launcher's store binds it (an ephemeral outbound connection, a socket of the
previous launch still closing): EADDRINUSE then says something about the harness, not about the code under test.  Every launch of the
suites goes through torchrun() below, which tries again on a fresh port when -- and only when -- the failure carries such a signature
(round 6: one full run of the GPU suite in about ten stopped on exactly that, in a test that launched the module directly)."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INFRA = ("address already in use", "eaddrinuse", "rendezvous", "connection refused", "connection reset", "failed to listen")


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def torchrun(nproc, script, args=(), env=None, timeout=600, tries=3, cwd=ROOT):
    """-> CompletedProcess of `python -m torch.distributed.run --nnodes=1 --nproc-per-node=<nproc> ... <script> <args>`."""
    r = None
    for _ in range(tries):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), script, *map(str, args)]
        r = subprocess.run(cmd, cwd=cwd, env=env, capture_output=True, text=True, timeout=timeout)
        if r.returncode == 0 or not any(k in (r.stdout + r.stderr).lower() for k in INFRA):
            break
    return r
