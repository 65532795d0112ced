"""One process per GPU, started from a parent that never touches the GPU.

What R:train.py:171-234 (``launch``: world size 1 runs in place; otherwise a free port on 127.0.0.1 and one worker per device
through ``mp.spawn``, every worker joined, the others terminated when one dies) and :114-168 (``_distributed_worker``:
``init_process_group("nccl")`` from rank / world size / URL) do for the reference's trainer, restated for this path's
one-rank-per-GPU layout:

  * the parent imports neither torch nor the HIP library -- a process that has initialised the GPU must not spawn ranks -- and
    starts N FRESH interpreters (``subprocess``; nothing is forked or re-exec'ed), each with the torchrun environment contract
    (RANK, LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT = a free port) that
    ``lsenerf_amd.dist.init_from_env`` reads;
  * rank 0's stdout is relayed line by line (the one JSON line of bench.py), every rank's stderr goes to the parent's stderr;
  * the ranks live in their own process groups; when one exits non-zero (or the parent is interrupted, or ``timeout`` runs out)
    exactly those groups are terminated -- by the pids this module started, never by a pattern -- and the parent exits non-zero
    with the failing rank's code.

``bench.py --gpus N`` uses it when it is started without a launcher (no WORLD_SIZE in the environment); started under
``python -m torch.distributed.run`` it finds WORLD_SIZE set and is a rank itself.
"""
from __future__ import annotations

import os
import signal
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence

RANK_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")


def find_free_port() -> int:
    """A free TCP port on 127.0.0.1 (R:train.py ``_find_free_port``)."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def needs_launch(n_ranks: int, environ: Optional[Dict[str, str]] = None) -> bool:
    """True when this process was asked for ``n_ranks`` > 1 ranks but is not a rank itself (no launcher set WORLD_SIZE)."""
    environ = os.environ if environ is None else environ
    return n_ranks > 1 and "WORLD_SIZE" not in environ


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    # the host driver of this pool only supports dmabuf IPC; without it RCCL fails with hipIpcGetMemHandle: invalid argument
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _kill_group(p: subprocess.Popen, sig: int) -> None:
    try:
        os.killpg(p.pid, sig)          # start_new_session=True: the child's pid is its process-group id
    except (ProcessLookupError, PermissionError):
        pass


def launch_ranks(argv: Sequence[str], n_ranks: int, timeout: Optional[float] = None, port: Optional[int] = None,
                 env: Optional[Dict[str, str]] = None, stdout=None, stderr=None, poll: float = 0.05, stdout_filter=None) -> int:
    """Start ``n_ranks`` children running ``argv`` (e.g. ``[sys.executable, "bench.py", "--gpus", "8", ...]``), wait for all of
    them and return the exit code: 0 when every rank exited 0, otherwise the first failing rank's code (124 on timeout) after
    the remaining ranks have been terminated.  Rank 0's stdout is copied to ``stdout`` (default: this process's);
    ``stdout_filter(line) -> bool`` keeps the result stream clean: lines of rank 0 it rejects (library chatter such as gloo's
    "[Gloo] Rank 0 is connected ...") go to ``stderr`` instead."""
    assert n_ranks >= 1
    stdout = sys.stdout if stdout is None else stdout
    stderr = sys.stderr if stderr is None else stderr
    port = find_free_port() if port is None else port
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []

    def pump(src, dst, prefix="", keep=None):
        for line in iter(src.readline, ""):
            if keep is not None and not keep(line):
                stderr.write("[rank 0] " + line)
                stderr.flush()
                continue
            dst.write(prefix + line)
            dst.flush()
        src.close()

    # SIGTERM to the parent (a job limit, a driver that gives up) must take the ranks along: they lead their own sessions
    def _on_term(signum, frame):
        raise KeyboardInterrupt
    old_term = None
    if threading.current_thread() is threading.main_thread():
        old_term = signal.signal(signal.SIGTERM, _on_term)
    try:
        for r in range(n_ranks):
            p = subprocess.Popen(list(argv), env=rank_env(r, n_ranks, port, env), stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                                 text=True, bufsize=1, start_new_session=True)
            procs.append(p)
            # only rank 0 prints the result line; what other ranks write to stdout would corrupt it, so it goes to stderr, tagged
            pumps.append(threading.Thread(target=pump, args=(p.stdout, stdout if r == 0 else stderr, "" if r == 0 else f"[rank {r}] ",
                                                             stdout_filter if r == 0 else None), daemon=True))
            pumps.append(threading.Thread(target=pump, args=(p.stderr, stderr, f"[rank {r}] " if n_ranks > 1 else ""), daemon=True))
        for t in pumps:
            t.start()
        t0 = time.monotonic()
        rc = 0
        alive = set(range(n_ranks))
        while alive and rc == 0:
            for r in sorted(alive):
                code = procs[r].poll()
                if code is None:
                    continue
                alive.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code        # killed by signal s: 128 + s, as a shell reports it
                    stderr.write(f"[launch] rank {r} exited with code {code}; terminating the other {len(alive)} rank(s)\n")
                    break
            if rc == 0 and alive:
                if timeout is not None and time.monotonic() - t0 > timeout:
                    rc = 124
                    stderr.write(f"[launch] timeout after {timeout:.0f} s; terminating {len(alive)} rank(s)\n")
                    break
                time.sleep(poll)
        return rc
    except KeyboardInterrupt:
        stderr.write("[launch] interrupted; terminating the ranks\n")
        return 130
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:
            _kill_group(p, signal.SIGTERM)
        deadline = time.monotonic() + 10.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                _kill_group(p, signal.SIGKILL)
                p.wait()
        for t in pumps:
            t.join(timeout=5.0)
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
