"""Bounded multi-process helpers: no test may wait for a rendezvous or a child longer than its deadline.

`spawn_bounded` replaces `mp.spawn(..., join=True)` (which waits forever for a child stuck in a rendezvous): the children are
started with join=False, polled until a deadline, then killed, and the test fails with whatever the children's `faulthandler`
wrote.  `run_bounded` is `subprocess.run` in its own process group, so that a launcher's grandchildren die with it."""
import faulthandler
import os
import signal
import socket
import subprocess
import sys
import tempfile
import time

import pytest

LOOPBACK_ENV = {"GLOO_SOCKET_IFNAME": "lo", "NCCL_SOCKET_IFNAME": "lo", "FIBER_DIST_TIMEOUT": "120"}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _entry(rank, fn, args, dump_dir, dump_after_s):
    # child: its own stack dump shortly before the parent's deadline, so a hang is attributable
    f = open(os.path.join(dump_dir, f"rank{rank}.txt"), "w")
    faulthandler.enable(file=f)
    faulthandler.dump_traceback_later(dump_after_s, exit=False, file=f)
    os.environ.update(LOOPBACK_ENV)
    try:
        fn(rank, *args)
    finally:
        faulthandler.cancel_dump_traceback_later()


def spawn_bounded(fn, args, nprocs=2, deadline_s=180, fail=True):
    """Run fn(rank, *args) in `nprocs` spawned processes; never hang: when they are not done after deadline_s they are killed and the
    test fails with their stack dumps (fail=False: returns False instead, for one retry on a fresh port).  Returns True when they finished."""
    import torch.multiprocessing as mp
    dump_dir = tempfile.mkdtemp(prefix="fiber_mp_")
    ctx = mp.spawn(_entry, args=(fn, args, dump_dir, max(deadline_s - 15, 5)), nprocs=nprocs, join=False)
    t_end = time.time() + deadline_s
    try:
        while time.time() < t_end:
            if ctx.join(timeout=5):          # True once every child has exited cleanly; raises if one failed
                return True
    except Exception:
        _kill(ctx)
        raise
    _kill(ctx)
    dumps = []
    for r in range(nprocs):
        try:
            dumps.append(f"--- rank {r} ---\n" + open(os.path.join(dump_dir, f"rank{r}.txt")).read()[-3000:])
        except OSError:
            pass
    if not fail:
        print(f"{fn.__name__}: {nprocs} ranks not finished after {deadline_s} s (killed); dumps:\n" + "\n".join(dumps), flush=True)
        return False
    pytest.fail(f"{fn.__name__}: {nprocs} ranks not finished after {deadline_s} s (killed)\n" + "\n".join(dumps), pytrace=False)


def _kill(ctx):
    for p in ctx.processes:
        if p.is_alive():
            p.kill()
    for p in ctx.processes:
        p.join(10)


def run_bounded(cmd, timeout, **kw):
    """subprocess.run(capture_output=True, text=True) in a new session; on timeout the whole process group is killed."""
    env = dict(kw.pop("env", None) or os.environ)
    for k, v in LOOPBACK_ENV.items():
        env.setdefault(k, v)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True, env=env, **kw)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = proc.communicate()
        pytest.fail(f"{' '.join(map(str, cmd[:6]))} ...: not finished after {timeout} s (process group killed)\n"
                    f"stdout tail: {out[-1500:]}\nstderr tail: {err[-1500:]}", pytrace=False)
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


def spawn_with_retry(fn, make_args, nprocs=2, deadline_s=150):
    """spawn_bounded with ONE retry on a fresh rendezvous port when the first attempt runs into its deadline (a rendezvous that never
    completes is an accident of the box, not of the code under test; a second one is reported).  make_args(port) -> the worker's arguments."""
    if spawn_bounded(fn, make_args(free_port()), nprocs=nprocs, deadline_s=deadline_s, fail=False):
        return
    spawn_bounded(fn, make_args(free_port()), nprocs=nprocs, deadline_s=deadline_s, fail=True)

