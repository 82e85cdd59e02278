"""Fail-fast launcher for the multi-rank checks under scripts/ (fresh `spawn` children only: nothing that has
touched the GPU is ever re-exec'd).

    run_ranks(worker, world, budget_s) -> [result of rank 0, result of rank 1, ...]

`worker(rank, world, port)` runs in a fresh process and returns a picklable result.  Every child reports through a
queue: ("ok", rank, result) or ("error", rank, traceback text).  The parent polls the queue with a short timeout and
looks at every child's exit code in between, so that

  * a rank that raises  -> its traceback is printed, the peers are terminated, the parent exits 1 within seconds;
  * a rank that dies without a word (signal, os._exit, the OOM killer) -> its exit code is printed, same;
  * a rank blocked in a collective whose peer is gone never holds the parent: the parent's own budget (which the
    caller keeps below its pytest timeout) ends the run with the list of ranks still alive.

Process-group timeouts inside the workers are kept short (`PG_TIMEOUT`) for the same reason."""
import datetime
import os
import queue as _queue
import socket
import sys
import time
import traceback

import torch.multiprocessing as mp

PG_TIMEOUT = datetime.timedelta(seconds=120)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _child(worker, rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        res = worker(rank, world, port)
        q.put(("ok", rank, res))
    except BaseException:                                   # noqa: BLE001  (report everything, then die non-zero)
        q.put(("error", rank, traceback.format_exc()))
        q.close()
        q.join_thread()                                      # the traceback must reach the pipe before the exit
        os._exit(1)


def _stop(procs):
    for p in procs:
        if p.is_alive():
            p.terminate()
    t_end = time.time() + 10
    for p in procs:
        p.join(max(0.1, t_end - time.time()))
        if p.is_alive():
            p.kill()
            p.join(5)


def run_ranks(worker, world, budget_s):
    port = free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_child, args=(worker, r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results, failure = {}, None
    deadline = time.time() + budget_s
    while len(results) < world and failure is None:
        try:
            kind, rank, payload = q.get(timeout=0.5)
            if kind == "ok":
                results[rank] = payload
            else:
                failure = f"rank {rank} raised:\n{payload}"
            continue
        except _queue.Empty:
            pass
        for r, p in enumerate(procs):
            if p.exitcode not in (None, 0) and r not in results:
                try:                                         # its own report may still be in the pipe
                    kind, rank, payload = q.get(timeout=2.0)
                    failure = f"rank {rank} raised:\n{payload}" if kind == "error" else None
                    if kind == "ok":
                        results[rank] = payload
                except _queue.Empty:
                    pass
                failure = failure or f"rank {r} exited with code {p.exitcode} without reporting (signal / hard exit)"
                break
        if failure is None and time.time() > deadline:
            alive = [r for r, p in enumerate(procs) if p.is_alive()]
            failure = f"budget of {budget_s:.0f} s spent; ranks still running: {alive}; ranks done: {sorted(results)}"
    if failure is not None:
        notes = [failure]                                    # a peer usually fails a moment later (connection reset): report every
        t_end = time.time() + 3.0                            # rank's own words and every exit code, the first cause is among them
        while time.time() < t_end:
            try:
                kind, rank, payload = q.get(timeout=0.3)
                if kind == "error" and f"rank {rank} raised" not in "".join(notes):
                    notes.append(f"rank {rank} raised:\n{payload}")
            except _queue.Empty:
                pass
        codes = {r: p.exitcode for r, p in enumerate(procs)}
        for r, c in codes.items():
            if c not in (None, 0) and not any(f"rank {r} " in n for n in notes):
                notes.append(f"rank {r} exited with code {c} without reporting (signal / hard exit)")
        _stop(procs)
        print("MULTI-RANK CHECK FAILED (exit codes before the stop: %s)" % codes, file=sys.stderr)
        for n in notes:
            print(n, file=sys.stderr)
        sys.stderr.flush()
        sys.exit(1)
    for p in procs:
        p.join(60)
    bad = [(r, p.exitcode) for r, p in enumerate(procs) if p.exitcode != 0]
    if bad:
        _stop(procs)
        print(f"MULTI-RANK CHECK FAILED: ranks reported but exited (rank, code) = {bad}", file=sys.stderr, flush=True)
        sys.exit(1)
    return [results[r] for r in range(world)]
