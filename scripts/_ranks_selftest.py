"""Self-test driver of scripts/_ranks.py (CPU, gloo): `python scripts/_ranks_selftest.py MODE` where MODE is
ok | raise | die | hang.  In the failing modes rank 0 fails right after the rendezvous while rank 1 sits in a collective --
the shape of the failure that hid a traceback for 580 s in round 4."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
from _ranks import PG_TIMEOUT, run_ranks        # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "ok"


def worker(rank, world, port):
    import torch
    import torch.distributed as dist
    mode = os.environ["RANKS_SELFTEST_MODE"]
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=PG_TIMEOUT)
    if rank == 0 and mode == "raise":
        raise RuntimeError("rank 0 fails on purpose")
    if rank == 0 and mode == "die":
        os._exit(9)
    if rank == 0 and mode == "hang":
        time.sleep(3600)
    x = torch.ones(4) * (rank + 1)
    dist.all_reduce(x)
    dist.destroy_process_group()
    return float(x[0])


if __name__ == "__main__":
    os.environ["RANKS_SELFTEST_MODE"] = MODE
    res = run_ranks(worker, 2, float(os.environ.get("RANKS_SELFTEST_BUDGET", "60")))
    print("selftest result:", res)
