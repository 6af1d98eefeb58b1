"""Out-of-process probe of the peer-to-peer exchange route: `python -m elegantrl_amd.p2p_probe`.

`parallel.gradient_comm` validates the one-shot peer-to-peer exchange (csrc/p2p.hip, csrc/grad_tail.hip) by a self-test before it
may carry a run -- but a self-test that FAULTS (an IPC mapping that does not work on this machine's fabric ends in a GPU memory
access fault, which kills the process) cannot report failure from inside the training process.  So the first contact with the
route happens here, in a throw-away child process per rank: its own gloo process group (address / port handed over by the
parent), its own stages, a few exchanges checked against gloo's sums.  Exit code 0 = the route works on this machine; anything
else (non-zero exit, a signal, a timeout in the parent) = every rank keeps RCCL.  Nothing of the child survives.
"""
from __future__ import annotations

import os
import sys


def main() -> int:
    import torch as th
    import torch.distributed as dist

    if os.environ.get("ERL_P2P_PROBE_FAIL"):      # fault injection (tests): the child dies the way a faulting peer mapping kills it -- before
        os._exit(int(os.environ["ERL_P2P_PROBE_FAIL"]) or 7)   # any collective, without cleaning up; every rank must fall back
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", rank)) % max(1, th.cuda.device_count())
    count = int(os.environ.get("ERL_P2P_PROBE_COUNT", "50848"))
    th.cuda.set_device(local)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from . import parallel
    comm = parallel.P2PComm.create(max_count=max(count, 1 << 16))
    ok = comm is not None
    if ok:
        res = parallel.selftest(comm, count, rounds=4, timed_calls=8)
        ok = bool(res["ok"])
        comm.close()
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 3


if __name__ == "__main__":
    sys.exit(main())
