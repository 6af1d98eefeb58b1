"""The out-of-process probe of the peer-to-peer route (elegantrl_amd/p2p_probe.py, parallel.probe_p2p_out_of_process): a child
that dies -- here: no GPU at all, the same exit path as a faulting peer mapping -- is reported as a reason on EVERY rank and
never raises in the parent.  Two gloo ranks on the CPU."""
import os
import socket

import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch as th
    import torch.distributed as dist
    from elegantrl_amd import parallel
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    th.cuda.current_device = lambda: 0                       # the parent only stringifies it for the child's LOCAL_RANK
    why = parallel.probe_p2p_out_of_process(1000, timeout_s=120.0)
    open(os.path.join(out_dir, f"why_{rank}.txt"), "w").write(why)
    dist.barrier()
    dist.destroy_process_group()


def test_probe_failure_is_a_reason_not_a_crash(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    whys = [open(tmp_path / f"why_{r}.txt").read() for r in range(world)]
    assert all(w != "ok" for w in whys), whys                # no GPU here: the child cannot come up
    assert any("probe process exited" in w or "another rank" in w for w in whys), whys
