"""N > 1 path on CPU: world_size-2 gloo run of the replica aggregation bench.py uses."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mgb_amd.replicas import aggregate, rate
    elapsed = 2.0 + rank          # rank 1 is the slow replica
    its = 100 + 10 * rank
    dist.barrier()
    tmax, itsum = aggregate(elapsed, its, dist, device="cpu")
    out[rank] = (tmax, itsum, rate(tmax, itsum))
    dist.barrier()
    dist.destroy_process_group()


def test_replica_aggregation_world2():
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    assert res[0] == res[1]
    tmax, itsum, r = res[0]
    assert tmax == 3.0 and itsum == 210.0 and abs(r - 70.0) < 1e-12


def test_single_process_aggregation():
    from mgb_amd.replicas import aggregate
    assert aggregate(1.5, 30) == (1.5, 30.0)
