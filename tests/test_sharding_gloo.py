"""CPU, world_size = 2, gloo: the N > 1 path (partition + one all-gather of defect slabs) is correct by
construction.  The propagation callable is injected by the test (the oracle stands in for the GPU sweep HERE
ONLY; the product's callable is lowthrustopt_amd.sharding.hip_indirect_defect)."""
import os
import socket

import numpy as np
import pytest

from lowthrustopt_amd import sharding, synth
from lowthrustopt_amd.constants import MU, DU, TU


def test_partition_covers_everything():
    for n in (0, 1, 7, 29, 4096, 4097):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                s, c = sharding.partition(n, world, r)
                seen += list(range(s, s + c))
            assert seen == list(range(n))
            counts = [sharding.partition(n, world, r)[1] for r in range(world)]
            assert max(counts) - min(counts) <= 1


def test_local_nodes_halo():
    XC, T = synth.indirect_problem(30)
    XC, t = XC[:, :, 0], T[:, 0]
    ln, lt, s, c = sharding.local_nodes(XC, t, 2, 1)
    assert (s, c) == (15, 14) and ln.shape == (12, 15) and np.array_equal(ln[:, 0], XC[:, 15]) and lt[-1] == t[-1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_nodes, q):
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    XC, T = synth.indirect_problem(n_nodes, seed=5)
    XC, t = XC[:, :, 0], T[:, 0]
    prm = [MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0]

    def sweep(ln, lt):
        d, _, rc = O.indirect_defect(ln, lt, prm, O.RK4, 8)
        assert rc == 0
        return d
    full = sharding.sharded_defect(sweep, XC, t, world, rank)
    q.put((rank, full.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_nodes", [30, 8, 3])
def test_world2_allgather_equals_single_process(oracle, n_nodes):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_nodes, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    XC, T = synth.indirect_problem(n_nodes, seed=5)
    d_ref, _, rc = oracle.indirect_defect(XC[:, :, 0], T[:, 0], [MU, DU, TU, 0.05, 1000.0, 1.0, 1.0, 1.0], oracle.RK4, 8)
    for r in (0, 1):
        assert res[r].shape == (12, n_nodes - 1)
        assert np.array_equal(res[r], d_ref)
