"""world_size-2 gloo test of the multi-GPU decomposition on CPU: every rank renders the (pass, block) pairs with
block_id % world == rank, one reduce(SUM) of the XYZAW film merges them.  The renders run on
the oracle here (no GPU); the sharding rule and the reduce are the ones bench.py uses with RCCL."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outfile, spp=8, spp_pass=4):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import tests.oracle_binding as ob
    scenes = importlib.import_module("eradiate-kernel_amd.scenes")
    d = scenes.c3_heterogeneous(64, 48, spp, res=16, samples_per_pass=spp_pass)          # spp / spp_pass passes x 4 blocks
    o = ob.OracleScene(d)
    film = torch.from_numpy(o.render(threads=1, shard_index=rank, shard_count=world))
    samples = torch.tensor([o.last_stats["samples"]], dtype=torch.int64)
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    dist.reduce(samples, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        full = ob.OracleScene(d).render(threads=1)
        np.savez(outfile, merged=film.numpy(), full=full, samples=samples.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_film_reduce(tmp_path):
    out = str(tmp_path / "r.npz")
    mp.spawn(_worker, args=(2, 29517, out), nprocs=2, join=True)
    z = np.load(out)
    assert z["samples"][0] == 64 * 48 * 8
    assert np.all(z["merged"][..., 4] == 8)                         # every pixel got all its samples exactly once
    assert np.allclose(z["merged"], z["full"], rtol=1e-6, atol=0)    # box filter: disjoint tiles, sum is exact up to pass order


def test_strong_scaling_partition(tmp_path):
    """bench.py --gpus N (strong scaling): the fixed job in N passes of spp / N (samples_per_pass, integrator.cpp:58-65), the
    (pass, block) pairs dealt block_id % N.  With 4 blocks and N = 2 every rank gets 4 of the 8 pairs -- as many as one
    rank renders of the one-pass job -- and the merged film equals the one-rank film of the same samples_per_pass."""
    out = str(tmp_path / "s.npz")
    mp.spawn(_worker, args=(2, 29519, out, 16, 8), nprocs=2, join=True)
    z = np.load(out)
    assert z["samples"][0] == 64 * 48 * 16
    assert np.all(z["merged"][..., 4] == 16)
    assert np.allclose(z["merged"], z["full"], rtol=1e-6, atol=0)
