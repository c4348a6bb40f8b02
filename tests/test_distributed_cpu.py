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


# ---------------------------------------------------------------------------------------------------------------------------------
# Round 4: `python bench.py --gpus N` as the driver calls it, and N = 8 without hardware.
def test_partition_arithmetic_of_the_baseline_configurations():
    """bench.py's strong-scaling cut for N in {1, 2, 4, 8} x {C3, C4, C5}: N passes of spp / N, every rank the same number of
    (pass, block) pairs, at least one workgroup per CU (256) per rank and launch, every sample accounted for."""
    sys.path.insert(0, ROOT)
    import bench
    for cfg in ("C3", "C4", "C5"):
        w, h, spp = bench.CONFIG_SIZES[cfg]
        for n in (1, 2, 4, 8):
            p = bench.partition(w, h, spp, n)
            assert p["spp_pass"] * p["passes"] == spp and p["passes"] == n
            assert p["blocks_total"] == (w // 32) * (h // 32) * n
            assert len(set(p["workgroups_per_rank"])) == 1 and sum(p["workgroups_per_rank"]) == p["blocks_total"]
            assert min(p["workgroups_per_rank"]) >= 256
    # a sample count N does not divide: the largest divisor of spp not above spp / N
    p = bench.partition(512, 512, 1000, 3)
    assert p["spp_pass"] == 250 and p["passes"] == 4 and sum(p["workgroups_per_rank"]) == 256 * 4
    assert max(p["workgroups_per_rank"]) - min(p["workgroups_per_rank"]) <= 1


def _worker8(rank, world, port, outfile):
    """The C3 partition at world size 8 on the oracle: 8 passes x 4 blocks of a 64 x 64 miniature, (pass, block) pairs dealt
    block_id % 8 -- bench.py's rule (spiral.cpp:27-72 ids, integrator.cpp:58-65 passes)."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import tests.oracle_binding as ob
    scenes = importlib.import_module("eradiate-kernel_amd.scenes")
    part = bench.partition(64, 64, 16, world)
    d = scenes.c3_heterogeneous(64, 64, 16, res=16, samples_per_pass=part["spp_pass"])
    o = ob.OracleScene(d)
    film = torch.from_numpy(o.render(threads=1, shard_index=rank, shard_count=world))
    mine = torch.tensor([o.last_stats["samples"]], dtype=torch.int64)
    counts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(counts, mine)
    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)
    if rank == 0:
        full = ob.OracleScene(d).render(threads=1)
        np.savez(outfile, merged=film.numpy(), full=full, counts=np.array([int(c.item()) for c in counts]),
                 per_rank=np.array(part["workgroups_per_rank"]), passes=part["passes"])
    dist.barrier()
    dist.destroy_process_group()


def test_eight_rank_partition_of_the_metric_job(tmp_path):
    out = str(tmp_path / "e.npz")
    mp.spawn(_worker8, args=(8, 29523, out), nprocs=8, join=True)
    z = np.load(out)
    assert int(z["passes"]) == 8 and (z["per_rank"] == 4).all()               # 4 blocks x 8 passes over 8 ranks
    assert (z["counts"] == 32 * 32 * 2 * 4).all()                             # every rank renders the same number of samples
    assert np.all(z["merged"][..., 4] == 16)                                  # every pixel got all its samples exactly once
    assert np.allclose(z["merged"], z["full"], rtol=1e-6, atol=0)             # = the one-rank film of the same samples_per_pass


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus N` with no launcher (how the driver calls it): the parent spawns N children with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_*, rank 0's line reaches stdout, the exit code is 0.  MTSAMD_BENCH_SELFTEST swaps the GPU work for one gloo
    all-reduce so that this runs without GPUs."""
    import json
    import subprocess
    env = dict(os.environ, MTSAMD_BENCH_SELFTEST="ok")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    for n in (2, 8):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1                                                # ONE JSON line, from rank 0
        line = json.loads(lines[0])
        assert line["n_gpus"] == n and line["sum_of_ranks_plus_one"] == n * (n + 1) // 2 and line["local_rank"] == 0
        assert line["partition"]["passes"] == n


def test_bench_fails_when_a_rank_fails():
    import subprocess
    env = dict(os.environ, MTSAMD_BENCH_SELFTEST="fail")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 3 and "rank 1 exited with status 3" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
