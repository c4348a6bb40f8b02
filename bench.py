#!/usr/bin/env python3
"""bench.py -- Msamples/s of the volpath hot path on the metric scene (BASELINE.json: C3, heterogeneous 128^3
grid + HG, 512x512x1024 spp) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete render of the workload with scene (grids) and film resident in HBM.  N = 1 renders
512x512x1024 spp.  N > 1 renders 512x512x(1024 N) spp in passes of 1024 spp (the reference's own
`samples_per_pass` mechanism, librender/integrator.cpp:58-65): the (pass, block) pairs are dealt round-robin
over the ranks (`block_id % N`), so every GPU always does one 512x512x1024 job (weak scaling), the per-pixel
random streams stay those of the reference, and the only exchange is one RCCL reduce of the 5 MB XYZAW film
over xGMI per step (inside the timed region).

Rank 0 prints ONE JSON line; see DESIGN.md for how `roofline` and `cpu_baseline` are derived.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md, chip-level parameters)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4"],
                    help="BASELINE.json configuration; C3 is the one the metric is quoted on, the others are side measurements")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU-baseline sample (0 = sized for ~15 s)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n = args.gpus
    if world != n:
        if world == 1 and n > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (n, n))
        n = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the gpu_rgb backend has no CPU fallback)")
    # Rehearsal switch: MTSAMD_BENCH_BACKEND=gloo runs the N-rank path with all ranks on ONE GPU and the film reduce on the
    # host (RCCL refuses two ranks on one device); numbers from such a run are not a measurement.
    backend = os.environ.get("MTSAMD_BENCH_BACKEND", "nccl")
    if backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if n > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=n)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=n, device_id=torch.device("cuda", local_rank))

    pkg = importlib.import_module("eradiate-kernel_amd")
    scenes = importlib.import_module("eradiate-kernel_amd.scenes")
    pkg.set_variant("gpu_rgb")

    cfg_w, cfg_h, cfg_spp = {"C1": (256, 256, 64), "C2": (512, 512, 256), "C3": (512, 512, 1024), "C4": (1024, 1024, 4096)}[args.config]
    args.width, args.height, args.spp = args.width or cfg_w, args.height or cfg_h, args.spp or cfg_spp
    spp_total = args.spp * n

    def make_scene(spp, samples_per_pass=-1):
        if args.config == "C1":
            d_ = scenes.c1_cornell(args.width, args.height, spp)
        elif args.config == "C2":
            d_ = scenes.c2_homogeneous_slab(args.width, args.height, spp)
        elif args.config == "C4":
            d_ = scenes.c4_atmosphere(args.width, args.height, spp, samples_per_pass=samples_per_pass)
        else:
            return scenes.c3_heterogeneous(args.width, args.height, spp, res=args.res, samples_per_pass=samples_per_pass)
        d_["integrator"]["samples_per_pass"] = samples_per_pass
        return d_
    d = make_scene(spp_total, args.spp)
    scene = pkg.load_dict(d, device=local_rank)            # grids uploaded to HBM here (outside the timed region)
    sensor = scene.sensors()[0]
    integ = scene.integrator()
    film = torch.zeros((args.height, args.width, 5), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        integ.render(scene, sensor, shard_index=rank, shard_count=n, device_film=film.data_ptr(), stream=stream)
        if n > 1 and backend == "gloo":
            host = film.cpu(); dist.reduce(host, dst=0, op=dist.ReduceOp.SUM); film.copy_(host)
        elif n > 1:
            dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)   # RCCL over xGMI: W*H*5 fp32
        return integ.last_stats

    def barrier():
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, launches, samples_rank = 0.0, 0, 0
    for _ in range(args.steps):
        st = step()
        kernel_ms += st["kernel_ms"]; launches += st["kernel_launches"]; samples_rank = st["samples"]
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
    if n > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    samples_step = args.width * args.height * spp_total      # all ranks, one step
    value = samples_step * args.steps / elapsed / 1e6

    # ---- roofline of the dominant kernel (render_kernel): algorithmic bytes / measured launch duration
    # counters come from one extra, untimed render with the counting kernel variant; they are
    # deterministic and equal the oracle's at the same seed (tests/test_gpu_parity.py).
    integ.render(scene, sensor, shard_index=rank, shard_count=n, device_film=film.data_ptr(), stream=stream, collect_counters=True)
    cs = integ.last_stats
    bytes_per_sample = (224.0 * cs["n_iter"] + 64.0 * cs["n_lookup"] + 128.0 * cs["n_nee_step"]) / cs["samples"] + 40.0
    avg_launch_ms = kernel_ms / max(launches, 1)
    bytes_per_launch = bytes_per_sample * samples_rank * (args.steps / max(launches, 1))
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    kv = os.environ.get("MTSAMD_KERNEL", "wga1024")
    if d["integrator"]["type"] == "path" or kv == "nested":
        kernel_name = "render_kernel<false, false, %d>" % {"path": 0, "volpath": 1, "volpathmis": 2}.get(d["integrator"]["type"], 0)
    elif kv == "flat":
        kernel_name = "render_kernel<false, true, 1>"
    else:
        paths = int(kv[3:]); nt = int(os.environ.get("MTSAMD_WG_THREADS", str(paths)))
        kernel_name = "render_kernel_wga<false, %d, %d, %d>" % (paths, nt, {1: 4, 0.75: 3, 0.5: 2}[nt / paths])
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": kernel_name, "avg_launch_ms": round(avg_launch_ms, 3),
                "bytes_per_sample": round(bytes_per_sample, 1),
                "n_iter_per_sample": round(cs["n_iter"] / cs["samples"], 3),
                "n_lookup_per_sample": round(cs["n_lookup"] / cs["samples"], 3),
                "n_nee_step_per_sample": round(cs["n_nee_step"] / cs["samples"], 3)}
    traffic_file = os.path.join(ROOT, "profiles", "traffic_bytes_per_launch.json")
    if os.path.exists(traffic_file) and args.config == "C3":                           # measured HBM bytes per launch from the rocprofv3 --pmc passes
        try:
            roofline["traffic"] = json.load(open(traffic_file)).get("bytes_per_launch")
        except Exception:
            pass

    cpu_baseline = None
    if rank == 0 and n == 1 and not args.no_cpu_baseline:
        # the oracle (CPU restatement of scalar_rgb) on all host cores, on a bounded sample of the same workload
        import tests.oracle_binding as ob
        cores = os.cpu_count() or 1
        def cpu_render(spp):
            osc = ob.OracleScene(make_scene(spp))
            tc0 = time.perf_counter()
            osc.render(threads=cores)
            return time.perf_counter() - tc0
        cpu_spp = args.cpu_spp
        if cpu_spp <= 0:                                         # calibrate with 2 spp, then size the sample for ~15 s
            tcal = cpu_render(2)
            cpu_spp = int(min(1024, max(2, round(15.0 / (tcal / 2.0)))))
        tcpu = cpu_render(cpu_spp)
        cpu_baseline = {"value": round(args.width * args.height * cpu_spp / tcpu / 1e6, 4), "unit": "Msamples/s", "cores": cores,
                        "kind": "port", "sample": "%dx%dx%dspp of the same scene, all pixels, seed 0 (%.1f s on %d threads)"
                                                  % (args.width, args.height, cpu_spp, tcpu, cores)}

    if rank == 0:
        workload = {"C1": "C1 path cornell box", "C2": "C2 volpath homogeneous slab", "C4": "C4 volpath layered atmosphere, blend/tabulated phase, RPV ground",
                    "C3": "C3 volpath heterogeneous %d^3 grid + HG g=0.8" % args.res}[args.config]
        out = {"metric": "Msamples/s volpath 512x512x1024spp plane-parallel atmosphere" if args.config == "C3" else "Msamples/s %s (side measurement)" % args.config,
               "value": round(value, 2), "unit": "Msamples/s",
               "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s, %dx%dx%dspp per GPU (%d spp total, passes of %d)"
                                      % (workload, args.width, args.height, args.spp, spp_total, args.spp),
                          "integrator": d["integrator"]["type"], "sampler": "independent seed 0", "block_size": 32, "rfilter": "box",
                          "sharding": "block_id %% %d round-robin + RCCL film reduce" % n if n > 1 else "single GPU"},
               "roofline": roofline, "cpu_baseline": cpu_baseline}
        print(json.dumps(out), flush=True)
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
