#!/usr/bin/env python3
"""bench.py -- Msamples/s of the volpath hot path on the metric scene (BASELINE.json: C3, heterogeneous 128^3
grid + HG, 512x512x1024 spp) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one complete render of the workload with scene (grids) and film resident in HBM.

STRONG scaling (the headline, `"scaling": "strong"`): every N renders the SAME 512x512x1024 job.  For N > 1 the job is cut
into N passes of 1024/N spp with the reference's own `samples_per_pass` mechanism (librender/integrator.cpp:58-65), which
gives N x 256 (pass, block) pairs; they are dealt round-robin over the ranks (`block_id % N`), so every GPU gets 256
workgroups (one per CU), the per-pixel random streams are those the reference draws for that `samples_per_pass`, and the
only exchange is one RCCL reduce of the 5 MB XYZAW film over xGMI per step (inside the timed region).  Rank 0 then checks
that the reduced film equals the film ONE rank renders of the same job (`"film_check"`).

WEAK scaling (side figure, `"weak"`): 512x512x(1024 N) spp in passes of 1024 spp, one 512x512x1024 job per GPU.

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

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md ("HBM3E ... 8 TB/s"): the contract's `peak`.  The device's own
                           # figure (memory clock x bus width from hipDeviceProp) rides along in the line as `peak_device_query`.
CONFIG_SIZES = {"C1": (256, 256, 64), "C2": (512, 512, 256), "C3": (512, 512, 1024), "C4": (1024, 1024, 4096), "C5": (1024, 1024, 4096),
                "C5S": (1024, 1024, 256), "C5SM": (1024, 1024, 256), "C5SB": (1024, 1024, 256), "C3M": (512, 512, 1024), "C1L": (512, 512, 256), "C1W": (256, 256, 64), "C1S": (512, 512, 256)}      # C5S: the same atmosphere in the spectral variant (gpu_spectral), 256 spp
C5_WAVELENGTHS = 16        # BASELINE.json configs[4]: the C4 atmosphere as a 16-wavelength batch


def c5_rayleigh_scale(k):
    """Rayleigh optical thickness ~ lambda^-4 over 16 wavelengths from 400 to 1000 nm, relative to 550 nm."""
    lam = 400.0 + 40.0 * k
    return (550.0 / lam) ** 4


def build_scene_dict(scenes, config, width, height, spp, samples_per_pass=-1, res=128, wavelength=0):
    if config in ("C1", "C1L", "C1W", "C1S"):                       # C1L: the same cornell box at a size that fills the chip
        d = scenes.c1_cornell(width, height, spp)
        if config == "C1S":                                         # C1L in the spectral variant: reflectances as regular spectra, a D65 light
            for v in d.values():
                if isinstance(v, dict) and "bsdf" in v:
                    rgb = v["bsdf"]["reflectance"]["value"]
                    v["bsdf"]["reflectance"] = {"type": "regular", "lambda_min": 400., "lambda_max": 700., "values": [rgb[2], rgb[1], rgb[0]]}
            d["light"]["emitter"]["radiance"] = {"type": "d65", "scale": 3.0}
        if config == "C1W":                                         # C1 with the streams of the reference's gpu_* variants: one per (pixel, sample)
            d["sensor"]["sampler"]["wavefront"] = True
    elif config == "C2":
        d = scenes.c2_homogeneous_slab(width, height, spp)
    elif config == "C4":
        d = scenes.c4_atmosphere(width, height, spp)
    elif config == "C5":
        d = scenes.c4_atmosphere(width, height, spp, rayleigh_scale=c5_rayleigh_scale(wavelength))
    elif config in ("C5S", "C5SM", "C5SB"):
        d = scenes.c5_atmosphere_spectral(width, height, spp)
        if config == "C5SB":                                        # the same under Eradiate's wavelength-bin integrator: 16 bins, 32 AOV channels
            d["integrator"] = {"type": "nbins", "wavelengths": ", ".join("%g" % (360.0 + 470.0 * (k + 0.5) / 16) for k in range(16)), "tolerance": 470.0 / 32,
                               "integrator": d["integrator"]}
        if config == "C5SM":                                        # the spectral atmosphere under volpathmis (4 x 4 weight matrices)
            d["integrator"]["type"] = "volpathmis"
    else:
        d = scenes.c3_heterogeneous(width, height, spp, res=res)
        if config == "C3M":                                         # the metric scene under volpathmis (side measurement)
            d["integrator"]["type"] = "volpathmis"
    d["integrator"]["samples_per_pass"] = samples_per_pass             # nbins / bins: the wrapper is the integrator that renders (its own property)
    return d


class Job:
    """One workload resident on this rank's GPU: scene(s), film, and the step() that renders this rank's share and reduces it."""

    def __init__(self, pkg, scenes, args, rank, n, local_rank, backend, spp_total, samples_per_pass):
        import torch
        self.torch, self.rank, self.n, self.backend = torch, rank, n, backend
        # C5: monochromatic batches (scalar_mono semantics), one per wavelength; C5S: the spectral variant
        variant = {"C5": "gpu_mono", "C5S": "gpu_spectral", "C5SM": "gpu_spectral", "C5SB": "gpu_spectral", "C1S": "gpu_spectral"}.get(args.config, "gpu_rgb")
        pkg.set_variant(variant)
        self.dicts = [build_scene_dict(scenes, args.config, args.width, args.height, spp_total, samples_per_pass, args.res, k)
                      for k in range(C5_WAVELENGTHS if args.config == "C5" else 1)]
        self.scenes = [pkg.load_dict(d, device=local_rank) for d in self.dicts]       # grids uploaded to HBM here (outside the timed region)
        # X, Y, Z, A, W (+ two AOV channels per spectral bin under nbins / bins)
        self.films = [torch.zeros((args.height, args.width, 5 + 2 * sc._desc.integrator.bin_count), dtype=torch.float32, device="cuda") for sc in self.scenes]
        self.stream = torch.cuda.current_stream().cuda_stream
        self.samples_step = args.width * args.height * spp_total * len(self.scenes)      # all ranks, one step

    def render(self, shard_index, shard_count, collect_counters=False):
        stats = []
        for scene, film in zip(self.scenes, self.films):
            integ = scene.integrator()
            integ.render(scene, scene.sensors()[0], shard_index=shard_index, shard_count=shard_count, device_film=film.data_ptr(),
                         device_film_floats=film.numel(), stream=self.stream, collect_counters=collect_counters)
            stats.append(integ.last_stats)
        return stats

    def step(self):
        import torch.distributed as dist
        stats = self.render(self.rank, self.n)
        if self.n > 1:
            for film in self.films:
                if self.backend == "gloo":                   # rehearsal: all ranks on one GPU, film reduce on the host
                    host = film.cpu(); dist.reduce(host, dst=0, op=dist.ReduceOp.SUM); film.copy_(host)
                else:
                    dist.reduce(film, dst=0, op=dist.ReduceOp.SUM)       # RCCL over xGMI: W*H*5 fp32
        return stats


def timed(job, steps, warmup, barrier, all_max):
    for _ in range(warmup):
        job.step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms, launches, samples_rank, cal_ms = 0.0, 0, 0, 0.0
    for _ in range(steps):
        sts = job.step()
        # kernel_ms / kernel_launches: the render launches only; the short block-cost calibration launch of scenes with more blocks than
        # CUs is reported apart (mts_stats.calibration_ms) -- it is inside the timed region and in `value`, not in the roofline's launch time
        kernel_ms += sum(s["kernel_ms"] for s in sts); launches += sum(s["kernel_launches"] for s in sts)
        cal_ms += sum(s["calibration_ms"] for s in sts)
        samples_rank = sum(s["samples"] for s in sts)
        job.kernel_variant = sts[0]["kernel_variant"]
    barrier()
    elapsed = all_max(time.perf_counter() - t0)
    return elapsed, kernel_ms, launches, samples_rank, cal_ms


def device_hbm_peak_gbs(torch, device):
    """Peak HBM bandwidth as the device reports it: 2 x memory clock x bus width (hipDeviceGetAttribute: MemoryClockRate in kHz,
    MemoryBusWidth in bits -- torch does not expose them, so the HIP runtime torch has already loaded is asked directly; the enum
    values are those of /opt/rocm/include/hip/hip_runtime_api.h and are cross-checked on MultiprocessorCount).  None when that
    fails.  Reported next to the guide's figure, which stays the contract's `peak`."""
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        def attr(a):
            v = ctypes.c_int(0)
            return v.value if hip.hipDeviceGetAttribute(ctypes.byref(v), a, int(device)) == 0 else None
        BUS_WIDTH, CLOCK_RATE, CU_COUNT = 59, 60, 63
        if attr(CU_COUNT) != torch.cuda.get_device_properties(device).multi_processor_count:
            return None
        khz, bits = attr(CLOCK_RATE), attr(BUS_WIDTH)
        if not khz or not bits:
            return None
        return {"memory_clock_khz": khz, "bus_width_bits": bits, "ddr_peak_gbs": round(2.0 * khz * 1e3 * bits / 8 / 1e9, 1)}
    except Exception:
        return None


def spawn_ranks(n, argv=None, env=None, timeout=None):
    """`python bench.py --gpus N` without a launcher: the parent -- BEFORE anything touches the GPU or imports torch -- starts N fresh
    child processes of this script, one per GPU, with the environment torch.distributed.run would give them (RANK, LOCAL_RANK,
    WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT), lets rank 0 write the JSON line to our stdout, and returns non-zero
    if any rank fails (the others are then ended: a rank that died leaves its peers waiting in a collective).  Never exec: children
    only.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    argv = list(sys.argv[1:] if argv is None else argv)
    base = dict(os.environ if env is None else env)
    base.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: what RCCL needs on this driver
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    t0 = time.monotonic()
    code = 0
    try:
        pending = set(range(n))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0 and code == 0:
                    code = rc if rc > 0 else 1
                    print("bench.py: rank %d exited with status %d; ending the other ranks" % (r, rc), file=sys.stderr, flush=True)
            if code != 0 or (timeout is not None and time.monotonic() - t0 > timeout):
                if code == 0:
                    code = 124
                    print("bench.py: ranks still running after %.0f s; ending them" % timeout, file=sys.stderr, flush=True)
                break
            time.sleep(0.05)
    finally:
        for p in procs:                                           # exactly the processes started here
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill(); p.wait()
    return code


def partition(width, height, spp, n, block=32):
    """The strong-scaling cut of `bench.py --gpus n` as plain arithmetic (tests/test_distributed_cpu.py holds it against the oracle's
    spiral): passes of the largest divisor of spp not above spp / n, (pass, block) pairs dealt block_id % n."""
    spp_pass = spp
    if n > 1:
        spp_pass = max(1, spp // n)
        while spp % spp_pass:
            spp_pass -= 1
    passes = spp // spp_pass if n > 1 else 1
    blocks_total = -(-width // block) * -(-height // block) * passes
    per_rank = [len(range(r, blocks_total, n)) for r in range(n)]
    return {"spp_pass": spp_pass, "passes": passes, "blocks_total": blocks_total, "workgroups_per_rank": per_rank}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3", choices=sorted(CONFIG_SIZES),
                    help="BASELINE.json configuration; C3 is the one the metric is quoted on, the others are side measurements")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--height", type=int, default=0)
    ap.add_argument("--spp", type=int, default=0)
    ap.add_argument("--res", type=int, default=128)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-weak", action="store_true", help="skip the weak-scaling side measurement (N > 1)")
    ap.add_argument("--no-film-check", action="store_true", help="skip the N-rank == 1-rank film check (N > 1)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="spp of the bounded CPU-baseline sample (0 = sized for ~15 s)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn_ranks(args.gpus))                  # `python bench.py --gpus N`: start the N ranks ourselves

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    selftest = os.environ.get("MTSAMD_BENCH_SELFTEST")
    if selftest:
        # tests/test_distributed_cpu.py: the launch plumbing of spawn_ranks on a box without GPUs -- rendezvous over gloo with the
        # environment the parent made, one all-reduce, rank 0 prints the line; "fail" lets rank 1 die before the rendezvous
        if selftest == "fail" and rank == 1:
            raise SystemExit(3)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        t = torch.tensor([rank + 1], dtype=torch.int64)
        dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"selftest": True, "n_gpus": world, "sum_of_ranks_plus_one": int(t.item()), "local_rank": local_rank,
                              "partition": partition(*CONFIG_SIZES[args.config], world)}), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    n = args.gpus
    if world != n:
        n = world                                                 # the launcher's world size wins over --gpus
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

    cfg_w, cfg_h, cfg_spp = CONFIG_SIZES[args.config]
    args.width, args.height, args.spp = args.width or cfg_w, args.height or cfg_h, args.spp or cfg_spp

    def barrier():
        if n > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_max(x):
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        if n > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ---- strong scaling: the fixed job, N passes of spp / N (the largest divisor of spp not above spp / N)
    part = partition(args.width, args.height, args.spp, n)
    spp_pass = part["spp_pass"]
    variant = {"C5": "gpu_mono", "C5S": "gpu_spectral", "C5SM": "gpu_spectral", "C5SB": "gpu_spectral", "C1S": "gpu_spectral"}.get(args.config, "gpu_rgb")
    job = Job(pkg, scenes, args, rank, n, local_rank, backend, args.spp, spp_pass if n > 1 else -1)
    # every rank should get at least one workgroup per CU (256) per launch, or the GPUs run partly empty: the reason the N-rank job is
    # cut into N passes.  Checked for the configurations at their BASELINE sizes (a rehearsal on a small film cannot meet it).
    workgroups_per_rank = min(part["workgroups_per_rank"])
    if (args.width, args.height, args.spp) == CONFIG_SIZES[args.config] and args.config not in ("C1", "C1W") and workgroups_per_rank < 256:
        raise SystemExit("bench.py: %d workgroups per rank (< 256 CUs) at --gpus %d" % (workgroups_per_rank, n))
    elapsed, kernel_ms, launches, samples_rank, cal_ms = timed(job, args.steps, args.warmup, barrier, all_max)
    value = job.samples_step * args.steps / elapsed / 1e6
    # kernel time per step of every rank (HIP events around the launches): a SCALE record shows load imbalance directly
    per_rank_ms = [kernel_ms / args.steps]
    if n > 1:
        t = torch.tensor([kernel_ms / args.steps], dtype=torch.float64, device="cpu" if backend == "gloo" else "cuda")
        gathered = [torch.zeros_like(t) for _ in range(n)]
        dist.all_gather(gathered, t)
        per_rank_ms = [float(g.item()) for g in gathered]

    film_check = None
    if n > 1 and not args.no_film_check:
        # the reduced film of the N ranks against the film ONE rank renders of the same job with the same samples_per_pass.  Box
        # filter: every film entry is the sum of the same per-pass values; only the order of those <= N additions differs.
        reduced = [f.clone() for f in job.films]
        if rank == 0:
            job.render(0, 1)
            worst = max(float(((a - b).abs() / b.abs().clamp_min(1e-6)).max().item()) for a, b in zip(reduced, job.films))
            film_check = {"max_rel_diff": worst, "ok": bool(worst < 1e-5)}
        barrier()
        if rank == 0 and not film_check["ok"]:
            raise SystemExit("bench.py: the %d-rank film differs from the 1-rank film (max rel %.3g)" % (n, film_check["max_rel_diff"]))

    # ---- roofline of the dominant kernel: algorithmic bytes / measured launch duration (SURVEY.md 8(d) convention).
    # Counters come from one extra, untimed render with the counting kernel variant; they are deterministic and equal the
    # oracle's at the same seed (tests/test_gpu_parity.py).
    cs = job.render(rank, n, collect_counters=True)
    c_iter, c_look, c_nee, c_samp = (sum(s[k] for s in cs) for k in ("n_iter", "n_lookup", "n_nee_step", "samples"))
    bytes_per_sample = (224.0 * c_iter + 64.0 * c_look + 128.0 * c_nee) / c_samp + 40.0
    avg_launch_ms = kernel_ms / max(launches, 1)
    bytes_per_launch = bytes_per_sample * samples_rank * (args.steps / max(launches, 1))
    achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
    integ_type = (job.dicts[0]["integrator"].get("integrator") or job.dicts[0]["integrator"])["type"]
    kv = os.environ.get("MTSAMD_KERNEL", "wga1024")
    if integ_type == "path" and kv != "nested":
        kernel_name = ("v_spectral::" if variant == "gpu_spectral" else "") + "render_kernel<false, true, 0>"      # path_pixel_flat: one flat loop with regeneration
    elif integ_type == "path" or kv == "nested":
        kernel_name = "render_kernel<false, false, %d>" % {"path": 0, "volpath": 1, "volpathmis": 2}.get(integ_type, 0)
    elif kv == "flat" and integ_type == "volpath":
        kernel_name = "render_kernel<false, true, 1>"
    elif integ_type == "volpathmis":
        kernel_name = "v_spectral::render_kernel_wga_mis<false, true, 256, 256>" if args.config == "C5SM" else "render_kernel_wga_mis<false, true, 512, 512>"
    elif args.config in ("C5S", "C5SB"):
        kernel_name = "v_spectral::render_kernel_wga<false, 256, 256, 2, false>"
    else:
        paths = int(kv[3:]); nt = int(os.environ.get("MTSAMD_WG_THREADS", str(paths)))
        kernel_name = "render_kernel_%s<false, %d, %d, %d%s>" % (kv[:3], paths, nt, {1: 4, 0.75: 3, 0.5: 2}[nt / paths], ", false" if kv[:3] == "wga" else "")
    # a scene that keeps the promises of a lean translation unit runs that unit's copy of the kernel (mts_stats.kernel_variant + 100000 / 200000)
    lean = {1: "v_rgb_lean_a::", 2: "v_rgb_lean_b::", 3: "v_spectral_lean::", 4: "v_rgb_lean_p::", 5: "v_spectral_lean_p::", 6: "v_rgb_lean_h::", 7: "v_rgb_lean_c::"}.get(getattr(job, "kernel_variant", 0) // 100000, "")
    if lean:
        kernel_name = lean + kernel_name.replace("v_spectral::", "")
        if lean == "v_spectral_lean::":
            kernel_name = kernel_name.replace("render_kernel_wga<false, 256, 256, 2, false>", "render_kernel_wga<false, 256, 256, 3, false>")       # budget of three waves per SIMD
        if lean == "v_rgb_lean_a::":
            kernel_name = kernel_name.replace("render_kernel_wga_mis<false, true, 512, 512>", "render_kernel_wga_mis<false, true, 512, 768>")   # three waves per SIMD
    # `achieved` / `frac` follow the contract: ALGORITHMIC bytes (a wavefront formulation's state round trips) over
    # the kernel's measured time.  This kernel keeps path state in LDS, so its real HBM traffic is several times lower and its
    # bound is latency at 4 waves per SIMD; `traffic*` and `valu_pipe_busy` (rocprofv3 --pmc, profiles/) say so whenever this run
    # matches the profiled configuration.
    # `bound`: what the counters say limits this kernel (rocprofv3 --pmc, profiles/: latency at 4 waves per SIMD, vector pipes half busy, HBM
    # at ~15 % of peak).  `achieved` / `peak` / `frac` stay the contract's HBM-roofline figure, which is a MODEL here (`model_bound`):
    # the algorithmic bytes of a wavefront formulation that round-trips path state through HBM, over the measured launch time.
    roofline = {"bound": "latency", "model_bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                "achieved_is": "algorithmic bytes per launch / measured launch time (SURVEY.md 8(d)): a modelling figure, not a bandwidth measurement",
                "kernel": kernel_name, "avg_launch_ms": round(avg_launch_ms, 3), "launches_per_step": round(launches / args.steps, 3),
                "calibration_ms_per_step": round(cal_ms / args.steps, 3),
                "peak_source": "MI355X_MICROARCH.md (HBM3E, 8 TB/s)", "peak_device_query": device_hbm_peak_gbs(torch, local_rank),
                "bytes_per_sample": round(bytes_per_sample, 1),
                "n_iter_per_sample": round(c_iter / c_samp, 3),
                "n_lookup_per_sample": round(c_look / c_samp, 3),
                "n_nee_step_per_sample": round(c_nee / c_samp, 3)}
    this_run = {"config": args.config, "width": args.width, "height": args.height, "spp": args.spp, "res": args.res, "n_gpus": n, "kernel": kernel_name}
    for fname in ("traffic_bytes_per_launch.json",):
        path = os.path.join(ROOT, "profiles", fname)
        try:
            prof = json.load(open(path))
        except Exception:
            continue
        if prof.get("run") == this_run:                          # counters describe THIS configuration and kernel only
            roofline["traffic"] = prof.get("bytes_per_launch")
            roofline["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this configuration; not measured in this run)" % fname
            roofline["traffic_gbs"] = round(prof["bytes_per_launch"] / (avg_launch_ms * 1e-3) / 1e9, 1)
            roofline["traffic_frac_of_peak"] = round(roofline["traffic_gbs"] / HBM_PEAK_GBS, 4)
            dv = prof.get("derived") or {}
            if "valu_pipe_busy" in dv:
                roofline["valu_pipe_busy"] = dv["valu_pipe_busy"]
                roofline["real_bound"] = ("latency at 4 waves per SIMD: vector pipes %.0f %% busy, a wave waits %.0f %% of its time in s_waitcnt, "
                                          "HBM at %.0f %% of peak" % (100 * dv["valu_pipe_busy"], 100 * dv.get("wave_wait_share", 0.0), 100 * roofline["traffic_frac_of_peak"]))

    # ---- weak scaling side figure (N > 1): one 512x512x1024 job per GPU
    weak = None
    if n > 1 and not args.no_weak:
        del job
        torch.cuda.empty_cache()
        wjob = Job(pkg, scenes, args, rank, n, local_rank, backend, args.spp * n, args.spp)
        w_elapsed = timed(wjob, args.steps, args.warmup, barrier, all_max)[0]
        weak = {"value": round(wjob.samples_step * args.steps / w_elapsed / 1e6, 2), "unit": "Msamples/s", "ms_per_step": round(w_elapsed / args.steps * 1e3, 2),
                "workload": "%dx%dx%d spp total, passes of %d: one %dx%dx%d job per GPU" % (args.width, args.height, args.spp * n, args.spp, args.width, args.height, args.spp)}
        del wjob

    cpu_baseline = None
    if rank == 0 and n == 1 and not args.no_cpu_baseline:
        # the oracle (CPU restatement of scalar_rgb) on all host cores, on a bounded sample of the same workload
        import tests.oracle_binding as ob
        cores = os.cpu_count() or 1
        def cpu_render(spp):
            osc = ob.OracleScene(build_scene_dict(scenes, args.config, args.width, args.height, spp, -1, args.res, 0), mono=args.config == "C5",
                                 spectral=args.config in ("C5S", "C5SM", "C5SB", "C1S"))
            tc0 = time.perf_counter()
            osc.render(threads=cores)
            return time.perf_counter() - tc0
        cpu_spp = args.cpu_spp
        if cpu_spp <= 0:                                         # calibrate with 2 spp, then size the sample for ~15 s
            tcal = cpu_render(2)
            cpu_spp = int(min(args.spp, max(2, round(15.0 / (tcal / 2.0)))))
        tcpu = cpu_render(cpu_spp)
        cpu_baseline = {"value": round(args.width * args.height * cpu_spp / tcpu / 1e6, 4), "unit": "Msamples/s", "cores": cores,
                        "kind": "port", "sample": "%dx%dx%dspp of the same scene%s, all pixels, seed 0 (%.1f s on %d threads)"
                                                  % (args.width, args.height, cpu_spp, " (first wavelength)" if args.config == "C5" else "", tcpu, cores)}

    if rank == 0:
        workload = {"C1": "C1 path cornell box", "C1W": "C1W = C1 with wavefront (gpu_*) streams, one per (pixel, sample): the samples of a pixel spread over several workgroups", "C1L": "C1L = the C1 cornell box at 512x512x256 (262144 pixel streams: one per lane of the chip)", "C2": "C2 volpath homogeneous slab", "C4": "C4 volpath layered atmosphere, blend/tabulated phase, RPV ground",
                    "C1S": "C1S = C1L in the spectral variant (gpu_spectral: regular reflectance spectra, D65 light), `path` as the flat loop with regeneration",
                    "C5": "C5 = C4 as %d monochromatic wavelength batches (Rayleigh ~ lambda^-4), gpu_mono" % C5_WAVELENGTHS,
                    "C5S": "C5S = the C4 atmosphere in the spectral variant (gpu_spectral: 4 wavelengths per sample, gridvolume_spectral grids, global majorant)",
                    "C5SB": "C5SB = C5S inside nbins (16 wavelength bins over 360 .. 830 nm: 32 AOV channels behind X, Y, Z, A, W), regrouping kernel",
                    "C5SM": "C5SM = C5S under volpathmis (spectral MIS with the 4 x 4 weight matrix of volpathmis.cpp:66-69), regrouping kernel compiled four wide",
                    "C3M": "C3M = the C3 scene under volpathmis (spectral MIS), regrouping kernel of volpathmis_flat.h",
                    "C3": "C3 volpath heterogeneous %d^3 grid + HG g=0.8" % args.res}[args.config]
        out = {"metric": "Msamples/s volpath 512x512x1024spp plane-parallel atmosphere" if args.config == "C3" else "Msamples/s %s (side measurement)" % args.config,
               "value": round(value, 2), "unit": "Msamples/s",
               "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2),
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "%s, %dx%dx%dspp%s" % (workload, args.width, args.height, args.spp,
                                                             " in %d passes of %d spp (samples_per_pass)" % (args.spp // spp_pass, spp_pass) if n > 1 else ""),
                          "integrator": integ_type, "sampler": "independent seed 0", "block_size": 32, "rfilter": "box",
                          "sharding": "(pass, block) pairs: block_id %% %d round-robin, %d workgroups per GPU + RCCL film reduce" % (n, -(-args.width // 32) * -(-args.height // 32) * (args.spp // spp_pass) // n)
                                      if n > 1 else "single GPU"},
               "kernel_ms_per_step": {"per_rank": [round(x, 3) for x in per_rank_ms], "max_over_min": round(max(per_rank_ms) / max(min(per_rank_ms), 1e-9), 4)},
               "workgroups_per_rank_per_launch": workgroups_per_rank,
               "roofline": roofline, "cpu_baseline": cpu_baseline}
        if film_check is not None:
            out["film_check"] = film_check
        if weak is not None:
            out["weak"] = weak
        print(json.dumps(out), flush=True)
    if n > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
