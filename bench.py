#!/usr/bin/env python3
"""bench.py -- refined samples/sec @ K refinement steps (BASELINE.json metric) on N MI355X GPUs.

A "step" = one pass of the hot path over one z-batch: propose (G head) + K-step collaborative
refinement (K+1 G-tail/D forwards, K backward-datas, momentum update, best-sample select) + final
render, images left in HBM.  Default workload = BASELINE configs[2]: DCGAN CelebA 64x64, batch 1024
per GPU, K = 20 (the configuration the 10k samples/s target is quoted on; it fits one GPU).
Inputs (z batches) and weights are resident in HBM before the timed region.  Two steps are in flight per GPU
(--streams), and for the small configurations several logical batches share one launch, each with its own
batch-norm statistics (--fuse; dcgan32 4 x 256, mnist 16 x 64): the work and the results of a step are unchanged.
The K-step program of a step is replayed as a hipGraph (the launch-bound inner loop: ~1300 dependent launches per step;
--no-graph launches them one by one); the per-kernel HIP-event timing behind `roofline` then comes from one extra step
launched eagerly on one stream right after the timed region (the same kernels with the same arguments).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), independent z-batches per rank
(seed 2019+rank, weak scaling), and ONE RCCL all-gather per step of the refined images into the
node-wide sample pool -- inside the timed region.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps 3 --warmup 1
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

PEAK_FP32_MATRIX_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of this same
# command, FETCH_SIZE doubled per the gfx950 correction of the guide); collected offline, see profiles/README.md
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")


def measured_traffic(kernel):
    """HBM bytes per launch of ``kernel`` from the committed PMC passes -- or None when profiles/traffic.json was collected
    on other kernel sources than the ones this run executes (a stale number is worse than none)."""
    try:
        from cgs_amd.lib import source_hash
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        if t.get("_source_sha256") != source_hash():
            return None
        return t.get(kernel, {}).get("hbm_bytes_per_launch_corrected")
    except (OSError, ValueError):
        return None


def cpu_baseline(arch, refine_steps, rate, budget_batch=64):
    """The CPU oracle (torch-CPU fp32 restatement of collaborator.build_refiner) timed on the host cores
    on a bounded sample of the same workload (same net, same K, a smaller batch: work is linear in B)."""
    from oracle import nets_ref as N
    from oracle import sampling_ref as S
    P = N.init_params(arch, 2019, True)
    zdim = N.ARCHS[arch]["z_dim"]
    gt, dd = (lambda f: N.feature_to_data(arch, P, f)), (lambda x: N.discriminator(arch, P, x))

    def run(B, K):
        z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (B, zdim)).astype(np.float32))
        with torch.no_grad():
            f0 = N.input_to_feature(arch, P, z)
        t = time.time()
        S.collaborative_refine(f0, gt, dd, K, rate)
        return time.time() - t
    run(4, 1)                                  # warm the thread pool / allocator
    # more threads is not faster on these layer sizes (8 cores beat 128 on the first boxes measured): calibrate the
    # thread count on a small run and time the sample with the best one
    ncpu = os.cpu_count() or 1
    best_n, best_t = torch.get_num_threads(), None
    for n in sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu}):
        torch.set_num_threads(n)
        run(8, 1)
        t = run(32, 1) / 4.0               # (per 8 samples; a 32-sample run is long enough to rank the thread counts reliably)
        if best_t is None or t < best_t:
            best_n, best_t = n, t
    torch.set_num_threads(best_n)
    # size the sample for ~15 s of CPU work from a 32-sample pilot at the full K
    pilot = run(32, refine_steps)
    budget_batch = int(min(512, max(budget_batch, 32 * round(15.0 / pilot))))
    dt = run(budget_batch, refine_steps)
    return {"value": round(budget_batch / dt, 3), "unit": "samples/s", "cores": best_n, "kind": "port",
            "sample": f"oracle.collaborative_refine (torch-CPU fp32), {arch}, batch {budget_batch}, K={refine_steps}, "
                      f"{dt:.1f} s wall, {best_n} of {ncpu} host threads (fastest of a 8/16/32/64/all calibration)"}


def run_synthetic2d(dev, rank, B, Ksteps, rate, steps, warmup, n_streams):
    """BASELINE config 1 on the fused device refiner: (seconds for ``steps`` batches, the pieces the cpu_baseline leg needs)."""
    from cgs_amd.synthetic import MLPDiscriminator
    from oracle import sampling_ref as S                      # only for the seeded weights + the cpu_baseline leg
    Ws, bs = S.mlp_init(64, 6, seed=2019, scale=2.0)
    D = MLPDiscriminator.from_lists([w.numpy() for w in Ws], [b.numpy() for b in bs], dev)
    n = steps + warmup
    x = torch.from_numpy((3.0 * np.random.RandomState(2019 + rank).randn(n, B, 2)).astype(np.float32)).to(dev)
    real = torch.from_numpy(S.toy_next_batch("Imbal-8Gaussians", 10.0, 0.9, B, np.random.RandomState(7)).astype(np.float32)).to(dev)

    # one wave per sample: a 512-sample batch occupies 6 % of the GPU's wave slots and is latency-bound (K+1 dependent MLP
    # evaluations), so independent batches are kept in flight on several streams (nothing synchronises with the host)
    streams = [torch.cuda.Stream(dev) for _ in range(n_streams)]

    def step(i):
        with torch.cuda.stream(streams[i % n_streams]):
            base = D.sigmoid_and_saliency(real, want_saliency=False)[0].mean()        # np.mean(real_sigmoid), refiner_cpu.py:23,28 (stays on the device)
            return D.refine(x[i], base, Ksteps, rate, "ladam")[0]
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(warmup, n):
        step(i)
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, (S, Ws, bs, x, real)


def bench_synthetic2d(args, dev, rank, world):
    """BASELINE config 1: Imbal-8Gaussians MLP-GAN (D: 2 -> 64 x 5 -> 1), batch 512, K = 10, ladam rate 0.1
    (synthetic/main.py:32-62) -- the whole refiner_cpu loop as one launch per batch (cgs_amd.synthetic)."""
    B, Ksteps = args.batch or 512, args.refine_steps or 10
    n_streams = args.streams if args.streams > 0 else 8
    dt, (S, Ws, bs, x, real) = run_synthetic2d(dev, rank, B, Ksteps, args.rate, args.steps, args.warmup, n_streams)
    if rank == 0:
        out = {"metric": f"refined samples/sec @ {Ksteps} refinement steps", "value": round(world * B * args.steps / dt, 1), "unit": "samples/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"synthetic2d Imbal-8Gaussians MLP-GAN D 2-64x5-1, batch {B}, K={Ksteps}, ladam rate {args.rate}: "
                                      f"real-batch baseline + fused K-step refine (3 launches per batch, no host synchronisation), {n_streams} batches in flight"},
               "roofline": None}
        if not args.no_cpu_baseline:
            fake = x[0].cpu().numpy(); rb = real.cpu().numpy().astype(np.float64)
            d_fn = lambda v: S.mlp_sigmoid_and_saliency(Ws, bs, v)
            torch.set_num_threads(min(8, os.cpu_count() or 1))          # 64-wide MLP layers: more threads only add overhead
            S.refine_2d(fake, rb, d_fn, Ksteps, args.rate, "ladam")
            t = time.time(); reps = 20
            for _ in range(reps):
                S.refine_2d(fake, rb, d_fn, Ksteps, args.rate, "ladam")
            c = (time.time() - t) / reps
            out["cpu_baseline"] = {"value": round(B / c, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": f"oracle.refine_2d (refiner_cpu.manipulate_sample restated; torch-CPU D), batch {B}, K={Ksteps}, {reps} reps"}
        print(json.dumps(out), flush=True)


def other_configs(dev, skip):
    """samples/s of the other single-GPU configurations ("always report MNIST"; BASELINE configs[0], [1]) at
    their bench.py defaults, timed AFTER and OUTSIDE the headline's timed region: a handful of steps each, value only."""
    from cgs_amd import nets
    from cgs_amd.engine import RefineEngine
    out = {}
    for arch, B, Ksteps, G, steps in (("mnist", 64, 50, 16, 6), ("dcgan32", 256, 20, 4, 6)):
        if arch == skip:
            continue
        A = nets.ARCHS[arch]
        P = nets.init_params(arch, dev, seed=2019)
        engines = [RefineEngine(arch, P, B * G, dev, use_graph=True, bn_groups=G) for _ in range(2)]
        streams = [torch.cuda.Stream(dev) for _ in engines]
        z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (steps + 2, B * G) + nets.g_input_shape(A)).astype(np.float32)).to(dev)

        def step(i):
            with torch.cuda.stream(streams[i % 2]):
                engines[i % 2].refine_from_z(z[i], Ksteps, 0.1)
        step(0); step(1)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(2, steps + 2):
            step(i)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        out[arch] = {"samples_per_s": round(B * G * steps / dt, 1), "batch": B, "refine_steps": Ksteps, "fused_per_launch": G,
                     "batches_in_flight": 2 * G, "steps": steps,
                     "algorithmic_tflops": round(B * G * steps / dt * nets.refine_flops_per_sample(arch, Ksteps) / 1e12, 2)}
        del engines, z, P
        torch.cuda.empty_cache()
    if skip != "synthetic2d":
        dt, _ = run_synthetic2d(dev, 0, 512, 10, 0.1, 256, 16, 8)
        out["synthetic2d"] = {"samples_per_s": round(512 * 256 / dt, 1), "batch": 512, "refine_steps": 10, "method": "ladam",
                              "batches_in_flight": 8, "steps": 256}
    return out


def self_launch(n_gpus):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N ranks as CHILD processes
    (`python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>`) before this process has touched the
    GPU -- never an in-place exec -- and relay rank 0's JSON line (the children inherit stdout) and the exit code."""
    import socket
    import subprocess
    with socket.socket() as so:                                 # a free rendezvous port on the loopback interface
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default 12: ms_per_step_median is then the median of 11 completion gaps; 256 for synthetic2d, whose step is 0.05 ms)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--arch", default="dcgan64", choices=["mnist", "dcgan32", "dcgan64", "synthetic2d", "cyclegan256"],
                    help="synthetic2d = BASELINE config 1 (2-D MLP GAN, batch 512, K=10, ladam) on the fused device refiner")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: 1024 dcgan64, 256 dcgan32, 64 mnist)")
    ap.add_argument("--refine-steps", type=int, default=0, help="K (default: 20; 50 for mnist)")
    ap.add_argument("--rate", type=float, default=0.1)
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="replay the K-step program as a hipGraph (the default for the refinement archs: +1.3 %% on the dcgan64 headline, the "
                         "launch gaps between the ~1300 dependent kernels of a step otherwise only partly hide behind the other batch in flight)")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch every kernel eagerly")
    ap.add_argument("--streams", type=int, default=0,
                    help="(default 2; 8 for synthetic2d) z-batches in flight per GPU, one engine + HIP stream each: the tail / small kernels of one "
                         "batch overlap the big kernels of the other (measured: 1 -> 5040, 2 -> 5500, 3 -> 5460 samples/s)")
    ap.add_argument("--fuse", type=int, default=0,
                    help="logical batches fused into one engine batch (conv launches G times larger, batch-norm statistics kept "
                         "per logical batch).  Default: as many as make ~1024 samples per launch (dcgan64 1, dcgan32 4, mnist 16)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="treat the N ranks' batches as ONE logical batch of N*B samples: all-reduce D's batch-norm sums "
                         "(needs the torch.distributed launch; eager only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the mnist / dcgan32 / synthetic2d samples/s that are measured after the headline's timed region")
    ap.add_argument("--by-layer", action="store_true", help="key the per-kernel timing records by layer shape too (diagnostic)")
    args = ap.parse_args()
    if args.steps <= 0:
        args.steps = 256 if args.arch == "synthetic2d" else 12

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; must be set before HIP initialises
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path to benchmark)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run (any world size)
    if use_dist:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    if args.arch == "synthetic2d":
        return bench_synthetic2d(args, dev, rank, world)

    from cgs_amd import kernels as K
    from cgs_amd import nets
    from cgs_amd.engine import RefineEngine

    B = args.batch or {"dcgan64": 1024, "dcgan32": 256, "mnist": 64, "cyclegan256": 8}[args.arch]
    Ksteps = args.refine_steps or (50 if args.arch == "mnist" else 20)
    A = nets.ARCHS[args.arch]
    P = nets.init_params(args.arch, dev, seed=2019)                     # same frozen weights on every rank
    if args.sync_bn and not use_dist:
        raise SystemExit("--sync-bn needs the torch.distributed launch (python -m torch.distributed.run ... bench.py)")
    G = args.fuse if args.fuse > 0 else {"dcgan32": 4, "mnist": 16}.get(args.arch, 1) if not args.sync_bn else 1
    if args.graph is None:
        # default: replay graphs on the single-GPU run; launch eagerly under the multi-process launch, where graph capture next to
        # RCCL's threads could not be exercised on hardware here (--graph turns it on there too), and with synchronised batch norm,
        # which has a collective inside the program
        args.graph = (not use_dist or world == 1) and not args.sync_bn
    engines = [RefineEngine(args.arch, P, B * G, dev, use_graph=args.graph, sync_bn=True if args.sync_bn else None, bn_groups=G)
               for _ in range(args.streams if args.streams > 0 else 2)]
    streams = [torch.cuda.Stream(dev) for _ in engines] if len(engines) > 1 else [torch.cuda.current_stream(dev)]
    eng = engines[0]
    n_batches = args.steps + args.warmup                                # a step = one engine call = G logical batches
    rs = np.random.RandomState(2019 + rank)                            # rank-offset seed: disjoint z shards
    z = torch.from_numpy(rs.uniform(-1, 1, (n_batches, B * G) + nets.g_input_shape(A)).astype(np.float32)).to(dev)   # z, or source images
    pools = [torch.empty((world * B * G,) + tuple(A["img"]), dtype=torch.float32, device=dev) if use_dist else None
             for _ in engines]                                           # one node-wide pool buffer per batch in flight

    done_ev = []                                                        # one event per timed step, recorded on the step's stream (no host sync)

    def step(i):
        e, st = engines[i % len(engines)], streams[i % len(engines)]
        with torch.cuda.stream(st):
            img = e.refine_from_z(z[i], Ksteps, args.rate)[0]
            if use_dist:
                dist.all_gather_into_tensor(pools[i % len(engines)], img)      # RCCL over xGMI: the refined sample pool
            if i >= args.warmup - 1 and not os.environ.get("CGS_BENCH_NO_EVENTS"):
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(st)
                done_ev.append(ev)
        return img

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    live_profile = rank == 0 and not args.graph and len(engines) == 1
    if live_profile:                      # one batch in flight: per-launch HIP events inside the timed region
        K.PROFILE = {}
        K.PROFILE_BY_LAYER = args.by_layer
    t0 = time.perf_counter()
    for i in range(args.warmup, n_batches):
        step(i)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    prof, K.PROFILE = K.PROFILE, None
    prof_ms, prof_note = dt * 1e3, "HIP events around every launch inside the timed region"
    if rank == 0 and not live_profile:
        # several batches in flight: kernels of different streams overlap, so a per-launch duration taken inside the
        # timed region would include the other stream's work; and a replayed hipGraph has no per-launch host hook.
        # Time ONE more step alone on one stream instead, launched eagerly (the same kernels with the same arguments).
        prof_eng = engines[0] if not args.graph else RefineEngine(args.arch, P, B * G, dev, use_graph=False,
                                                                  sync_bn=True if args.sync_bn else None, bn_groups=G)
        if args.graph:                                                   # (untimed first pass: packs weights, sizes workspaces)
            with torch.cuda.stream(streams[0]):
                prof_eng.refine_from_z(z[args.warmup], Ksteps, args.rate)
            torch.cuda.synchronize(dev)
        K.PROFILE = {}
        K.PROFILE_BY_LAYER = args.by_layer
        tp = time.perf_counter()
        with torch.cuda.stream(streams[0]):
            prof_eng.refine_from_z(z[args.warmup], Ksteps, args.rate)
        torch.cuda.synchronize(dev)
        prof_ms = (time.perf_counter() - tp) * 1e3
        prof, K.PROFILE = K.PROFILE, None
        prof_note = "HIP events around every launch of one extra single-stream step right after the timed region" + (" (launched eagerly; the timed region replays hipGraphs)" if args.graph else "")
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = world * B * G * args.steps / dt
        flops_per_sample = nets.refine_flops_per_sample(args.arch, Ksteps)
        out = {
            "metric": f"refined samples/sec @ {Ksteps} refinement steps", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.arch} collaborative refinement (propose + K-step refine + render), "
                                   f"batch {B}/GPU{f' (x{G} logical batches per launch, batch-norm statistics per logical batch)' if G > 1 else ''}, "
                                   f"K={Ksteps}, momentum rate {args.rate}, refine at feature "
                                   f"{list(A['feature'])}, random-init weights, z~U(-1,1) seed 2019+rank",
                       "global_batch": world * B * G, "refine_steps": Ksteps, "parallelism": f"z-shards x{world} + RCCL all-gather of the pool" if world > 1 else "single GPU",
                       "hipgraph": bool(args.graph), "batches_in_flight": len(engines) * G, "fused_per_launch": G, "sync_bn": bool(args.sync_bn)},
            "algorithmic_tflops": round(value * flops_per_sample / 1e12, 2),
        }
        ns = len(engines)
        if len(done_ev) >= 3 * ns:   # SURVEY.md 8d "median of >= 10": median gap between a stream's consecutive step completions
            gaps = sorted(a.elapsed_time(b) for a, b in zip(done_ev[:-ns], done_ev[ns:]))     # (GPU timestamps; the ns batches in flight
            out["ms_per_step_median"] = round(gaps[len(gaps) // 2] / ns, 3)                    # finish together, so per stream, / ns)
        if prof:
            dom = max(prof.items(), key=lambda kv: sum(a.elapsed_time(b) for a, b in kv[1][1]))
            per = {}
            for name, (fl, evs, ex) in prof.items():
                ms = sum(a.elapsed_time(b) for a, b in evs)
                per[name] = {"launches": len(evs), "avg_us": round(1e3 * ms / len(evs), 2), "tflops": round(fl / ms / 1e9, 2),
                             "executed_tflops": round(ex / ms / 1e9, 2), "share_of_step": round(ms / prof_ms, 3)}
            name, (fl, evs, ex) = dom
            ms = sum(a.elapsed_time(b) for a, b in evs)
            ach = fl / ms / 1e9
            # "achieved" divides the ALGORITHMIC flop count (zero-padding taps included, the contract's definition) by the
            # launch time; "executed_frac" counts only what the kernel issues to the matrix cores: it skips the K tiles of
            # taps that fall into the padding on the <= 16x16 grids (exact: they add zeros), 7-28 % of those layers
            out["roofline"] = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_FP32_MATRIX_TFLOPS,
                               "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_MATRIX_TFLOPS, 4),
                               "executed": round(ex / ms / 1e9, 2), "executed_frac": round(ex / ms / 1e9 / PEAK_FP32_MATRIX_TFLOPS, 4),
                               "traffic": measured_traffic(name) if args.arch == "dcgan64" and B == 1024 else None,
                               "launches": len(evs), "avg_launch_us": round(1e3 * ms / len(evs), 2), "timing": prof_note,
                               "flop_per_launch_avg": round(fl / len(evs), 0), "executed_flop_per_launch_avg": round(ex / len(evs), 0),
                               "note": "achieved/frac = ALGORITHMIC flops (zero-padding taps included) / launch time, so it can pass 1.0: the kernel "
                                       "skips the K tiles of padding taps (exact); executed/executed_frac = what is issued to the matrix cores"}
            out["kernels"] = per
        if world == 1 and not args.no_other_configs:
            out["other_configs"] = other_configs(dev, args.arch)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.arch, Ksteps, args.rate)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
