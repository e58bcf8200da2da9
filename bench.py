#!/usr/bin/env python3
"""bench.py -- refined samples/sec @ K refinement steps (BASELINE.json metric) on N MI355X GPUs.

A "step" = one pass of the hot path over one z-batch: propose (G head) + K-step collaborative
refinement (K+1 G-tail/D forwards, K backward-datas, momentum update, best-sample select) + final
render, images left in HBM.  Default workload = BASELINE configs[2]: DCGAN CelebA 64x64, batch 1024
per GPU, K = 20 (the configuration the 10k samples/s target is quoted on; it fits one GPU).
Inputs (z batches) and weights are resident in HBM before the timed region.  Two steps are in flight per GPU
(--streams), and for the small configurations several logical batches share one launch, each with its own
batch-norm statistics (--fuse; dcgan32 8 x 256, mnist 32 x 64: ~2048 samples per launch): the work and the results of a step are unchanged.
The K-step program of a step is replayed as a hipGraph (the launch-bound inner loop: ~1300 dependent launches per step;
--no-graph launches them one by one); the per-kernel HIP-event timing behind `roofline` then comes from one extra step
launched eagerly on one stream right after the timed region (the same kernels with the same arguments).

N > 1: one process per GPU (torch.distributed, backend nccl = RCCL), independent z-batches per rank
(seed 2019+rank, weak scaling), and ONE RCCL all-gather per step of the refined images into the
node-wide sample pool -- inside the timed region, and timed by itself with HIP events (`dist.gather_ms_per_step`).
The `dist` object of the line says what the collective layer really saw: backend, dist.get_world_size(), the number of
distinct ranks an all-gather of rank ids returned, pool bytes, per-rank throughput.
`--backend gloo --share-gpu` is the debug transport: all ranks on device 0, the pool staged through the host -- the whole
N-rank control flow (self-launch, rank-offset seeds, gather, MAX-reduce, one JSON line) on a 1-GPU box.

Besides the driver's keys the line carries: `roofline` (dominant kernel, flops issued / HIP-event launch time / the peak of the instructions it runs on), `kernels`,
`hbm`, `cpu_baseline`, `other_configs` (mnist, dcgan32, cyclegan256, synthetic2d: samples/s, roofline, cpu_baseline and the bx6 samples/s of each), `class_surface`
(the reference's verbatim Refiner wiring, engine and generic path) and `bx6` (the opt-in split-bf16 contraction on the headline's workload: never the headline).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node 8 ... bench.py --gpus 8 --steps 3 --warmup 1
    python bench.py --gpus 2 --backend gloo --share-gpu --arch mnist --steps 2 --warmup 1      (1-GPU box)
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
PEAK_HBM_GBPS = 8000.0              # same guide: HBM3E ~ 8 TB/s
PEAK_BF16_MATRIX_TFLOPS = 2500.0    # same guide: "Peak BF16/FP16 MFMA ~2.5 PF dense"
# the opt-in split-bf16 contraction (--contraction bx6, csrc/igemm_bx6.hip) issues SIX bf16 multiply-accumulates per fp32 one: its
# kernels' fp32-equivalent rate is priced against the dense bf16 peak / 6
PEAK_BX6_TFLOPS = PEAK_BF16_MATRIX_TFLOPS / 6.0


def kernel_peak(name):
    return PEAK_BX6_TFLOPS if name.startswith("igemm_bx6") else PEAK_FP32_MATRIX_TFLOPS
# HBM-side bytes per launch from rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs of this same
# command, FETCH_SIZE doubled per the gfx950 correction of the guide); collected offline, see profiles/README.md
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "traffic.json")


# HBM-bound multi-kernel entry points -> the kernels one call launches (rocprofv3 names; the last one runs once per call)
HBM_OPS = {
    "bn_train_lrelu_fwd_from_partials": ["bn_slice_sums_kernel", "bn_finalize_slices_kernel", "bn_apply_fwd_kernel"],
    "bn_train_lrelu_bwd_data": ["bn_partial_kernel<1>", "bn_finalize_kernel<1>", "bn_apply_bwd_kernel"],
    # (the fused configurations' norms: the same kernels per group of rows; where a configuration also runs an un-fused norm -- mnist's bn
    # behind its linear layer -- the forward kernels' per-launch averages mix the two tensor sizes: an approximation there)
    "groupnorm_lrelu_fwd_from_partials": ["bn_finalize_kernel<0>", "bn_apply_fwd_kernel"],
    "instnorm_lrelu_bwd_data": ["bn_partial_kernel<1>", "bn_finalize_kernel<1>", "bn_apply_bwd_kernel"],
    # (round 6: the sums pass rides on the backward-data launch of the conv above the norm -- cgs_*_bwd_data_nstats -- where the library offers it)
    "norm_lrelu_bwd_from_partials": ["bn_finalize_kernel<1>", "bn_apply_bwd_kernel"],
    "refine_update": ["refine_update_kernel"],
    "linear_out1_fwd": ["linear_out1_fwd_kernel"],
    "linear_out1_bwd": ["linear_out1_bwd_kernel"],
}
# logical batches fused per launch by default (tools/sessions/r03_sweep_fuse.sh on MI355X, two launches in flight: mnist 16 -> 24.3 k,
# 24 -> 26.1 k, 32 -> 26.3 k, 40 -> 27.1 k samples/s; dcgan32 4 -> 26.5 k, 6 -> 27.5 k, 8 -> 28.6 k): the tails and the per-launch
# fixed costs of the ~13-27 GFLOP layers amortise over more rows
# (config 5's instance norms have no batch coupling, so any number of samples per launch would be exact, but 8 / 16 / 32 / 64 rows x 4
# calls in flight gave 170.3 / 172.5 / 172.2 / 169.0 samples/s in one session and 171-174 / 173 / 173-175 in another: it keeps its own 8)
FUSE = {"dcgan32": 8, "mnist": 32}
# engine calls in flight per GPU (one engine + HIP stream each).  dcgan64's 700-us launches: 1 -> 6258, 2 -> 6640..6720, 3 -> +0.5 %,
# 4 -> -0.7 % (round 2).  The small configurations' short launches gain from more (round 3, same session: cyclegan256 2 -> 155.3,
# 3 -> 163.8, 4 -> 166.5, 6 -> 164.2 samples/s; mnist 28.1 / 29.4 / 29.8 / 29.1 k; dcgan32 28.7 / 29.4 / 29.6 / 29.1 k)
IN_FLIGHT = {"dcgan64": 2, "dcgan32": 4, "mnist": 4, "cyclegan256": 4}
THREE_CHANNEL = ("convt_rows_kernel", "conv_patch2_kernel", "conv_patch_kernel", "convt_quad", "convt_taps_kernel", "conv_taps_kernel")


def lib_stamp():
    """Which library produced the line: the sha256 of the kernel sources the LOADED .so was built from -- its embedded stamp
    (``cgs_source_sha``, csrc/stamp.hip), not a hash of whatever sources lie beside it: the key profiles/traffic.json is valid for --,
    whether that stamp differs from the tree (``stale``: only loadable at all under CGS_LIB / CGS_ALLOW_STALE), the ABI version the
    loaded object reports, and whether CGS_LIB pointed the run at another build."""
    from cgs_amd import lib as L
    lib = L.load()
    return {"source_sha16": L.built_from()[:16], "stale": L.stale, "cgs_version": int(lib.cgs_version()), "so": os.path.relpath(L.LIB_PATH, ROOT),
            "cgs_lib_override": bool(os.environ.get("CGS_LIB"))}


def _traffic_table():
    """profiles/traffic.json -- or {} when it was collected on other kernel sources than the ones this run executes (a stale
    number is worse than none), or when CGS_LIB points the run at some other build of the library."""
    try:
        from cgs_amd.lib import built_from
        if os.environ.get("CGS_LIB"):
            return {}
        with open(TRAFFIC_JSON) as f:
            t = json.load(f)
        return t if t.get("_source_sha256") == built_from() else {}
    except (OSError, ValueError):
        return {}


def measured_traffic(name, table=None, arch=None):
    """HBM bytes per launch of a kernel (or per call of an HBM_OPS entry point: its kernels' bytes summed) from the committed
    PMC passes; None when unknown.  ``arch``: the table of a non-headline configuration (``_by_arch`` of traffic.json)."""
    t = _traffic_table() if table is None else table
    if arch is not None:
        t = t.get("_by_arch", {}).get(arch, {})
    if name in HBM_OPS:
        ks = HBM_OPS[name]
        if not all(k in t for k in ks):
            return None
        calls = t[ks[-1]]["launches"]
        return int(sum(t[k]["hbm_bytes_per_launch_corrected"] * t[k]["launches"] for k in ks) / max(1, calls))
    return t.get(name, {}).get("hbm_bytes_per_launch_corrected")


def profile_records(prof, prof_ms, step_ms, timing_note, with_traffic, prof_steps=1, traffic_arch=None):
    """The per-kernel HIP-event records of one profiled step -> (roofline, kernels, hbm, executed flops of the step).

    roofline: the dominant kernel.  ``achieved`` / ``frac`` count what is ISSUED to the matrix cores: the implicit-GEMM
    kernels skip the K tiles of taps that fall into the zero padding (exact -- they only add zeros), so the dense count of
    SURVEY.md 8(d) (``nominal`` / ``nominal_frac``) bills 7-28 % work that no exact implementation has to do and can exceed
    the machine's peak; frac <= 1 by construction.
    hbm: the HBM-bound launches (3-channel conv kernels, batch-norm passes, the momentum update): algorithmic bytes (every
    operand once) / launch time against the 8 TB/s peak, and the PMC-measured bytes over the algorithmic ones."""
    table = _traffic_table() if with_traffic else {}
    per, hbm, ex_total, ideal_ms = {}, {}, 0.0, 0.0
    for name, (fl, evs, ex, nb) in prof.items():
        ms = sum(a.elapsed_time(b) for a, b in evs)
        ex_total += ex
        ideal_ms += ex / kernel_peak(name) / 1e9            # what this kernel's issued flops take at the peak of the instructions it runs on
        n = len(evs)
        if fl > 0 and name not in ("linear_out1_fwd", "linear_out1_bwd"):
            per[name] = {"launches": n, "avg_us": round(1e3 * ms / n, 2), "tflops": round(ex / ms / 1e9, 2),
                         "nominal_tflops": round(fl / ms / 1e9, 2), "share_of_step": round(ms / prof_ms, 3)}
        if fl == 0 or name.startswith(THREE_CHANNEL) or name.startswith("linear_out1"):
            gbps = nb / ms / 1e6
            tr = measured_traffic(name, table, traffic_arch)
            hbm[name] = {"launches": n, "avg_us": round(1e3 * ms / n, 2), "algorithmic_bytes": int(nb / n),
                         "achieved": round(gbps, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": round(gbps / PEAK_HBM_GBPS, 4),
                         "traffic": tr, "traffic_over_algorithmic": round(tr / (nb / n), 3) if tr else None,
                         "share_of_step": round(ms / prof_ms, 3)}
    mf = {k: v for k, v in prof.items() if v[0] > 0}
    name, (fl, evs, ex, nb) = max(mf.items(), key=lambda kv: sum(a.elapsed_time(b) for a, b in kv[1][1]))
    ms = sum(a.elapsed_time(b) for a, b in evs)
    n = len(evs)
    tr = measured_traffic(name, table, traffic_arch)
    peak = kernel_peak(name)
    roof = {"kernel": name, "bound": "mfma", "achieved": round(ex / ms / 1e9, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(ex / ms / 1e9 / peak, 4),
            "nominal": round(fl / ms / 1e9, 2), "nominal_frac": round(fl / ms / 1e9 / peak, 4),
            "traffic": tr, "algorithmic_bytes": int(nb / n), "traffic_over_algorithmic": round(tr / (nb / n), 3) if tr else None,
            "launches": n, "avg_launch_us": round(1e3 * ms / n, 2), "share_of_step": round(ms / prof_ms, 3), "timing": timing_note,
            "profile_passes": {"kept": "median", "wall_ms": list(PROFILE_PASSES_MS)} if PROFILE_PASSES_MS else None,
            "flop_per_launch": round(ex / n, 0), "nominal_flop_per_launch": round(fl / n, 0),
            "step_executed_tflops": round(ex_total / prof_steps / step_ms / 1e9, 2),
            "step_executed_frac": round(ideal_ms / prof_steps / step_ms, 4),
            "note": "achieved/frac = flops issued to the matrix cores / launch time (padding taps the kernel skips exactly are not "
                    "counted, so frac <= 1); nominal* = the dense count of SURVEY 8(d), padding taps included; step_executed_tflops = all "
                    "contraction launches of one step / the measured ms_per_step; step_executed_frac = the time those launches would take "
                    "at the peak of the instructions they run on / ms_per_step"}
    if peak != PEAK_FP32_MATRIX_TFLOPS:
        roof["peak_note"] = "split-bf16 kernel: fp32-equivalent flops against the dense bf16 MFMA peak / 6 (six bf16 products per fp32 product)"
    return roof, per, hbm, ex_total / prof_steps


PROFILE_PASSES_MS = []          # walls of the last profile_one_step's three passes (the median one is the record)
_CPU_THREADS = [None]          # the calibrated thread count of the headline's cpu_baseline, reused for the other configs


def cpu_baseline(arch, refine_steps, rate, seconds=15.0, max_batch=512):
    """The CPU oracle (torch-CPU fp32 restatement of collaborator.build_refiner) timed on the host cores
    on a bounded sample of the same workload (same net, same K, a smaller batch: work is linear in B)."""
    from oracle import nets_ref as N
    from oracle import sampling_ref as S
    P = N.init_params(arch, 2019, True)
    A = N.ARCHS[arch]
    in_shape = tuple(A["g_in"]) if A.get("g_in") else (A["z_dim"],)
    gt, dd = (lambda f: N.feature_to_data(arch, P, f)), (lambda x: N.discriminator(arch, P, x))

    def run(B, K):
        z = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (B,) + in_shape).astype(np.float32))
        with torch.no_grad():
            f0 = N.input_to_feature(arch, P, z)
        t = time.time()
        S.collaborative_refine(f0, gt, dd, K, rate)
        return time.time() - t
    unit = 2 if arch == "cyclegan256" else 32                  # pilot batch
    run(min(4, unit), 1)                                       # warm the thread pool / allocator
    # more threads is not faster on these layer sizes (8 cores beat 128 on the first boxes measured): calibrate the
    # thread count once (on the headline's net) on a small run and time the samples with the best one
    ncpu = os.cpu_count() or 1
    if _CPU_THREADS[0] is None:
        best_n, best_t = torch.get_num_threads(), None
        for n in sorted({c for c in (8, 16, 32, 64, ncpu) if c <= ncpu}):
            torch.set_num_threads(n)
            run(8 if unit > 8 else unit, 1)
            t = run(unit, 1)               # (a 32-sample run is long enough to rank the thread counts reliably)
            if best_t is None or t < best_t:
                best_n, best_t = n, t
        _CPU_THREADS[0] = best_n
    best_n = _CPU_THREADS[0]
    torch.set_num_threads(best_n)
    # size the sample for ~`seconds` of CPU work from a pilot at the full K
    pilot = run(unit, refine_steps)
    batch = int(min(max_batch, max(unit, unit * round(seconds / pilot))))
    dt = run(batch, refine_steps) if batch != unit else pilot
    return {"value": round(batch / dt, 3), "unit": "samples/s", "cores": best_n, "kind": "port",
            "sample_short": f"oracle (torch-CPU fp32) {arch} batch {batch} K={refine_steps}, {dt:.1f} s, {best_n} of {ncpu} threads",
            "sample": f"oracle.collaborative_refine (torch-CPU fp32), {arch}, batch {batch}, K={refine_steps}, "
                      f"{dt:.1f} s wall, {best_n} of {ncpu} host threads (fastest of a 8/16/32/64/all calibration; a reported "
                      f"baseline that moves 10-20 % from run to run with the box's other tenants, never a target)"}


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
    # Which HIP streams the batches in flight ride on matters here: the runtime folds streams onto a few hardware queues, and with
    # 3-launch steps of microseconds each, two "streams" on one queue simply serialise.  Measured in fresh processes (round 4): the first
    # 2 / 3 / 4 / 6 / 8 / 16 streams of torch's pool give 2.55 / 3.81 / 2.57 / 3.81 / 3.41 / 4.05 M samples/s, the SECOND eight 5.05 M with
    # the same tensors (round 3's line happened to get those: 5.05 M).  So: three candidate sets of n_streams pool streams, a short untimed
    # trial on each, the timed run on the fastest.
    sets = [[torch.cuda.Stream(dev) for _ in range(n_streams)] for _ in range(3)]
    streams = sets[0]

    def step(i):
        with torch.cuda.stream(streams[i % n_streams]):
            base = D.sigmoid_and_saliency(real, want_saliency=False)[0].mean()        # np.mean(real_sigmoid), refiner_cpu.py:23,28 (stays on the device)
            return D.refine(x[i], base, Ksteps, rate, "ladam")[0]
    def trial_ms(cand, m=min(n, 64)):
        nonlocal streams
        streams = cand
        torch.cuda.synchronize(dev)
        tt = time.perf_counter()
        for i in range(m):
            step(i)
        torch.cuda.synchronize(dev)
        return time.perf_counter() - tt
    for cand in sets:                                 # round 1: allocator, clocks, lazily created queues -- discarded
        trial_ms(cand)
    times = [trial_ms(cand) for cand in sets]         # round 2 counts
    streams = sets[times.index(min(times))]
    for i in range(warmup):
        step(i)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(warmup, n):
        step(i)
    torch.cuda.synchronize(dev)
    return time.perf_counter() - t0, (S, Ws, bs, x, real)


def cpu_baseline_synthetic2d(S, Ws, bs, x, real, B, Ksteps, rate, reps=20):
    """BASELINE config 1 on the host: oracle.refine_2d (refiner_cpu.manipulate_sample restated, torch-CPU D) on one of the timed batches."""
    fake = x[0].cpu().numpy(); rb = real.cpu().numpy().astype(np.float64)
    d_fn = lambda v: S.mlp_sigmoid_and_saliency(Ws, bs, v)
    keep = torch.get_num_threads()
    torch.set_num_threads(min(8, os.cpu_count() or 1))          # 64-wide MLP layers: more threads only add overhead
    try:
        S.refine_2d(fake, rb, d_fn, Ksteps, rate, "ladam")
        t = time.time()
        for _ in range(reps):
            S.refine_2d(fake, rb, d_fn, Ksteps, rate, "ladam")
        c = (time.time() - t) / reps
        return {"value": round(B / c, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                "sample_short": f"oracle.refine_2d (torch-CPU D) batch {B} K={Ksteps}, {reps} reps",
                "sample": f"oracle.refine_2d (refiner_cpu.manipulate_sample restated; torch-CPU D), batch {B}, K={Ksteps}, {reps} reps"}
    finally:
        torch.set_num_threads(keep)


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
               "roofline": None, "lib": lib_stamp()}
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_synthetic2d(S, Ws, bs, x, real, B, Ksteps, args.rate)
        emit(out, args.detail)


def profile_one_step(arch, P, B, G, Ksteps, rate, z1, dev, stream, by_layer=False, sync_bn=None, engine=None):
    """One step launched eagerly on ONE stream with HIP events around every launch (kernels.PROFILE): what `roofline`,
    `kernels` and `hbm` are computed from when the timed region overlaps streams or replays hipGraphs."""
    from cgs_amd import kernels as K
    from cgs_amd.engine import RefineEngine
    eng = engine if engine is not None else RefineEngine(arch, P, B * G, dev, use_graph=False, sync_bn=sync_bn, bn_groups=G)
    # an engine of the timed region is launched eagerly here on the SAME buffers its hipGraph replays (a fresh engine's fresh
    # allocations read the dominant kernel 4-5 % slower than the kernel table of the profiled runs: 702-718 against 678-684 us)
    was_graph, eng.use_graph = eng.use_graph, False
    try:
        with torch.cuda.stream(stream):
            eng.refine_from_z(z1, Ksteps, rate)                         # (untimed first pass: packs weights, sizes workspaces)
        torch.cuda.synchronize(dev)
        # three profiled passes, the MEDIAN one (by wall time) kept: a single pass now and then catches a transient (round 5: one kernel
        # at 1.5x its usual time in ONE pass of one run, 749 -> 1109 us, nothing of it in the timed region or in the rocprofv3 table of
        # the same box); the fastest of several would bias the roofline upward against the mean-based headline it explains.
        # The averages are those of the kept pass; `profile_passes` in the record lists all three walls.
        passes = []
        for _ in range(3):
            K.PROFILE, K.PROFILE_BY_LAYER = {}, by_layer
            tp = time.perf_counter()
            with torch.cuda.stream(stream):
                eng.refine_from_z(z1, Ksteps, rate)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - tp) * 1e3
            prof, K.PROFILE = K.PROFILE, None
            passes.append((ms, prof))
        passes.sort(key=lambda t: t[0])
        ms, prof = passes[1]
        PROFILE_PASSES_MS[:] = [round(t[0], 2) for t in passes]
    finally:
        K.PROFILE = None
        eng.use_graph = was_graph
    return prof, ms


def other_configs(dev, skip, want_cpu):
    """The other single-GPU configurations ("always report MNIST"; BASELINE configs[0], [1]) at their bench.py
    defaults, timed AFTER and OUTSIDE the headline's timed region: a handful of steps each, with their own `roofline`
    (one extra eager single-stream step, as for the headline) and a short `cpu_baseline`."""
    from cgs_amd import nets
    from cgs_amd.engine import RefineEngine
    out = {}
    for arch, B, Ksteps, G, steps in (("mnist", 64, 50, FUSE["mnist"], 8), ("dcgan32", 256, 20, FUSE["dcgan32"], 8),
                                      ("cyclegan256", 8, 20, 1, 12)):          # (BASELINE config 5's per-GPU share: 64 over 8 GPUs)
        if arch == skip:
            continue
        A = nets.ARCHS[arch]
        P = nets.init_params(arch, dev, seed=2019)
        nf = IN_FLIGHT[arch]
        streams = [torch.cuda.Stream(dev) for _ in range(nf)]
        z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (steps + nf, B * G) + nets.g_input_shape(A)).astype(np.float32)).to(dev)

        def timed(contraction):
            engs = [RefineEngine(arch, P, B * G, dev, use_graph=True, bn_groups=G, contraction=contraction) for _ in range(nf)]

            def step(i):
                with torch.cuda.stream(streams[i % nf]):
                    engs[i % nf].refine_from_z(z[i], Ksteps, 0.1)
            for i in range(nf):
                step(i)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for i in range(nf, steps + nf):
                step(i)
            torch.cuda.synchronize(dev)
            return engs, time.perf_counter() - t0
        # (the opt-in split-bf16 contraction first, as a bare samples/s; then the exact-fp32 default, whose engines the roofline step re-uses)
        _, dt_bx6 = timed("bx6")
        torch.cuda.empty_cache()
        engines, dt = timed("f32")
        out[arch] = {"samples_per_s": round(B * G * steps / dt, 1), "batch": B, "refine_steps": Ksteps, "fused_per_launch": G,
                     "batches_in_flight": nf * G, "steps": steps, "hipgraph": True,
                     "algorithmic_tflops": round(B * G * steps / dt * nets.refine_flops_per_sample(arch, Ksteps) / 1e12, 2)}
        prof, prof_ms = profile_one_step(arch, P, B, G, Ksteps, 0.1, z[0], dev, streams[0], engine=engines[0])
        del engines
        # (traffic: the PMC passes of THIS configuration at these very launch sizes, profiles/traffic.json `_by_arch`; null while the
        # table belongs to other kernel sources)
        roof, _, hbm, _ = profile_records(prof, prof_ms, 1e3 * dt / steps, "an extra eager single-stream step (the median of three passes)", True, traffic_arch=arch)
        roof.pop("note")
        out[arch]["roofline"] = roof
        out[arch]["hbm"] = {k: {f: v[f] for f in ("avg_us", "achieved", "frac", "share_of_step", "traffic", "traffic_over_algorithmic")} for k, v in hbm.items()}
        out[arch]["bx6"] = {"samples_per_s": round(B * G * steps / dt_bx6, 1), "dtype": DTYPE["bx6"], "contraction": "bx6",
                            "note": "opt-in (--contraction bx6): the layers with a multiple of 64 output channels, Cred % 32 == 0 and GPU-filling grids (>= 256 blocks, K >= 512) on split-bf16 MFMA"}
        if want_cpu:
            out[arch]["cpu_baseline"] = cpu_baseline(arch, Ksteps, 0.1, seconds=4.0, max_batch=1024)
        del z, P
        torch.cuda.empty_cache()
    if skip != "synthetic2d":
        dt, (S, Ws, bs, x, real) = run_synthetic2d(dev, 0, 512, 10, 0.1, 256, 16, 8)
        out["synthetic2d"] = {"samples_per_s": round(512 * 256 / dt, 1), "batch": 512, "refine_steps": 10, "method": "ladam",
                              "batches_in_flight": 8, "steps": 256, "roofline": None}
        if want_cpu:
            out["synthetic2d"]["cpu_baseline"] = cpu_baseline_synthetic2d(S, Ws, bs, x, real, 512, 10, 0.1)
    return out


def class_surface(dev, configs=(("dcgan64", 1024, 20, 6, 3), ("mnist", 64, 50, 24, 4), ("dcgan32", 64, 20, 24, 4), ("dcgan32", 256, 20, 16, 6))):
    """What a caller of the reference's CLASS SURFACE gets, as opposed to the engines the headline drives directly:
    ``model.GAN`` + ``collaborator.Refiner`` wired with the very lines of nsgan/GAN.py:171-181 -- a ``functools.partial`` of the
    discriminator and a local loss closure -- then per z batch ``input_to_feature`` (operator API) and ``build_refiner``: one batch
    at a time on one stream, hipGraph replay (Refiner.use_graph's default).  ``generic``: the same wiring with the discriminator
    wrapped in a lambda, which the engine detection cannot see through: the ops + torch.autograd loop (the same HIP kernels,
    nothing fused across layers), captured into a hipGraph at its second call and replayed (round 6: the two untimed warm-up
    calls are the eager one and the capturing one).  Timed after and outside the headline's timed region."""
    from functools import partial
    from cgs_amd import nets, ops
    from cgs_amd.model import GAN
    from cgs_amd.sampling.collaborator import Refiner
    out = {"wiring": "nsgan/GAN.py:171-181 verbatim: Refiner(K, rate).set_env(partial(gan.discriminator, is_training=True, reuse=True), "
                     "gan.feature_to_data, <local BCE-vs-ones closure>); per batch gan.input_to_feature(z) + refiner.build_refiner(feature, "
                     "inputs, 'deterministic'); one batch in flight, one stream"}
    for arch, B, Ksteps, steps, gsteps in configs:
        A = nets.ARCHS[arch]
        ops.reset_variables()
        self = GAN(arch, batch_size=B, device=dev, params=nets.init_params(arch, dev, seed=2019))
        rollout_steps, rollout_rate = Ksteps, 0.1
        discriminator_refine = partial(self.discriminator, is_training=True, reuse=True)

        def loss_refine(logits):
            return ops.sigmoid_cross_entropy_with_logits(logits=logits, labels=ops.ones_like(logits))
        refiner = Refiner(rollout_steps=rollout_steps, rollout_rate=rollout_rate)
        refiner.set_env(discriminator_refine, self.feature_to_data, loss_refine)
        generic = Refiner(rollout_steps=rollout_steps, rollout_rate=rollout_rate)
        generic.set_env(lambda x: self.discriminator(x, is_training=True, reuse=True), self.feature_to_data, loss_refine)
        n = max(steps, gsteps) + 2
        z = torch.from_numpy(np.random.RandomState(2019).uniform(-1, 1, (n, B) + nets.g_input_shape(A)).astype(np.float32)).to(dev)
        inputs = torch.empty((B,) + tuple(A["img"]), dtype=torch.float32, device=dev).uniform_(-1, 1)
        rec = {"batch": B, "refine_steps": Ksteps}
        for name, r, k in (("engine", refiner, steps), ("generic", generic, gsteps)):
            import warnings
            with torch.no_grad(), warnings.catch_warnings():
                warnings.simplefilter("ignore", RuntimeWarning)          # (the generic leg is meant to take the generic path: no need to be told)
                for i in range(2):                                   # first call: packs, sizes workspaces, captures the graph
                    r.build_refiner(self.input_to_feature(z[i]), inputs, mode="deterministic")
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(2, 2 + k):
                    r.build_refiner(self.input_to_feature(z[i]), inputs, mode="deterministic")
                torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            rec[name] = {"samples_per_s": round(B * k / dt, 1), "ms_per_batch": round(1e3 * dt / k, 3), "steps": k, "path": r.path,
                         "hipgraph": bool(r.use_graph) if r.path == "engine" else bool(r._generic_graphs)}
            if r.graph_fallback:
                rec[name]["hipgraph_fallback"] = r.graph_fallback
        if B == 64:
            # the reference's callers run the refiner one batch_size-64 batch per sess.run (nsgan/main.py:32, nsgan/GAN.py:270-272,398-426):
            # ``refiner.logical_batch = 64`` hands build_refiner G such batches at once -- one launch per layer, D's batch statistics and
            # the step bookkeeping per logical batch -- through the same verbatim wiring
            Gf = 32
            refiner.logical_batch = B
            zf = torch.from_numpy(np.random.RandomState(2020).uniform(-1, 1, (steps // 4 + 2, B * Gf) + nets.g_input_shape(A)).astype(np.float32)).to(dev)
            with torch.no_grad():
                for i in range(2):
                    refiner.build_refiner(self.input_to_feature(zf[i]), None, mode="deterministic")
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for i in range(2, len(zf)):
                    refiner.build_refiner(self.input_to_feature(zf[i]), None, mode="deterministic")
                torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            kf = len(zf) - 2
            rec["fused"] = {"samples_per_s": round(B * Gf * kf / dt, 1), "ms_per_call": round(1e3 * dt / kf, 3), "calls": kf, "logical_batch": B,
                            "logical_batches_per_call": Gf, "path": refiner.path, "hipgraph": bool(refiner.use_graph),
                            "how": f"refiner.logical_batch = {B}; build_refiner(feature[{B * Gf}], ...): one call, one stream"}
            refiner.logical_batch = None
            del zf
        # (keyed by the net; a second batch size of a net -- BASELINE config 2 at its literal batch 256, one call at a time -- as <net>_b<batch>)
        out[arch if arch not in out else f"{arch}_b{B}"] = rec
        del self, refiner, generic, z, inputs
        ops.reset_variables()
        torch.cuda.empty_cache()
    return out


def shaping_record(dev, configs=(("mnist", 64, 50, 30), ("dcgan64", 64, 20, 12))):
    """SURVEY 8f-2 measured: one iteration of the reference's D-shaping loop (nsgan/GAN.py:266-272) at the reference's batch size
    (nsgan/main.py:32) -- sess.run(g_refine_proba) then sess.run([d_optim, d_loss]) -- as ``shaping.shape_step``: probabilistic
    refine on the engine (hipGraph replay), one Adam step of D (two D passes with batch statistics, weight / bias / gamma / beta
    gradients, Adam), then ``refresh_weights`` (re-fold + re-pack for the next refine).  Strictly sequential by construction (every
    refine sees the D the previous step shaped): nothing to fuse across iterations.  ``kernels.wgrad_kernel``: the weight-gradient
    GEMMs of one D step (HIP events, eager), flops issued / launch time against the fp32 matrix peak."""
    from cgs_amd import kernels as K
    from cgs_amd import nets
    from cgs_amd.engine import RefineEngine
    from cgs_amd.shaping import DShaper, shape_step
    out = {}
    for arch, B, Ksteps, iters in configs:
        A = nets.ARCHS[arch]
        P = nets.init_params(arch, dev, seed=2019)
        eng = RefineEngine(arch, P, B, dev, use_graph=True)
        sh = DShaper(arch, P, B, dev, learning_rate=1e-5)                     # run_shaping.sh: --learning_rate 1e-5
        rs = np.random.RandomState(2019)
        z = torch.from_numpy(rs.uniform(-1, 1, (iters + 3, B) + nets.g_input_shape(A)).astype(np.float32)).to(dev)
        real = torch.from_numpy(rs.uniform(-1, 1, (B,) + tuple(A["img"])).astype(np.float32)).to(dev)
        idx = rs.randint(Ksteps + 1, size=B)                                  # ONE draw baked into the graph (collaborator.py:54-56)
        for i in range(3):
            shape_step(eng, sh, z[i], real, Ksteps, 0.1, indices=idx)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for i in range(3, 3 + iters):
            shape_step(eng, sh, z[i], real, Ksteps, 0.1, indices=idx)
        torch.cuda.synchronize(dev)
        it_ms = 1e3 * (time.perf_counter() - t0) / iters
        refined = eng.refine_from_z(z[0], Ksteps, 0.1, mode="probabilistic", indices=idx)[0].clone()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            sh.step(real, refined)
        torch.cuda.synchronize(dev)
        d_ms = 1e3 * (time.perf_counter() - t0) / iters
        t0 = time.perf_counter()
        for _ in range(iters):
            eng.refresh_weights()
        torch.cuda.synchronize(dev)
        refresh_ms = 1e3 * (time.perf_counter() - t0) / iters
        K.PROFILE = {}
        tp = time.perf_counter()
        sh.step(real, refined)
        torch.cuda.synchronize(dev)
        prof_ms = 1e3 * (time.perf_counter() - tp)
        prof, K.PROFILE = K.PROFILE, None
        eng.refresh_weights()
        kern = {}
        for name, (fl, evs, ex, nb) in prof.items():
            ms = sum(a.elapsed_time(b) for a, b in evs)
            if fl > 0 and ms > 0:
                kern[name] = {"launches": len(evs), "avg_us": round(1e3 * ms / len(evs), 2), "tflops": round(ex / ms / 1e9, 2),
                              "frac_of_fp32_matrix_peak": round(ex / ms / 1e9 / PEAK_FP32_MATRIX_TFLOPS, 4), "share_of_d_step": round(ms / prof_ms, 3)}
        out[arch] = {"batch": B, "refine_steps": Ksteps, "iterations": iters, "iteration_ms": round(it_ms, 3), "iterations_per_s": round(1e3 / it_ms, 2),
                     "samples_per_s": round(B * 1e3 / it_ms, 1), "d_step_ms": round(d_ms, 3), "refresh_weights_ms": round(refresh_ms, 3),
                     "refine_ms": round(it_ms - d_ms - refresh_ms, 3), "hipgraph": True, "kernels": kern}
        del eng, sh, P, z
        torch.cuda.empty_cache()
    out["note"] = ("nsgan/GAN.py:266-272 per iteration: probabilistic K-step refine (hipGraph replay) + one Adam step of D (real + refined pass, batch "
                   "statistics) + refresh (re-fold, re-pack); batch 64 as in nsgan/main.py:32: launch-latency-bound by construction")
    return out


def f1_fill(dev, eval_size=50000, B=64, Ksteps=50, G=None):
    """SURVEY 8f-1 measured: the reference's evaluate fill loop (nsgan/GAN.py:384-426, "collaborate": eval_size = 50 000 at
    batch_size 64, nsgan/main.py:32-33) on mnist -- score the real set, standard samples as the chain's first pass, then refine ->
    D-score -> MH independence chain (T = 20) until eval_size samples are accepted (MIN_EFFICIENCY = 0.2 cut-off) -- as
    ``evaluate.collaborate_fused`` over ``FusedProposer``: G logical batches per device round, the host chain one round behind.
    Random-init weights and uniform noise as the "real" set (no dataset here): the acceptance statistics are those of this D, the
    bookkeeping and the device work per proposal are the reference's."""
    from cgs_amd import nets, ops
    from cgs_amd.evaluate import MIN_EFFICIENCY, FusedProposer, collaborate_fused
    from cgs_amd.model import GAN
    from cgs_amd.sampling import IndependenceSampler
    G = G or FUSE["mnist"]
    ops.reset_variables()
    gan = GAN("mnist", batch_size=B, device=dev, params=nets.init_params("mnist", dev, seed=2019))
    prop = FusedProposer(gan, Ksteps, 0.1, batch=B, groups=G, depth=2)
    n_batch = eval_size // B
    n_eval = n_batch * B                                     # (the reference's int(eval_size / batch_size) whole batches, :386)
    np.random.seed(2019)                                     # nsgan/GAN.py:257
    real = np.random.RandomState(1).uniform(-1, 1, (n_eval, 28, 28, 1)).astype(np.float32)
    z0 = np.random.uniform(-1, 1, [prop.b * prop.G, gan.z_dim]).astype(np.float32)
    prop.result(prop.launch(z0)); prop.result(prop.launch(z0))            # both slots: graph capture, workspaces (untimed)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    sigmoid_real = prop.score_real(real)                                   # :388-390
    t1 = time.perf_counter()
    eval_z = np.random.uniform(-1, 1, [n_eval, gan.z_dim]).astype(np.float32)      # :391
    std_img = np.empty((n_eval, 28, 28, 1), dtype=np.float32)
    std_sig = np.empty((n_eval, 1), dtype=np.float32)
    n = prop.b * prop.G
    for i in range(0, n_eval, n):                                          # :394-400 (standard samples + their sigmoids), G batches per launch
        zb = eval_z[i:i + n]
        m = len(zb)
        if m < n:
            zb = np.concatenate([zb, np.zeros((n - m, gan.z_dim), np.float32)])
        im, sg = prop.result(prop.launch(zb, refine=False))
        std_img[i:i + m], std_sig[i:i + m] = im[:m], sg[:m]
    t2 = time.perf_counter()
    st = {}
    samples, eff = collaborate_fused(prop, IndependenceSampler(T=20), n_eval, float(np.mean(sigmoid_real)), base=(std_img, std_sig),
                                     min_efficiency=MIN_EFFICIENCY, stats=st)
    t3 = time.perf_counter()
    assert samples.shape[0] == n_eval and np.isfinite(samples).all()
    fill_s = t3 - t2
    rec = {"arch": "mnist", "eval_size": n_eval, "batch": B, "refine_steps": Ksteps, "logical_batches_per_round": G, "rounds_in_flight": 2,
           "accepted_samples_per_s": round(n_eval / fill_s, 1), "fill_s": round(fill_s, 3), "efficiency": round(eff, 4),
           "proposed": int(st["proposed"]), "proposals_per_s": round(st["proposed"] / fill_s, 1), "rounds": int(st["rounds"]),
           "host_chain_s": round(st["host_chain_s"], 3), "host_chain_share_of_wall": round(st["host_chain_s"] / fill_s, 3),
           "device_wait_s": round(st["device_wait_s"], 3), "score_real_s": round(t1 - t0, 3), "standard_pass_s": round(t2 - t1, 3),
           "note": "nsgan/GAN.py:384-426 (method 'collaborate'): real-set scoring, standard pass, then refine -> score -> MH chain (T=20) fill with "
                   "the MIN_EFFICIENCY=0.2 cut-off; the host chain of round r runs while the device refines round r+1; random-init weights, "
                   "uniform noise as the real set"}
    del prop, gan
    ops.reset_variables()
    torch.cuda.empty_cache()
    return rec


DTYPE = {"f32": "f32", "bx6": "f32 (3xbf16 split, fp32 accumulate)"}


def bx6_line(args, P, B, G, Ksteps, z, dev, n_flight, streams, flops_per_sample):
    """The headline's workload with RefineEngine(contraction="bx6"): the layers cgs_igemm_bx6_ok admits (N % 64 == 0, Cred % 32 == 0, >= 256 blocks, K >= 512) on split-bf16 MFMA
    (csrc/igemm_bx6.hip).  Same engines-in-flight / hipGraph structure and the same z batches as the f32 region; opt-in, reported
    beside the headline, priced against the bf16 peak / 6 for its kernels."""
    from cgs_amd.engine import RefineEngine
    engines = [RefineEngine(args.arch, P, B * G, dev, use_graph=bool(args.graph), bn_groups=G, contraction="bx6") for _ in range(n_flight)]

    def step(i):
        j = i % n_flight
        with torch.cuda.stream(streams[j]):
            engines[j].refine_from_z(z[i], Ksteps, args.rate)
    for i in range(max(args.warmup, n_flight)):
        step(i % len(z))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(args.warmup, args.warmup + args.steps):
        step(i)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    prof, prof_ms = profile_one_step(args.arch, P, B, G, Ksteps, args.rate, z[args.warmup], dev, streams[0], engine=engines[0])
    roof, kern, _, ex_step = profile_records(prof, prof_ms, 1e3 * dt / args.steps, "an extra eager single-stream step right after this mode's timed steps (the median of three passes)",
                                             args.arch == "dcgan64" and B == 1024 and G == 1)
    value = B * G * args.steps / dt
    return {"value": round(value, 2), "unit": "samples/s", "ms_per_step": round(1e3 * dt / args.steps, 3), "steps": args.steps,
            "dtype": DTYPE["bx6"], "contraction": "bx6", "hipgraph": bool(args.graph), "batches_in_flight": n_flight * G,
            "algorithmic_tflops": round(value * flops_per_sample / 1e12, 2), "executed_tflops": round(ex_step / (1e3 * dt / args.steps) / 1e9, 2),
            "roofline": roof, "kernels": {k: v for k, v in kern.items() if k.startswith("igemm")},
            "selected_by": "python bench.py --contraction bx6   (RefineEngine(..., contraction='bx6'); the C ABI: cgs_set_contraction)",
            "accuracy": "six of the nine products of an exact three-way bf16 split of both fp32 operands, fp32 accumulate: error of the "
                        "size of an fp32 fma chain's own; the full parity suite (operator tests at 2e-5, reference goldens, full-size "
                        "oracle cases) runs in this mode too (tests/conftest.py, `contraction`): identical bars except two that measure chaotic "
                        "amplification of rounding over K steps (K-step image drift 60x instead of 25x the trajectory tolerance; K=50 trajectory 1.25e-2)"}


LINE_BUDGET = 6000          # bytes: the driver keeps 8 kB of stdout tail (BENCH_r05's 22 kB line was cut and parsed as null)
DETAIL_PATH = os.path.join(ROOT, "bench_detail.json")
ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "nominal_frac", "traffic", "algorithmic_bytes", "traffic_over_algorithmic",
             "launches", "avg_launch_us", "share_of_step", "step_executed_frac", "profile_passes")
DIST_KEYS = ("backend", "world_size", "ranks_seen", "pool_bytes", "pool_buffers_per_rank", "pool_bytes_per_rank_total", "device_mem_free_before_pools",
             "device_mem_total", "gathers_per_step", "gather_ms_per_step", "per_rank_samples_per_s", "hipgraph_ranks", "pool_rows_match_ranks",
             "pool_rank_sums_distinct", "rank0_profile_step_wall_s")


def _short_cpu(c):
    """cpu_baseline with a sample description of <= 120 characters (the long form stays in the sidecar)."""
    return {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"], "sample": str(c.get("sample_short") or c["sample"])[:120]}


def compact_line(full, detail_path=None):
    """The ONE stdout line the driver parses, from the full record of a run: the contract keys, the dominant kernel's `roofline`,
    `cpu_baseline`, `lib`, `dist` (N > 1) and a `summary` of ONE number per extra measurement.  Everything else (kernel / hbm tables,
    per-configuration rooflines, notes) lives in the sidecar file the line names (`detail`).  Pure function of the record: the CPU suite
    holds it to LINE_BUDGET on a canned record (tests/test_host_cpu.py)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "algorithmic_tflops", "executed_tflops")
    line = {k: full[k] for k in keep if k in full}
    cfg = dict(full.get("config", {}))
    if len(cfg.get("workload", "")) > 240:
        cfg["workload"] = cfg["workload"][:240]
    if "hipgraph_fallback" in cfg:
        cfg["hipgraph_fallback"] = str(cfg["hipgraph_fallback"])[:160]
    line["config"] = cfg
    r = full.get("roofline")
    line["roofline"] = {k: r[k] for k in ROOF_KEYS if k in r} if r else None
    if full.get("cpu_baseline"):
        line["cpu_baseline"] = _short_cpu(full["cpu_baseline"])
    if "lib" in full:
        line["lib"] = full["lib"]
    if full.get("dist"):
        line["dist"] = {k: full["dist"][k] for k in DIST_KEYS if k in full["dist"]}
    s = {}
    if "bx6" in full:
        s["bx6"] = {"samples_per_s": full["bx6"]["value"], "roofline_frac": full["bx6"]["roofline"]["frac"],
                    "step_executed_frac": full["bx6"]["roofline"]["step_executed_frac"]}
    for arch, o in full.get("other_configs", {}).items():
        e = {"samples_per_s": o["samples_per_s"]}
        if o.get("roofline"):
            e["roofline_frac"] = o["roofline"]["frac"]
            e["step_executed_frac"] = o["roofline"]["step_executed_frac"]
        if o.get("bx6"):
            e["bx6_samples_per_s"] = o["bx6"]["samples_per_s"]
        if o.get("cpu_baseline"):
            e["cpu_samples_per_s"] = o["cpu_baseline"]["value"]
        s.setdefault("other_configs", {})[arch] = e
    for name, o in full.get("class_surface", {}).items():
        if isinstance(o, dict):
            s.setdefault("class_surface", {})[name] = {k: o[k]["samples_per_s"] for k in ("engine", "generic", "fused") if k in o}
    if "f1" in full:
        s["f1"] = {"accepted_samples_per_s": full["f1"]["accepted_samples_per_s"], "proposals_per_s": full["f1"]["proposals_per_s"],
                   "efficiency": full["f1"]["efficiency"]}
    if "shaping" in full:
        s["shaping_iteration_ms"] = {a: o["iteration_ms"] for a, o in full["shaping"].items() if isinstance(o, dict)}
    if s:
        line["summary"] = s
    if detail_path:
        line["detail"] = detail_path
    return line


def emit(full, detail_path):
    """Write the full record to the sidecar (best effort: a read-only tree costs the sidecar, not the line) and print the compact line."""
    rel = None
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(full, f, indent=1)
            rel = os.path.relpath(detail_path, ROOT)
        except OSError as ex:
            print(f"[bench] could not write {detail_path}: {ex}", file=sys.stderr, flush=True)
    line = compact_line(full, rel)
    text = json.dumps(line)
    assert len(text) < LINE_BUDGET + 2000, f"bench line grew to {len(text)} bytes: move the new object into the sidecar"
    print(text, flush=True)


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
                    help="replay the K-step program as a hipGraph (the default, at every world size: +1.3 %% on the dcgan64 headline, +14 %% on "
                         "mnist -- the launch gaps between the ~1300 dependent kernels of a step otherwise only partly hide behind the other "
                         "batch in flight).  By default a failed capture falls back to eager launches in the same process and the line says so "
                         "(config.hipgraph false + config.hipgraph_fallback); with --graph given explicitly a failed capture is an error")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="launch every kernel eagerly")
    ap.add_argument("--streams", type=int, default=0,
                    help="(default: dcgan64 2, the other nets 4, synthetic2d 8) engine calls in flight per GPU, one engine + HIP stream each: the tail "
                         "/ small kernels of one batch overlap the big kernels of the other (IN_FLIGHT above holds the measurements)")
    ap.add_argument("--fuse", type=int, default=0,
                    help="logical batches fused into one engine batch (conv launches G times larger, batch-norm statistics kept "
                         "per logical batch).  Default: dcgan64 1, dcgan32 8, mnist 32 (~2048 samples per launch; measured sweep in DESIGN.md 6)")
    ap.add_argument("--sync-bn", action="store_true",
                    help="treat the N ranks' batches as ONE logical batch of N*B samples: all-reduce D's batch-norm sums "
                         "(needs the torch.distributed launch; eager only)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend: nccl = RCCL over xGMI (the product path); gloo = debug transport, the pool is staged "
                         "through the host (what a 1-GPU box can run at world size 2 together with --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank uses device 0 (debug: the N-rank control flow on a 1-GPU box; RCCL refuses two ranks on one device, so "
                         "this needs --backend gloo)")
    ap.add_argument("--contraction", default="f32", choices=["f32", "bx6"],
                    help="f32 (default, the headline): every contraction on the exact-fp32 matrix instructions.  bx6: opt-in -- the layers with "
                         "a multiple of 64 output channels, Cred %% 32 == 0 and GPU-filling grids run on the bf16 matrix cores with every fp32 operand split exactly into three bf16 pieces "
                         "(six products, fp32 accumulate: an fp32 chain's accuracy, DESIGN.md); the line's dtype says so.  The default run "
                         "reports this mode as a second object `bx6` next to the f32 headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the mnist / dcgan32 / synthetic2d samples/s that are measured after the headline's timed region")
    ap.add_argument("--detail", default=DETAIL_PATH,
                    help="sidecar file for the full record (kernel / hbm tables, per-configuration rooflines, class surface, f1, shaping: everything the "
                         "compact stdout line summarises in one number each); '' = do not write it")
    ap.add_argument("--by-layer", action="store_true", help="key the per-kernel timing records by layer shape too (diagnostic)")
    args = ap.parse_args()
    if args.steps <= 0:
        args.steps = 256 if args.arch == "synthetic2d" else 12
    if args.share_gpu and args.backend == "nccl" and args.gpus > 1:
        raise SystemExit("--share-gpu puts every rank on device 0, which RCCL refuses: add --backend gloo")

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC for RCCL; must be set before HIP initialises
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)
    env_world = int(os.environ.get("WORLD_SIZE", 1))
    if env_world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={env_world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (there is no CPU path to benchmark)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = "RANK" in os.environ and "MASTER_ADDR" in os.environ      # launched by torch.distributed.run (any world size)
    rank, world = 0, 1
    if use_dist:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        rank, world = dist.get_rank(), dist.get_world_size()             # what the process group says, not the environment
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {world} ranks")

    if args.arch == "synthetic2d":
        return bench_synthetic2d(args, dev, rank, world)

    from cgs_amd import dist as D
    from cgs_amd import kernels as K
    from cgs_amd import nets
    from cgs_amd.engine import RefineEngine

    B = args.batch or {"dcgan64": 1024, "dcgan32": 256, "mnist": 64, "cyclegan256": 8}[args.arch]
    Ksteps = args.refine_steps or (50 if args.arch == "mnist" else 20)
    A = nets.ARCHS[args.arch]
    P = nets.init_params(args.arch, dev, seed=2019)                     # same frozen weights on every rank
    if args.sync_bn and not use_dist:
        raise SystemExit("--sync-bn needs the torch.distributed launch (python -m torch.distributed.run ... bench.py)")
    G = args.fuse if args.fuse > 0 else FUSE.get(args.arch, 1) if not args.sync_bn else 1
    graph_forced = args.graph is True
    if args.graph is None:
        # default: replay graphs at every world size (the engine captures in thread-local mode on its own side stream, so RCCL's
        # proxy / watchdog threads may make HIP calls meanwhile); eager only with synchronised batch norm, which has a
        # collective inside the program
        args.graph = not args.sync_bn
    n_flight = args.streams if args.streams > 0 else IN_FLIGHT.get(args.arch, 2)
    sync = True if args.sync_bn else None

    def build_engines(use_graph):
        return [RefineEngine(args.arch, P, B * G, dev, use_graph=use_graph, sync_bn=sync, bn_groups=G, contraction=args.contraction)
                for _ in range(n_flight)]

    streams = [torch.cuda.Stream(dev) for _ in range(n_flight)] if n_flight > 1 else [torch.cuda.current_stream(dev)]
    n_batches = args.steps + args.warmup                                # a step = one engine call = G logical batches
    rs = np.random.RandomState(D.rank_seed(rank))                       # rank-offset seed (2019 + rank): disjoint z shards
    z = torch.from_numpy(rs.uniform(-1, 1, (n_batches, B * G) + nets.g_input_shape(A)).astype(np.float32)).to(dev)   # z, or source images
    # one node-wide pool buffer per batch in flight (config 4: 8 ranks x 1024 x 64x64x3 fp32 = 402.7 MB per buffer, x 2 in flight per rank):
    # sized against the device's free memory BEFORE allocating, so a pool that cannot fit is a message, not an OOM inside the timed region
    pool_bytes = world * B * G * int(np.prod(A["img"])) * 4
    mem_free, mem_total = torch.cuda.mem_get_info(dev)
    if use_dist and pool_bytes * n_flight > 0.5 * mem_free:
        raise SystemExit(f"[bench rank {rank}] the sample pool needs {n_flight} x {pool_bytes / 1e6:.1f} MB but only {mem_free / 1e6:.1f} MB of device "
                         f"memory are free (half of it is kept for the engines): lower --batch / --streams")
    pools = [torch.empty((world * B * G,) + tuple(A["img"]), dtype=torch.float32, device=dev) if use_dist else None
             for _ in range(n_flight)]

    # -- prepare: every engine's first call (packs weights, sizes workspaces, captures its hipGraph).  No collective in here, so
    # a rank whose capture fails can fall back to eager launches by itself without unbalancing the ranks' gather counts.
    graph_fallback = None

    def prepare(engs):
        for e, st in zip(engs, streams):
            with torch.cuda.stream(st):
                e.refine_from_z(z[0], Ksteps, args.rate)
        torch.cuda.synchronize(dev)
    if os.environ.get("CGS_BENCH_BREAK_CAPTURE") and args.graph:     # test hook: the first hipGraph capture fails (exercises the fallback below)
        class _Refused:
            def __init__(self, *a, **k):
                raise RuntimeError("hipGraph capture refused (CGS_BENCH_BREAK_CAPTURE test hook)")
        torch.cuda.graph = _Refused
    engines = build_engines(args.graph)
    try:
        prepare(engines)
    except Exception as ex:                                              # noqa: BLE001 (any capture failure: HIP, RCCL threads, allocator)
        if not args.graph or graph_forced:
            raise
        graph_fallback = f"{type(ex).__name__}: {str(ex)[:300]}"
        print(f"[bench rank {rank}] hipGraph capture failed, launching eagerly instead: {graph_fallback}", file=sys.stderr, flush=True)
        args.graph = False
        try:
            torch.cuda.synchronize(dev)
        except Exception:                                                # noqa: BLE001
            pass
        del engines
        engines = build_engines(False)
        prepare(engines)

    done_ev, gather_ev = [], []                                         # per timed step: completion event; (before, after) the gather

    def step(i):
        j = i % n_flight
        e, st = engines[j], streams[j]
        timed = i >= args.warmup
        with torch.cuda.stream(st):
            img = e.refine_from_z(z[i], Ksteps, args.rate)[0]
            if use_dist:
                if timed:
                    g0 = torch.cuda.Event(enable_timing=True); g0.record(st)
                D.gather_pool(img, out=pools[j])                          # RCCL over xGMI: the refined sample pool
                if timed:
                    g1 = torch.cuda.Event(enable_timing=True); g1.record(st)
                    gather_ev.append((g0, g1))
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
    live_profile = rank == 0 and not args.graph and n_flight == 1
    if live_profile:                      # one batch in flight: per-launch HIP events inside the timed region
        K.PROFILE = {}
        K.PROFILE_BY_LAYER = args.by_layer
    t0 = time.perf_counter()
    last_img = None
    for i in range(args.warmup, n_batches):
        last_img = step(i)
    torch.cuda.synchronize(dev)
    dt_own = time.perf_counter() - t0                                   # this rank's own work, before it waits for the others
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    prof, K.PROFILE = K.PROFILE, None
    # (this rank's last images, summed BEFORE the profiling step below re-uses engine 0 and its output buffer)
    own_sum = float(last_img.double().sum().item()) if (use_dist and last_img is not None) else 0.0
    prof_ms, prof_note, prof_wall_s = dt * 1e3, "HIP events around every launch inside the timed region", None
    if rank == 0 and not live_profile and not (args.sync_bn and world > 1):      # (synchronised batch norm: a step is a collective of ALL ranks)
        # several batches in flight: kernels of different streams overlap, so a per-launch duration taken inside the
        # timed region would include the other stream's work; and a replayed hipGraph has no per-launch host hook.
        # Time ONE more step alone on one stream instead, launched eagerly (the same kernels with the same arguments).
        t_prof = time.perf_counter()
        prof, prof_ms = profile_one_step(args.arch, P, B, G, Ksteps, args.rate, z[args.warmup], dev, streams[0], args.by_layer, sync,
                                         engine=engines[0])
        prof_wall_s = time.perf_counter() - t_prof       # (ranks 1.. wait in the all-gather below meanwhile: must stay far below the collective timeout)
        prof_note = "HIP events around every launch of an extra single-stream step right after the timed region (the median of three passes)" + (" (launched eagerly; the timed region replays hipGraphs)" if args.graph else "")
    dist_rec = None
    if use_dist:
        # what the collective layer really saw (all ranks take part in these two small all-gathers)
        rows = D.all_gather_floats([rank, dt_own, dt, 1.0 if args.graph else 0.0, own_sum], device=dev)
        dt = float(rows[:, 2].max())                                    # MAX over ranks of the barrier-to-barrier time
        if rank == 0:
            jl = (n_batches - 1) % n_flight
            pool_sums = pools[jl].view(world, -1).double().sum(dim=1).cpu().numpy() if n_batches > 0 else np.zeros(world)
            per_rank = B * G * args.steps / rows[:, 1]
            dist_rec = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": len({int(r) for r in rows[:, 0]}),
                        "devices": "shared cuda:0 (debug)" if args.share_gpu else "one per rank",
                        "pool_bytes": int(pools[0].numel() * 4), "pool_buffers_per_rank": n_flight,
                        "pool_bytes_per_rank_total": int(pool_bytes * n_flight), "device_mem_free_before_pools": int(mem_free),
                        "device_mem_total": int(mem_total), "gathers_per_step": 1,
                        "gather_ms_per_step": round(sum(a.elapsed_time(b) for a, b in gather_ev) / max(1, len(gather_ev)), 4),
                        "gather": "all_gather_into_tensor on the step's stream, HIP events around it" + ("" if args.backend == "nccl" else " (gloo: staged through the host)"),
                        "per_rank_samples_per_s": [round(float(per_rank.min()), 1), round(float(per_rank.max()), 1)],
                        "hipgraph_ranks": int(rows[:, 3].sum()),
                        # rank r's rows of the last pool are rank r's own images (disjoint z shards -> distinct sums)
                        "pool_rows_match_ranks": bool(np.allclose(pool_sums, rows[:, 4], rtol=1e-9, atol=1e-6)),
                        "pool_rank_sums_distinct": len({round(float(v), 3) for v in pool_sums}) == world}

    if rank == 0:
        value = world * B * G * args.steps / dt
        flops_per_sample = nets.refine_flops_per_sample(args.arch, Ksteps)
        all_graph = bool(args.graph) if dist_rec is None else dist_rec["hipgraph_ranks"] == world
        out = {
            "metric": f"refined samples/sec @ {Ksteps} refinement steps", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE[args.contraction], "data": "synthetic",
            "config": {"workload": f"{args.arch} collaborative refinement (propose + K-step refine + render), "
                                   f"batch {B}/GPU{f' (x{G} logical batches per launch, batch-norm statistics per logical batch)' if G > 1 else ''}, "
                                   f"K={Ksteps}, momentum rate {args.rate}, refine at feature "
                                   f"{list(A['feature'])}, random-init weights, z~U(-1,1) seed 2019+rank",
                       "global_batch": world * B * G, "refine_steps": Ksteps, "parallelism": f"z-shards x{world} + RCCL all-gather of the pool" if world > 1 else "single GPU",
                       "hipgraph": all_graph, "batches_in_flight": n_flight * G, "fused_per_launch": G, "sync_bn": bool(args.sync_bn),
                       "contraction": args.contraction},
            "algorithmic_tflops": round(value * flops_per_sample / 1e12, 2),
            "lib": lib_stamp(),
        }
        if graph_fallback:
            out["config"]["hipgraph_fallback"] = graph_fallback
        if dist_rec is not None:
            if prof_wall_s is not None:
                dist_rec["rank0_profile_step_wall_s"] = round(prof_wall_s, 3)
            out["dist"] = dist_rec
        ns = n_flight
        if len(done_ev) >= 3 * ns:   # SURVEY.md 8d "median of >= 10": median gap between a stream's consecutive step completions
            gaps = sorted(a.elapsed_time(b) for a, b in zip(done_ev[:-ns], done_ev[ns:]))     # (GPU timestamps; the ns batches in flight
            out["ms_per_step_median"] = round(gaps[len(gaps) // 2] / ns, 3)                    # finish together, so per stream, / ns)
        if prof:
            out["roofline"], out["kernels"], out["hbm"], ex_step = profile_records(
                prof, prof_ms, 1e3 * dt / args.steps, prof_note, args.arch == "dcgan64" and B == 1024 and G == 1,
                prof_steps=args.steps if live_profile else 1)
            out["executed_tflops"] = round(world * ex_step / (1e3 * dt / args.steps) / 1e9, 2)      # whole job, padding taps not counted
        cpu = cpu_baseline(args.arch, Ksteps, args.rate) if world == 1 and not args.no_cpu_baseline else None   # (calibrates the host thread count)
        if world == 1 and not args.no_other_configs:
            if args.contraction == "f32":
                # the opt-in split-bf16 mode on the SAME workload, engines, batches in flight and z batches, timed right after
                # (and outside) the headline's region: its own samples/s, dtype and roofline (never the headline)
                del engines
                torch.cuda.empty_cache()
                out["bx6"] = bx6_line(args, P, B, G, Ksteps, z, dev, n_flight, streams, flops_per_sample)
            out["other_configs"] = other_configs(dev, args.arch, not args.no_cpu_baseline)
            out["class_surface"] = class_surface(dev)
            out["f1"] = f1_fill(dev)
            out["shaping"] = shaping_record(dev)
        if cpu is not None:
            out["cpu_baseline"] = cpu
        emit(out, args.detail)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
