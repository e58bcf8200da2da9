"""bench.py / __graft_entry__ contract on the GPU box: one JSON line with the driver's keys, the roofline and the
cpu_baseline objects; smoke() runs and checks itself against the oracle."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def free_port():
    """A free rendezvous port on the loopback interface (the way bench.self_launch picks its own)."""
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


LINE_BUDGET = 6000          # bytes (bench.LINE_BUDGET): the driver keeps 8 kB of stdout tail


def run_bench(*extra, detail=None):
    """One `python bench.py --gpus 1 ...` run -> the ONE stdout line (held to the size budget), and with ``detail`` the sidecar record too."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *extra] + (["--detail", str(detail)] if detail else ["--detail", ""])
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.strip().splitlines()[-1] == lines[0], out.stdout
    assert len(lines[0]) < LINE_BUDGET, len(lines[0])
    line = json.loads(lines[0])
    if detail:
        with open(detail) as f:
            return line, json.load(f)
    return line


def test_bench_line_small_config(tmp_path):
    """The compact line (contract keys + roofline + cpu_baseline + lib + one-number summary, < 6 kB: VERDICT r5 #1) and the full record
    in the sidecar it names."""
    line, d = run_bench("--arch", "mnist", "--steps", "2", "--warmup", "1", "--refine-steps", "5", detail=tmp_path / "bench_detail.json")
    for k in REQUIRED + ["lib", "summary", "detail"]:
        assert k in line, k
    for k in REQUIRED:
        assert k in d, k
    assert line["value"] == d["value"] and line["ms_per_step"] == d["ms_per_step"] and line["roofline"]["frac"] == d["roofline"]["frac"]
    assert "kernels" not in line and "hbm" not in line and "other_configs" not in line and "note" not in line["roofline"]
    assert len(line["cpu_baseline"]["sample"]) <= 120 and line["cpu_baseline"]["value"] == d["cpu_baseline"]["value"]
    sm = line["summary"]
    assert set(sm["other_configs"]) == {"dcgan32", "cyclegan256", "synthetic2d"} and set(sm["class_surface"]) >= {"dcgan64", "mnist", "dcgan32", "dcgan32_b256"}
    assert d["class_surface"]["dcgan32_b256"]["batch"] == 256 and d["class_surface"]["dcgan32_b256"]["engine"]["path"] == "engine"
    for arch in ("mnist", "dcgan32"):         # the generic path replays its captured hipGraph (round 6)
        assert d["class_surface"][arch]["generic"]["hipgraph"] is True
    assert sm["other_configs"]["dcgan32"]["samples_per_s"] == d["other_configs"]["dcgan32"]["samples_per_s"]
    assert sm["f1"]["accepted_samples_per_s"] == d["f1"]["accepted_samples_per_s"] and set(sm["shaping_iteration_ms"]) == {"mnist", "dcgan64"}
    assert d["roofline"]["profile_passes"]["kept"] == "median" and len(d["roofline"]["profile_passes"]["wall_ms"]) == 3
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "samples/s" and d["value"] > 0 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert 0 < r["frac"] <= 1.0 and r["nominal_frac"] >= r["frac"] - 1e-3     # frac counts what is issued to the matrix cores: never above the machine
    assert 0 < r["step_executed_frac"] <= 1.0
    for h in d["hbm"].values():
        assert 0 < h["frac"] <= 1.0 and h["unit"] == "GB/s" and h["algorithmic_bytes"] > 0
    # every BASELINE workload that fits one GPU is in the line: the headline + dcgan32, dcgan64-or-mnist, cyclegan256, synthetic2d
    assert set(d["other_configs"]) == {"dcgan32", "cyclegan256", "synthetic2d"}
    for name, o in d["other_configs"].items():
        assert o["samples_per_s"] > 0 and o["cpu_baseline"]["value"] > 0 and o["cpu_baseline"]["cores"] >= 1
        if name != "synthetic2d":
            assert 0 < o["roofline"]["frac"] <= 1.0 and 0 < o["roofline"]["step_executed_frac"] <= 1.0 and o["hbm"]
    # the class surface: the reference's verbatim wiring reaches the fused engine; the opaque-callable form stays generic
    cs = d["class_surface"]
    for arch in ("dcgan64", "mnist", "dcgan32"):
        e, g = cs[arch]["engine"], cs[arch]["generic"]
        assert e["path"] == "engine" and e["hipgraph"] is True and "hipgraph_fallback" not in e and g["path"] == "generic"
        assert e["samples_per_s"] > g["samples_per_s"] > 0
    for arch in ("mnist", "dcgan32"):         # the reference's batch-64 calls, G per launch through the same wiring (Refiner.logical_batch)
        f = cs[arch]["fused"]
        assert cs[arch]["batch"] == 64 and f["path"] == "engine" and f["logical_batch"] == 64 and f["samples_per_s"] > 2.0 * cs[arch]["engine"]["samples_per_s"]
    f1 = d["f1"]                              # the evaluate fill loop (nsgan/GAN.py:384-426), measured
    assert f1["eval_size"] == 49984 and f1["accepted_samples_per_s"] > 0 and 0 < f1["efficiency"] <= 1.0 and 0 <= f1["host_chain_share_of_wall"] < 1
    assert len(d["lib"]["source_sha16"]) == 16
    for arch in ("mnist", "dcgan64"):         # the D shaping iteration (nsgan/GAN.py:266-272) at batch 64, with the weight-gradient kernel's record
        sr = d["shaping"][arch]
        assert sr["batch"] == 64 and sr["iteration_ms"] > sr["d_step_ms"] > 0 and 0 < sr["kernels"]["wgrad_kernel"]["frac_of_fp32_matrix_peak"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


def test_a_failed_graph_capture_falls_back_to_eager_launches_in_the_same_process():
    """The default is hipGraph replay at every world size; if the capture raises (here: a test hook refuses it) the run must go on
    with eager launches in the SAME process and say so in the line -- and with an explicit --graph the failure is an error."""
    env = dict(os.environ, CGS_BENCH_BREAK_CAPTURE="1")
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--arch", "mnist", "--steps", "2", "--warmup", "1", "--refine-steps", "3",
            "--no-cpu-baseline", "--no-other-configs"]
    out = subprocess.run(args, cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["value"] > 0 and d["config"]["hipgraph"] is False and "capture refused" in d["config"]["hipgraph_fallback"]
    assert "launching eagerly instead" in out.stderr
    out = subprocess.run(args + ["--graph"], cwd=ROOT, capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode != 0 and "capture refused" in out.stderr


def test_bench_synthetic2d_line():
    d = run_bench("--arch", "synthetic2d", "--steps", "3", "--warmup", "1")
    for k in REQUIRED:
        assert k in d, k
    assert d["value"] > 0


def test_smoke_entry():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    g.smoke()


def test_bench_under_torchrun_runs_the_rccl_gather():
    """`python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1`: the launch form the driver uses for N > 1,
    at the one world size a 1-GPU box can hold -- init_process_group("nccl") (= RCCL), all_gather_into_tensor of the refined
    pool inside the timed region, barrier, MAX all-reduce of the time."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--arch", "mnist", "--steps", "2",
                          "--warmup", "1", "--refine-steps", "3", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "RCCL" not in d["config"]["parallelism"]
    x = d["dist"]
    assert x["backend"] == "nccl" and x["world_size"] == 1 and x["ranks_seen"] == 1 and x["gather_ms_per_step"] > 0
    assert x["pool_rows_match_ranks"] is True and d["config"]["hipgraph"] is True and "hipgraph_fallback" not in d["config"]


def test_two_ranks_on_the_one_gpu_through_the_whole_multi_rank_path():
    """`python bench.py --gpus 2 --backend gloo --share-gpu` on the 1-GPU box -- self_launch -> torch.distributed.run
    children (started before any GPU call) -> rank-offset seeds -> hipGraph replay per rank -> gather of the pool (gloo: staged
    through the host; the same dist.gather_pool call RCCL serves on the 8-GPU node) -> MAX-reduce of the time -> ONE JSON line from
    rank 0 -> the child's exit code relayed."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("MASTER_ADDR", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--arch", "mnist",
                          "--steps", "4", "--warmup", "1", "--refine-steps", "3", "--no-cpu-baseline"], cwd=ROOT, capture_output=True,
                         text=True, timeout=900, env=env)        # (5 batches over 4 engines: the LAST one runs on engine 0, the one the roofline's extra step re-uses)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout                                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["global_batch"] == 2 * 64 * 32 and "x2" in d["config"]["parallelism"]
    x = d["dist"]
    assert x["backend"] == "gloo" and x["world_size"] == 2 and x["ranks_seen"] == 2
    assert x["pool_bytes"] == 2 * 64 * 32 * 28 * 28 * 4 and x["gather_ms_per_step"] > 0
    assert x["pool_rows_match_ranks"] is True and x["pool_rank_sums_distinct"] is True     # rank r's rows ARE rank r's images; seeds differ
    assert x["hipgraph_ranks"] == 2 and d["config"]["hipgraph"] is True
    lo, hi = x["per_rank_samples_per_s"]
    assert 0 < lo <= hi and d["value"] <= 2.0 * hi * 1.01                                   # whole-job value = both ranks over the MAX time
    assert "other_configs" not in d and d["roofline"]["frac"] <= 1.0
    # a wrong launch is refused, not silently run at another size
    bad = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--arch", "mnist"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in (bad.stderr + bad.stdout)


@pytest.mark.parametrize("arch,contraction", [("dcgan64", "f32"), ("dcgan64", "bx6"), ("cyclegan256", "f32")])
def test_two_ranks_on_the_one_gpu_with_the_drivers_own_workload(arch, contraction):
    """the first 8-GPU run is the driver's `bench.py --gpus N` with the dcgan64 DEFAULT workload (batch 1024 per rank,
    K = 20, hipGraph x 2 in flight).  The same command at world 2 on the one GPU: graphs captured on every rank with RCCL-free gloo
    threads around, no eager fallback, rank 0's extra profiling step (the other rank waits for it inside an all-gather) far below
    the collective timeout -- with the exact-fp32 default and with the opt-in split-bf16 contraction; and BASELINE config 5 in its
    multi-rank form (cyclegan256, batch 8 per rank, four calls in flight per rank)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("MASTER_ADDR", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--arch", arch,
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--contraction", contraction], cwd=ROOT, capture_output=True, text=True,
                         timeout=1200, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    B, img = (1024, 64 * 64 * 3) if arch == "dcgan64" else (8, 256 * 256 * 3)        # (config 5's per-GPU share: 64 over 8 GPUs)
    assert d["n_gpus"] == 2 and arch in d["config"]["workload"] and d["config"]["global_batch"] == 2 * B and d["config"]["refine_steps"] == 20
    assert d["config"]["hipgraph"] is True and "hipgraph_fallback" not in d["config"]
    x = d["dist"]
    assert x["world_size"] == 2 and x["ranks_seen"] == 2 and x["hipgraph_ranks"] == 2
    assert x["pool_bytes"] == 2 * B * img * 4 and x["pool_rows_match_ranks"] is True and x["pool_rank_sums_distinct"] is True
    assert 0 < x["rank0_profile_step_wall_s"] < 5.0
    assert 0 < d["roofline"]["frac"] <= 1.0 and ("igemm_bx6" if contraction == "bx6" else "igemm_kernel") in d["roofline"]["kernel"]
    assert d["config"]["contraction"] == contraction and d["dtype"] == ("f32" if contraction == "f32" else "f32 (3xbf16 split, fp32 accumulate)")


@pytest.mark.parametrize("arch,extra,B,G,img", [("mnist", ["--fuse", "4", "--refine-steps", "3"], 64, 4, 28 * 28), ("dcgan64", [], 1024, 1, 64 * 64 * 3)],
                         ids=["mnist", "config4_dcgan64_8x1024"])
def test_eight_ranks_on_the_one_gpu(arch, extra, B, G, img):
    """world 8 has never run anywhere.  `python bench.py --gpus 8 --backend gloo --share-gpu`: eight children started
    before any GPU call, rendezvous on the loopback, eight DISTINCT pool shards (seeds 2019 + rank), ranks_seen == 8, hipGraphs on
    every rank, the MAX-reduce, ONE JSON line, the exit code relayed.  With the dcgan64 default this is BASELINE config 4 in its own
    form -- 8 x 1024 samples, a 402.7 MB node-wide pool per batch in flight -- minus the transport (gloo through the host instead of
    RCCL over xGMI) and with the eight ranks time-slicing one GPU; the pool arithmetic is asserted against the device's memory."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("MASTER_ADDR", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--share-gpu", "--arch", arch,
                          "--steps", "3", "--warmup", "1", "--no-cpu-baseline"] + extra, cwd=ROOT, capture_output=True, text=True,
                         timeout=1500, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["value"] > 0 and d["config"]["global_batch"] == 8 * B * G
    assert "x8" in d["config"]["parallelism"] and d["config"]["hipgraph"] is True and "hipgraph_fallback" not in d["config"]
    x = d["dist"]
    assert x["world_size"] == 8 and x["ranks_seen"] == 8 and x["hipgraph_ranks"] == 8
    assert x["pool_bytes"] == 8 * B * G * img * 4
    assert x["pool_bytes_per_rank_total"] == x["pool_bytes"] * x["pool_buffers_per_rank"] < 0.5 * x["device_mem_free_before_pools"]
    if arch == "dcgan64":
        assert x["pool_bytes"] == 402653184 and x["pool_buffers_per_rank"] == 2          # 8 x 50.3 MB, two batches in flight
        assert x["pool_bytes_per_rank_total"] < 0.01 * x["device_mem_total"]              # < 1 % of the 288 GB of one MI355X
    assert x["pool_rows_match_ranks"] is True and x["pool_rank_sums_distinct"] is True
    assert 0 < x["rank0_profile_step_wall_s"] < 60.0
    lo, hi = x["per_rank_samples_per_s"]
    assert 0 < lo <= hi and d["value"] <= 8.0 * hi * 1.01
    assert len(d["lib"]["source_sha16"]) == 16 and d["lib"]["cgs_version"] >= 100 and d["lib"]["cgs_lib_override"] is False


def test_gather_pool_on_rccl_world1_goes_through_the_collective():
    """cgs_amd.dist.gather_pool on the nccl (RCCL) backend, in-process at world size 1: the library's own gather function runs
    all_gather_into_tensor (no world-1 short-circuit), into a fresh and into a caller-owned pool buffer."""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import cgs_amd.dist as D\n"
        f"os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='{free_port()}', RANK='0', WORLD_SIZE='1')\n"
        "dev = torch.device('cuda:0'); torch.cuda.set_device(dev)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "calls = []\n"
        "real = dist.all_gather_into_tensor\n"
        "def spy(*a, **k):\n"
        "    calls.append(1); return real(*a, **k)\n"
        "dist.all_gather_into_tensor = spy\n"
        "x = torch.arange(24, dtype=torch.float32, device=dev).view(4, 6)\n"
        "pool = D.gather_pool(x)\n"
        "assert torch.equal(pool, x) and pool.data_ptr() != x.data_ptr() and len(calls) == 1\n"
        "mine = torch.zeros_like(x); got = D.gather_pool(x, out=mine)\n"
        "assert got is mine and torch.equal(mine, x) and len(calls) == 2\n"
        "assert D.gather_pool(x, skip_trivial=True) is x and len(calls) == 2\n"
        "rows = D.all_gather_floats([0, 3.5], device=dev); assert rows.shape == (1, 2) and rows[0, 1] == 3.5\n"
        "t = torch.tensor([3.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == 3.5\n"
        "dist.destroy_process_group(); print('RCCL_OK')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stderr[-3000:]
