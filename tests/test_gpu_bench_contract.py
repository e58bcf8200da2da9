"""bench.py / __graft_entry__ contract on the GPU box: one JSON line with the driver's keys, the roofline and the
cpu_baseline objects; smoke() runs and checks itself against the oracle."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", *extra], cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_line_small_config():
    d = run_bench("--arch", "mnist", "--steps", "2", "--warmup", "1", "--refine-steps", "5")
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "samples/s" and d["value"] > 0 and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] in ("mfma", "hbm") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]


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
                          "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--arch", "mnist", "--steps", "2",
                          "--warmup", "1", "--refine-steps", "3", "--no-cpu-baseline"], cwd=ROOT, capture_output=True, text=True,
                         timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and "RCCL" not in d["config"]["parallelism"]


def test_gather_pool_on_rccl_world1_is_identity_through_the_collective():
    """dist.gather_pool's all_gather_into_tensor on the nccl (RCCL) backend, in-process at world size 1."""
    code = (
        "import os, sys, torch, torch.distributed as dist\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29534', RANK='0', WORLD_SIZE='1')\n"
        "dev = torch.device('cuda:0'); torch.cuda.set_device(dev)\n"
        "dist.init_process_group('nccl', device_id=dev)\n"
        "x = torch.arange(24, dtype=torch.float32, device=dev).view(4, 6)\n"
        "pool = torch.empty_like(x); dist.all_gather_into_tensor(pool, x)\n"
        "assert torch.equal(pool, x)\n"
        "t = torch.tensor([3.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); assert t.item() == 3.5\n"
        "dist.destroy_process_group(); print('RCCL_OK')\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stderr[-3000:]
