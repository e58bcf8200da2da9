"""The drop-in boundary is a C ABI, not a Python extension: a plain C++ host (tests/abi/abi_smoke.cpp, HIP runtime only,
no torch) links libcgs_hip.so, runs conv forward / backward-data and checks the adjoint identity on its own."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cxx_host_links_and_runs(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.join(ROOT, "collaborative-gan-sampling_amd")
    assert os.path.exists(os.path.join(libdir, "libcgs_hip.so")), "build the library first (__graft_entry__.build())"
    exe = str(tmp_path / "abi_smoke")
    cc = subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "abi", "abi_smoke.cpp"), "-o", exe, "-L", libdir, "-lcgs_hip"],
                        capture_output=True, text=True, timeout=600)
    assert cc.returncode == 0, cc.stderr[-3000:]
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0 and run.stdout.strip().endswith("OK"), run.stdout + run.stderr
    assert "igemm_kernel" in run.stdout
