"""AddressSanitizer run of the C ABI's HOST side (argument checking, geometry, tile planning) -- CPU only: the library is
rebuilt with -fsanitize=address (host code; gpurun refuses GPU sanitizers) and a plain C++ driver calls every conv-family
entry point with the BASELINE layer geometries and fake device pointers; without a GPU each call runs its host logic and
returns CGS_ELAUNCH at the first HIP call.  SURVEY.md section 5 (race detection / sanitizers)."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
CSRC = os.path.join(ROOT, "collaborative-gan-sampling_amd", "csrc")


def _sources():
    """The library's sources, from its own Makefile (SRC := a.hip b.hip ...)."""
    with open(os.path.join(CSRC, "Makefile")) as f:
        line = next(l for l in f if l.startswith("SRC :="))
    return [s[:-4] for s in line.split(":=")[1].split()]


SOURCES = _sources()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_host_side_is_asan_clean(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("meant for the GPU-less build container (with a GPU the calls would launch kernels on fake pointers)")
    flags = ["-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fsanitize=address", "-fno-gpu-sanitize",
             f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}", "-Wno-unused-value", "-Wno-pass-failed"]
    procs = [subprocess.Popen([HIPCC, *flags, "-c", os.path.join(CSRC, f"{s}.hip"), "-o", str(tmp_path / f"{s}.o")],
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for s in SOURCES]
    for s, p in zip(SOURCES, procs):
        out, err = p.communicate(timeout=600)
        assert p.returncode == 0, f"{s}.hip: {err[-2000:]}"
    lib = tmp_path / "libcgs_hip_asan.so"
    r = subprocess.run([HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-fsanitize=address", "-fno-gpu-sanitize",
                        *[str(tmp_path / f"{s}.o") for s in SOURCES], "-o", str(lib)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    exe = tmp_path / "asan_host"
    r = subprocess.run([HIPCC, "-std=c++17", "-fsanitize=address", "-fno-gpu-sanitize", f"-I{os.path.join(ROOT, 'include')}",
                        os.path.join(ROOT, "tests", "abi", "asan_host.cpp"), f"-L{tmp_path}", "-lcgs_hip_asan", f"-Wl,-rpath,{tmp_path}",
                        "-o", str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert "AddressSanitizer" not in r.stderr and "ASAN_HOST_OK" in r.stdout and r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    shutil.rmtree(tmp_path, ignore_errors=True)
