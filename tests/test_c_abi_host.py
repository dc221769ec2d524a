"""The C ABI driven from a plain C program (tests/c_abi/abi_host.c: no Python, no torch in the process): hipMalloc'd
buffers, workspace queries, a caller-owned stream, status codes -- checked inside the program against the C restatement
of the reference (oracle/c).  The CPU half proves the program builds against include/anyloc_hip.h, links the library and
refuses to run without a GPU; the GPU half is the parity run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "c_abi"))


def _binary():
    import build_host
    lib = os.path.join(ROOT, "anyloc_amd", "libanyloc_hip.so")
    assert os.path.isfile(lib), f"{lib} not built (python -m anyloc_amd.build)"
    return build_host.build_host()              # gcc, ~1 s, only when the binary is missing or older than its C sources


def test_c_host_builds_links_and_fails_loudly_without_a_gpu():
    exe = _binary()
    ldd = subprocess.run(["ldd", exe], capture_output=True, text=True).stdout
    assert "libanyloc_hip.so" in ldd and "not found" not in ldd, ldd
    assert "libtorch" not in ldd and "libpython" not in ldd, "the C caller must not pull in torch or Python"
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the parity run is the gpu-marked test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no HIP device" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c_host_parity_against_the_c_restatement():
    exe = _binary()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout)
    sys.stderr.write(r.stderr)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert last["abi_host"] == "ok" and last["failures"] == 0
    assert r.stdout.count(" ok  ") >= 16 and "FAIL" not in r.stdout
