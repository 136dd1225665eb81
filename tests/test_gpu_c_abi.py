"""GPU: a pure C/HIP program (no Python, no torch) drives the C ABI of libgcfr_hip.so -- hipMalloc'd buffers,
its own stream -- and checks one-call == three-call (bit-equal) and both against the C oracle."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_program_drives_the_abi(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    assert os.path.exists(hipcc)
    lib_dir = os.path.join(ROOT, "geomconsistentfr_amd", "lib")
    ora_dir = os.path.join(ROOT, "oracle", "_build")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    exe = str(tmp_path / "abi_smoke")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "c", "abi_smoke.cpp"),
                           "-o", exe, "-L" + lib_dir, "-lgcfr_hip", "-L" + ora_dir, "-lgcfr_oracle",
                           "-Wl,-rpath," + lib_dir, "-Wl,-rpath," + ora_dir])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "C-ABI SMOKE OK" in out.stdout
