"""GPU: sqrt_rn_normal() (csrc/gcfr_device.hpp) -- the march epilogue's correctly rounded square root without the parts of the
compiler's IEEE expansion that serve denormal-range arguments -- against __builtin_sqrtf over EVERY float of its domain
(x >= 2^-96, +inf, NaN): 1.887 G bit patterns, compared on the device (tests/c/sqrt_rn_check.hip, built here with hipcc; the
product source is included, not restated).  min_dist = sqrt(min S) / |BC| is pinned bit for bit to the reference, so this
helper must be THE square root, not a close one."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_sqrt_rn_normal_equals_ieee_sqrt_on_its_whole_domain(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not on this box")
    exe = str(tmp_path / "sqrt_rn_check")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                    os.path.join(ROOT, "tests", "c", "sqrt_rn_check.hip"), "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "mismatches 0" in r.stdout, r.stdout
    assert "checked 1.887 G" in r.stdout, r.stdout    # every float from 2^-96 to +inf and the positive NaNs
