"""CPU: register / scratch budget of the built march kernels, read from the code objects inside libgcfr_hip.so.

The inference march is forced to six waves per SIMD (80 VGPRs) and the training (argmin) march to five (96); both
must fit WITHOUT scratch: a kernel with a private segment writes every resident wave's arena back to HBM once per
launch (round 1: +30 MB per launch, round 2 start: +15 MB) whether or not the spill sits inside the sample loop.
The metadata is what the loader uses (AMDGPU code-object notes), so this is the shipped binary, not a compile log."""
import os
import re
import shutil
import subprocess

import pytest

LLVM = "/opt/rocm/lib/llvm/bin"


def _kernel_metadata(tmp_path):
    from geomconsistentfr_amd import _lib
    _lib.load()
    so = shutil.copy(_lib.lib_path(), tmp_path / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    kernels = {}
    for f in sorted(tmp_path.glob("lib.so.*gfx950")):
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            get = lambda key: re.search(r"\.%s:\s+(\S+)" % key, blk).group(1)
            name = subprocess.run(["c++filt", get("name")], capture_output=True, text=True).stdout.strip()
            kernels[name.split("(")[0].replace("void ", "").replace("gcfr::", "")] = {
                "vgpr": int(get("vgpr_count")), "sgpr": int(get("sgpr_count")), "scratch": int(get("private_segment_fixed_size")),
                "lds": int(get("group_segment_fixed_size"))}
    return kernels


@pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-readelf")), reason="needs the ROCm LLVM tools")
def test_march_kernels_fit_their_forced_occupancy_without_scratch(tmp_path):
    k = _kernel_metadata(tmp_path)
    inference = {n: v for n, v in k.items() if n.startswith(("shadow_fwd_quad_kernel<", "shadow_fwd_quad_lds_kernel<"))}
    training = {n: v for n, v in k.items() if n.startswith(("shadow_fwd_quad_argmin_kernel<", "shadow_fwd_quad_argmin_lds_kernel<",
                                                             "shadow_fwd_quad_argmin_own_kernel<"))}   # own: pixels = mask (round 4)
    assert len(inference) >= 16 + 4 and len(training) >= 16 + 4 + 16, sorted(k)      # + the LDS-staged variants of the default shape
    for n, v in inference.items():
        assert v["vgpr"] <= 80, (n, v)                      # 512 / 6 waves, granule 8
    for n, v in training.items():
        assert v["vgpr"] <= 96, (n, v)                      # 5 waves
    # the kernels the library launches by default (16 x 4 tiles, groups of four), fused or not, either half parity
    for n, v in list(inference.items()) + list(training.items()):
        if re.search(r"<16, (true|false), 4, (true|false), 0>", n):
            assert v["scratch"] == 0, (n, v)
    # everything else that runs per pixel or per texel is spill-free too
    for n in ("build_quad_kernel", "normals_fwd_kernel", "inference_images_kernel", "fix_border_kernel"):
        hit = [v for m, v in k.items() if m.startswith(n)]
        assert hit and all(v["scratch"] == 0 for v in hit), (n, hit)
    # the staged backward runs at four waves per SIMD (128 VGPRs) with its ~40 pointer arguments parked in two VGPRs'
    # lanes (v_writelane) and a few dwords of scratch outside its hot stages: bounded, not zero (round 3: the run key of
    # the corner-atomic merge carries the wrapped-column flag, one dword more than round 2's 16 B)
    bwd = [v for m, v in k.items() if m.startswith("render_bwd_single_light_kernel")]
    # (round 4: + the corner window's box and flush bookkeeping, 32 B)
    assert bwd and all(v["vgpr"] <= 128 and v["scratch"] <= 32 for v in bwd), bwd
