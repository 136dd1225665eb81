"""TEST INFRASTRUCTURE -- pins SURVEY.md 8f-3 (the image side of the inference scripts) to the reference itself.

Run in the authoring container only (needs /root/reference):   python oracle/make_golden_slt_main.py

Runs the UNMODIFIED `main()` of test_relight_single_image_lighting_transfer.py (SLT:516-579) on CPU through
oracle/ref_shim.py with exactly two seams, both at the script's file I/O:
  * `imageio.imread(path)`  -> returns the 256x256 uint8 arrays stored in the fixture (the shipped FFHQ samples
                               resized 1024 -> 256 with PIL bilinear, and the shipped 256x256 skin masks);
  * `cv2.imwrite(path, a)`  -> captures the six float arrays the script hands to OpenCV (SLT:574-579).
`sys.argv[1:4]` carries the three paths (SLT:524-526), the working directory is /root/reference for the duration of
the call because the script loads 'model_lighting_transfer/model_epoch106.pth' by relative path (SLT:518); nothing is
written there (bytecode writing is off, imwrite is captured).  The checkpoint was saved from CUDA tensors, so
`torch.load` gets map_location="cpu" for the call (an environment shim like ref_shim's identity `.cuda()`: no GPU here).  RelightNet.forward is wrapped only to RECORD the two
passes' estimated light / ambient (SLT:543-545), which main() does not expose.

Files written under tests/golden/ (data only):
  slt_main_<case>.npz            inputs (input / reference image, mask: uint8), the relighting pass's model outputs the images
                                 are made from (`model_*`, dtypes as the reference holds them), the six arrays as the script computed them
                                 (float64, BGR where the script flips; `*_f64`) and the bytes OpenCV stores for them
                                 (`*_u8`: saturate_cast<uchar>(cvRound(v)) = round-half-even, clip to [0,255]; cv2 itself
                                 is not installed here, that conversion is OpenCV's documented one), estimated light / ambient
  slt_checkpoint_epoch106.npz    the shipped lighting-transfer weights (state_dict tensors by name) -- the GPU box has
                                 no /root/reference, and the `-m gpu` test has to run the same network
"""
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shim  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
CASES = {"a": ("00295.png", "00110.png"), "b": ("00508.png", "00295.png")}     # (input image + its mask, reference image)
KEYS = ["rendered_image", "shadow_mask", "albedo", "depth", "shading", "surface_normals"]      # SLT:574-579 order


def load_inputs(name):
    from PIL import Image
    img = Image.open(os.path.join(ref_shim.REFERENCE_ROOT, "sample_test_images_FFHQ", name)).convert("RGB")
    img = np.asarray(img.resize((256, 256), Image.BILINEAR), dtype=np.uint8)
    mask = np.asarray(Image.open(os.path.join(ref_shim.REFERENCE_ROOT, "FFHQ_skin_masks", name)), dtype=np.uint8)
    return img, mask


def saturate_u8(a):
    return np.clip(np.rint(np.asarray(a, dtype=np.float64)), 0, 255).astype(np.uint8)


def run_case(SLT, input_name, reference_name):
    import cv2
    import imageio
    xin, mask = load_inputs(input_name)
    xref, _ = load_inputs(reference_name)
    files = {"in.png": xin, "ref.png": xref, "mask.png": mask}
    written, passes = [], []
    imageio.imread = lambda p: files[p].copy()
    cv2.imwrite = lambda p, a: written.append((p, np.array(a, copy=True)))
    orig_forward = SLT.RelightNet.forward

    def recording_forward(self, *a, **k):
        out = orig_forward(self, *a, **k)
        passes.append(tuple(o.detach().numpy().copy() for o in out))
        return out

    SLT.RelightNet.forward = recording_forward
    argv, cwd, orig_load = sys.argv, os.getcwd(), torch.load
    torch.load = lambda f, *a, **k: orig_load(f, *a, **dict(k, map_location="cpu"))
    try:
        sys.argv = ["test_relight_single_image_lighting_transfer.py", "in.png", "ref.png", "mask.png"]
        os.chdir(ref_shim.REFERENCE_ROOT)
        with contextlib.redirect_stdout(io.StringIO()):          # main() prints the model
            SLT.main()
    finally:
        os.chdir(cwd)
        sys.argv = argv
        torch.load = orig_load
        SLT.RelightNet.forward = orig_forward
    assert len(written) == 6 and len(passes) == 2, (len(written), len(passes))
    suffixes = ["_rendered_image.png", "_shadow_mask.png", "_albedo.png", "_depth.png", "_shading.png", "_surface_normals.png"]
    p2 = passes[1]     # SLT:514 tuple of the relighting pass: what SLT:547-579 turns into images
    out = dict(input_u8=xin, reference_u8=xref, mask_u8=mask,
               estimated_light=passes[0][10].reshape(3).astype(np.float32),
               estimated_ambient=passes[0][11].reshape(1).astype(np.float32),
               # dtypes as the reference holds them (final_shading / surface_normals are f64 by promotion)
               model_albedo=p2[0], model_depth=p2[1], model_shadow_mask_weights=p2[2], model_rendered_images=p2[5],
               model_final_shading=p2[8], model_surface_normals=p2[9])
    for key, suf, (path, arr) in zip(KEYS, suffixes, written):
        assert path == "lighting_transfer_result/in" + suf, path
        out[key + "_f64"] = np.asarray(arr, dtype=np.float64)
        out[key + "_u8"] = saturate_u8(arr)
    return out


def main():
    SLT = ref_shim.load("SLT")
    torch.manual_seed(0)
    for tag, (inp, ref) in CASES.items():
        res = run_case(SLT, inp, ref)
        np.savez_compressed(os.path.join(OUT, "slt_main_%s.npz" % tag), input_name=inp, reference_name=ref, **res)
        print(tag, inp, "<-", ref, "light", res["estimated_light"], "ambient", res["estimated_ambient"],
              {k: res[k + "_f64"].shape for k in KEYS})
    sd = torch.load(os.path.join(ref_shim.REFERENCE_ROOT, "model_lighting_transfer", "model_epoch106.pth"), map_location="cpu")
    np.savez_compressed(os.path.join(OUT, "slt_checkpoint_epoch106.npz"), **{k: v.numpy() for k, v in sd.items()})
    print("checkpoint:", len(sd), "tensors")


if __name__ == "__main__":
    main()
