"""TEST INFRASTRUCTURE -- pins the single-image relighting script's image side (S1:601-620, the f64-mask flow) to the reference.

Run in the authoring container only (needs /root/reference):   python oracle/make_golden_s1_main.py

Runs the UNMODIFIED `main()` of test_relight_single_image.py (S1:507-620) on CPU through oracle/ref_shim.py.  The script
needs `model/model_epoch99.pth`, which the reference does not ship (`.MISSING_LARGE_BLOBS`): `torch.load` is answered with
the state_dict of a freshly constructed `RelightNet()` under torch.manual_seed(1234) -- the network's WEIGHTS are not what
this fixture pins (the render block given a network's outputs is pinned by s1_*.npz), the lines after the forward are.
Other seams, all at the script's file I/O: `imageio.imread` returns the stored 256 x 256 uint8 arrays (image: the shipped
FFHQ sample resized with PIL; mask: the shipped skin mask), `cv2.resize` of a 256 x 256 array to (256, 256) returns it
unchanged, `cv2.imwrite` captures the array (S1:620).  RelightNet.forward is wrapped only to record its 10-tuple.

Writes tests/golden/s1_main.npz (data only): inputs, the script's light / ambient row, `model_rendered_images` (f32, what
S1:616 multiplies), the composite as the script computed it (`rendered_image_f64`, BGR) and the bytes OpenCV stores for it.
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
from make_golden_slt_main import load_inputs, saturate_u8  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    S1 = ref_shim.load("S1")        # (installs the cv2 / imageio stub modules the script imports)
    import cv2
    import imageio
    img, mask = load_inputs("00295.png")                      # S1:514: img_name = '00295.png'
    files = {"sample_test_images_FFHQ/00295.png": img, "FFHQ_skin_masks/00295.png": mask}
    written, outs = [], []
    imageio.imread = lambda p: files[p].copy()
    cv2.resize = lambda a, size: a if tuple(a.shape[:2]) == tuple(size[::-1]) else (_ for _ in ()).throw(ValueError(a.shape))
    cv2.imwrite = lambda p, a: written.append((p, np.array(a, copy=True)))
    orig_forward, orig_load = S1.RelightNet.forward, torch.load

    def recording_forward(self, *a, **k):
        out = orig_forward(self, *a, **k)
        outs.append(tuple(o.detach().numpy().copy() for o in out))
        return out

    def seeded_state_dict(_path, *a, **k):
        torch.manual_seed(1234)
        return S1.RelightNet().state_dict()

    S1.RelightNet.forward, torch.load = recording_forward, seeded_state_dict
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            S1.main()
    finally:
        S1.RelightNet.forward, torch.load = orig_forward, orig_load
    assert len(written) == 1 and len(outs) == 1 and written[0][0] == "FFHQ_relighting_results/00295_rendered_image.png", written
    arr = np.asarray(written[0][1], dtype=np.float64)
    np.savez_compressed(os.path.join(OUT, "s1_main.npz"), input_u8=img, mask_u8=mask,
                        model_rendered_images=outs[0][5], rendered_image_f64=arr, rendered_image_u8=saturate_u8(arr))
    print("s1_main:", arr.shape, arr.dtype, "rendered range", float(outs[0][5].min()), float(outs[0][5].max()))


if __name__ == "__main__":
    main()
