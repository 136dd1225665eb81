"""Writes a small dataset in the reference's on-disk layout (the directories, file names and .mat keys that
train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:535-556 reads) from synthetic faces: test data for
geomconsistentfr_amd/dataset.py.  `python tests/make_dataset_fixture.py <dir> [n]`, or `write_dataset(dir, n)` from a test.
The real 'MP_data/' is a Google-Drive download the reference's README names and is not in the repository."""
import os
import sys

import numpy as np
import scipy.io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_dataset(root: str, n: int = 5, H: int = 256, W: int = 256):
    from PIL import Image
    from geomconsistentfr_amd.dataset import DIRS
    from geomconsistentfr_amd.train import synthetic_batch
    for d in DIRS.values():
        os.makedirs(os.path.join(root, d), exist_ok=True)
    b = synthetic_batch(n, 4242, H, W)
    rng = np.random.default_rng(9)
    truth = []
    for i in range(n):
        ident = "%05d" % (10 * i + 3)
        img = np.rint(b["images"][i].numpy() * 255).astype(np.uint8)
        depth = b["depths"][i, ..., 0].numpy().astype(np.float64)
        light = b["lightings"][i, 1:4].numpy().astype(np.float64)
        dmask = np.where(b["masks"][i, ..., 0].numpy() > 0, 255, 0).astype(np.uint8)
        dmask[H // 2 - 8:H // 2 + 8, W // 2 - 6:W // 2 + 6] = 0                      # the depth mask has a hole at nose / mouth ...
        fmask = np.where(b["masks"][i, ..., 0].numpy() > 0, rng.choice([120, 128, 129, 200, 255], size=(H, W)), 0).astype(np.uint8)
        alb = np.rint(b["albedo"][i, ..., 0].numpy() * 255).astype(np.uint8)          # ... which the face mask fills (T8:552-556)
        Image.fromarray(img).save(os.path.join(root, DIRS["images"], ident + ".jpg"), quality=92)
        Image.fromarray(alb).save(os.path.join(root, DIRS["albedo"], ident + ".jpg"), quality=92)
        Image.fromarray(fmask).save(os.path.join(root, DIRS["face_masks"], ident + ".jpg"), quality=95)
        Image.fromarray(dmask).save(os.path.join(root, DIRS["masks"], ident + "_mask.png"))
        scipy.io.savemat(os.path.join(root, DIRS["depths"], ident + "_depth.mat"), {"depth_img": depth}, do_compression=True)
        scipy.io.savemat(os.path.join(root, DIRS["lightings"], ident + ".jpg.mat"), {"lighting_direction": light.reshape(1, 3)})
        truth.append(dict(id=ident, depth=depth, light=light, dmask=dmask))
    return truth


if __name__ == "__main__":
    write_dataset(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5)
