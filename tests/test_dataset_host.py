"""CPU: the dataset reader (geomconsistentfr_amd/dataset.py) against load_data()'s own lines
(train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:535-556) restated with numpy, on a small dataset written in the
reference's layout by tests/make_dataset_fixture.py."""
import os
import sys

import numpy as np
import pytest
import scipy.io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from make_dataset_fixture import write_dataset  # noqa: E402


def load_data_statement(root, H=256, W=256):
    """T8:527-558 line for line (float64, as the script), on whatever the directories hold."""
    from PIL import Image
    imread = lambda p: np.asarray(Image.open(p))
    depths = sorted(os.listdir(os.path.join(root, "depth_maps_CelebA-HQ")))
    masks = sorted(os.listdir(os.path.join(root, "depth_masks_CelebA-HQ_DFNRMVS")))
    n = len(depths)
    images, lightings = np.zeros((n, H, W, 3)), np.zeros((n, 4))
    deps, msk, alb, fill = np.zeros((n, H, W, 1)), np.zeros((n, H, W, 1)), np.zeros((n, H, W)), np.zeros((n, H, W, 1))
    lightings[:, 0] = 0.5
    for i in range(n):
        deps[i] = np.reshape(scipy.io.loadmat(os.path.join(root, "depth_maps_CelebA-HQ", depths[i]))["depth_img"], (H, W, 1))
        msk[i] = np.reshape(imread(os.path.join(root, "depth_masks_CelebA-HQ_DFNRMVS", masks[i])), (H, W, 1))
        name_parts = depths[i].split("_")
        lightings[i, 1:4] = scipy.io.loadmat(os.path.join(root, "lighting_directions_CelebAHQ_DFNRMVS", name_parts[0] + ".jpg.mat"))["lighting_direction"]
        images[i] = imread(os.path.join(root, "CelebA-HQ_DFNRMVS_cropped", name_parts[0] + ".jpg")) / 255.0
        alb[i] = imread(os.path.join(root, "CelebA-HQ_albedo_grayscale", name_parts[0] + ".jpg"))
        tmp = np.reshape(imread(os.path.join(root, "CelebAHQ_face_masks", name_parts[0] + ".jpg")), (H, W, 1))
        tmp = np.maximum(tmp, msk[i])
        tmp[tmp > 128] = 255.0
        tmp[tmp <= 128] = 0.0
        fill[i] = tmp
    return images, lightings, deps, msk, alb, fill


def test_reader_holds_the_bytes_load_data_turns_into_float64(tmp_path):
    from geomconsistentfr_amd.dataset import RelightDataset
    truth = write_dataset(str(tmp_path), 4)
    ds = RelightDataset(str(tmp_path))
    images, lightings, deps, msk, alb, fill = load_data_statement(str(tmp_path))
    assert len(ds) == 4 and ds.ids == [t["id"] for t in truth]
    assert ds.images.dtype == np.uint8 and ds.masks.dtype == np.uint8 and ds.albedo.dtype == np.uint8
    np.testing.assert_array_equal(ds.images / 255.0, images)                      # T8:550
    np.testing.assert_array_equal(ds.masks[..., None].astype(np.float64), msk)    # T8:546
    np.testing.assert_array_equal(ds.albedo.astype(np.float64), alb)              # T8:551
    np.testing.assert_array_equal(ds.depths.astype(np.float64), deps.astype(np.float32).astype(np.float64))   # f32 on the host
    np.testing.assert_allclose(ds.lightings, lightings, rtol=1e-7)
    assert np.all(ds.lightings[:, 0] == 0.5)                                      # T8:542
    # the fill-nose-and-mouth rule from the two byte masks the reader keeps (what the device kernel evaluates)
    np.testing.assert_array_equal(np.where(np.maximum(ds.face_masks, ds.masks) > 128, 255.0, 0.0)[..., None], fill)
    assert (fill[:, 120:136, 122:134] == 255.0).all() or (fill != msk).any()       # the face mask really fills the hole
    # bytes, not doubles: the reference holds 8x (images, masks, albedo) of this
    assert ds.host_bytes() < 0.2 * sum(a.nbytes for a in (images, lightings, deps, msk, alb, fill))
    assert RelightDataset(str(tmp_path), limit=2).images.shape[0] == 2


def test_reader_pairs_masks_by_position_like_the_script(tmp_path):
    from geomconsistentfr_amd.dataset import DIRS, RelightDataset
    write_dataset(str(tmp_path), 3)
    os.remove(os.path.join(str(tmp_path), DIRS["masks"], sorted(os.listdir(os.path.join(str(tmp_path), DIRS["masks"])))[0]))
    with pytest.raises(ValueError, match="by position"):
        RelightDataset(str(tmp_path))


def test_batches_need_a_device(tmp_path):
    import torch
    from geomconsistentfr_amd._lib import GcfrError
    from geomconsistentfr_amd.dataset import RelightDataset, assemble_batch, masked_metrics
    write_dataset(str(tmp_path), 2)
    ds = RelightDataset(str(tmp_path))
    with pytest.raises(GcfrError):
        ds.batch([0, 1], device="cpu")
    z = torch.zeros(1, 8, 8, 3, dtype=torch.uint8)
    with pytest.raises(GcfrError):
        assemble_batch(z, z[..., 0], z[..., 0], z[..., 0])
    with pytest.raises(GcfrError):
        masked_metrics(z, z, z[..., 0])


def test_reader_against_the_references_own_load_data(tmp_path, monkeypatch, capsys):
    """Authoring container only (skipped without /root/reference): the reference's OWN `load_data()`
    (train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:527-558), imported unmodified through oracle/ref_shim.py and run in a
    working directory whose 'MP_data/' is the synthetic dataset of this test.  Seams: `imageio.imread` -> Pillow (what
    imageio uses for jpg / png), and `np.zeros` shrinks the hard-coded 29,890 leading dimension to the number of files (the
    script allocates 110 GB of float64 up front).  Its six float64 arrays against the byte arrays of RelightDataset:
    identical after the script's own conversions."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import ref_shim
    if not ref_shim.reference_available():
        pytest.skip("/root/reference not present")
    from PIL import Image
    from geomconsistentfr_amd.dataset import RelightDataset
    n = 4
    root = tmp_path / "MP_data"
    write_dataset(str(root), n)
    T8 = ref_shim.load("T8")
    import imageio
    monkeypatch.setattr(imageio, "imread", lambda p: np.asarray(Image.open(p)), raising=False)
    real_zeros = np.zeros
    monkeypatch.setattr(np, "zeros", lambda shape, *a, **k: real_zeros((n,) + tuple(shape[1:]) if isinstance(shape, tuple) and shape and shape[0] == 29890 else shape, *a, **k))
    monkeypatch.chdir(tmp_path)
    images, lightings, depths, masks, albedo, fill = T8.load_data()              # prints the index of every face (T8:544)
    capsys.readouterr()
    ds = RelightDataset(str(root))
    assert images.dtype == np.float64 and images.shape == (n, 256, 256, 3)
    np.testing.assert_array_equal(ds.images / 255.0, images)                       # T8:550
    np.testing.assert_array_equal(ds.masks[..., None].astype(np.float64), masks)   # T8:546
    np.testing.assert_array_equal(ds.albedo.astype(np.float64), albedo)            # T8:551
    np.testing.assert_array_equal(ds.depths.astype(np.float64), depths.astype(np.float32).astype(np.float64))
    np.testing.assert_allclose(ds.lightings, lightings, rtol=1e-7)                 # ambient target 0.5 in column 0 (T8:542)
    np.testing.assert_array_equal(np.where(np.maximum(ds.face_masks, ds.masks) > 128, 255.0, 0.0)[..., None], fill)   # T8:552-556
    # and this test's line-for-line statement of load_data is the reference's function
    for a, b in zip(load_data_statement(str(root)), (images, lightings, depths, masks, albedo, fill)):
        np.testing.assert_array_equal(a, b)
