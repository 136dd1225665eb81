"""GPU: batch assembly from bytes (gcfr_assemble_batch_u8) against load_data()'s float64 arithmetic
(train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:545-556, 607-615), the assembled batch through one Trainer.step, and the
device reductions of the MATLAB metrics (gcfr_masked_metrics_u8) against the oracle's numpy statements of MSE_MP.m:24 /
DSSIM_MP_RGB.m:24-26."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from make_dataset_fixture import write_dataset  # noqa: E402
from test_dataset_host import load_data_statement  # noqa: E402
import postprocess_statements as st  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_batch_from_bytes_equals_the_scripts_float64_batch(tmp_path):
    from geomconsistentfr_amd.dataset import RelightDataset
    write_dataset(str(tmp_path), 5)
    ds = RelightDataset(str(tmp_path))
    images, lightings, deps, msk, alb, fill = load_data_statement(str(tmp_path))
    idx = [3, 0, 4]
    b = ds.batch(idx, device=DEV)
    torch.cuda.synchronize()
    # T8:607-615: slices of the f64 arrays, /255.0 where load_data had not divided, then .float() at the call (T8:618)
    np.testing.assert_array_equal(b["images"].cpu().numpy(), images[idx].astype(np.float32))
    np.testing.assert_array_equal(b["masks"].cpu().numpy(), (msk[idx] / 255.0).astype(np.float32))
    np.testing.assert_array_equal(b["masks_fill"].cpu().numpy(), (fill[idx] / 255.0).astype(np.float32))
    np.testing.assert_array_equal(b["albedo"].cpu().numpy()[..., 0], (alb[idx] / 255.0).astype(np.float32))
    np.testing.assert_allclose(b["lightings"].cpu().numpy(), lightings[idx], rtol=1e-7)
    np.testing.assert_array_equal(b["depths"].cpu().numpy(), deps[idx].astype(np.float32))
    assert set(np.unique(b["masks_fill"].cpu().numpy())) <= {0.0, 1.0}


def test_assembled_batch_drives_a_training_step(tmp_path):
    from geomconsistentfr_amd.dataset import RelightDataset
    from geomconsistentfr_amd.train import TrainConfig, Trainer
    write_dataset(str(tmp_path), 3)
    ds = RelightDataset(str(tmp_path))
    torch.manual_seed(0)
    tr = Trainer(TrainConfig(miopen_find=False), device=DEV)
    logs = tr.step(ds.batch([0, 1, 2], device=DEV), 0, 0)                      # T8:617-656 on a batch of 3, epoch 0
    for k in ("recon", "depth", "ambient", "lighting", "albedo", "generator", "DSSIM", "total", "discriminator"):
        assert np.isfinite(logs[k]), (k, logs)


@pytest.mark.parametrize("B,H,W,shared", [(3, 64, 48, False), (2, 256, 256, True), (1, 37, 29, False)])
def test_masked_metrics_match_the_matlab_statements(B, H, W, shared):
    from geomconsistentfr_amd.dataset import masked_metrics
    rng = np.random.default_rng(H + B)
    gt = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    recon = np.clip(gt.astype(int) + rng.integers(-25, 26, gt.shape), 0, 255).astype(np.uint8)
    recon[0, : H // 3] = gt[0, : H // 3]                                            # an identical region
    mask = rng.choice([0, 64, 255], size=(1 if shared else B, H, W), p=[0.5, 0.1, 0.4]).astype(np.uint8)
    t = lambda a: torch.from_numpy(a).to(DEV)
    mse, dssim = masked_metrics(t(recon), t(gt), t(mask))
    mse, dssim = mse.cpu().numpy(), dssim.cpu().numpy()
    for b in range(B):
        m = mask[0 if shared else b]
        np.testing.assert_allclose(mse[b], st.masked_mse(recon[b], gt[b], m), rtol=1e-12)
        np.testing.assert_allclose(dssim[b], st.masked_dssim(recon[b], gt[b], m), rtol=1e-9, atol=1e-12)
    same, _ = masked_metrics(t(gt), t(gt), t(mask))
    assert float(same.abs().max()) == 0.0
