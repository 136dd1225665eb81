"""GPU: the SSIM term's depthwise blurs through ATen's own kernels (train._DepthwiseBlur, round 6) against the MIOpen form of rounds
2-5 and against the CPU restatement (which tests/test_train_host.py holds to a naive numpy SSIM): the same fp32 sums in another
order -- value within 2e-6 relative, gradient within 2e-5 of its largest entry -- and the training step's eleven logged losses of one
step agree between the two forms to the noise MIOpen's own solver choice has (tools/diag_determinism.py)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _pair(B=4, H=256, W=256, seed=0):
    rng = np.random.default_rng(seed)
    Y = rng.random((B, 3, H, W), dtype=np.float32)
    X = np.clip(Y + 0.08 * rng.standard_normal(Y.shape).astype(np.float32), 0, 1)
    return X, Y


@pytest.mark.parametrize("size_average", [True, False])
def test_aten_blur_equals_miopen_blur_and_the_cpu_restatement(size_average):
    from geomconsistentfr_amd.train import ssim
    X, Y = _pair()
    res = {}
    for where, kern in (("cpu", "miopen"), ("gpu", "miopen"), ("gpu", "aten")):
        dev = torch.device("cpu") if where == "cpu" else DEV
        x = torch.from_numpy(X).to(dev).requires_grad_()
        v = ssim(x, torch.from_numpy(Y).to(dev), size_average=size_average, blur_kernels=kern)
        v.sum().backward()
        res[(where, kern)] = (v.detach().cpu().numpy(), x.grad.cpu().numpy())
    ref_v, ref_g = res[("cpu", "miopen")]
    for key in (("gpu", "miopen"), ("gpu", "aten")):
        v, g = res[key]
        np.testing.assert_allclose(v, ref_v, rtol=2e-6, err_msg=str(key))
        assert np.abs(g - ref_g).max() <= 2e-5 * np.abs(ref_g).max(), (key, np.abs(g - ref_g).max(), np.abs(ref_g).max())
    assert np.abs(ref_g).max() > 0


def test_odd_sizes_and_channel_counts():
    from geomconsistentfr_amd.train import ssim
    rng = np.random.default_rng(5)
    for B, C, H, W in ((1, 1, 33, 47), (2, 3, 64, 40), (3, 2, 21, 128)):
        Y = rng.random((B, C, H, W), dtype=np.float32)
        X = np.clip(Y + 0.1 * rng.standard_normal(Y.shape).astype(np.float32), 0, 1)
        vals = []
        for kern in ("miopen", "aten"):
            x = torch.from_numpy(X).to(DEV).requires_grad_()
            v = ssim(x, torch.from_numpy(Y).to(DEV), blur_kernels=kern)
            v.backward()
            vals.append((float(v), x.grad.cpu().numpy()))
        assert abs(vals[0][0] - vals[1][0]) <= 2e-6 * abs(vals[0][0])
        assert np.abs(vals[0][1] - vals[1][1]).max() <= 2e-5 * np.abs(vals[0][1]).max()


def test_one_training_step_logs_the_same_losses_with_either_blur():
    """One step of the Trainer from the same seed with ssim_blur = "miopen" and "aten": every logged loss agrees to 1e-4 relative
    (the convolutions in front of the losses are MIOpen's and differ from run to run by more than the blur does)."""
    from geomconsistentfr_amd.train import TrainConfig, Trainer, synthetic_batch
    logs = {}
    for kern in ("miopen", "aten"):
        torch.manual_seed(77)
        tr = Trainer(TrainConfig(ssim_blur=kern), device=DEV)
        batch = synthetic_batch(4, 0, device=DEV)
        logs[kern] = tr.step(batch, 200, 0, log=True)
    assert set(logs["miopen"]) == set(logs["aten"]) and "DSSIM" in logs["aten"]
    for k, v in logs["miopen"].items():
        assert abs(v - logs["aten"][k]) <= 1e-4 * max(abs(v), 1e-3), (k, v, logs["aten"][k])
