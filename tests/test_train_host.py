"""CPU: host-side logic of the training harness -- sharding, SSIM restatement, and the N>1 path
(world_size-2 gloo DistributedDataParallel gradient all-reduce on the network's CPU-runnable part)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions_faces_exactly():
    from geomconsistentfr_amd.train import shard_range
    for n in (0, 1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _ssim_naive(X, Y, L=1.0):
    """Independent numpy SSIM: explicit 11x11 Gaussian window, valid region, K=(0.01,0.03)."""
    g = np.exp(-((np.arange(11) - 5) ** 2) / (2 * 1.5 ** 2))
    g /= g.sum()
    w2 = np.outer(g, g)
    C1, C2 = (0.01 * L) ** 2, (0.03 * L) ** 2
    B, C, H, W = X.shape
    out = np.zeros((B, C))
    for b in range(B):
        for c in range(C):
            vals = []
            for i in range(H - 10):
                for j in range(W - 10):
                    x, y = X[b, c, i:i + 11, j:j + 11], Y[b, c, i:i + 11, j:j + 11]
                    mx, my = (w2 * x).sum(), (w2 * y).sum()
                    sx, sy = (w2 * x * x).sum() - mx * mx, (w2 * y * y).sum() - my * my
                    sxy = (w2 * x * y).sum() - mx * my
                    vals.append(((2 * mx * my + C1) / (mx * mx + my * my + C1)) * ((2 * sxy + C2) / (sx + sy + C2)))
            out[b, c] = max(np.mean(vals), 0.0)
    return out.mean()


def test_ssim_restatement():
    from geomconsistentfr_amd.train import ssim
    rng = np.random.default_rng(0)
    X = rng.random((2, 3, 24, 20))
    Y = np.clip(X + 0.1 * rng.standard_normal(X.shape), 0, 1)
    Xt, Yt = torch.from_numpy(X), torch.from_numpy(Y)
    assert abs(float(ssim(Xt, Xt)) - 1.0) < 1e-12
    assert abs(float(ssim(Xt, Yt)) - float(ssim(Yt, Xt))) < 1e-12
    assert abs(float(ssim(Xt, Yt)) - _ssim_naive(X, Y)) < 1e-10


def test_ssim_stacked_convolutions_are_bit_equal_to_the_five_separate_ones():
    """Round 6 (the training step's breakdown, profiles/r06_train_step_breakdown.md): the SSIM's five blurred maps from ONE pair of
    depthwise convolutions over the stacked inputs.  A depthwise convolution filters each channel on its own, so value and gradient
    are the same numbers either way -- the value bit for bit on the CPU, where the convolution is deterministic."""
    from geomconsistentfr_amd.train import ssim
    torch.manual_seed(3)
    X0 = torch.rand(2, 3, 48, 40)
    Y = torch.rand(2, 3, 48, 40)
    res = []
    for stacked in (False, True):
        X = X0.clone().requires_grad_()
        v = ssim(X, Y, stacked=stacked)
        v.backward()
        res.append((v.detach().clone(), X.grad.clone()))
    assert torch.equal(res[0][0], res[1][0])                                  # the loss value: the same bits
    # the gradient: X's three contributions (through blur(X), blur(X*X), blur(X*Y)) are added by autograd in another order
    assert float((res[0][1] - res[1][1]).abs().max()) <= 4e-7 * float(res[0][1].abs().max())
    per_image = [ssim(X0, Y, size_average=False, stacked=s) for s in (False, True)]
    assert torch.equal(per_image[0], per_image[1])


def test_synthetic_batch_is_deterministic_and_rank_disjoint():
    from geomconsistentfr_amd.train import synthetic_batch
    a = synthetic_batch(2, 5, 64, 64)
    b = synthetic_batch(2, 5, 64, 64)
    c = synthetic_batch(2, 1_000_005, 64, 64)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert not torch.equal(a["images"], c["images"])
    assert a["images"].shape == (2, 64, 64, 3) and a["lightings"].shape == (2, 4)
    assert a["masks"].shape == (2, 64, 64, 1) and a["depths"].shape == (2, 64, 64, 1)
    np.testing.assert_allclose(a["lightings"][:, 1:].norm(dim=1).numpy(), 1.0, atol=1e-6)


class _FeatureLoss(torch.nn.Module):
    """RelightNet up to the T8:352 seam (CPU-runnable) with a scalar loss on its three heads."""

    def __init__(self, epoch=200):
        super().__init__()
        from geomconsistentfr_amd.relightnet import RelightNet
        torch.manual_seed(1234)
        self.net = RelightNet()
        self.epoch = epoch

    def forward(self, img):
        albedo, depth, SL = self.net.features(img, self.epoch)
        return albedo.mean() + 1e-3 * depth.abs().mean() + SL.pow(2).mean()


def _ddp_worker(rank, world, port, q):
    try:
        _ddp_worker_body(rank, world, port, q)
    except Exception as e:  # surface the failure instead of letting the parent wait for its timeout
        q.put((rank, "error: %r" % (e,), 0.0, 0))
        raise


def _ddp_worker_body(rank, world, port, q):
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from geomconsistentfr_amd.train import shard_range, synthetic_batch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    m = _FeatureLoss()
    ddp = DDP(m, bucket_cap_mb=32)
    lo, hi = shard_range(4, rank, world)                     # 4 faces over 2 ranks, whole faces per rank
    full = synthetic_batch(4, 0, 256, 256)["images"]         # the lighting head pools 16x16 at 1/16 scale (T8:85)
    ddp(full[lo:hi]).backward()
    g = torch.cat([p.grad.flatten() for p in m.parameters()])
    # single-process reference: mean of the per-shard gradients (BatchNorm statistics stay per shard)
    ref = []
    for r in range(world):
        m2 = _FeatureLoss()
        a, b = shard_range(4, r, world)
        m2(full[a:b]).backward()
        ref.append(torch.cat([p.grad.flatten() for p in m2.parameters()]))
    ref = torch.stack(ref).mean(0)
    q.put((rank, float((g - ref).abs().max()), float(ref.abs().max()), g.numel()))
    dist.destroy_process_group()


def _ddp_epoch0_worker(rank, world, port, q):
    """Two optimiser steps at epoch 0 through DDP with the Trainer's own arguments (train.ddp_kwargs): the
    epoch-gated skip convolutions run in forward but get no gradient (T8:245-283 gates 8/10/12/14)."""
    try:
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel as DDP
        from geomconsistentfr_amd.train import TrainConfig, ddp_kwargs, shard_range, synthetic_batch
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        m = _FeatureLoss(epoch=0)
        ddp = DDP(m, **ddp_kwargs(TrainConfig(), torch.device("cpu"), generator=True))
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        lo, hi = shard_range(2, rank, world)
        full = synthetic_batch(2, 0, 256, 256)["images"]
        for _ in range(2):                                   # the failure mode only shows on the SECOND step
            opt.zero_grad(set_to_none=True)
            ddp(full[lo:hi]).backward()
            opt.step()
        named = dict(m.net.named_parameters())
        unused = [n for n, p in named.items() if p.grad is None or float(p.grad.abs().max()) == 0.0]
        used = torch.cat([p.grad.flatten() for n, p in named.items() if p.grad is not None])
        gathered = [torch.zeros_like(used) for _ in range(world)]
        dist.all_gather(gathered, used)
        q.put((rank, float((gathered[0] - gathered[1]).abs().max()), float(used.abs().max()),
               sum("skip" in n for n in unused)))
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "error: %r" % (e,), 0.0, 0))
        raise


def test_gloo_world2_two_steps_at_epoch_zero_with_gated_skips():
    """Regression (round-1 advisor finding): multi-GPU training from scratch.  At epoch 0 every additive skip is
    gated off, so the skip convolutions' parameters are unused; DDP must be told (find_unused_parameters) or the
    second step raises.  Both ranks end with identical averaged gradients, and the skip parameters got none."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_epoch0_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    assert all(not isinstance(r[1], str) for r in res), res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, diff, scale, n_unused_skip in res:
        assert diff == 0.0 and scale > 0
        assert n_unused_skip >= 16          # 2 branches x 4 stages x (conv + bn weights/biases) of the gated skips


def _ddp_disc_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel as DDP
        from geomconsistentfr_amd.relightnet import PatchGAN
        from geomconsistentfr_amd.train import discriminator_losses
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        torch.manual_seed(7)
        disc = PatchGAN()
        ddp = DDP(disc, bucket_cap_mb=32, gradient_as_bucket_view=True, broadcast_buffers=False)
        g = torch.Generator().manual_seed(100 + rank)
        fake, real = torch.rand(1, 3, 64, 64, generator=g), torch.rand(1, 3, 64, 64, generator=g)
        d_fake, d_real = discriminator_losses(ddp, fake, real)          # two forwards, then ONE backward (T8:619-625)
        (d_fake + d_real).backward()
        grad = torch.cat([p.grad.flatten() for p in disc.parameters()])
        gathered = [torch.zeros_like(grad) for _ in range(world)]
        dist.all_gather(gathered, grad)
        q.put((rank, float((gathered[0] - gathered[1]).abs().max()), float(grad.abs().max()), grad.numel()))
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "error: %r" % (e,), 0.0, 0))
        raise


def test_gloo_world2_discriminator_step_two_forwards_one_backward():
    """Regression: DDP's default per-forward buffer broadcast breaks the D step (two forwards before one backward);
    Trainer wraps with broadcast_buffers=False.  After backward every rank holds the same averaged gradient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_disc_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    assert all(not isinstance(r[1], str) for r in res), res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, diff, scale, n in res:
        assert n == 2_766_529 and scale > 0 and diff == 0.0


def test_gloo_world2_gradient_allreduce_matches_mean_of_shards():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    assert all(not isinstance(r[1], str) for r in res), res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, scale, n in res:
        assert n == 1_204_796
        assert err <= 1e-5 * max(scale, 1e-3), (rank, err, scale)


def _ddp_rewrap_worker(rank, world, port, q):
    """Trainer._wrap_generator's mechanism on CPU: one step at the last gated epoch under find_unused_parameters=True,
    then the wrapper is dropped and the SAME module re-wrapped without the flag for two steps at the next epoch."""
    try:
        import torch.distributed as dist
        from torch.nn.parallel import DistributedDataParallel as DDP
        from geomconsistentfr_amd.train import LAST_GATED_EPOCH, TrainConfig, ddp_kwargs, shard_range, synthetic_batch
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        m = _FeatureLoss(epoch=LAST_GATED_EPOCH)
        kw0 = ddp_kwargs(TrainConfig(), torch.device("cpu"), generator=True, epoch=LAST_GATED_EPOCH)
        kw1 = ddp_kwargs(TrainConfig(), torch.device("cpu"), generator=True, epoch=LAST_GATED_EPOCH + 1)
        flags = (kw0["find_unused_parameters"], kw1["find_unused_parameters"])
        opt = torch.optim.Adam(m.parameters(), lr=1e-4)
        lo, hi = shard_range(2, rank, world)
        full = synthetic_batch(2, 0, 256, 256)["images"]
        ddp = DDP(m, **kw0)
        opt.zero_grad(set_to_none=True)
        ddp(full[lo:hi]).backward()
        opt.step()
        ddp = None                                            # the old reducer goes, with its autograd hooks
        m.epoch = LAST_GATED_EPOCH + 1                        # every skip branch now reaches the loss
        ddp = DDP(m, **kw1)
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            ddp(full[lo:hi]).backward()
            opt.step()
        grads = [p.grad for p in m.parameters()]
        n_none = sum(g is None for g in grads)
        flat = torch.cat([g.flatten() for g in grads if g is not None])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        q.put((rank, float((gathered[0] - gathered[1]).abs().max()), float(flat.abs().max()), (flags, n_none)))
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, "error: %r" % (e,), 0.0, 0))
        raise


def test_gloo_world2_generator_is_rewrapped_without_find_unused_after_the_last_gated_epoch():
    """find_unused_parameters is needed only while epoch <= 14 (T8:245-283); after that every
    parameter gets a gradient, the re-wrapped DDP (flag off) keeps the ranks' gradients identical."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_rewrap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    assert all(not isinstance(r[1], str) for r in res), res
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, diff, scale, (flags, n_none) in res:
        assert flags == (True, False)
        assert diff == 0.0 and scale > 0 and n_none == 0
