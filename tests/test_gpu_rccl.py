"""GPU: the real RCCL / DistributedDataParallel path of the training harness on the leased GPU.

The driver's GPU box has one GPU, so the process group has world_size 1 -- but it IS the `nccl` backend (= RCCL on
ROCm): communicator creation, DDP's bucketed gradient all-reduce over RCCL, barrier and all_reduce all run through
librccl, which is what an 8-GPU launch uses with more ranks.  Multi-rank behaviour is covered on CPU with gloo
(tests/test_train_host.py); an 8-GPU curve is the driver's to measure.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def test_trainer_distributed_over_rccl_world1_d_step_and_g_step_from_epoch_zero():
    import torch.distributed as dist
    from geomconsistentfr_amd.train import TrainConfig, Trainer, synthetic_batch
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 34500 + (os.getpid() % 2000)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1, device_id=DEV)
    try:
        assert dist.get_backend() == "nccl"
        torch.manual_seed(0)
        tr = Trainer(TrainConfig(miopen_find=False), device=DEV, distributed=True)
        from torch.nn.parallel import DistributedDataParallel as DDP
        assert isinstance(tr.net, DDP) and isinstance(tr.disc, DDP)
        before = torch.cat([p.detach().flatten().clone() for p in tr.model.parameters()])
        d_before = torch.cat([p.detach().flatten().clone() for p in tr.patchgan.parameters()])
        batch = synthetic_batch(4, 0, device=DEV)
        # epoch 0: every epoch-gated skip is off (T8:245-283) -- the case DDP needs find_unused_parameters for;
        # j = 0 runs the discriminator step AND the generator step (T8:624), j = 1 the generator step alone
        import warnings
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            logs0 = tr.step(batch, epoch=0, j=0)
            logs1 = tr.step(batch, epoch=0, j=1)
            logs2 = tr.step(batch, epoch=200, j=5)              # all skips on, D step again
        # (round 5) the [1, 16, 1, 1] head weights' gradients reach the reducer with canonical strides: no layout warning
        assert not [w for w in caught if "Grad strides do not match" in str(w.message)], [str(w.message)[:200] for w in caught]
        assert "discriminator" in logs0 and "discriminator" not in logs1 and "discriminator" in logs2
        for lg in (logs0, logs1, logs2):
            assert all(np.isfinite(v) for v in lg.values()), lg
        after = torch.cat([p.detach().flatten() for p in tr.model.parameters()])
        d_after = torch.cat([p.detach().flatten() for p in tr.patchgan.parameters()])
        assert not torch.equal(before, after) and not torch.equal(d_before, d_after)
        assert torch.isfinite(after).all() and torch.isfinite(d_after).all()
        # the collectives bench.py uses around its timed region, on the same communicator
        t = torch.tensor([3.5], dtype=torch.float64, device=DEV)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        torch.cuda.synchronize()
        assert float(t.item()) == 3.5
    finally:
        dist.destroy_process_group()


def _rccl_world2_worker(rank, world, port, q, backend="nccl", share_gpu=False):
    """One rank of the 2-rank tests below (spawned; its own process; its own GPU over `nccl`, or -- the rehearsal on a 1-GPU
    box -- GPU 0 shared by both ranks with the collectives over `gloo`)."""
    try:
        import copy
        import torch.distributed as dist
        from geomconsistentfr_amd.train import TrainConfig, Trainer, shard_range, synthetic_batch
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"            # dmabuf IPC: what RCCL needs on this driver
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
        dev = torch.device("cuda", 0 if share_gpu else rank)
        torch.cuda.set_device(dev)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        assert dist.get_backend() == backend
        # MIOpen's default convolution algorithms are not run-to-run reproducible here (the hourglass' depth output differs by
        # ~1e-3 between two forwards of the same input, tools/diag_determinism.py), and at epoch 0 the shadow of a rough depth
        # map amplifies that into per-cent differences of the gradients: the deterministic algorithms for this comparison
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
        torch.manual_seed(0)                                       # the same initial weights on every rank
        tr = Trainer(TrainConfig(miopen_find=False), device=dev, distributed=True)
        full = synthetic_batch(2 * world, 0, device=dev)           # whole faces per rank, no data-path collective
        lo, hi = shard_range(2 * world, rank, world)
        mine = {k: v[lo:hi] for k, v in full.items()}

        def loss_of(net, b):
            out = net(b["images"], 0, tr.K, b["masks_fill"])
            return out[5].mean() + out[2].mean() + 1e-3 * out[1].abs().mean()

        # (a) DDP's all-reduced gradient == mean of the per-shard gradients (BatchNorm statistics stay per shard)
        ref = []
        for r in range(world):
            m2 = copy.deepcopy(tr.model)
            a, b = shard_range(2 * world, r, world)
            loss_of(m2, {k: v[a:b] for k, v in full.items()}).backward()
            ref.append(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m2.parameters()]))
            del m2
        ref = torch.stack(ref).mean(0)
        tr.opt.zero_grad(set_to_none=True)
        loss_of(tr.net, mine).backward()
        got = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in tr.model.parameters()])
        g_err, g_scale = float((got - ref).abs().max()), float(ref.abs().max())
        tr.opt.zero_grad(set_to_none=True)
        # (b) two optimiser steps at epoch 0 (every skip gated off: find_unused_parameters): D + G, then G alone
        logs0 = tr.step(mine, epoch=0, j=0)
        logs1 = tr.step(mine, epoch=0, j=1)
        ok_logs = ("discriminator" in logs0 and "discriminator" not in logs1
                   and all(np.isfinite(v) for lg in (logs0, logs1) for v in lg.values()))
        flat = torch.cat([p.detach().flatten() for p in list(tr.model.parameters()) + list(tr.patchgan.parameters())])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        p_diff = max(float((g - gathered[0]).abs().max()) for g in gathered[1:])
        torch.cuda.synchronize()
        q.put((rank, None, g_err, g_scale, p_diff, ok_logs))
        dist.destroy_process_group()
    except Exception as e:                                         # surface the failure instead of a parent time-out
        q.put((rank, "error: %r" % (e,), 0.0, 0.0, 0.0, False))
        raise


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (the driver's lease has one): RCCL with world size 2")
def test_trainer_over_rccl_world2_gradients_are_the_mean_of_the_shards_and_parameters_stay_equal():
    """tests/test_train_host.py's world-2 gloo checks on the real backend: two ranks, one GPU each, `nccl` (= RCCL over
    xGMI).  Skips on a 1-GPU lease; runs the day a node with >= 2 GPUs runs `pytest -m gpu`."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 36500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rccl_world2_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] is None for r in res), res
    assert all(p.exitcode == 0 for p in procs)
    _check_world2(res)


def _check_world2(res):
    for rank, _, g_err, g_scale, p_diff, ok_logs in res:
        assert g_scale > 0 and g_err <= 1e-4 * g_scale, (rank, g_err, g_scale)   # (depth-gradient atomics: order jitter)
        assert p_diff == 0.0, (rank, p_diff)
        assert ok_logs


def test_trainer_world2_rehearsal_two_ranks_share_the_gpu_over_gloo():
    """The same two-rank check on the hardware there is: both ranks on GPU 0 (RCCL refuses two ranks on one device), DDP's
    gradient all-reduce over `gloo`.  It exercises everything of the test above except the transport -- the HIP render block
    forward and backward inside DistributedDataParallel on two processes, whole faces per rank, gradients = mean of the shards,
    parameters equal after a D+G step and a G step at epoch 0."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 38500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_rccl_world2_worker, args=(r, 2, port, q, "gloo", True)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=120)
    assert all(r[1] is None for r in res), res
    assert all(p.exitcode == 0 for p in procs)
    _check_world2(res)
