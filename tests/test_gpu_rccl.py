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
        logs0 = tr.step(batch, epoch=0, j=0)
        logs1 = tr.step(batch, epoch=0, j=1)
        logs2 = tr.step(batch, epoch=200, j=5)              # all skips on, D step again
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
