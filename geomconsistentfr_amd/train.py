"""Training-step harness around the HIP render block (BASELINE configs 3-4).

Mirrors the reference's step (train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:582-656): PatchGAN
discriminator step every GD_ratio iterations, then the generator step with seven losses -- masked L2
reconstruction x20, masked L1 depth, ambient L1 x2.5, 1-cos light direction, grey-albedo L1 x5,
GAN BCE x0.01, DSSIM x8/2.  Everything here is stock PyTorch-ROCm (consumer of the render block's
outputs); the only hand-written device code on the step is the render block itself.

Data parallelism: faces shard over ranks (whole faces per GPU, `shard_range`), the render block needs no
collective, and the step adds one RCCL gradient all-reduce per optimiser step through
DistributedDataParallel -- RelightNet 1,204,796 f32 = 4.8 MB every step, PatchGAN 2,766,529 f32 = 11 MB
every GD_ratio-th step; both fit one flat bucket (bucket_cap_mb=32), which is what a latency-bound
all-reduce on point-to-point xGMI wants.  BatchNorm statistics stay per GPU, as in the reference (B=3).

Deviations that do not change any parameter update: the D step uses rendered.detach() (the reference
back-propagates d_loss into the generator with retain_graph=True and then zeroes those grads, T8:624,631);
PatchGAN's parameters are frozen during the G backward (the reference accumulates and later zeroes them).
pytorch_msssim is un-vendored: `ssim` below restates its published algorithm (PARITY UNPINNED, off the hot path).
"""
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .relightnet import PatchGAN, RelightNet


# ------------------------------------------------------------------------------------------------
# sharding
# ------------------------------------------------------------------------------------------------
def shard_range(n_items: int, rank: int, world: int):
    """Contiguous, balanced [lo, hi) of `n_items` faces for `rank` (first n%world ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# ------------------------------------------------------------------------------------------------
# SSIM (restatement of pytorch_msssim.ssim as called at T8:643)
# ------------------------------------------------------------------------------------------------
def _gauss_window(size: int, sigma: float, device, dtype):
    c = torch.arange(size, dtype=dtype, device=device) - size // 2
    g = torch.exp(-(c ** 2) / (2 * sigma ** 2))
    return g / g.sum()


class _DepthwiseBlur(torch.autograd.Function):
    """Separable 'valid' Gaussian blur of every channel (the SSIM's window), forward AND backward through ATen's own depthwise
    convolution kernels instead of MIOpen (round 6, profiles/r06_train_step_breakdown.md).  MIOpen has no tuned solver for a
    depthwise 11 x 1 convolution over 3 channels: it runs CK's grouped-convolution kernel at 3 groups (70 us per 25-MB map, five
    times what the bytes need) or, for the 15-group stacked form, its NAIVE reference kernel (`naive_conv_ab_nonpacked_*`, 0.7 ms
    forward and 1.25 ms backward per launch) -- the SSIM term cost ~3.4 ms of the 33-ms step.  With `cudnn.flags(enabled=False)`
    (which is what switches MIOpen off in PyTorch-ROCm) the dispatcher takes `conv_depthwise2d`, a bandwidth-bound kernel.  The
    window is symmetric, so the backward of a valid correlation is the same correlation of the zero-padded upstream gradient:
    two more depthwise convolutions, under the same switch (autograd's own convolution backward re-selects the backend when it
    RUNS, outside the forward's context).  Same fp32 sums of the same eleven products per output; the order of the additions is
    the kernel's, as it is MIOpen's in the other form (which differs between its solvers too).
    (`cudnn.flags` is PROCESS-wide state, switched for the few microseconds these four calls take to enqueue: a convolution issued
    by another host thread of the same process in that window would take ATen's path too -- correct, slower.  One process per GPU
    with one training thread, as `Trainer` runs, has no such thread; autograd's backward thread runs while the main thread waits.)"""

    @staticmethod
    def forward(ctx, t, wh, ww):
        C = t.shape[1]
        ctx.save_for_backward(wh, ww)
        with torch.backends.cudnn.flags(enabled=False):
            return F.conv2d(F.conv2d(t, wh, groups=C), ww, groups=C)

    @staticmethod
    def backward(ctx, g):
        wh, ww = ctx.saved_tensors
        C = g.shape[1]
        p = wh.shape[2] - 1
        with torch.backends.cudnn.flags(enabled=False):
            g = F.conv2d(F.conv2d(g.contiguous(), ww, groups=C, padding=(0, p)), wh, groups=C, padding=(p, 0))
        return g, None, None


def ssim(X: torch.Tensor, Y: torch.Tensor, data_range: float = 1.0, size_average: bool = True,
         nonnegative_ssim: bool = True, win_size: int = 11, win_sigma: float = 1.5, stacked: bool = False,
         blur_kernels: str = "aten") -> torch.Tensor:
    """Gaussian-window SSIM, separable 'valid' filtering, per-channel mean, K = (0.01, 0.03).
    `blur_kernels` (device tensors only): "aten" = the depthwise convolutions through ATen's own kernels (`_DepthwiseBlur`),
    "miopen" = plain F.conv2d as in rounds 2-5 (MIOpen picks the solver).  On the CPU both are F.conv2d.
    `stacked` (round 6): the five blurred maps -- X, Y, X*X, Y*Y, X*Y -- come from ONE pair of depthwise convolutions over the
    five inputs stacked on the channel axis (15 groups) instead of five pairs over 3 groups: a depthwise convolution filters
    every channel on its own, so each map is the same sum of the same eleven products either way (value bit-equal on the CPU,
    tests/test_train_host.py; the gradient with respect to X is the same three terms added in another order by autograd: 1 ulp);
    ten convolution launches become two in the forward, and as many in the backward.  MEASURED AND NOT KEPT as the default:
    with MIOpen's kernels the stacked form is 0.45 ms per step SLOWER (forward -0.45 ms, backward +0.97 ms: it lands on the naive
    solver; profiles/r06_ssim_ab.txt)."""
    C = X.shape[1]
    g = _gauss_window(win_size, win_sigma, X.device, X.dtype)
    n_maps = 5 if stacked else 1
    wh = g.view(1, 1, -1, 1).repeat(n_maps * C, 1, 1, 1)
    ww = g.view(1, 1, 1, -1).repeat(n_maps * C, 1, 1, 1)

    if blur_kernels not in ("aten", "miopen"):
        raise ValueError("blur_kernels must be 'aten' or 'miopen'")

    def blur(t):
        if blur_kernels == "aten" and t.is_cuda:
            return _DepthwiseBlur.apply(t, wh, ww)
        return F.conv2d(F.conv2d(t, wh, groups=n_maps * C), ww, groups=n_maps * C)

    C1, C2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    if stacked:
        mu1, mu2, xx, yy, xy = blur(torch.cat([X, Y, X * X, Y * Y, X * Y], dim=1)).split(C, dim=1)
    else:
        mu1, mu2, xx, yy, xy = blur(X), blur(Y), blur(X * X), blur(Y * Y), blur(X * Y)
    s1 = xx - mu1 * mu1
    s2 = yy - mu2 * mu2
    s12 = xy - mu1 * mu2
    cs = (2 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2 * mu1 * mu2 + C1) / (mu1 * mu1 + mu2 * mu2 + C1)) * cs
    per_channel = ssim_map.flatten(2).mean(-1)
    if nonnegative_ssim:
        per_channel = torch.relu(per_channel)
    val = per_channel.mean(1)
    return val.mean() if size_average else val


# ------------------------------------------------------------------------------------------------
# synthetic data (no dataset exists in the repository: SURVEY.md 8d)
# ------------------------------------------------------------------------------------------------
def synthetic_batch(B: int, seed0: int, H: int = 256, W: int = 256, device="cpu") -> Dict[str, torch.Tensor]:
    """Deterministic stand-in for load_data() (T8:527-558): images, [ambient, light] targets, depth,
    masks (skin, fill-nose-and-mouth), grey albedo; face i uses numpy default_rng(seed0 + i)."""
    r, c = np.mgrid[0:H, 0:W]
    x, y = c - W / 2.0, r - H / 2.0
    sx, sy = W / 256.0, H / 256.0
    imgs, lights, depths, masks, masks_fill, albedos = [], [], [], [], [], []
    for i in range(B):
        rng = np.random.default_rng(seed0 + i)
        ax, ay = (85 + 10 * rng.random()) * sx, (105 + 10 * rng.random()) * sy
        d = 80 * sx * np.sqrt(np.maximum(1 - (x / ax) ** 2 - (y / ay) ** 2, 0)) \
            + (30 + 10 * rng.random()) * sx * np.exp(-((x / sx) ** 2 / 288 + ((y / sy) - 12) ** 2 / 648))
        m = (((x / (ax - 8 * sx)) ** 2 + (y / (ay - 8 * sy)) ** 2) < 1).astype(np.float32)
        l = rng.standard_normal(3)
        l[2] = abs(l[2]) + 0.3
        l /= np.linalg.norm(l)
        base = 0.35 + 0.4 * rng.random()
        alb = np.clip(base + 0.08 * np.sin(c / 11.0 + i) * np.cos(r / 13.0), 0.05, 0.95)
        shade = 0.5 + 0.5 * np.clip(d / (80 * sx), 0, 1)
        img = np.clip(alb[..., None] * shade[..., None] * (0.8 + 0.2 * rng.random(3)), 0, 1)
        imgs.append(img.astype(np.float32))
        lights.append(np.concatenate([[0.4 + 0.2 * rng.random()], l]).astype(np.float32))
        depths.append(d.astype(np.float32)[..., None])
        masks.append(m[..., None])
        masks_fill.append(m[..., None])
        albedos.append(alb.astype(np.float32)[..., None])
    t = lambda a: torch.from_numpy(np.stack(a)).to(device)
    return dict(images=t(imgs), lightings=t(lights), depths=t(depths), masks=t(masks),
                masks_fill=t(masks_fill), albedo=t(albedos))


# ------------------------------------------------------------------------------------------------
# losses and the step
# ------------------------------------------------------------------------------------------------
def generator_losses(out, batch, logits_fake_for_g, ssim_stacked: bool = False, ssim_blur: str = "aten") -> Dict[str, torch.Tensor]:
    """The seven generator-side terms of T8:633-645.  `out` is RelightNet.forward's 8-tuple."""
    albedo, depth, _w, _amb_l, _full, rendered, unit_light, ambient_values = out
    B = rendered.shape[0]
    img = batch["images"].permute(0, 3, 1, 2)
    m3 = batch["masks_fill"].permute(0, 3, 1, 2).expand(-1, 3, -1, -1)     # (a view: the reference's .repeat copies the mask three times, T8:619)
    L = {}
    L["recon"] = 20.0 * F.mse_loss(rendered * m3, img * m3, reduction="sum") / m3.sum()                     # T8:633
    L["depth"] = F.l1_loss(depth.permute(0, 2, 3, 1) * batch["masks"], batch["depths"] * batch["masks"],
                           reduction="sum") / batch["masks"].sum()                                         # T8:634
    L["ambient"] = 2.5 * F.l1_loss(ambient_values, batch["lightings"][:, 0].reshape(B, 1, 1))               # T8:635
    L["lighting"] = torch.sum(1 - torch.sum(unit_light * batch["lightings"][:, 1:4].reshape(B, 3, 1, 1), dim=1)) / B
    grey = albedo.mean(1).reshape(B, albedo.shape[2], albedo.shape[3], 1)
    L["albedo"] = 5.0 * F.l1_loss(grey * batch["masks_fill"], batch["albedo"] * batch["masks_fill"],
                                  reduction="sum") / batch["masks_fill"].sum()                              # T8:639
    L["generator"] = 0.01 * F.binary_cross_entropy_with_logits(logits_fake_for_g, torch.ones_like(logits_fake_for_g))
    composite = rendered * m3 + (1.0 - m3) * img
    L["DSSIM"] = 8.0 * (1 - ssim(composite, img, data_range=1.0, size_average=True, nonnegative_ssim=True, stacked=ssim_stacked,
                                blur_kernels=ssim_blur)) / 2.0
    L["total"] = sum(L.values())
    return L


def discriminator_losses(disc: nn.Module, fake: torch.Tensor, real: torch.Tensor):
    """T8:619-623: two separate forwards (batch-stat BatchNorm sees fake and real separately), BCE x0.01 each."""
    lf = disc(fake)
    lr_ = disc(real)
    d_fake = 0.01 * F.binary_cross_entropy_with_logits(lf, torch.zeros_like(lf))
    d_real = 0.01 * F.binary_cross_entropy_with_logits(lr_, torch.ones_like(lr_))
    return d_fake, d_real


@dataclass
class TrainConfig:
    lr: float = 1e-4            # T8:44
    gd_ratio: int = 5           # T8:49
    focal: float = 1570.0       # T8:572
    H: int = 256
    W: int = 256
    shortcut: str = "3x3"
    bucket_cap_mb: int = 32     # one flat bucket per model (see module docstring)
    miopen_find: bool = True    # torch.backends.cudnn.benchmark: MIOpen picks the fastest conv algorithm (+9 % step rate)
    ssim_stacked: bool = False  # the SSIM's five blurs as one pair of depthwise convolutions over stacked inputs (`ssim`): measured
                                # 0.45 ms per step slower with MIOpen's kernels (profiles/r06_ssim_ab.txt): off
    ssim_blur: str = "aten"     # "aten": the SSIM's depthwise blurs through ATen's conv_depthwise2d kernels (`_DepthwiseBlur`);
                                # "miopen": F.conv2d as in rounds 2-5 (CK grouped / naive solvers: ~3 ms per step more)
    render_pixels: str = "all"  # "mask": the render block leaves out the pixels outside the mask (RenderParams.pixels; every loss
                                # multiplies them by the mask, T8:619-643: bit-equal losses, half the training march)


LAST_GATED_EPOCH = 14      # T8:245, 258, 271, 283: the decoders' skip additions switch on after epochs 8 / 10 / 12 / 14


def ddp_kwargs(cfg: "TrainConfig", device: torch.device, generator: bool, epoch: int = 0) -> dict:
    """DistributedDataParallel arguments of the two models.
    broadcast_buffers=False: BatchNorm running statistics stay per GPU (module docstring), and the discriminator
    runs two forwards (fake, real) before one backward -- DDP's per-forward buffer broadcast would overwrite BN
    buffers in place between them and break autograd's version check.
    find_unused_parameters=True for the generator: its skip convolutions are epoch-gated (relightnet._decode,
    T8:245, 258, 271, 283: added only when epoch > 8 / 10 / 12 / 14).  The reference trains from epoch 0 (T8:592),
    where those branches run in forward but never reach the loss; without the flag DDP's reducer waits for their
    gradients for ever and the SECOND step raises "Expected to have finished reduction in the prior iteration".
    Once epoch > LAST_GATED_EPOCH every parameter takes part in every step and the flag only costs a traversal of the
    autograd graph per iteration: Trainer re-wraps the generator without it when training crosses that epoch."""
    kw = dict(device_ids=[device.index] if device.type == "cuda" else None, bucket_cap_mb=cfg.bucket_cap_mb,
              gradient_as_bucket_view=True, broadcast_buffers=False)
    if generator:
        kw["find_unused_parameters"] = epoch <= LAST_GATED_EPOCH
    return kw


def _canonical_grad_strides(module: nn.Module):
    """Gradients of convolution weights with a size-1 dimension (the [1, 16, 1, 1] heads) come out of MIOpen's backward with
    channels-last-looking strides ([16, 1, 16, 16]); DDP's reducer compares them with its bucket view's ([16, 1, 1, 1]) and warns
    "Grad strides do not match bucket view strides" on every rank -- harmless at these sizes, but noise in the one path that
    cannot be tested on hardware here.  A size-1 dimension's stride is free: re-viewing the gradient gives it the canonical
    strides without a copy."""
    for p_ in module.parameters():
        if p_.dim() == 4 and 1 in p_.shape:
            p_.register_hook(lambda g: g.reshape(-1).view(g.shape) if g.is_contiguous() else g.contiguous())


class Trainer:
    """One process per GPU.  `step(batch, epoch, j)` = T8:617-656 for one batch of any size."""

    def __init__(self, cfg: TrainConfig = TrainConfig(), device="cuda", distributed: bool = False,
                 model: Optional[nn.Module] = None, patchgan: Optional[nn.Module] = None):
        self.cfg, self.device = cfg, torch.device(device)
        if cfg.miopen_find and self.device.type == "cuda":
            torch.backends.cudnn.benchmark = True
        if cfg.render_pixels not in ("all", "mask"):
            raise ValueError("TrainConfig.render_pixels must be 'all' or 'mask', got %r" % (cfg.render_pixels,))
        self.model = (model or RelightNet(cfg.shortcut)).float().to(self.device)
        if cfg.render_pixels != "all":
            # NB: a `model` passed in by the caller is MUTATED here (its render_params are replaced): the Trainer owns the model
            import dataclasses
            self.model.render_params = dataclasses.replace(self.model.render_params, pixels=cfg.render_pixels)
        self.patchgan = (patchgan or PatchGAN()).float().to(self.device)
        self.net, self.disc = self.model, self.patchgan
        self.distributed, self._net_find_unused = distributed, None
        if distributed:
            from torch.nn.parallel import DistributedDataParallel as DDP
            _canonical_grad_strides(self.model)
            _canonical_grad_strides(self.patchgan)
            self._wrap_generator(0)
            self.disc = DDP(self.patchgan, **ddp_kwargs(cfg, self.device, generator=False))
        self.opt = torch.optim.Adam(self.model.parameters(), lr=cfg.lr)                  # T8:589
        self.opt_d = torch.optim.Adam(self.patchgan.parameters(), lr=cfg.lr)             # T8:590
        K = torch.zeros(1, 3, 3, dtype=torch.float64)
        K[:, 0, 0] = K[:, 1, 1] = cfg.focal
        K[:, 2, 2] = 1.0
        K[:, 0, 2], K[:, 1, 2] = cfg.W / 2.0, cfg.H / 2.0
        self.K = K.to(self.device)

    def _wrap_generator(self, epoch: int):
        """(Re-)wrap the generator for `epoch`: find_unused_parameters only while some skip branch is still gated off
        (epoch <= LAST_GATED_EPOCH).  The old wrapper is dropped first -- its reducer removes its autograd hooks when it
        is destroyed -- and the new one broadcasts rank 0's parameters, which are already equal on every rank."""
        from torch.nn.parallel import DistributedDataParallel as DDP
        kw = ddp_kwargs(self.cfg, self.device, generator=True, epoch=epoch)
        if self._net_find_unused == kw["find_unused_parameters"]:
            return
        # BatchNorm running statistics stay per GPU (module docstring).  DDP's constructor synchronises module state from
        # rank 0 (_sync_module_states); whether that includes BUFFERS under broadcast_buffers=False has differed between torch
        # releases, so on a RE-wrap (training crosses epoch 14: every rank holds its own statistics by then) the buffers are
        # snapshotted and put back instead of trusting the installed one.  The FIRST wrap is left to DDP: if rank 0 alone loaded a
        # checkpoint before the Trainer was built, its buffers reach the other ranks with its parameters (advisor r04).
        rewrap = self._net_find_unused is not None
        keep = [(b_, b_.detach().clone()) for b_ in self.model.buffers()] if rewrap else []
        self.net = None                                                       # release the previous reducer's hooks
        self.net = DDP(self.model, **kw)                                      # (callers must not cache `trainer.net`: a
        with torch.no_grad():                                                 #  still-referenced old wrapper keeps its hooks)
            for b_, saved in keep:
                b_.copy_(saved)
        self._net_find_unused = kw["find_unused_parameters"]

    def step(self, batch: Dict[str, torch.Tensor], epoch: int, j: int, log: bool = True) -> Dict[str, float]:
        """log=False skips the per-iteration .item() syncs the reference pays for its 11 prints (T8:657-669)."""
        if self.distributed:
            self._wrap_generator(epoch)
        img = batch["images"].permute(0, 3, 1, 2)
        m3 = batch["masks_fill"].permute(0, 3, 1, 2).expand(-1, 3, -1, -1)
        out = self.net(batch["images"], epoch, self.K, batch["masks_fill"])              # T8:618
        rendered = out[5]
        composite = rendered * m3 + (1.0 - m3) * img
        logs = {}
        # ---- discriminator (T8:617-629) ----
        if j % self.cfg.gd_ratio == 0:
            self.opt_d.zero_grad(set_to_none=True)
            d_fake, d_real = discriminator_losses(self.disc, composite.detach(), img)
            (d_fake + d_real).backward()
            self.opt_d.step()
            if log:
                logs.update(discriminator=float((d_fake + d_real).detach()))
        # ---- generator (T8:631-656) ----
        self.opt.zero_grad(set_to_none=True)
        for p in self.patchgan.parameters():
            p.requires_grad_(False)
        try:
            L = generator_losses(out, batch, self.patchgan(composite), self.cfg.ssim_stacked, self.cfg.ssim_blur)
            L["total"].backward()
        finally:
            for p in self.patchgan.parameters():
                p.requires_grad_(True)
        self.opt.step()
        if log:
            logs.update({k: float(v.detach()) for k, v in L.items()})
        return logs
