"""Host-side mirror of the reference's only callable boundary: RelightNet.forward.

The reference re-declares one network in five flat scripts, each with a hard-coded batch size and the
render block inlined after the depth decoder (train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:38-524,
"T8").  Here the encoder/decoders stay plain PyTorch-ROCm (MIOpen convolutions -- out of the HIP scope,
SURVEY.md section 2), are batch-agnostic, and hand over to the HIP render block at the T8:352 seam.

Drop-in properties kept:
  * module / parameter names equal the reference's, so its state_dicts load unchanged
    (e.g. model_lighting_transfer/model_epoch106.pth with shortcut="1x1");
  * forward signatures and the order / shapes of the returned tuples:
      RelightNet.forward(img NHWC, epoch, intrinsic_matrix, masks (B,H,W,1))      -> 8-tuple  (T8:196, 524)
      RelightNetSingleImage.forward(img, epoch, K, mask (H,W,1), target_lighting (B,3,1,1),
                                    target_ambient_values (B,1,1), batch_mask)     -> 10-tuple (S1:169, 505)
      RelightNetLightingTransfer.forward(img, epoch, K, mask, target_lighting,
                                    target_ambient_values)                         -> 12-tuple (SLT:169, 514)
Documented deviation: the reference's full_shading / final_shading / surface_normals are float64 only
because its camera matrix is float64 (torch promotion); here they are float32.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .block import RenderParams, render_from_depth, render_from_depth_prepass


def _lrelu(x):
    return F.leaky_relu(x, 0.2)


class _Hourglass(nn.Module):
    """Residual encoder -> {lighting MLP, albedo decoder, depth decoder} (T8:57-194, 199-350)."""

    # (name, in, out): residual encoder stages with a projected shortcut (T8:61-69)
    _ENC = [("h2", "h1_out", 16, 32), ("h3", "h2_out", 32, 64), ("h4", "h3_out", 64, 155)]
    # decoder stages: (stage, in, out, shortcut name, skip name, skip source channels, epoch gate)
    _DEC = [("h5", 128, 64, "shortcut_all_features", "skip_s1", 8),
            ("h6", 64, 32, "shortcut_h5_out", "skip_s2", 10),
            ("h7", 32, 16, "shortcut_h6_out", "skip_s3", 12),
            ("h8", 16, 16, None, "skip_s4", 14)]

    def __init__(self, shortcut: str = "3x3"):
        super().__init__()
        assert shortcut in ("3x3", "1x1")     # "1x1" = train_lighting_transfer.py:63-69 (bias-free 1x1)
        sc = (dict(kernel_size=3, padding=1) if shortcut == "3x3" else dict(kernel_size=1, bias=False))

        def conv_bn(name, cin, cout, k=3, **kw):
            setattr(self, "conv_" + name, nn.Conv2d(cin, cout, k, padding=k // 2, **kw))
            setattr(self, "bn_" + name, nn.BatchNorm2d(cout))

        conv_bn("c1_og", 3, 16, 5)
        conv_bn("h1_1", 16, 16)
        conv_bn("h1_2", 16, 16)
        for st, src, cin, cout in self._ENC:
            conv_bn(st + "_1", cin, cout)
            conv_bn(st + "_2", cout, cout)
            setattr(self, "conv_shortcut_" + src, nn.Conv2d(cin, cout, **sc))
            setattr(self, "bn_shortcut_" + src, nn.BatchNorm2d(cout))
        self.AvgPool_LF = nn.AvgPool2d((16, 16), (1, 1))     # T8:85
        self.linear_SL1 = nn.Linear(27, 128)
        self.linear_SL2 = nn.Linear(128, 4)
        for br, cout_final in (("albedo", 3), ("depth", 1)):
            for st, cin, cout, scname, skip, _ in self._DEC:
                setattr(self, "deconv_%s_%s_1" % (br, st), nn.ConvTranspose2d(cin, cout, 3, padding=1))
                setattr(self, "deconv_%s_%s_2" % (br, st), nn.ConvTranspose2d(cout, cout, 3, padding=1))
                setattr(self, "bn_%s_%s_1" % (br, st), nn.BatchNorm2d(cout))
                setattr(self, "bn_%s_%s_2" % (br, st), nn.BatchNorm2d(cout))
                if scname:
                    setattr(self, "deconv_%s_%s" % (br, scname), nn.ConvTranspose2d(cin, cout, **sc))
                    setattr(self, "bn_%s_%s" % (br, scname), nn.BatchNorm2d(cout))
                for j in (1, 2):
                    conv_bn("%s_%s_%d" % (br, skip, j), cout, cout)
                setattr(self, "upsample_%s_%s_out" % (br, st), nn.Upsample(scale_factor=2, mode="nearest"))
            conv_bn(br + "_c2_1", 16, 16, 3)
            conv_bn(br + "_c2_2", 16, 16, 1)
            conv_bn(br + "_c2_3", 16, 16, 1)
            setattr(self, "conv_%s_c2_o" % br, nn.Conv2d(16, cout_final, 1))

    # --- helpers -------------------------------------------------------------------------------
    def _cb(self, name, x, kind="conv"):
        return getattr(self, "bn_" + name)(getattr(self, kind + "_" + name)(x))

    def _decode(self, br, identity, skips, epoch):
        x = identity
        for (st, _, _, scname, skip, gate), src in zip(self._DEC, skips):
            a = _lrelu(self._cb("%s_%s_1" % (br, st), x, "deconv"))
            a = self._cb("%s_%s_2" % (br, st), a, "deconv")
            res = self._cb("%s_%s" % (br, scname), x, "deconv") if scname else x
            x = getattr(self, "upsample_%s_%s_out" % (br, st))(_lrelu(res + a))
            s = _lrelu(self._cb("%s_%s_1" % (br, skip), src))
            s = _lrelu(src + self._cb("%s_%s_2" % (br, skip), s))
            if epoch > gate:                     # epoch-gated additive skips, T8:245, 258, 271, 283
                x = x + s
        for j in (1, 2, 3):
            x = _lrelu(self._cb("%s_c2_%d" % (br, j), x))
        return getattr(self, "conv_%s_c2_o" % br)(x)

    _pinned_camera = None

    def pin_camera(self, intrinsic_matrix):
        """Fix the camera for every later forward from a HOST copy of the matrix (None = unpin).
        The reference's call site uploads a fresh `intrinsic_matrix.cuda()` every step (T8:618, S1:588); a fresh device
        tensor cannot be recognised without reading it back -- one device-to-host synchronisation per step.  A drop-in
        that keeps the call site untouched pins the (constant, T8:571-577) matrix once and the per-step argument is then
        ignored; a call site that can change passes the host tensor itself (no `.cuda()`), which is read without a sync."""
        if intrinsic_matrix is None:
            self._pinned_camera = None
            return self
        K = intrinsic_matrix.detach().to("cpu", torch.float64)
        if K.shape[0] != 1 and not bool((K == K[:1]).all()):
            raise ValueError("pin_camera needs one camera matrix for the whole batch")
        self._pinned_camera = K[:1].clone()
        return self

    def _camera(self, intrinsic_matrix):
        return self._pinned_camera if self._pinned_camera is not None else intrinsic_matrix

    def render_lights(self, img, epoch, intrinsic_matrix, mask, lights, ambient_of):
        """MANY target lights per face from ONE network pass: features() once, one prepass (on the side stream, under the
        albedo decoder), one normals stage, L marches -- where the reference's scripts re-run the whole model per light
        (S1:582-588 inside the loop over the light directions).  lights (L,3) shared by the batch, or (B,L,3);
        `ambient_of(SL, B, L)` -> (B,L) ambient values.  Returns (albedo, depth, SL, lights (B,L,3), render dict with a light axis)."""
        cam = self._camera(intrinsic_matrix)
        B, H, W, _ = img.shape
        lights = torch.as_tensor(lights, dtype=torch.float32, device=img.device)
        if lights.dim() == 2:
            lights = lights[None].expand(B, -1, 3)
        lights = lights.reshape(B, -1, 3).contiguous()
        L = lights.shape[1]
        m1 = mask.reshape(1, H, W)
        early = []

        def on_depth(depth, SL):
            early.append(render_from_depth_prepass(depth, lights, cam, m1, self.render_params))

        albedo, depth, SL = self.features(img, epoch, on_depth if (self.hoist_prepass and img.is_cuda) else None)
        r = render_from_depth(depth, albedo, lights, ambient_of(SL, B, L), cam, self.normal_z_offset, m1, self.render_params,
                              prepared=early[0] if early else None)
        return albedo, depth, SL, lights, r

    hoist_prepass = True   # the render block's prepass on a side stream, under the albedo decoder (block.render_prepass)

    def features(self, img_nhwc, epoch, on_depth=None):
        """-> (albedo (B,3,H,W) in (0,1), depth (B,1,H,W) x100, SL_lin2 (B,1,1,4)).  T8:197-350.
        The DEPTH decoder runs first (the reference runs the albedo decoder first, T8:290 then T8:350; the two are independent
        sub-graphs off the same encoder features, so the order changes no value): `on_depth(depth, SL_lin2)`, if given, is called
        as soon as both exist -- RelightNet.forward uses it to put the render block's prepass on a side stream, under the
        albedo decoder's convolutions."""
        img = img_nhwc.permute(0, 3, 1, 2)
        c1_og = _lrelu(self._cb("c1_og", img))
        c1 = F.max_pool2d(c1_og, 2)
        h = _lrelu(self._cb("h1_1", c1))
        h1_out_og = _lrelu(c1 + self._cb("h1_2", h))
        outs = {"h1_out": F.max_pool2d(h1_out_og, 2)}
        ogs = {"h1": h1_out_og}
        for st, src, _, _ in self._ENC:
            x = outs[src]
            a = self._cb(st + "_2", _lrelu(self._cb(st + "_1", x)))
            og = _lrelu(self._cb("shortcut_" + src, x) + a)
            ogs[st] = og
            if st != "h4":
                outs[st + "_out"] = F.max_pool2d(og, 2)
        h4_out = ogs["h4"]
        identity, lighting = h4_out[:, 0:128], h4_out[:, 128:155]                      # T8:225-226
        lf = self.AvgPool_LF(lighting).permute(0, 2, 3, 1)
        SL_lin2 = self.linear_SL2(_lrelu(self.linear_SL1(lf)))                          # (B,1,1,4)
        skips = [ogs["h3"], ogs["h2"], ogs["h1"], c1_og]
        depth = 100.0 * self._decode("depth", identity, skips, epoch)                   # T8:350
        if on_depth is not None:
            on_depth(depth, SL_lin2)
        albedo = torch.sigmoid(self._decode("albedo", identity, skips, epoch))          # T8:290
        return albedo, depth, SL_lin2


class RelightNet(_Hourglass):
    """Training form (T8 / train_lighting_transfer.py): light and ambient predicted by the network."""

    def __init__(self, shortcut: str = "3x3", params: RenderParams = None, normal_z_offset: float = 1610.0):
        super().__init__(shortcut)
        self.render_params = params or RenderParams.training()
        self.normal_z_offset = normal_z_offset                                          # T8:353

    def forward(self, img, epoch, intrinsic_matrix, masks):
        cam = self._camera(intrinsic_matrix)
        m3 = masks.reshape(masks.shape[0], masks.shape[1], masks.shape[2])
        early = []

        def on_depth(depth, SL):  # depth, light and mask exist: everything the prepass reads (gcfr_options.phase = 1)
            early.append(render_from_depth_prepass(depth, SL[:, 0, 0, 1:4], cam, m3, self.render_params))

        albedo, depth, SL = self.features(img, epoch, on_depth if (self.hoist_prepass and img.is_cuda) else None)
        B = depth.shape[0]
        r = render_from_depth(depth, albedo, SL[:, 0, 0, 1:4], SL[:, 0, 0, 0], cam, self.normal_z_offset,
                              m3, self.render_params, prepared=early[0] if early else None)   # T8:353-522
        return (albedo, depth, r["shadow_mask_weights"], r["ambient_light"], r["full_shading"],
                r["rendered_images"], r["unit_light_direction"], r["ambient_values"])   # T8:524


class RelightNetSingleImage(_Hourglass):
    """Inference form with a target light (test_relight_single_image.py, ambient_offset=-0.1 at S1:342;
    test_raytracing_relighting_CelebAHQ_DSSIM_8x.py, ambient_offset=0 at S8:342)."""

    def __init__(self, shortcut: str = "3x3", ambient_offset: float = -0.1, img_height: int = 256,
                 img_width: int = 256, normal_z_offset: float = 1610.0):
        super().__init__(shortcut)
        self.render_params = RenderParams.single_image(img_height, img_width)
        self.ambient_offset = ambient_offset
        self.normal_z_offset = normal_z_offset

    def forward(self, img, epoch, intrinsic_matrix, mask, target_lighting, target_ambient_values, batch_mask=None):
        albedo, depth, SL = self.features(img, epoch)
        B, _, H, W = depth.shape
        ambient = SL[:, 0, 0, 0] + self.ambient_offset                                  # S1:342
        r = render_from_depth(depth, albedo, target_lighting.reshape(B, 3), ambient, self._camera(intrinsic_matrix),
                              self.normal_z_offset, mask.reshape(1, H, W), self.render_params)
        normals = r["surface_normals"]
        return (albedo, depth, r["shadow_mask_weights"], r["ambient_light"], r["full_shading"],
                r["rendered_images"], r["unit_light_direction"], r["ambient_values"], r["final_shading"],
                F.normalize(normals, p=2, dim=1))                                        # S1:505

    def forward_lights(self, img, epoch, intrinsic_matrix, mask, target_lightings):
        """`forward` for L target lights per face at once: target_lightings (L,3) (shared by the batch) or (B,L,3).  The same
        10-tuple with a light axis behind the batch axis on every per-light entry: shadow_mask_weights / ambient_light /
        full_shading / final_shading (B,L,H,W), rendered_images (B,L,3,H,W), unit_light_direction (B,L,3,1,1),
        ambient_values (B,L,1,1); albedo, depth and the normals are per face.  Entry [:, l] equals `forward` with light l
        (bit for bit given the same network outputs: tests/test_gpu_relight_lights.py)."""
        amb_of = lambda SL, B, L: (SL[:, 0, 0, 0] + self.ambient_offset)[:, None].expand(B, L)       # S1:342, every light
        albedo, depth, SL, _, r = self.render_lights(img, epoch, intrinsic_matrix, mask, target_lightings, amb_of)
        return (albedo, depth, r["shadow_mask_weights"], r["ambient_light"], r["full_shading"],
                r["rendered_images"], r["unit_light_direction"], r["ambient_values"], r["final_shading"],
                F.normalize(r["surface_normals"], p=2, dim=1))


class RelightNetLightingTransfer(_Hourglass):
    """Lighting-transfer inference form (test_relight_single_image_lighting_transfer.py)."""

    def __init__(self, shortcut: str = "1x1", img_height: int = 256, img_width: int = 256,
                 normal_z_offset: float = 1410.0, estimate_z_min: float = 0.16):
        super().__init__(shortcut)
        self.render_params = RenderParams.lighting_transfer(img_height, img_width)
        self.normal_z_offset = normal_z_offset                                          # SLT:325
        self.estimate_z_min = estimate_z_min                                            # SLT:332

    def forward(self, img, epoch, intrinsic_matrix, mask, target_lighting, target_ambient_values):
        albedo, depth, SL = self.features(img, epoch)
        B, _, H, W = depth.shape
        est = SL[:, 0, 0, 1:4]
        est = torch.stack([est[:, 0], est[:, 1], torch.clamp_min(est[:, 2], self.estimate_z_min)], 1)
        est_unit = F.normalize(est, p=2, dim=1).reshape(B, 3, 1, 1)                     # SLT:329-335
        r = render_from_depth(depth, albedo, target_lighting.reshape(B, 3), target_ambient_values.reshape(B),
                              self._camera(intrinsic_matrix), self.normal_z_offset, mask.reshape(1, H, W), self.render_params)
        normals = r["surface_normals"]
        return (albedo, depth, r["shadow_mask_weights"], r["ambient_light"], r["full_shading"],
                r["rendered_images"], r["unit_light_direction"], r["ambient_values"], r["final_shading"],
                F.normalize(normals, p=2, dim=1), est_unit, SL[:, :, :, 0])              # SLT:514

    def forward_lights(self, img, epoch, intrinsic_matrix, mask, target_lightings, target_ambient_values):
        """`forward` for L target lights per face at once (see RelightNetSingleImage.forward_lights): target_lightings (L,3) or
        (B,L,3); target_ambient_values a scalar, (L,) or (B,L).  The same 12-tuple with a light axis on the per-light entries."""
        def amb_of(SL, B, L):
            a = torch.as_tensor(target_ambient_values, dtype=torch.float32, device=SL.device)
            return (a.reshape(1, -1) if a.numel() in (1, L) else a.reshape(B, L)).expand(B, L)
        albedo, depth, SL, _, r = self.render_lights(img, epoch, intrinsic_matrix, mask, target_lightings, amb_of)
        B = depth.shape[0]
        est = SL[:, 0, 0, 1:4]
        est = torch.stack([est[:, 0], est[:, 1], torch.clamp_min(est[:, 2], self.estimate_z_min)], 1)
        est_unit = F.normalize(est, p=2, dim=1).reshape(B, 3, 1, 1)                     # SLT:329-335
        return (albedo, depth, r["shadow_mask_weights"], r["ambient_light"], r["full_shading"],
                r["rendered_images"], r["unit_light_direction"], r["ambient_values"], r["final_shading"],
                F.normalize(r["surface_normals"], p=2, dim=1), est_unit, SL[:, :, :, 0])


class PatchGAN(nn.Module):
    """70x70-style discriminator of the training script (T8:15-35); stock convolutions, (B,1,15,15) out."""

    def __init__(self):
        super().__init__()
        chans = [3, 64, 128, 256, 512]
        for i in range(4):
            setattr(self, "conv%d" % (i + 1), nn.Conv2d(chans[i], chans[i + 1], 4, stride=2, padding=1))
            if i > 0:
                setattr(self, "bn%d" % (i + 1), nn.BatchNorm2d(chans[i + 1]))
        self.conv5 = nn.Conv2d(512, 1, 4, stride=1, padding=1)

    def forward(self, img):
        x = _lrelu(self.conv1(img))
        for i in (2, 3, 4):
            x = _lrelu(getattr(self, "bn%d" % i)(getattr(self, "conv%d" % i)(x)))
        return self.conv5(x)
