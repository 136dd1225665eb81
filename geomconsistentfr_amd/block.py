"""Host-side entry points of the render block (inner boundary of SURVEY.md 8b).

`render()` is what a RelightNet.forward calls at the T8:352 seam.  Everything numeric happens in the
HIP kernels of libgcfr_hip.so; torch is used for device memory, streams and (later) autograd glue.
All tensors must live on a ROCm device; there is no CPU path.
"""
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib


@dataclass(frozen=True)
class RenderParams:
    """Constants of the block across the reference's five scripts.
    Defaults are the training script's (T8:41-48)."""
    n_samples: int = 160                  # T8:48   (SLT:22 -> 159)
    t0: float = 0.025                     # T8:468  (SLT:451 -> 0.03)
    dt: float = 0.005
    light_distance: float = 4013.0        # T8:47
    directional_intensity: float = 0.5    # T8:46   (SLT:20 -> 0.41)
    clamp_light_z_min: Optional[float] = 0.0   # T8:358; None = target light given, no clamp (S1:332)
    inside_bonus: float = 0.0             # S1:495-496 / SLT:503-504 -> 5.0
    bonus_box: Optional[Tuple[float, float, float, float]] = None   # (x_lo, x_hi, y_lo, y_hi)
    # Which pixels are marched.  "all": every pixel, as the reference (T8:371-515).  "mask" (opt-in, DEVIATES from the
    # reference's returned tensors): pixels whose own mask cell is zero are not marched and carry the masked value
    # (minimum distance 1e6 -> shadow weight 1, final = full shading); every consumer of the training script multiplies
    # them by the mask (T8:619, 633, 641, 643), so the losses are bit-equal -- see gcfr_options.pixels in include/gcfr.h.
    pixels: str = "all"

    @staticmethod
    def training() -> "RenderParams":
        return RenderParams()

    @staticmethod
    def single_image(H: int = 256, W: int = 256) -> "RenderParams":
        """test_relight_single_image.py / test_raytracing_relighting_..._8x.py form (S1:495)."""
        return RenderParams(clamp_light_z_min=None, inside_bonus=5.0,
                            bonus_box=(-(W / 2.0), W - W / 2.0 - 1, 1 - H / 2.0, H / 2.0))

    @staticmethod
    def lighting_transfer(H: int = 256, W: int = 256) -> "RenderParams":
        """test_relight_single_image_lighting_transfer.py form (SLT:20-22, 451, 503)."""
        return RenderParams(n_samples=159, t0=0.03, directional_intensity=0.41, clamp_light_z_min=None,
                            inside_bonus=5.0, bonus_box=(-4.0 * W, 4.0 * W, 4.0 * (1 - H), 4.0 * H))


_TABLE_CACHE = {}


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.GcfrError("geomconsistentfr_amd has no CPU path: tensors must be on a ROCm device")


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def sample_table(params: RenderParams, device) -> torch.Tensor:
    """Device copy of the f64 sample-fraction table (np.arange value rule, T8:468)."""
    key = (params.t0, params.dt, params.n_samples, str(device))
    t = _TABLE_CACHE.get(key)
    if t is None:
        host = np.empty(params.n_samples, dtype=np.float64)
        _lib.check(_lib.load().gcfr_sample_table(params.t0, params.dt, params.n_samples, host.ctypes.data),
                   "gcfr_sample_table")
        t = torch.from_numpy(host).to(device)
        _TABLE_CACHE[key] = t
    return t


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def light_prep(light: torch.Tensor, params: RenderParams = RenderParams()):
    """(..., 3) raw / target light -> (unit direction, light point), same leading shape.  T8:357-363."""
    _require_device(light)
    L = _lib.load()
    lr = _f32c(light).reshape(-1, 3)
    unit = torch.empty_like(lr)
    pt = torch.empty_like(lr)
    clamp = params.clamp_light_z_min is not None
    with torch.cuda.device(lr.device):
        _lib.check(L.gcfr_light_prep(lr.data_ptr(), lr.shape[0], int(clamp),
                                     float(params.clamp_light_z_min or 0.0), float(params.light_distance),
                                     unit.data_ptr(), pt.data_ptr(), _stream_ptr(lr.device)), "gcfr_light_prep")
    return unit.reshape(light.shape), pt.reshape(light.shape)


def mask_to_u8(mask: torch.Tensor) -> torch.Tensor:
    """The reference tests `mask == 0` on a float mask (T8:510); the kernels take u8 {0,1}."""
    if mask.dtype == torch.uint8:
        return mask.contiguous()
    return (mask != 0).to(torch.uint8).contiguous()


def shadow_min_distance(depth: torch.Tensor, mask: torch.Tensor, light_pt: torch.Tensor,
                        params: RenderParams = RenderParams(), want_argmin: bool = True,
                        use_workspace: bool = True, options=None):
    """depth (B,H,W) f32, mask (B|1,H,W), light_pt (B,L,3) -> min_dist (B,L,H,W) f32, argmin i32|None.
    Replaces T8:371-515.  use_workspace=False selects the direct-gather kernel (same bits, slower).
    `options`: a `_lib.Options` (kernel / schedule selection and hooks; none changes a result bit EXCEPT `pixels`, see
    include/gcfr.h).  `params.pixels = "mask"` reaches the library as `options.pixels = 1`, lives in the training (argmin) march
    and therefore FORCES want_argmin: an argmin tensor is returned (and the five-wave argmin kernel runs) even if the caller
    passed want_argmin=False."""
    _require_device(depth, mask, light_pt)
    if params.pixels != "all":
        if not use_workspace:
            raise _lib.GcfrError("RenderParams.pixels = %r needs the workspace path (use_workspace=True)" % (params.pixels,))
        want_argmin, options = _pixels_options(params, want_argmin, options)
    L_ = _lib.load()
    depth = _f32c(depth)
    B, H, W = depth.shape
    mask_u8 = mask_to_u8(mask).reshape(-1, H, W)
    light_pt = _f32c(light_pt).reshape(B, -1, 3)
    L = light_pt.shape[1]
    tt = sample_table(params, depth.device)
    md = torch.empty((B, L, H, W), dtype=torch.float32, device=depth.device)
    am = torch.empty((B, L, H, W), dtype=torch.int32, device=depth.device) if want_argmin else None
    box = (ctypes_float4(params.bonus_box) if params.bonus_box is not None else None)
    ws_bytes = int(L_.gcfr_shadow_workspace_bytes(B, H, W)) if use_workspace else 0
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=depth.device) if ws_bytes else None
    with torch.cuda.device(depth.device):
        _lib.check(L_.gcfr_shadow_fwd(depth.data_ptr(), mask_u8.data_ptr(), mask_u8.shape[0], light_pt.data_ptr(),
                                      B, L, H, W, params.n_samples, tt.data_ptr(), float(params.inside_bonus),
                                      box, md.data_ptr(), am.data_ptr() if am is not None else None,
                                      ws.data_ptr() if ws is not None else None, ws_bytes,
                                      _stream_ptr(depth.device), _lib.opt_ref(options)), "gcfr_shadow_fwd")
    return md, am


def ctypes_float4(v):
    import ctypes
    arr = (ctypes.c_float * 4)(*[float(x) for x in v])
    return ctypes.cast(arr, ctypes.c_void_p)


def shade(normals, depth, albedo, light_pt, ambient, min_dist, params: RenderParams = RenderParams()):
    """Replaces T8:364-369, 517-522.  Shapes as include/gcfr.h; returns dict of f32 tensors."""
    _require_device(normals, depth, albedo, light_pt, ambient, min_dist)
    L_ = _lib.load()
    depth = _f32c(depth)
    B, H, W = depth.shape
    normals = _f32c(normals).reshape(B, 3, H, W)
    albedo = _f32c(albedo).reshape(B, 3, H, W)
    light_pt = _f32c(light_pt).reshape(B, -1, 3)
    L = light_pt.shape[1]
    ambient = _f32c(ambient).reshape(B, L)
    min_dist = _f32c(min_dist).reshape(B, L, H, W)
    dev = depth.device
    w = torch.empty((B, L, H, W), dtype=torch.float32, device=dev)
    full = torch.empty_like(w)
    fin = torch.empty_like(w)
    ren = torch.empty((B, L, 3, H, W), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        _lib.check(L_.gcfr_shade_fwd(normals.data_ptr(), depth.data_ptr(), albedo.data_ptr(), light_pt.data_ptr(),
                                     ambient.data_ptr(), min_dist.data_ptr(), B, L, H, W,
                                     float(params.directional_intensity), w.data_ptr(), full.data_ptr(),
                                     fin.data_ptr(), ren.data_ptr(), _stream_ptr(dev)), "gcfr_shade_fwd")
    return dict(shadow_mask_weights=w, full_shading=full, final_shading=fin, rendered_images=ren)


# Lights per face from which the normals stage runs as its own launch (gcfr_normals_fwd, then gcfr_render_fwd reads its
# output) instead of inside the march's epilogue (gcfr_render_from_depth_fwd): the stencil does not depend on the light, the
# fused epilogue evaluates it once per (pixel, light).  Measured, own launch against fused (512 x 512 x 320, interleaved,
# profiles/r05_normals_stage_crossover.txt): four batches in flight +0.5 % at 4 lights per face, +1.3 % at 8, +2.1 % at 18
# (BASELINE configs[4]); one batch at a time -1.9 % at 4, -0.8 % at 8, +0.6 % at 18; one light per face: fused wins both ways
# (profiles/r04_normals_stage_ab.txt).  16: where both rates gain.  Bit-identical either way (tests/test_gpu_normals.py).
NORMALS_KERNEL_MIN_LIGHTS = 16


def normals_stage_for(n_lights: int, normals_stage: str = "auto") -> str:
    """'fused' | 'kernel' for a render of `n_lights` lights per face ('auto': by NORMALS_KERNEL_MIN_LIGHTS)"""
    if normals_stage == "auto":
        return "kernel" if n_lights >= NORMALS_KERNEL_MIN_LIGHTS else "fused"
    if normals_stage not in ("fused", "kernel"):
        raise _lib.GcfrError("normals_stage must be 'auto', 'fused' or 'kernel'")
    return normals_stage


_SIDE_STREAMS = {}


def side_stream(device) -> torch.cuda.Stream:
    """the device's stream for hoisted prepasses (one per device, created on first use)"""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    s = _SIDE_STREAMS.get(key)
    if s is None:
        s = _SIDE_STREAMS[key] = torch.cuda.Stream(device=torch.device("cuda", key))
    return s


class Prepared:
    """What `render_prepass()` leaves behind for the `render_fwd(..., prepared=...)` / `render_from_depth(..., prepared=...)`
    that follows: the workspace the prepass filled, the light outputs it wrote, the event that marks its end on the side
    stream, and the arguments it was issued with (the march must be issued with the same ones)."""
    __slots__ = ("ws", "ws_bytes", "unit", "pt", "event", "key", "depth", "mask_u8", "light", "src")


def source_signature(*tensors):
    """What identifies the CALLER's depth / mask / light between a prepass and its march: storage address, in-place version
    counter (shared by every view of a tensor, bumped by every in-place write), shape, strides and dtype of the tensors AS THE
    CALLER HANDED THEM OVER -- before any f32 / u8 / contiguous conversion, whose copies have fresh addresses every time."""
    return tuple((t.data_ptr(), t._version, tuple(t.shape), tuple(t.stride()), t.dtype) for t in tensors)


def _prepass_key(shapes, params, options):
    import ctypes
    knobs = None
    if options is not None:
        o = _lib.with_phase(options, 0)
        o.event_start = o.event_stop = None
        knobs = bytes(ctypes.string_at(ctypes.byref(o), ctypes.sizeof(o)))
    return (shapes, params, knobs)


def render_prepass(depth, mask, light, params: RenderParams = RenderParams(), want_argmin: bool = True, options=None,
                   stream: Optional[torch.cuda.Stream] = None, src=None) -> Prepared:
    """The FIRST of the forward's two launches on its own (gcfr_options.phase = 1): depth repack, mask statistics, depth
    bounds, horizon tables, light preparation, sample-table check -- everything that depends on depth (B,H,W), mask (B|1,H,W)
    and light (B,L,3) only.  Enqueued on `stream` (default: the device's side stream) behind whatever the CURRENT stream
    has enqueued so far, so that it runs under the work the caller enqueues next on the current stream (RelightNet.forward:
    the albedo decoder's convolutions, T8:226-290).  Pass the result as `prepared=` to render_fwd / render_from_depth with the
    same depth, mask, light, params, want_argmin and options: that call waits for the prepass and enqueues the march alone.
    Bit-identical to the one-call form (the same two launches with the same arguments).
    `src`: the `source_signature()` of the caller's own tensors where a wrapper reshaped them before this call (default: of
    depth, mask, light as given); the march call must present the same signature -- a different tensor, or the same one written
    in place in between, is an error there, not a render of stale data."""
    _require_device(depth, mask, light)
    want_argmin, options = _pixels_options(params, want_argmin, options)
    L_ = _lib.load()
    if src is None:
        src = source_signature(depth, mask, light)
    depth = _f32c(depth)
    B, H, W = depth.shape
    dev = depth.device
    mask_u8 = mask_to_u8(mask).reshape(-1, H, W)
    light = _f32c(light).reshape(B, -1, 3)
    L = light.shape[1]
    tt = sample_table(params, dev)
    p = Prepared()
    p.depth, p.mask_u8, p.light = depth, mask_u8, light          # (kept alive until the march has been enqueued)
    p.unit = torch.empty((B, L, 3), dtype=torch.float32, device=dev)
    p.pt = torch.empty((B, L, 3), dtype=torch.float32, device=dev)
    p.ws_bytes = int(L_.gcfr_shadow_workspace_bytes(B, H, W))
    p.ws = torch.empty(p.ws_bytes, dtype=torch.uint8, device=dev)
    p.src = src
    p.key = _prepass_key((tuple(depth.shape), tuple(mask_u8.shape), tuple(light.shape)), params, options)
    box = ctypes_float4(params.bonus_box) if params.bonus_box is not None else None
    clamp = params.clamp_light_z_min is not None
    side = stream if stream is not None else side_stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))              # depth / light are produced on the current stream
    for t in (depth, mask_u8, light, p.unit, p.pt, p.ws):         # allocated on the current stream, used on `side`: the
        t.record_stream(side)                                     # caching allocator must not recycle them under the prepass
    opt1 = _lib.with_phase(options, 1)
    with torch.cuda.device(dev), torch.cuda.stream(side):
        _lib.check(L_.gcfr_render_fwd(
            light.data_ptr(), int(clamp), float(params.clamp_light_z_min or 0.0), float(params.light_distance),
            depth.data_ptr(), mask_u8.data_ptr(), mask_u8.shape[0], None, None, None, B, L, H, W, params.n_samples,
            tt.data_ptr(), float(params.inside_bonus), box, float(params.directional_intensity), p.unit.data_ptr(),
            p.pt.data_ptr(), None, None, None, None, None, None, p.ws.data_ptr(), p.ws_bytes, side.cuda_stream,
            _lib.opt_ref(opt1)), "gcfr_render_fwd (prepass)")
        p.event = torch.cuda.Event()
        p.event.record(side)
    return p


def render_fwd(depth, mask, light, ambient, normals, albedo, params: RenderParams = RenderParams(),
               want_argmin: bool = True, camera=None, options=None, prepared: Optional[Prepared] = None, src=None):
    """One enqueue for the whole forward block (gcfr_render_fwd): light prep, depth repack, ray march with
    the shading fused into its epilogue.  depth (B,H,W), mask (B|1,H,W), light (B,L,3) raw/target,
    ambient (B,L), normals/albedo (B,3,H,W).  Returns a dict of f32 tensors (B,L,...).
    normals=None with camera=(fx, fy, cx, cy, z_offset): the normals stage (T8:353-354) is fused into the
    epilogue as well (gcfr_render_from_depth_fwd); the dict then also carries "surface_normals".
    `options`: a `_lib.Options`; only `pixels` changes a result bit, and `params.pixels = "mask"` forces the argmin plane (see
    shadow_min_distance).  `prepared`: the result of `render_prepass()` on the same depth / mask / light / params / options --
    the prepass has been enqueued already (side stream), this call waits for it and enqueues the march only.  The caller's
    depth / mask / light are compared with what the prepass was given (`source_signature`: address, in-place version, shape,
    strides, dtype; `src` = the signature of the caller's own tensors where a wrapper reshaped them): a mismatch raises."""
    _require_device(depth, mask, light, ambient, albedo)
    if normals is None and camera is None:
        raise _lib.GcfrError("render_fwd needs either normals or camera=(fx, fy, cx, cy, z_offset)")
    want_argmin, options = _pixels_options(params, want_argmin, options)
    L_ = _lib.load()
    if prepared is not None:
        # the march reads the very tensors the prepass read (its f32 / u8 copies where the caller's were converted): what the
        # caller hands over NOW must be what it handed to the prepass, unmodified
        if (src if src is not None else source_signature(depth, mask, light)) != prepared.src:
            raise _lib.GcfrError("render_fwd(prepared=...): depth / mask / light are not the tensors the prepass was issued with "
                                 "(a different tensor, another view, or written in place since)")
        depth, mask_u8, light = prepared.depth, prepared.mask_u8, prepared.light
        B, H, W = depth.shape
        dev = depth.device
        L = light.shape[1]
        if prepared.key != _prepass_key((tuple(depth.shape), tuple(mask_u8.shape), tuple(light.shape)), params, options):
            raise _lib.GcfrError("render_fwd(prepared=...): params / options differ from the prepass call's")
    else:
        depth = _f32c(depth)
        B, H, W = depth.shape
        dev = depth.device
        mask_u8 = mask_to_u8(mask).reshape(-1, H, W)
        light = _f32c(light).reshape(B, -1, 3)
        L = light.shape[1]
    ambient = _f32c(ambient).reshape(B, L)
    if normals is not None:
        _require_device(normals)
        normals = _f32c(normals).reshape(B, 3, H, W)
    albedo = _f32c(albedo).reshape(B, 3, H, W)
    tt = sample_table(params, dev)
    f32 = dict(dtype=torch.float32, device=dev)
    md = torch.empty((B, L, H, W), **f32)
    am = torch.empty((B, L, H, W), dtype=torch.int32, device=dev) if want_argmin else None
    w = torch.empty((B, L, H, W), **f32)
    full = torch.empty((B, L, H, W), **f32)
    fin = torch.empty((B, L, H, W), **f32)
    ren = torch.empty((B, L, 3, H, W), **f32)
    if prepared is None:
        unit = torch.empty((B, L, 3), **f32)
        pt = torch.empty((B, L, 3), **f32)
        ws_bytes = int(L_.gcfr_shadow_workspace_bytes(B, H, W))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    else:
        unit, pt, ws, ws_bytes = prepared.unit, prepared.pt, prepared.ws, prepared.ws_bytes
        torch.cuda.current_stream(dev).wait_event(prepared.event)     # the march starts behind the prepass
        options = _lib.with_phase(options, 2)
    box = ctypes_float4(params.bonus_box) if params.bonus_box is not None else None
    clamp = params.clamp_light_z_min is not None
    out = dict(unit_light_direction=unit, light_pt=pt, minimum_distance=md, argmin=am, shadow_mask_weights=w,
               full_shading=full, final_shading=fin, rendered_images=ren)
    with torch.cuda.device(dev):
        if normals is not None:
            _lib.check(L_.gcfr_render_fwd(
                light.data_ptr(), int(clamp), float(params.clamp_light_z_min or 0.0), float(params.light_distance),
                depth.data_ptr(), mask_u8.data_ptr(), mask_u8.shape[0], normals.data_ptr(), albedo.data_ptr(),
                ambient.data_ptr(), B, L, H, W, params.n_samples, tt.data_ptr(), float(params.inside_bonus), box,
                float(params.directional_intensity), unit.data_ptr(), pt.data_ptr(), md.data_ptr(), _opt_ptr(am),
                w.data_ptr(), full.data_ptr(), fin.data_ptr(), ren.data_ptr(), ws.data_ptr(), ws_bytes,
                _stream_ptr(dev), _lib.opt_ref(options)), "gcfr_render_fwd")
        elif normals_stage_for(L) == "kernel":
            # many lights per face: the light-independent stencil once, in its own launch, instead of once per light in the
            # march's epilogue (the same bits; NORMALS_KERNEL_MIN_LIGHTS)
            fx, fy, cx, cy, z_off = [float(v) for v in camera]
            nout = torch.empty((B, 3, H, W), **f32)
            _lib.check(L_.gcfr_normals_fwd(depth.data_ptr(), B, H, W, fx, fy, cx, cy, z_off, 1, nout.data_ptr(), _stream_ptr(dev)),
                       "gcfr_normals_fwd")
            _lib.check(L_.gcfr_render_fwd(
                light.data_ptr(), int(clamp), float(params.clamp_light_z_min or 0.0), float(params.light_distance),
                depth.data_ptr(), mask_u8.data_ptr(), mask_u8.shape[0], nout.data_ptr(), albedo.data_ptr(),
                ambient.data_ptr(), B, L, H, W, params.n_samples, tt.data_ptr(), float(params.inside_bonus), box,
                float(params.directional_intensity), unit.data_ptr(), pt.data_ptr(), md.data_ptr(), _opt_ptr(am),
                w.data_ptr(), full.data_ptr(), fin.data_ptr(), ren.data_ptr(), ws.data_ptr(), ws_bytes,
                _stream_ptr(dev), _lib.opt_ref(options)), "gcfr_render_fwd")
            out["surface_normals"] = nout
        else:
            fx, fy, cx, cy, z_off = [float(v) for v in camera]
            nout = torch.empty((B, 3, H, W), **f32)
            _lib.check(L_.gcfr_render_from_depth_fwd(
                light.data_ptr(), int(clamp), float(params.clamp_light_z_min or 0.0), float(params.light_distance),
                depth.data_ptr(), mask_u8.data_ptr(), mask_u8.shape[0], fx, fy, cx, cy, z_off, 1, albedo.data_ptr(),
                ambient.data_ptr(), B, L, H, W, params.n_samples, tt.data_ptr(), float(params.inside_bonus), box,
                float(params.directional_intensity), unit.data_ptr(), pt.data_ptr(), md.data_ptr(), _opt_ptr(am),
                nout.data_ptr(), w.data_ptr(), full.data_ptr(), fin.data_ptr(), ren.data_ptr(), ws.data_ptr(),
                ws_bytes, _stream_ptr(dev), _lib.opt_ref(options)), "gcfr_render_from_depth_fwd")
            out["surface_normals"] = nout
    return out


def _pixels_options(params: RenderParams, want_argmin: bool, options):
    """RenderParams.pixels -> gcfr_options.pixels.  "mask" lives in the library's training (argmin) march, so the argmin
    plane is requested whether or not the caller wants it."""
    if params.pixels == "all":
        return want_argmin, options
    if params.pixels != "mask":
        raise _lib.GcfrError("RenderParams.pixels must be 'all' or 'mask', got %r" % (params.pixels,))
    return True, _lib.with_pixels(options, 1)


def _zeros(shape, dtype, device):
    return torch.zeros(shape, dtype=dtype, device=device)


def _opt_ptr(t):
    return t.data_ptr() if t is not None else None


def _light_shapes(B, light, ambient):
    """The caller's light / ambient as (B,L,3) / (B,L) views and whether the call was the reference's one-light form
    (light (B,3), ambient (B,): outputs without the light axis) or the many-lights form (light (B,L,3), ambient (B,L))."""
    if light.dim() == 3 and light.shape[0] == B and light.shape[2] == 3:
        L = light.shape[1]
        if ambient.numel() != B * L:
            raise _lib.GcfrError("light is (B,L,3) = %s: ambient must hold B*L = %d values, got %s"
                                 % (tuple(light.shape), B * L, tuple(ambient.shape)))
        return light, ambient.reshape(B, L), L, True
    if light.numel() != B * 3:
        raise _lib.GcfrError("light must be (B,3) or (B,L,3); got %s for B = %d" % (tuple(light.shape), B))
    return light.reshape(B, 1, 3), ambient.reshape(B, 1), 1, False


def _result_dict(B, H, W, multi, ambient, w, full, fin, ren, unit, md, normals=None):
    """The reference's tensors by name (T8:524 / S1:505).  One-light form: the reference's shapes.  Many-lights form: a light
    axis after the batch axis (w / full / final / minimum_distance (B,L,H,W), rendered_images (B,L,3,H,W),
    unit_light_direction (B,L,3,1,1), ambient_values (B,L,1,1), ambient_light (B,L,H,W))."""
    if multi:
        L = w.shape[1]
        amb = ambient.to(torch.float32).reshape(B, L, 1, 1)
        r = dict(shadow_mask_weights=w, ambient_light=amb.expand(B, L, H, W), full_shading=full, rendered_images=ren,
                 unit_light_direction=unit.reshape(B, L, 3, 1, 1), ambient_values=amb, final_shading=fin, minimum_distance=md)
    else:
        amb = ambient.to(torch.float32).reshape(B, 1, 1)
        r = dict(shadow_mask_weights=w[:, 0], ambient_light=amb.expand(B, H, W), full_shading=full[:, 0],
                 rendered_images=ren[:, 0], unit_light_direction=unit.reshape(B, 3, 1, 1), ambient_values=amb,
                 final_shading=fin[:, 0], minimum_distance=md[:, 0])
    if normals is not None:
        r["surface_normals"] = normals
    return r


class _RenderFunction(torch.autograd.Function):
    """autograd glue around the HIP forward/backward kernels, L >= 1 lights per image.

    Differentiable inputs: depth, albedo, light (B,L,3), ambient (B,L), normals -- the leaves autograd reaches in the
    reference (T8:352-524).  The mask and the constants are not differentiable."""

    @staticmethod
    def forward(ctx, depth, albedo, light, ambient, normals, mask_u8, params):
        B, _, H, W = depth.shape
        L = light.shape[1]
        depth3 = _f32c(depth).reshape(B, H, W)
        light3 = _f32c(light).reshape(B, L, 3)
        amb = _f32c(ambient).reshape(B, L)
        albedo_c = _f32c(albedo)
        normals_c = _f32c(normals)
        need_grad = any(ctx.needs_input_grad[:5])
        out = render_fwd(depth3, mask_u8, light3, amb, normals_c, albedo_c, params, want_argmin=need_grad)
        unit, pt, md, am = out["unit_light_direction"], out["light_pt"], out["minimum_distance"], out["argmin"]
        ctx.params = params
        if need_grad:
            ctx.save_for_backward(depth3, albedo_c, light3, amb, normals_c, pt, md, am)
        ctx.mark_non_differentiable(md)
        return (out["shadow_mask_weights"], out["full_shading"], out["final_shading"], out["rendered_images"], unit, md)

    @staticmethod
    def backward(ctx, g_w, g_full, g_fin, g_ren, g_unit, _g_md):
        depth3, albedo, light3, amb, normals, pt, md, am = ctx.saved_tensors
        prm = ctx.params
        L_ = _lib.load()
        B, H, W = depth3.shape
        L = light3.shape[1]
        dev = depth3.device
        gw, gfull, gfin, gren = [None if g is None else _f32c(g) for g in (g_w, g_full, g_fin, g_ren)]
        grad_normals = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        grad_albedo = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        grad_depth = _zeros((B, H, W), torch.float32, dev)
        grad_pt = _zeros((B, L, 3), torch.float64, dev)
        grad_amb = _zeros((B, L), torch.float64, dev)
        grad_md = torch.empty((B, L, H, W), dtype=torch.float32, device=dev)
        tt = sample_table(prm, dev)
        with torch.cuda.device(dev):
            st = _stream_ptr(dev)
            _lib.check(L_.gcfr_shade_bwd(normals.data_ptr(), depth3.data_ptr(), albedo.data_ptr(), pt.data_ptr(),
                                         amb.data_ptr(), md.data_ptr(), B, L, H, W, float(prm.directional_intensity),
                                         _opt_ptr(gw), _opt_ptr(gfull), _opt_ptr(gfin), _opt_ptr(gren),
                                         grad_normals.data_ptr(), grad_albedo.data_ptr(), grad_depth.data_ptr(),
                                         grad_pt.data_ptr(), grad_amb.data_ptr(), grad_md.data_ptr(), st),
                       "gcfr_shade_bwd")
            _lib.check(L_.gcfr_shadow_bwd(grad_md.data_ptr(), depth3.data_ptr(), pt.data_ptr(), am.data_ptr(),
                                          B, L, H, W, prm.n_samples, tt.data_ptr(), grad_depth.data_ptr(),
                                          grad_pt.data_ptr(), st), "gcfr_shadow_bwd")
            grad_light = torch.empty((B, L, 3), dtype=torch.float32, device=dev)
            gu = None if g_unit is None else _f32c(g_unit).reshape(B * L, 3)
            clamp = prm.clamp_light_z_min is not None
            _lib.check(L_.gcfr_light_prep_bwd(light3.data_ptr(), B * L, int(clamp), float(prm.clamp_light_z_min or 0.0),
                                              float(prm.light_distance), _opt_ptr(gu), grad_pt.data_ptr(),
                                              grad_light.data_ptr(), st), "gcfr_light_prep_bwd")
        return (grad_depth.reshape(B, 1, H, W), grad_albedo, grad_light, grad_amb.float(), grad_normals, None, None)


def render(depth, albedo, light, ambient, normals, mask, params: RenderParams = RenderParams()):
    """Render block for a batch, differentiable; one light per image (the reference's call shape) or many.

      depth (B,1,H,W) f32      c2_o_depth (x100 already applied, T8:350)
      albedo (B,3,H,W) f32     c2_o_albedo
      light (B,3)              SL_lin2[...,1:4] (T8:357) or target_lighting (S1:332)   | (B,L,3): L lights per face
      ambient (B,)             SL_lin2[...,0] (T8:367) / target ambient                | (B,L)
      normals (B,3,H,W)        depth_to_normals(depth+offset, K), y negated (T8:353-354)
      mask (B,H,W) or (1,H,W)  0 = outside the face (T8:510)

    Returns the reference's tensors by name (T8:524 / S1:505), all f32; with (B,L,3) lights every per-light tensor carries a
    light axis after the batch axis (`_result_dict`).  Gradients flow to depth, albedo, light, ambient and normals through
    the HIP backward kernels (for L lights: summed over the lights where the input has no light axis)."""
    _require_device(depth, albedo, light, ambient, normals, mask)
    B, _, H, W = depth.shape
    mask_u8 = mask_to_u8(mask).reshape(-1, H, W)
    light3, amb2, L, multi = _light_shapes(B, light, ambient)
    w, full, fin, ren, unit, md = _RenderFunction.apply(depth, albedo, light3, amb2, normals, mask_u8, params)
    return _result_dict(B, H, W, multi, amb2, w, full, fin, ren, unit, md)


class RenderFwdPlan:
    """Steady-state forward for fixed shapes: outputs and workspace are allocated once and every call is ONE
    ctypes call on the current stream (no per-call allocation or argument marshalling: the eager `render_fwd`
    spends ~60 us of host time per call, which bounds a B=8 step once two streams keep the GPU full).
    Results are overwritten by the next call on the same plan -- use one plan per stream / per batch in flight.
    Inputs must already be device tensors of the planned shapes and dtypes (f32, mask u8)."""

    def __init__(self, B, L, H, W, params: RenderParams = RenderParams(), device="cuda", want_argmin=False,
                 mask_batch=None, camera=None, options=None, normals_stage="auto"):
        self.L_ = _lib.load()
        want_argmin, options = _pixels_options(params, want_argmin, options)
        self.options = options          # _lib.Options or None; kept alive here, read by the library at every call
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.GcfrError("RenderFwdPlan needs a HIP device (there is no CPU path); got %s" % dev)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.dev, self.params, self.shape, self.camera = dev, params, (B, L, H, W), camera
        # "fused": the march epilogue evaluates the normals stencil (gcfr_render_from_depth_fwd, two launches);
        # "kernel": gcfr_normals_fwd first, then gcfr_render_fwd reads its output (three launches; the same bits);
        # "auto": by the number of lights per face (normals_stage_for)
        self.normals_stage = normals_stage_for(L, normals_stage)
        f32 = dict(dtype=torch.float32, device=dev)
        self.tt = sample_table(params, dev)
        o = dict(unit_light_direction=torch.empty((B, L, 3), **f32), light_pt=torch.empty((B, L, 3), **f32),
                 minimum_distance=torch.empty((B, L, H, W), **f32),
                 argmin=torch.empty((B, L, H, W), dtype=torch.int32, device=dev) if want_argmin else None,
                 shadow_mask_weights=torch.empty((B, L, H, W), **f32), full_shading=torch.empty((B, L, H, W), **f32),
                 final_shading=torch.empty((B, L, H, W), **f32), rendered_images=torch.empty((B, L, 3, H, W), **f32))
        if camera is not None:
            o["surface_normals"] = torch.empty((B, 3, H, W), **f32)
        self.out = o
        self.ws_bytes = int(self.L_.gcfr_shadow_workspace_bytes(B, H, W))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.box = ctypes_float4(params.bonus_box) if params.bonus_box is not None else None
        self.mask_batch = B if mask_batch is None else mask_batch
        self._tail = (B, L, H, W, params.n_samples, self.tt.data_ptr(), float(params.inside_bonus), self.box,
                      float(params.directional_intensity), o["unit_light_direction"].data_ptr(),
                      o["light_pt"].data_ptr(), o["minimum_distance"].data_ptr(), _opt_ptr(o["argmin"]))
        self._outs = (o["shadow_mask_weights"].data_ptr(), o["full_shading"].data_ptr(),
                      o["final_shading"].data_ptr(), o["rendered_images"].data_ptr(), self.ws.data_ptr(),
                      self.ws_bytes)
        self._head = (int(params.clamp_light_z_min is not None), float(params.clamp_light_z_min or 0.0),
                      float(params.light_distance))
        self._validated = None
        self.graph = None

    def _validate(self, depth, mask_u8, light, ambient, normals, albedo):
        B, L, H, W = self.shape
        for name, t, dt, n in (("depth", depth, torch.float32, B * H * W), ("mask_u8", mask_u8, torch.uint8, self.mask_batch * H * W),
                               ("light", light, torch.float32, B * L * 3), ("ambient", ambient, torch.float32, B * L),
                               ("albedo", albedo, torch.float32, B * 3 * H * W)) + \
                              ((("normals", normals, torch.float32, B * 3 * H * W),) if self.camera is None else ()):
            if t is None or t.dtype != dt or t.numel() != n or not t.is_contiguous() or t.device != self.dev:
                raise _lib.GcfrError("RenderFwdPlan: %s must be a contiguous %s tensor of %d elements on %s"
                                     % (name, dt, n, self.dev))

    def capture(self, depth, mask_u8, light, ambient, normals, albedo):
        """Capture one call on these (static) input tensors into a hipGraph; `replay()` then re-runs it on the
        current stream for ~10 us of host time instead of ~55 (two kernel launches, marshalling 31 arguments).
        The entry points neither allocate nor synchronise, so they are capture-safe.  New data goes into the same
        input tensors (copy_) before a replay."""
        self._static = (depth, mask_u8, light, ambient, normals, albedo)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):                      # warm-up outside the capture, as torch requires
            self(*self._static)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self(*self._static)
        return self

    def replay(self):
        self.graph.replay()
        return self.out

    def capture_split(self, depth, mask_u8, light, ambient, normals, albedo):
        """The two launches as two hipGraphs (gcfr_options.phase 1 / 2): `replay_prepass()` -- on whatever stream is current,
        typically a side stream, as soon as depth, mask and light are in the captured tensors -- and `replay_march()` behind it
        (the caller orders the two: same stream or an event).  The same bits as `replay()`."""
        self._static = (depth, mask_u8, light, ambient, normals, albedo)
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            self(*self._static)
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        self.graph_pre, self.graph_march = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_pre):
            self(*self._static, phase=1)
        with torch.cuda.graph(self.graph_march):
            self(*self._static, phase=2)
        return self

    def replay_prepass(self):
        self.graph_pre.replay()

    def replay_march(self):
        self.graph_march.replay()
        return self.out

    def __call__(self, depth, mask_u8, light, ambient, normals, albedo, phase: int = 0):
        """depth (B,H,W) f32, mask_u8 (B|1,H,W) u8, light (B,L,3) f32, ambient (B,L) f32, albedo (B,3,H,W) f32,
        normals (B,3,H,W) f32 or None (plan built with camera=...).  All contiguous, on the plan's device.
        phase: 0 both launches; 1 the prepass only; 2 the march only (gcfr_options.phase)."""
        # (the checks cost ~14 us of host time: once per set of buffers, not per call.  The key is what the kernels
        # actually consume -- address, shape, dtype, device -- so a new tensor that happens to reuse a Python id,
        # or a tensor whose storage was reassigned, is validated again.)
        key = tuple(None if t is None else (t.data_ptr(), tuple(t.shape), t.dtype, t.device, t.is_contiguous())
                    for t in (depth, mask_u8, light, ambient, normals, albedo))
        if key != self._validated:
            self._validate(depth, mask_u8, light, ambient, normals, albedo)
            self._validated = key
        st = torch.cuda.current_stream(self.dev).cuda_stream
        opts = self.options if phase == 0 else _lib.with_phase(self.options, phase)
        if self.camera is None:
            rc = self.L_.gcfr_render_fwd(light.data_ptr(), *self._head, depth.data_ptr(), mask_u8.data_ptr(),
                                         self.mask_batch, normals.data_ptr(), albedo.data_ptr(), ambient.data_ptr(),
                                         *self._tail, *self._outs, st, _lib.opt_ref(opts))
        elif self.normals_stage == "kernel":
            fx, fy, cx, cy, z_off = [float(v) for v in self.camera]
            nrm = self.out["surface_normals"]
            if phase != 1:
                _lib.check(self.L_.gcfr_normals_fwd(depth.data_ptr(), self.shape[0], self.shape[2], self.shape[3], fx, fy, cx, cy, z_off, 1,
                                                    nrm.data_ptr(), st), "gcfr_normals_fwd (plan)")
            rc = self.L_.gcfr_render_fwd(light.data_ptr(), *self._head, depth.data_ptr(), mask_u8.data_ptr(),
                                         self.mask_batch, nrm.data_ptr(), albedo.data_ptr(), ambient.data_ptr(),
                                         *self._tail, *self._outs, st, _lib.opt_ref(opts))
        else:
            fx, fy, cx, cy, z_off = [float(v) for v in self.camera]
            rc = self.L_.gcfr_render_from_depth_fwd(light.data_ptr(), *self._head, depth.data_ptr(),
                                                    mask_u8.data_ptr(), self.mask_batch, fx, fy, cx, cy, z_off, 1,
                                                    albedo.data_ptr(), ambient.data_ptr(), *self._tail,
                                                    self.out["surface_normals"].data_ptr(), *self._outs, st,
                                                    _lib.opt_ref(opts))
        _lib.check(rc, "gcfr_render_fwd (plan)")
        return self.out


class _RenderFromDepthFunction(torch.autograd.Function):
    """Whole T8:353-522 seam, L >= 1 lights per image, differentiable w.r.t. depth, albedo, light (B,L,3), ambient (B,L).
    Forward: gcfr_render_from_depth_fwd (two launches; from NORMALS_KERNEL_MIN_LIGHTS lights per face on the normals stage
    as its own launch in front of gcfr_render_fwd).  Backward: gcfr_render_bwd (one launch: shading, ray-march and
    normals-stencil backward fused per pixel, every light of a pixel in the same thread) + gcfr_light_prep_bwd."""

    @staticmethod
    def forward(ctx, depth, albedo, light, ambient, mask, cam, params, prepared=None, src=None):
        B, _, H, W = depth.shape
        L = light.shape[1]
        amb = _f32c(ambient).reshape(B, L)
        albedo_c = _f32c(albedo)
        need_grad = any(ctx.needs_input_grad[:4])
        if prepared is None:
            depth3 = _f32c(depth).reshape(B, H, W)
            light3 = _f32c(light).reshape(B, L, 3)
            o = render_fwd(depth3, mask, light3, amb, None, albedo_c, params, want_argmin=need_grad, camera=cam)
        else:       # the march reads the prepass's own copies; the caller's tensors are only compared with its record
            o = render_fwd(depth.reshape(B, H, W), mask, light, amb, None, albedo_c, params, want_argmin=need_grad, camera=cam,
                           prepared=prepared, src=src)
            depth3, light3 = prepared.depth, prepared.light     # (what the kernels read: saved for the backward)
        ctx.params, ctx.cam = params, cam
        if need_grad:
            ctx.save_for_backward(depth3, albedo_c, light3, amb, o["light_pt"], o["minimum_distance"], o["argmin"],
                                  o["surface_normals"])
        md = o["minimum_distance"]
        ctx.mark_non_differentiable(md)
        return (o["shadow_mask_weights"], o["full_shading"], o["final_shading"], o["rendered_images"],
                o["unit_light_direction"], o["surface_normals"], md)

    @staticmethod
    def backward(ctx, g_w, g_full, g_fin, g_ren, g_unit, g_nrm, _g_md):
        depth3, albedo, light3, amb, pt, md, am, nrm_fwd = ctx.saved_tensors
        prm, cam = ctx.params, ctx.cam
        L_ = _lib.load()
        B, H, W = depth3.shape
        L = light3.shape[1]
        dev = depth3.device
        gw, gfull, gfin, gren, gnrm = [None if g is None else _f32c(g) for g in (g_w, g_full, g_fin, g_ren, g_nrm)]
        grad_albedo = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev)
        grad_depth = _zeros((B, H, W), torch.float32, dev)
        grad_pt = _zeros((B, L, 3), torch.float64, dev)
        grad_amb = _zeros((B, L), torch.float64, dev)
        grad_light = torch.empty((B, L, 3), dtype=torch.float32, device=dev)
        tt = sample_table(prm, dev)
        fx, fy, cx, cy, z_off = cam
        with torch.cuda.device(dev):
            st = _stream_ptr(dev)
            _lib.check(L_.gcfr_render_bwd(depth3.data_ptr(), albedo.data_ptr(), pt.data_ptr(), amb.data_ptr(),
                                          md.data_ptr(), am.data_ptr(), nrm_fwd.data_ptr(), B, L, H, W, prm.n_samples,
                                          tt.data_ptr(),
                                          fx, fy, cx, cy, z_off, 1, float(prm.directional_intensity),
                                          _opt_ptr(gw), _opt_ptr(gfull), _opt_ptr(gfin), _opt_ptr(gren), _opt_ptr(gnrm),
                                          grad_albedo.data_ptr(), grad_depth.data_ptr(), grad_pt.data_ptr(),
                                          grad_amb.data_ptr(), st), "gcfr_render_bwd")
            gu = None if g_unit is None else _f32c(g_unit).reshape(B * L, 3)
            clamp = prm.clamp_light_z_min is not None
            _lib.check(L_.gcfr_light_prep_bwd(light3.data_ptr(), B * L, int(clamp), float(prm.clamp_light_z_min or 0.0),
                                              float(prm.light_distance), _opt_ptr(gu), grad_pt.data_ptr(),
                                              grad_light.data_ptr(), st), "gcfr_light_prep_bwd")
        return (grad_depth.reshape(B, 1, H, W), grad_albedo, grad_light, grad_amb.float(), None, None, None, None, None)


_CAMERA_CACHE = {}      # id(tensor) -> (weakref to the tensor, its in-place version, scalars)


def camera_scalars(camera_matrix: torch.Tensor):
    """(fx, fy, cx, cy) of a (1|B,3,3) camera matrix as host floats, or None if the B matrices differ.
    The reference builds K on the host and passes `intrinsic_matrix.cuda()` to every forward (T8:571-577, 618).
    A host tensor is simply read (10 us of tensor arithmetic; cached like a device tensor's so that a caller that keeps one K
    pays it once).  A DEVICE tensor has to be copied back -- a device-to-host sync -- so its scalars are cached per tensor OBJECT: the entry holds a weak reference and the tensor's in-place version counter, and is
    only trusted while that very object is alive and unmodified (an address- or id-keyed cache would hand a freed
    tensor's scalars to whatever is allocated in its place).  Callers that keep one K on the device (Trainer does)
    therefore synchronise once; callers that upload a fresh K every step should pass the host tensor instead."""
    import weakref

    def read(K):
        K = K.detach().to("cpu", torch.float64)
        same = K.shape[0] == 1 or bool((K == K[:1]).all())
        return (float(K[0, 0, 0]), float(K[0, 1, 1]), float(K[0, 0, 2]), float(K[0, 1, 2])) if same else None

    key = id(camera_matrix)
    ent = _CAMERA_CACHE.get(key)
    if ent is not None and ent[0]() is camera_matrix and ent[1] == camera_matrix._version:
        return ent[2]
    scalars = read(camera_matrix)                                   # the only sync: first use of this tensor object
    _CAMERA_CACHE[key] = (weakref.ref(camera_matrix, lambda _r, k=key: _CAMERA_CACHE.pop(k, None)),
                          camera_matrix._version, scalars)
    return scalars


def render_from_depth_prepass(depth, light, camera_matrix, mask, params: RenderParams = RenderParams()):
    """The prepass of the `render_from_depth()` call that is about to follow, enqueued NOW on the device's side stream (see
    `render_prepass`): call it as soon as depth (B,1,H,W), light (B,3) | (B,L,3) and the mask exist -- in RelightNet.forward
    that is before the albedo decoder runs -- and hand the result to `render_from_depth(..., prepared=...)` together with the
    SAME depth / light / mask tensors (checked there: `source_signature`).  Returns None where the one-call form has to be
    used (per-image camera matrices: the three-stage path).  (Whether the march will track the argmin -- autograd or not --
    does not concern the prepass.)"""
    B, _, H, W = depth.shape
    if camera_scalars(camera_matrix) is None:
        return None
    _require_device(depth, light, mask)
    light3 = light if (light.dim() == 3 and light.shape[0] == B) else light.reshape(B, 1, 3)
    return render_prepass(depth.reshape(B, H, W), mask.reshape(-1, H, W), light3, params,
                          src=source_signature(depth, mask, light))


def render_from_depth(depth, albedo, light, ambient, camera_matrix, z_offset, mask,
                      params: RenderParams = RenderParams(), prepared: Optional[Prepared] = None):
    """The whole T8:353-522 seam: normals from depth, shading, ray march, composite -- for one light per image (light (B,3),
    ambient (B,): the reference's call shape and output shapes) or MANY lights per face (light (B,L,3), ambient (B,L): one
    prepass and one normals stage per face, L marches; every per-light output carries a light axis after the batch axis,
    `_result_dict`).  Two launches forward (`gcfr_render_from_depth_fwd`: prepass, march with normals + shading in its
    epilogue; from NORMALS_KERNEL_MIN_LIGHTS lights per face the normals stage is its own launch) and, with autograd active,
    one fused backward launch (`gcfr_render_bwd`, all L lights of a pixel in one thread) plus the tiny light-prep backward.
    Same dict as `render()` plus "surface_normals" (unit, y negated; (B,3,H,W) whatever L).
    `prepared`: the result of `render_from_depth_prepass()` on the same depth / light / mask tensors and params: the prepass
    is already on its way on the side stream and this call enqueues the march only (bit-identical)."""
    from .normals import depth_to_normals
    B, _, H, W = depth.shape
    needs_grad = torch.is_grad_enabled() and any(t.requires_grad for t in (depth, albedo, light, ambient))
    k4 = camera_scalars(camera_matrix)
    if k4 is None:  # per-image camera matrices: the three-stage path handles them
        normals = depth_to_normals(depth, camera_matrix, z_offset=z_offset)
        r = render(depth, albedo, light, ambient, normals, mask, params)
        r["surface_normals"] = normals
        return r
    cam = k4 + (float(z_offset),)
    _require_device(depth, albedo, light, ambient, mask)
    light3, amb2, L, multi = _light_shapes(B, light, ambient)
    src = source_signature(depth, mask, light) if prepared is not None else None
    mask3 = mask.reshape(-1, H, W)                 # (converted to u8 inside render_fwd; with `prepared` the prepass's copy is used)
    if needs_grad:
        w, full, fin, ren, unit, nrm, md = _RenderFromDepthFunction.apply(
            depth, albedo, light3, amb2, mask3, cam, params, prepared, src)
        return _result_dict(B, H, W, multi, amb2, w, full, fin, ren, unit, md, normals=nrm)
    o = render_fwd(depth.reshape(B, H, W), mask3, light3, amb2, None, albedo, params, want_argmin=False, camera=cam,
                   prepared=prepared, src=src)
    return _result_dict(B, H, W, multi, amb2.detach(), o["shadow_mask_weights"], o["full_shading"], o["final_shading"],
                        o["rendered_images"], o["unit_light_direction"], o["minimum_distance"], normals=o["surface_normals"])
