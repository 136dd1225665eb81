#!/usr/bin/env python3
"""bench.py -- headline benchmark of the render block on MI355X.

Metric (BASELINE.json): ray-steps/s (whole job) + relit faces/s on 256x256 faces, 160 march steps.
Workload at N=1: BASELINE configs[1] -- batch of 8 synthetic 256x256 faces, one light each,
forward-only shadow + shade.  One "step" = one pass of the hot path over one batch:
one gcfr_render_fwd enqueue (depth repack + light prep, then the ray march with the shading fused into
its epilogue), inputs resident in HBM.  --from-depth also fuses the normals stencil (+3 %).
ray_steps = B*L*H*W*N nominal, never "steps executed".

Multi-GPU (`torchrun --nproc-per-node N bench.py --gpus N`): faces are independent, so each rank
renders its own batch of 8 with no data-path collective ("weak" scaling); the timed region is
bracketed by barrier + synchronize on both sides and the max over ranks is reported.

The JSON line also carries
  roofline     -- for the dominant kernel (shadow_fwd_kernel): algorithmic bytes (17.4 B per ray-step,
                  SURVEY.md 8d) per launch / average launch duration measured with HIP events on the
                  launch stream, against the 8 TB/s HBM peak; `traffic` = HBM bytes per launch from the
                  rocprofv3 PMC pass committed under profiles/ (null if that file is absent);
  cpu_baseline -- the oracle's materialised-torch port (same op sequence as the reference, which cannot
                  travel to the GPU box) timed on this host's cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 256
N_SAMPLES = 160
FACES_PER_GPU = 8
ALGO_BYTES_PER_RAY_STEP = 17.4      # 4 f32 depth corners + 1 u8 mask cell + 0.4 B amortised pixel I/O
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec


LIGHTS18 = np.array([[.7518, 0, .6594], [.6893, .3991, .6047], [.5145, 0, .8575], [-.5843, 0, .8115],
                     [-.7574, 0, .6529], [-.7076, .3892, .5897], [-.5151, .4722, .7154], [.4478, .4925, .7463],
                     [0, .7071, .7071], [-.8138, -.3420, .4698], [.8138, -.3420, .4698],      # 11 from S1:519-562
                     [.3, .3, .9], [-.3, .3, .9], [.2, -.5, .84], [-.2, -.5, .84], [.9, .1, .42], [-.9, .1, .42],
                     [0, .2, .98]], np.float32)                                                # 7 synthetic


def synth_faces_sized(B, seed0, size, n_lights, mask_kind="ellipse"):
    """Config-5 style inputs: `size` x `size` faces (surface scaled), `n_lights` lights per face."""
    r, c = np.mgrid[0:size, 0:size]
    s = size / 256.0
    x, y = (c - size / 2.0) / s, (r - size / 2.0) / s
    depth, mask, albedo, normals = [], [], [], []
    for i in range(B):
        rng = np.random.default_rng(seed0 + i)
        ax, ay, nose = 85 + 10 * rng.random(), 105 + 10 * rng.random(), 30 + 10 * rng.random()
        d = s * (80 * np.sqrt(np.maximum(1 - (x / ax) ** 2 - (y / ay) ** 2, 0))
                 + nose * np.exp(-(x ** 2 / 288 + (y - 12) ** 2 / 648)) + 3 * np.sin(x / 7) * np.cos(y / 9))
        depth.append(d.astype(np.float32))
        m = (((x / (ax - 8)) ** 2 + (y / (ay - 8)) ** 2) < 1) if mask_kind == "ellipse" else np.ones_like(x, bool)
        mask.append(m.astype(np.uint8))
        albedo.append((0.15 + 0.7 * rng.random((3, size, size))).astype(np.float32))
        gy, gx = np.gradient(d)
        n = np.stack([-gx, gy, np.ones_like(d)])
        normals.append((n / np.linalg.norm(n, axis=0)).astype(np.float32))
    light = np.stack([np.roll(LIGHTS18, seed0 + i, axis=0)[:n_lights] for i in range(B)])
    amb = np.full((B, n_lights), 0.5, np.float32)
    return np.stack(depth), np.stack(mask), np.stack(albedo), np.stack(normals), light, amb


def synth_faces(B, seed0):
    """Deterministic synthetic faces: jittered ellipsoid + nose + ripple."""
    r, c = np.mgrid[0:H, 0:W]
    x, y = c - 128.0, r - 128.0
    lights11 = np.array([[.7518, 0, .6594], [.6893, .3991, .6047], [.5145, 0, .8575], [-.5843, 0, .8115],
                         [-.7574, 0, .6529], [-.7076, .3892, .5897], [-.5151, .4722, .7154], [.4478, .4925, .7463],
                         [0, .7071, .7071], [-.8138, -.3420, .4698], [.8138, -.3420, .4698]], np.float32)
    depth, mask, albedo, normals, light, amb = [], [], [], [], [], []
    for i in range(B):
        rng = np.random.default_rng(seed0 + i)
        ax, ay, nose = 85 + 10 * rng.random(), 105 + 10 * rng.random(), 30 + 10 * rng.random()
        d = 80 * np.sqrt(np.maximum(1 - (x / ax) ** 2 - (y / ay) ** 2, 0)) \
            + nose * np.exp(-(x ** 2 / 288 + (y - 12) ** 2 / 648)) + 3 * np.sin(c / 7) * np.cos(r / 9)
        depth.append(d.astype(np.float32))
        mask.append((((x / (ax - 8)) ** 2 + (y / (ay - 8)) ** 2) < 1).astype(np.uint8))
        albedo.append((0.15 + 0.7 * rng.random((3, H, W))).astype(np.float32))
        gy, gx = np.gradient(d)
        n = np.stack([-gx, gy, np.ones_like(d)])
        normals.append((n / np.linalg.norm(n, axis=0)).astype(np.float32))
        light.append(lights11[(seed0 + i) % 11])
        amb.append(np.float32(0.5))
    return (np.stack(depth), np.stack(mask), np.stack(albedo), np.stack(normals), np.stack(light),
            np.asarray(amb, np.float32))


def cpu_baseline(seed0=0, runs=3, with_backward=True):
    """The CPU baseline of record: oracle/materialised.py -- the op-for-op torch-CPU port
    of T8:352-524, bit-equal to the imported reference (tests/test_oracle_vs_reference.py); the reference's own .py
    cannot travel to the GPU box -- in the reference's training form: a batch of B = 3 faces, normals from depth
    inside the timed region (T8:353), 256 x 256 x 160.  Forward under no_grad at 8, 32 and all host threads, median
    of `runs` after a warm-up each, the BEST thread count reported (all 256 hyper-threads of the GPU host are ~10x
    slower than 8-32 for these memory-bound elementwise ops); then forward+backward (autograd through the port, as
    loss.backward() replays the reference's graph, T8:655) at that thread count, median of `runs`.
    Checker code used strictly as the reported baseline; bounded: about 2-3 minutes of host time."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import materialised as M
    from normals_restatement import depth_to_normals
    B = 3
    cores = os.cpu_count() or 1
    depth, mask, albedo, _, light, amb = synth_faces(B, seed0)
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = 1570.0
    K[:, 2, 2] = 1.0
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    t_depth, t_alb, t_light, t_amb, t_mask = (torch.from_numpy(depth)[:, None], torch.from_numpy(albedo),
                                              torch.from_numpy(light), torch.from_numpy(amb), torch.from_numpy(mask))

    def block(d, al, li, am):
        n = depth_to_normals(d + 1610.0, K)                                  # T8:353 (kornia restatement, f64)
        n = torch.cat([n[:, 0:1], -n[:, 1:2], n[:, 2:3]], 1)                 # T8:354
        return M.render_block(d, al, li, am, n, t_mask)

    def forward():
        with torch.no_grad():
            block(t_depth, t_alb, t_light, t_amb)

    def forward_backward():
        leaves = [t.clone().requires_grad_() for t in (t_depth, t_alb, t_light, t_amb)]
        o = block(*leaves)
        (o["rendered_images"].sum() + o["shadow_mask_weights"].sum()).backward()

    def median_time(fn, n):
        ts = []
        for _ in range(n):
            t = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t)
        return float(np.median(ts)), ts

    steps = B * H * W * N_SAMPLES
    by_threads = {}
    for th in sorted({min(8, cores), min(32, cores), cores}):
        torch.set_num_threads(th)
        forward()                                                            # warm-up (thread pool, allocator)
        med, ts = median_time(forward, runs)
        by_threads[th] = {"median_s": med, "runs_s": ts, "ray_steps_per_sec": steps / med}
    best = max(by_threads, key=lambda th: by_threads[th]["ray_steps_per_sec"])
    out = {"value": by_threads[best]["ray_steps_per_sec"], "unit": "ray-steps/s", "cores": best, "kind": "port",
           "host_threads_available": cores, "faces_per_s": B / by_threads[best]["median_s"],
           "forward_by_threads": {str(k): v for k, v in by_threads.items()},
           "sample": "T8 form: batch of 3 faces 256x256x160 incl. normals from depth, forward no_grad, "
                     "oracle/materialised.py (op-for-op torch-CPU port of T8:352-524), median of %d runs after a "
                     "warm-up at each of %s threads; best = %d threads, %.2f s per batch"
                     % (runs, sorted(by_threads), best, by_threads[best]["median_s"])}
    if with_backward:
        torch.set_num_threads(best)
        med, ts = median_time(forward_backward, runs)
        out["forward_backward"] = {"value": steps / med, "unit": "ray-steps/s", "cores": best, "median_s": med,
                                   "runs_s": ts, "faces_per_s": B / med,
                                   "sample": "same batch, forward with autograd graph + backward of sum(rendered) + "
                                             "sum(shadow weights), median of %d runs" % runs}
    torch.set_num_threads(cores)
    return out


def cpu_baseline_c(sample_faces=8, seed0=0):
    """Second, much stronger CPU reference point: the scalar C oracle under OpenMP on all host threads
    (shadow march + shade).  Reported next to `cpu_baseline`, which stays the op-for-op port of the reference."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    depth, mask, albedo, normals, light, amb = synth_faces(sample_faces, seed0)
    tt = c_oracle.sample_table(0.025, 0.005, N_SAMPLES)
    _, pt = c_oracle.light_prep(light, clamp_z_min=0.0)
    c_oracle.shadow_min_distance(depth[:1], mask[:1], pt[:1, None, :], tt)          # warm-up
    t = time.perf_counter()
    md, _ = c_oracle.shadow_min_distance(depth, mask, pt[:, None, :], tt)
    c_oracle.shade(normals.astype(np.float64), depth, albedo, pt[:, None, :], amb[:, None], md)
    dt = time.perf_counter() - t
    return {"value": sample_faces * H * W * N_SAMPLES / dt, "unit": "ray-steps/s", "cores": c_oracle.num_threads(),
            "kind": "port", "sample": "%d faces 256x256x160, oracle/gcfr_oracle.c (scalar C, OpenMP), %.2f s" % (sample_faces, dt)}


def measured_copy_bandwidth_gbs(dev, mb=1024, iters=5):
    """Device-to-device copy rate (read + write bytes) -- the achievable-HBM denominator SURVEY.md 8d asks to
    report beside the 8 TB/s spec peak."""
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    a.fill_(1.0)
    b.copy_(a)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * 4 * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9


def pmc_traffic_bytes():
    """HBM bytes per shadow_fwd launch from the committed rocprofv3 PMC pass, or None."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p)).get("shadow_fwd_hbm_bytes_per_launch")
    except Exception:
        return None


def pmc_valu_insts():
    """VALU wave-instructions per shadow_fwd launch from the committed PMC pass (SQ_INSTS_VALU), or None."""
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        return json.load(open(p)).get("valu_insts_per_launch")
    except Exception:
        return None


VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0     # wave-instructions/s: 1024 SIMDs, one wave64 VALU op per 4 cycles at 2.4 GHz


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--faces", type=int, default=FACES_PER_GPU, help="faces per GPU per step (configs[1]: 8)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--direct", action="store_true", help="A/B: direct-gather kernel (no workspace prepass)")
    ap.add_argument("--unfused", action="store_true", help="A/B: three separate entry points instead of gcfr_render_fwd")
    ap.add_argument("--from-depth", action="store_true",
                    help="also compute the normals (T8:353-354) inside the march epilogue instead of reading them "
                         "(SURVEY 8d's 17.4 B/ray-step accounting counts normals as a 12 B/pixel input, the default)")
    ap.add_argument("--streams", type=int, default=4,
                    help="issue successive steps round-robin on this many HIP streams, one RenderFwdPlan (own outputs "
                         "and workspace) per stream: independent batches overlap, the next step's prepass and "
                         "prologue fill the previous march's tail (B=8 is 8192 waves for 256 CUs).  1 = one stream")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue every step as a plan call (two kernel launches + event records, ~55 us of host time) "
                         "instead of replaying the plan's captured hipGraph (~10 us)")
    ap.add_argument("--eager", action="store_true",
                    help="call render_fwd (allocates its outputs per call, ~60 us of host time) instead of a plan")
    ap.add_argument("--size", type=int, default=256, help="other workloads: image side (config 5: 512)")
    ap.add_argument("--lights", type=int, default=1, help="lights per face (config 5: 18)")
    ap.add_argument("--samples", type=int, default=160, help="march steps (config 5: 320)")
    ap.add_argument("--depth-noise", type=float, default=0.0,
                    help="worst case for the depth-bound skip: add uniform noise of this amplitude to the depth maps "
                         "(an untrained network's output; the bounds then never separate ray and surface)")
    ap.add_argument("--mask", choices=["ellipse", "ones"], default="ellipse",
                    help="'ones' = worst case: no fully masked wave-step exists, nothing is skipped")
    ap.add_argument("--tune", type=str, default="",
                    help="A/B: comma list of gcfr_options knobs, e.g. tile_w=32,schedule=0,tile_order=2,ksplit=1,"
                         "depth_bound_skip=0,group=2 (never changes a result bit)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 --master-port 29500 bench.py --gpus %d" % (a.gpus, a.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product has no CPU path)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)   # nccl == RCCL on ROCm

    from geomconsistentfr_amd import RenderParams
    from geomconsistentfr_amd import block as R

    from geomconsistentfr_amd import _lib
    knobs = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)}
    base_opt = _lib.options(**knobs) if knobs else None
    B = a.faces
    headline = (a.size == 256 and a.lights == 1 and a.samples == 160 and a.mask == "ellipse" and a.depth_noise == 0.0)
    if headline:
        prm = RenderParams()
        depth, mask, albedo, normals, light, amb = synth_faces(B, seed0=rank * 1_000_000)
    else:
        prm = RenderParams(n_samples=a.samples, dt=0.8 / a.samples)
        depth, mask, albedo, normals, light, amb = synth_faces_sized(B, rank * 1_000_000, a.size, a.lights, a.mask)
    if a.depth_noise > 0.0:
        depth = depth + (a.depth_noise * np.random.default_rng(7).random(depth.shape)).astype(np.float32)
    Hh = Ww = a.size
    Ll, Nn = a.lights, a.samples
    d_depth = torch.from_numpy(depth).to(dev)
    d_mask = torch.from_numpy(mask).to(dev)
    d_albedo = torch.from_numpy(albedo).to(dev)
    d_normals = torch.from_numpy(normals).to(dev)
    d_light = torch.from_numpy(light).to(dev)
    d_amb = torch.from_numpy(amb).to(dev)

    # HIP events around the dominant (march) kernel alone, recorded on the launch stream by the library
    # itself (gcfr_options.event_start / event_stop); created through the same HIP runtime torch loaded.
    import ctypes
    L_ = _lib.load()
    # the exact file torch loaded (same inode -> the same runtime instance, never a second HIP runtime)
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    ev_pairs = []

    def new_event():
        e = ctypes.c_void_p()
        assert hip.hipEventCreate(ctypes.byref(e)) == 0
        return e

    cam = (1570.0 * Hh / 256.0, 1570.0 * Hh / 256.0, Ww / 2.0, Hh / 2.0, 1610.0)
    plans = None
    if not (a.eager or a.direct or a.unfused):
        d_mask_u8 = R.mask_to_u8(d_mask).reshape(-1, Hh, Ww).contiguous()
        d_light3, d_amb2 = d_light.reshape(B, Ll, 3).contiguous(), d_amb.reshape(B, Ll).contiguous()
        plans = [R.RenderFwdPlan(B, Ll, Hh, Ww, prm, dev, want_argmin=False, mask_batch=d_mask_u8.shape[0],
                                 camera=cam if a.from_depth else None, options=base_opt)
                 for _ in range(max(1, a.streams))]

    use_graph = plans is not None and not a.no_graph
    graph_error = None
    if use_graph:
        try:
            for p_ in plans:
                p_.capture(d_depth, d_mask_u8, d_light3, d_amb2, None if a.from_depth else d_normals, d_albedo)
        except Exception as e:      # a runtime that cannot capture: same kernels, issued call by call
            graph_error, use_graph = repr(e), False
            torch.cuda.synchronize()

    def step(timed):
        if use_graph and step.graph_ok:      # timed region: one hipGraph replay per step (no per-launch events)
            return plans[step.i % len(plans)].replay()
        opt = base_opt
        if timed:        # this call's options carry an event pair the library records around the march kernel
            e0, e1 = new_event(), new_event()
            opt = _lib.options(**knobs, event_start=e0, event_stop=e1)
            ev_pairs.append((e0, e1, opt))
        if plans is not None:
            pl_ = plans[step.i % len(plans)]
            pl_.options = opt
            out = pl_(d_depth, d_mask_u8, d_light3, d_amb2, None if a.from_depth else d_normals, d_albedo)
            pl_.options = base_opt
        elif a.direct or a.unfused:
            _, pt = R.light_prep(d_light, prm)
            md, _ = R.shadow_min_distance(d_depth, d_mask, pt.reshape(B, Ll, 3), prm, want_argmin=False,
                                          use_workspace=not a.direct, options=opt)
            out = R.shade(d_normals, d_depth, d_albedo, pt.reshape(B, Ll, 3), d_amb.reshape(B, Ll), md, prm)
        else:
            out = R.render_fwd(d_depth, d_mask, d_light.reshape(B, Ll, 3), d_amb.reshape(B, Ll),
                               None if a.from_depth else d_normals, d_albedo, prm, want_argmin=False,
                               camera=cam if a.from_depth else None, options=opt)
        return out

    streams = [torch.cuda.Stream(device=dev) for _ in range(a.streams)] if a.streams > 1 else None
    step.graph_ok = True
    step.i = 0

    def run_step(i, timed):
        step.i = i
        if streams is None:
            return step(timed)
        with torch.cuda.stream(streams[i % len(streams)]):
            return step(timed)

    for i in range(a.warmup):
        run_step(i, False)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for i in range(a.steps):
        run_step(i, True)
    host_issue = time.perf_counter() - t0      # host time to enqueue every step (before waiting for the GPU)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    def single_stream_reference(n=100):
        """the same steps on ONE stream, after the timed region (information only): per-step time and the
        march's un-overlapped launch duration"""
        saved = ev_pairs[:]
        del ev_pairs[:]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n):
            step.i = 0
            step(True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        ms = float(np.mean([elapsed_ms(e0, e1) for e0, e1, _ in ev_pairs]))
        ev_pairs[:] = saved
        return {"ms_per_step": 1e3 * dt / n, "ray_steps_per_sec": B * Ll * Hh * Ww * Nn * n / dt, "avg_launch_ms": ms}

    ray_steps_per_step = world * B * Ll * Hh * Ww * Nn
    value = ray_steps_per_step * a.steps / elapsed
    def elapsed_ms(e0, e1):
        ms = ctypes.c_float()
        assert hip.hipEventElapsedTime(ctypes.byref(ms), e0, e1) == 0
        return ms.value

    if use_graph:   # graph replays carry no per-launch events: measured below, on one stream
        shadow_ms = None
    else:
        shadow_ms = float(np.mean([elapsed_ms(e0, e1) for e0, e1, _ in ev_pairs]))
    # With several streams the launches of successive steps overlap: an event pair then brackets a kernel that
    # shares the GPU (and rocprofv3's tracing perturbs that overlap, so its average could not agree).  The
    # roofline therefore uses the kernel's UN-overlapped duration, measured live right after the timed region
    # with the same events over 100 launches of the same step on one stream; profiles/ holds the rocprofv3
    # summary of `bench.py --streams 1`, which that number agrees with.  The overlapped mean is kept beside it.
    overlapped_ms = None
    single = None
    if (streams is not None or use_graph) and not a.direct:
        step.graph_ok = False                      # plan calls with the library's event hook, one stream
        single = single_stream_reference()
        step.graph_ok = True
        overlapped_ms, shadow_ms = shadow_ms, single["avg_launch_ms"]
    algo_bytes = B * Ll * Hh * Ww * Nn * ALGO_BYTES_PER_RAY_STEP          # per launch (one rank)
    achieved = algo_bytes / (shadow_ms * 1e-3) / 1e9

    if rank == 0:
        out = {
            "metric": "ray_steps_per_sec", "value": value, "unit": "ray-steps/s", "n_gpus": world,
            "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64+f32",
            "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: batch=%d synthetic 256x256 faces per GPU, 1 light each, "
                                    "160 march steps, forward-only shadow+shade" % B) if headline else
                                   ("non-headline: batch=%d synthetic %dx%d faces per GPU, %d light(s) each, %d march "
                                    "steps, mask=%s, depth noise %g, forward-only shadow+shade" % (B, Hh, Ww, Ll, Nn, a.mask, a.depth_noise)),
                       "faces_per_gpu": B, "H": Hh, "W": Ww, "lights_per_face": Ll, "n_samples": Nn,
                       "parallelism": "dp%d" % world, "hip_streams": (a.streams if plans is not None or streams else 1),
                       "host_path": ("RenderFwdPlan, hipGraph replay" if use_graph else "RenderFwdPlan (preallocated outputs)")
                       if plans is not None else "render_fwd (eager)"},
            "faces_per_sec": world * B * Ll * a.steps / elapsed,
            "host_issue_ms_per_step": 1e3 * host_issue / a.steps,
            "ray_steps_per_sec_per_gpu": value / world,
            "roofline": {"bound": "hbm", "note": "north_star's HBM accounting; the gathers are cache-served (traffic << "
                         "algorithmic bytes, so frac can exceed 1) and the kernel is VALU-issue bound -- DESIGN.md 4.1",
                         "kernel": "shadow_fwd_quad_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_bytes(),
                         "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": shadow_ms,
                         "kernel_ray_steps_per_sec": B * Ll * Hh * Ww * Nn / (shadow_ms * 1e-3),
                         "measured_copy_GBs": measured_copy_bandwidth_gbs(dev)},
        }
        if graph_error:
            out["config"]["graph_capture_failed"] = graph_error
        vi = pmc_valu_insts()
        if vi:   # what actually bounds the kernel: VALU issue, not HBM
            out["roofline"]["valu"] = {"insts_per_launch": vi, "insts_per_nominal_ray_step": vi / (B * Ll * Hh * Ww * Nn),
                                       "achieved_insts_per_s": vi / (shadow_ms * 1e-3), "peak_insts_per_s": VALU_ISSUE_PEAK,
                                       "frac": vi / (shadow_ms * 1e-3) / VALU_ISSUE_PEAK,
                                       "note": "wave64 VALU instructions (rocprofv3 SQ_INSTS_VALU, headline config) per "
                                               "un-overlapped launch against 1024 SIMDs x 1 issue / 4 cycles"} if headline else None
        if single is not None:
            out["roofline"]["note"] += ("; avg_launch_ms is the kernel's un-overlapped duration (100 launches on one "
                                        "stream right after the timed region, HIP events around the kernel)")
            if overlapped_ms is not None:
                out["roofline"]["note"] += "; with %d streams in flight an event pair spans %.4f ms" % (a.streams, overlapped_ms)
                out["roofline"]["avg_launch_ms_overlapped"] = overlapped_ms
            out["single_stream"] = single
        if world == 1 and not a.no_cpu_baseline and headline:
            out["cpu_baseline"] = cpu_baseline()
            out["cpu_baseline_c_openmp"] = cpu_baseline_c()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
