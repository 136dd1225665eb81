#!/usr/bin/env python3
"""bench.py -- benchmarks of the render block on MI355X.

Metric (BASELINE.json): ray-steps/s (whole job) + relit faces/s on 256x256 faces, 160 march steps;
ray_steps = B*L*H*W*N nominal, never "steps executed".

--workload render (default; BASELINE configs[1]): a batch of 8 synthetic 256x256 faces per GPU, one light each,
    forward-only shadow + shade.  One "step" = one pass of the hot path over one batch: one gcfr_render_fwd enqueue
    (prepass: depth repack, statistics, depth bounds, light prep; then the ray march with the shading fused into its
    epilogue), inputs resident in HBM.  Steps are independent batches: `--streams S` (default 4) keeps S of them in
    flight on S HIP streams, each a hipGraph replay of a preallocated RenderFwdPlan -- `value` is that throughput;
    `single_stream` in the same line is the one-batch-at-a-time rate (= the batch-8 latency).
--workload train (BASELINE configs[2]; configs[3] under torchrun): batch of 32 faces per GPU, one full training step
    (RelightNet forward incl. the fused render block, PatchGAN every 5th step, seven losses, backward through the
    fused backward kernel, two Adam steps; DistributedDataParallel over RCCL when WORLD_SIZE > 1).

Multi-GPU (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`): faces are independent, so each
rank works on its own batch with no data-path collective ("weak" scaling; training adds DDP's gradient all-reduce);
the timed region is bracketed by barrier + synchronize on both sides and the max over ranks is reported.

The JSON line also carries
  roofline     -- for the dominant kernel, measured live with HIP events on the launch stream (un-captured plan calls
                  on one stream, library-recorded events around the kernel):
                  bound "valu": the SIMD issue time the kernel's instruction mix needs (rocprofv3 SQ_INSTS_VALU_* per
                  launch from profiles/pmc_summary.json x the per-class issue cost measured by tools/ubench_valu on this
                  chip) / launch duration, against 1024 SIMDs x 2.4 GHz -- a fraction <= 1 of a limit that binds;
                  `hbm`: north_star's accounting kept beside it (17.4 algorithmic B per nominal ray-step vs 8 TB/s; the
                  gathers are cache-served and ~91 % of the nominal ray-steps are provably skipped, so it exceeds 1);
                  `traffic` = HBM bytes per launch from the PMC passes (separate --pmc runs, KiB units, read side x2);
  cpu_baseline -- oracle/materialised.py (op-for-op torch-CPU port of the reference, which cannot travel to the GPU
                  box), T8 form B=3, forward and forward+backward, best of {8, 32, all} host threads (rank 0, N=1 only).
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


def synth_faces_sized(B, seed0, size, n_lights, mask_kind="ellipse", light_seed0=None):
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
    ls0 = seed0 if light_seed0 is None else light_seed0
    light = np.stack([np.roll(LIGHTS18, ls0 + i, axis=0)[:n_lights] for i in range(B)])
    amb = np.full((B, n_lights), 0.5, np.float32)
    return np.stack(depth), np.stack(mask), np.stack(albedo), np.stack(normals), light, amb


def synth_faces(B, seed0, light_seed0=None):
    """Deterministic synthetic faces: jittered ellipsoid + nose + ripple.
    Face i takes light (light_seed0 + i) mod 11 of the reference's eleven shipped directions (light_seed0 = seed0 unless given)."""
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
        light.append(lights11[((seed0 if light_seed0 is None else light_seed0) + i) % 11])
        amb.append(np.float32(0.5))
    return (np.stack(depth), np.stack(mask), np.stack(albedo), np.stack(normals), np.stack(light),
            np.asarray(amb, np.float32))


def cpu_baseline(seed0=0, runs=3, with_backward=True):
    """The CPU baseline of record: oracle/materialised.py -- the op-for-op torch-CPU port
    of T8:352-524, bit-equal to the imported reference (tests/test_oracle_vs_reference.py); the reference's own .py
    cannot travel to the GPU box -- in the reference's training form: a batch of B = 3 faces, normals from depth
    inside the timed region (T8:353), 256 x 256 x 160.  Forward under no_grad at 8, 32 and 64 host threads, median
    of `runs` after a warm-up each, the BEST thread count reported (all 256 hyper-threads of the GPU host are 14x
    slower than 32 for these memory-bound elementwise ops); then forward+backward (autograd through the port, as
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
    # (all 256 hyper-threads of the GPU host: 60 s per batch, 14x slower than 32 -- measured once in round 2,
    #  profiles/r02_bench_render.json; the sweep stops at 64 so that the default run stays within minutes)
    for th in sorted({min(8, cores), min(32, cores), min(64, cores)}):
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


def pmc_summary():
    """profiles/pmc_summary.json (tools/summarize_profile.py), or {}."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    except Exception:
        return {}


N_SIMD, NOMINAL_HZ = 1024, 2.4e9      # 256 CUs x 4 SIMDs; MI355X_MICROARCH.md max clock


def valu_roofline(kernel_entry, launch_ms):
    """VALU-issue roofline of one kernel from its committed PMC instruction mix and a LIVE launch duration."""
    v = kernel_entry["valu"]
    demanded = v["issue_cycles_per_launch"] / (launch_ms * 1e-3)            # SIMD issue cycles needed per second
    peak = N_SIMD * NOMINAL_HZ
    return {"bound": "valu", "kernel": kernel_entry["kernel"], "achieved": demanded / 1e9, "peak": peak / 1e9,
            "unit": "G SIMD-issue-cycles/s", "frac": demanded / peak, "avg_launch_ms": launch_ms,
            "valu_insts_per_launch": v["insts_per_launch"], "mean_issue_cycles_per_inst": v["mean_issue_cycles_per_inst"],
            "traffic": kernel_entry["hbm"]["bytes_per_launch"],
            "note": "achieved = sum over VALU instruction classes of (wave-instructions per launch, rocprofv3 "
                    "SQ_INSTS_VALU_* of this workload, profiles/pmc_summary.json) x (sustained issue cost of the class on "
                    "this chip, tools/ubench_valu -> profiles/r02_valu_cost_table.json), divided by the kernel's "
                    "un-overlapped launch duration measured live (HIP events recorded by the library around the "
                    "kernel, 100 plan calls on one stream); peak = 1024 SIMDs x 2.4 GHz"}


class HipEvents:
    """hipEvent_t handles through the HIP runtime torch already loaded (same inode -> the same runtime instance)."""

    def __init__(self):
        import ctypes
        self.ct = ctypes
        self.hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))

    def new(self):
        e = self.ct.c_void_p()
        assert self.hip.hipEventCreate(self.ct.byref(e)) == 0
        return e

    def elapsed_ms(self, e0, e1):
        ms = self.ct.c_float()
        assert self.hip.hipEventElapsedTime(self.ct.byref(ms), e0, e1) == 0
        return ms.value


def setup_distributed(a):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    return rank, world, dev, dist


def fence(dist):
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(dist, dev, seconds):
    if dist is None:
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ------------------------------------------------------------------------------------------------
# workload "render": BASELINE configs[1]
# ------------------------------------------------------------------------------------------------
def run_render(a, rank, world, dev, dist):
    from geomconsistentfr_amd import RenderParams, _lib
    from geomconsistentfr_amd import block as R

    knobs = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)}
    base_opt = _lib.options(**knobs) if knobs else None
    B = a.faces
    headline = (a.size == 256 and a.lights == 1 and a.samples == 160 and a.mask == "ellipse" and a.depth_noise == 0.0
                and B == FACES_PER_GPU and not knobs and not (a.direct or a.unfused or a.from_depth))
    n_streams = max(1, a.streams)
    Hh = Ww = a.size
    Ll, Nn = a.lights, a.samples
    default_shape = a.size == 256 and a.lights == 1 and a.samples == 160 and a.mask == "ellipse"
    prm = RenderParams() if default_shape else RenderParams(n_samples=a.samples, dt=0.8 / a.samples)

    def device_batch(j):
        """the j-th batch of B synthetic faces of this rank.  Every stream renders its OWN faces (geometry, mask, albedo:
        batches in flight are different data, as in serving) under the SAME light assignment as batch 0 (face i takes
        light i of the list), so that every batch is the workload BASELINE.md section 4 defines and not a harder or
        easier mix of grazing and overhead lights."""
        seed0 = rank * 1_000_000 + j * B
        if default_shape:
            depth, mask, albedo, normals, light, amb = synth_faces(B, seed0=seed0, light_seed0=rank * 1_000_000)
        else:
            depth, mask, albedo, normals, light, amb = synth_faces_sized(B, seed0, a.size, a.lights, a.mask,
                                                                         light_seed0=rank * 1_000_000)
        if a.depth_noise > 0.0:
            depth = depth + (a.depth_noise * np.random.default_rng(7 + j).random(depth.shape)).astype(np.float32)
        t = [torch.from_numpy(x).to(dev) for x in (depth, mask, albedo, normals, light, amb)]
        t[1] = R.mask_to_u8(t[1]).reshape(-1, Hh, Ww).contiguous()
        return t

    batches = [device_batch(j) for j in range(n_streams)]
    d_depth, d_mask_u8, d_albedo, d_normals, d_light, d_amb = batches[0]
    d_mask = d_mask_u8
    ev = HipEvents()
    cam = (1570.0 * Hh / 256.0, 1570.0 * Hh / 256.0, Ww / 2.0, Hh / 2.0, 1610.0)

    def inputs_of(bt):
        dd, mm, al, nr, li, am = bt
        return (dd, mm, li.reshape(B, Ll, 3).contiguous(), am.reshape(B, Ll).contiguous(), None if a.from_depth else nr, al)

    stream_inputs = [inputs_of(bt) for bt in batches]
    plan_inputs = stream_inputs[0]

    def new_plan():
        return R.RenderFwdPlan(B, Ll, Hh, Ww, prm, dev, want_argmin=False, mask_batch=d_mask_u8.shape[0],
                               camera=cam if a.from_depth else None, options=base_opt)

    use_plans = not (a.eager or a.direct or a.unfused)
    plans = [new_plan() for _ in range(n_streams)] if use_plans else None
    use_graph, graph_error = use_plans and not a.no_graph, None
    if use_graph:
        try:
            for p_, inp in zip(plans, stream_inputs):
                p_.capture(*inp)
        except Exception as e:      # a runtime that cannot capture: same kernels, issued call by call
            graph_error, use_graph = repr(e), False
            torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(n_streams)]

    def eager_step(opt):
        if a.direct or a.unfused:
            _, pt = R.light_prep(d_light, prm)
            md, _ = R.shadow_min_distance(d_depth, d_mask, pt.reshape(B, Ll, 3), prm, want_argmin=False,
                                          use_workspace=not a.direct, options=opt)
            return R.shade(d_normals, d_depth, d_albedo, pt.reshape(B, Ll, 3), d_amb.reshape(B, Ll), md, prm)
        return R.render_fwd(d_depth, d_mask, d_light.reshape(B, Ll, 3), d_amb.reshape(B, Ll),
                            None if a.from_depth else d_normals, d_albedo, prm, want_argmin=False,
                            camera=cam if a.from_depth else None, options=opt)

    def issue(i):
        """enqueue step i (no events): graph replay, plan call or eager call on stream i % S"""
        with torch.cuda.stream(streams[i % n_streams]):
            if use_graph:
                plans[i % n_streams].replay()
            elif use_plans:
                plans[i % n_streams](*stream_inputs[i % n_streams])
            else:
                eager_step(base_opt)

    def timed_run(n_steps, stream_count):
        """n_steps steps round-robin over the first `stream_count` streams, fenced on both sides"""
        nonlocal n_streams
        saved, n_streams = n_streams, stream_count
        fence(dist)
        t0 = time.perf_counter()
        for i in range(n_steps):
            issue(i)
        host = time.perf_counter() - t0
        fence(dist)
        dt = time.perf_counter() - t0
        n_streams = saved
        return max_over_ranks(dist, dev, dt), host

    for i in range(a.warmup):
        issue(i)
    elapsed, host_issue = timed_run(a.steps, n_streams)

    # the dominant kernel's un-overlapped launch duration: plan calls (not graph replays) on ONE stream, each with
    # its own event pair recorded by the library immediately before / after the march kernel on that stream
    def kernel_launch_ms(n=100):
        pairs = []
        torch.cuda.synchronize()
        with torch.cuda.stream(streams[0]):
            for _ in range(n):
                e0, e1 = ev.new(), ev.new()
                opt = _lib.options(**knobs, event_start=e0, event_stop=e1)
                pairs.append((e0, e1, opt))
                if use_plans:
                    plans[0].options = opt
                    plans[0](*plan_inputs)
                    plans[0].options = base_opt
                else:
                    eager_step(opt)
        torch.cuda.synchronize()
        return float(np.mean([ev.elapsed_ms(e0, e1) for e0, e1, _ in pairs]))

    shadow_ms = kernel_launch_ms()
    single_steps = max(50, min(a.steps, 1000))
    single_elapsed, _ = timed_run(single_steps, 1) if n_streams > 1 else (elapsed * single_steps / a.steps, None)
    ray_steps_per_step_rank = B * Ll * Hh * Ww * Nn
    value = world * ray_steps_per_step_rank * a.steps / elapsed
    single = {"ms_per_step": 1e3 * single_elapsed / single_steps,
              "ray_steps_per_sec": world * ray_steps_per_step_rank * single_steps / single_elapsed,
              "note": "the same steps one at a time on ONE stream (hipGraph replay): the latency of one batch"}
    if rank != 0:
        return None
    algo_bytes = ray_steps_per_step_rank * ALGO_BYTES_PER_RAY_STEP          # per launch (one rank)
    achieved_gbs = algo_bytes / (shadow_ms * 1e-3) / 1e9
    pm = pmc_summary()
    hbm_line = {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": algo_bytes, "measured_copy_GBs": measured_copy_bandwidth_gbs(dev),
                "note": "north_star's accounting: 17.4 algorithmic B per NOMINAL ray-step (SURVEY 8d) / launch duration; the "
                        "gathers are cache-served and most nominal ray-steps are provably skipped, so this is not a "
                        "fraction of a limit (it exceeds 1) -- the binding roofline is the VALU one"}
    fwd = pm.get("kernels", {}).get("fwd")
    if headline and fwd:
        roof = valu_roofline(fwd, shadow_ms)
        roof["hbm"] = hbm_line
        roof["kernel_ray_steps_per_sec"] = ray_steps_per_step_rank / (shadow_ms * 1e-3)
        if "work" in fwd:
            roof["executed_fraction_of_nominal_ray_steps"] = fwd["work"]["executed_fraction_of_nominal"]
            roof["valu_wave_insts_per_executed_wave_step"] = fwd["work"]["valu_wave_insts_per_executed_wave_step"]
        if "l1" in fwd:
            roof["l1_frac_of_peak_under_rocprofv3"] = fwd["l1"]["frac"]
        # the same mix against the overlapped rate: what the chip's VALU does when `streams` launches share it
        roof["frac_at_throughput"] = fwd["valu"]["issue_cycles_per_launch"] / (N_SIMD * NOMINAL_HZ * elapsed / a.steps)
    else:       # no PMC mix for this workload / kernel selection: HBM accounting only
        roof = dict(hbm_line, kernel="shadow_fwd_quad_kernel" if not a.direct else "shadow_fwd_kernel",
                    avg_launch_ms=shadow_ms, traffic=None,
                    kernel_ray_steps_per_sec=ray_steps_per_step_rank / (shadow_ms * 1e-3))
    out = {
        "metric": "ray_steps_per_sec", "value": value, "unit": "ray-steps/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64+f32", "data": "synthetic",
        "config": {"workload": ("BASELINE configs[1]: batch=%d synthetic 256x256 faces per GPU, 1 light each, 160 march "
                                "steps, forward-only shadow+shade; %d batch(es) in flight" % (B, n_streams)) if headline else
                               ("non-headline: batch=%d synthetic %dx%d faces per GPU, %d light(s) each, %d march steps, mask=%s, "
                                "depth noise %g, knobs %s, forward-only shadow+shade; %d batch(es) in flight"
                                % (B, Hh, Ww, Ll, Nn, a.mask, a.depth_noise, knobs or "default", n_streams)),
                   "faces_per_gpu": B, "H": Hh, "W": Ww, "lights_per_face": Ll, "n_samples": Nn,
                   "parallelism": "dp%d" % world, "hip_streams": n_streams, "batches_in_flight": n_streams,
                   "distinct_face_batches": n_streams,
                   "host_path": ("RenderFwdPlan, hipGraph replay" if use_graph else "RenderFwdPlan (preallocated outputs)")
                   if use_plans else "render_fwd (eager)"},
        "faces_per_sec": world * B * Ll * a.steps / elapsed,
        "host_issue_ms_per_step": 1e3 * host_issue / a.steps,
        "ray_steps_per_sec_per_gpu": value / world,
        "single_stream": single,
        "latency_one_batch_ms": single["ms_per_step"],
        # `value` is a throughput with `hip_streams` batches in flight (faces_in_flight below), not the rate of one batch
        # of `faces_per_gpu` on its own -- that one is `single_stream` (VERDICT r01 asked for both to be named)
        "faces_in_flight_per_gpu": B * n_streams,
        "throughput_in_flight": {"faces_in_flight_per_gpu": B * n_streams, "ray_steps_per_sec": value},
        "roofline": roof,
    }
    if graph_error:
        out["config"]["graph_capture_failed"] = graph_error
    if world == 1 and not a.no_cpu_baseline and headline:
        out["cpu_baseline"] = cpu_baseline()
        out["cpu_baseline_c_openmp"] = cpu_baseline_c()
    return out


# ------------------------------------------------------------------------------------------------
# workload "train": BASELINE configs[2] (1 GPU) / configs[3] (8 GPUs, DDP over RCCL)
# ------------------------------------------------------------------------------------------------
def run_train(a, rank, world, dev, dist):
    from geomconsistentfr_amd import _lib
    from geomconsistentfr_amd import block as R
    from geomconsistentfr_amd.train import TrainConfig, Trainer, synthetic_batch

    B = a.faces if a.faces != FACES_PER_GPU else 32                          # configs[2]: batch=32 per GPU
    torch.manual_seed(1234 + rank)
    tr = Trainer(TrainConfig(), device=dev, distributed=dist is not None)
    batch = synthetic_batch(B, rank * 1_000_000, device=dev)
    epoch = 200                                                               # every epoch-gated skip on (T8:245-283)
    for j in range(max(a.warmup, 6)):                                        # MIOpen find mode tunes on first use
        tr.step(batch, epoch, j, log=False)
    fence(dist)
    t0 = time.perf_counter()
    for j in range(a.steps):
        tr.step(batch, epoch, j, log=False)                                  # D step every 5th (T8:624), G step always
    fence(dist)
    elapsed = max_over_ranks(dist, dev, time.perf_counter() - t0)

    # the render block's own kernels on this batch, measured live on the current stream: forward (prepass + march with
    # fused normals + shading, argmin variant) and the fused backward, from the tensors of a real step's forward
    with torch.no_grad():
        albedo, depth, SL = tr.model.features(batch["images"], epoch)
    prm = tr.model.render_params
    masks = R.mask_to_u8(batch["masks_fill"].reshape(B, 256, 256))
    cam = R.camera_scalars(tr.K) + (tr.model.normal_z_offset,)
    plan = R.RenderFwdPlan(B, 1, 256, 256, prm, dev, want_argmin=True, camera=cam)
    ins = (depth.reshape(B, 256, 256).contiguous(), masks, SL[:, 0, 0, 1:4].reshape(B, 1, 3).contiguous(),
           SL[:, 0, 0, 0].reshape(B, 1).contiguous(), None, albedo.contiguous())
    ev = HipEvents()
    pairs = []
    for _ in range(30):
        e0, e1 = ev.new(), ev.new()
        plan.options = _lib.options(event_start=e0, event_stop=e1)
        pairs.append((e0, e1, plan.options))
        o = plan(*ins)
    torch.cuda.synchronize()
    march_ms = float(np.mean([ev.elapsed_ms(e0, e1) for e0, e1, _ in pairs[5:]]))
    L_ = _lib.load()
    g_ren = torch.rand((B, 1, 3, 256, 256), device=dev) * masks[:, None, None].float()  # the losses mask the rendered image
    g_alb, g_depth = torch.empty((B, 3, 256, 256), device=dev), torch.zeros((B, 256, 256), device=dev)
    g_pt, g_amb = torch.zeros((B, 1, 3), dtype=torch.float64, device=dev), torch.zeros((B, 1), dtype=torch.float64, device=dev)
    tt = R.sample_table(prm, dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    t_ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def bwd_once():
        _lib.check(L_.gcfr_render_bwd(ins[0].data_ptr(), ins[5].data_ptr(), o["light_pt"].data_ptr(), ins[3].data_ptr(),
                                      o["minimum_distance"].data_ptr(), o["argmin"].data_ptr(), o["surface_normals"].data_ptr(),
                                      B, 1, 256, 256, prm.n_samples, tt.data_ptr(), *cam[:4], cam[4], 1,
                                      float(prm.directional_intensity), None, None, None, g_ren.data_ptr(), None,
                                      g_alb.data_ptr(), g_depth.data_ptr(), g_pt.data_ptr(), g_amb.data_ptr(), st), "gcfr_render_bwd")

    for _ in range(5):
        bwd_once()
    t_ev[0].record()
    for _ in range(30):
        bwd_once()
    t_ev[1].record()
    torch.cuda.synchronize()
    bwd_ms = t_ev[0].elapsed_time(t_ev[1]) / 30
    if rank != 0:
        return None
    ray_steps = B * 256 * 256 * N_SAMPLES
    value = world * ray_steps * a.steps / elapsed
    pm = pmc_summary().get("kernels", {})
    px_bytes = 72.0     # per pixel: reads depth 4 + albedo 12 + g_rendered 12 + min_dist 4 + argmin 4 + normals 12 (+ stencil
    #                     neighbours from cache), writes grad_albedo 12 + grad_depth read-modify-write 8 + light partials ~0
    bwd_roof = {"bound": "hbm", "kernel": "gcfr::render_bwd_single_light_kernel", "avg_launch_ms": bwd_ms,
                "achieved": B * 65536 * px_bytes / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": B * 65536 * px_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": B * 65536 * px_bytes,
                "traffic": pm.get("bwd", {}).get("hbm", {}).get("bytes_per_launch"),
                "note": "one backward sample per pixel: compulsory I/O 72 B per pixel; what holds the kernel back is the L2's "
                        "f32 atomic rate (4 bilinear-corner atomics per pixel: 78 of 167 us before same-texel merging, "
                        "profiles/r02_bwd_stage_trace.txt, DESIGN.md 4.5), then f64 VALU issue"}
    # (no VALU line here: the committed instruction mixes are those of tools/bwd_bench.py -- dense upstream gradient,
    #  depth noise 2 -- not of this step's masked gradient and untrained-network depth; profiles/pmc_summary.json has
    #  them with their own launch times: backward 0.40, training march 0.66 of the VALU issue capacity)
    return {
        "metric": "ray_steps_per_sec", "value": value, "unit": "ray-steps/s", "n_gpus": world, "steps": a.steps,
        "warmup": max(a.warmup, 6), "ms_per_step": 1e3 * elapsed / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 network (MIOpen) + f64/f32 render block", "data": "synthetic",
        "config": {"workload": "BASELINE configs[%d]: batch=%d per GPU, full training step (RelightNet forward with the fused HIP "
                               "render block, PatchGAN step every 5th iteration, seven losses, backward through the fused HIP "
                               "backward, two Adam steps)%s" % (2 if world == 1 else 3, B,
                                                               "" if world == 1 else ", DistributedDataParallel over RCCL"),
                   "faces_per_gpu": B, "global_batch": B * world, "H": 256, "W": 256, "n_samples": N_SAMPLES,
                   "parallelism": "dp%d" % world, "epoch": epoch},
        "faces_per_sec": world * B * a.steps / elapsed,
        "render_block_ms": {"forward_march_kernel": march_ms, "fused_backward_kernel": bwd_ms,
                            "share_of_step": (march_ms + bwd_ms) / (1e3 * elapsed / a.steps),
                            "note": "the step is MIOpen-bound (fp32 convolutions of the hourglass and PatchGAN); the render "
                                    "block's two big kernels are this share of it"},
        "roofline": bwd_roof,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 3000 (render), 20 (train)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 50 (render), 6 (train)")
    ap.add_argument("--workload", choices=["render", "train"], default="render",
                    help="render = BASELINE configs[1] (batch 8, forward); train = configs[2]/[3] (batch 32, full step)")
    ap.add_argument("--faces", type=int, default=FACES_PER_GPU, help="faces per GPU per step (configs[1]: 8; train: 32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--direct", action="store_true", help="A/B: direct-gather kernel (no workspace prepass)")
    ap.add_argument("--unfused", action="store_true", help="A/B: three separate entry points instead of gcfr_render_fwd")
    ap.add_argument("--from-depth", action="store_true",
                    help="also compute the normals (T8:353-354) inside the march epilogue instead of reading them "
                         "(SURVEY 8d's 17.4 B/ray-step accounting counts normals as a 12 B/pixel input, the default)")
    ap.add_argument("--streams", type=int, default=4,
                    help="batches in flight: successive steps go round-robin to this many HIP streams, one RenderFwdPlan (own "
                         "outputs and workspace) per stream.  One launch cannot fill the chip to its end -- its duration is "
                         "that of its heaviest tile (profiles/r02_schedule_experiments.md) -- a second batch at a different "
                         "phase does.  1 = one batch at a time (also always reported as `single_stream`)")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue every step as a plan call (two kernel launches, ~55 us of host time) instead of replaying "
                         "the plan's captured hipGraph (~10 us)")
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
                    help="A/B: comma list of gcfr_options knobs, e.g. tile_w=32,ksplit=1,depth_bound_skip=0,group=2 "
                         "(never changes a result bit)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 3000 if a.workload == "render" else 20
    if a.warmup is None:
        a.warmup = 50 if a.workload == "render" else 6
    rank, world, dev, dist = setup_distributed(a)
    out = (run_render if a.workload == "render" else run_train)(a, rank, world, dev, dist)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
