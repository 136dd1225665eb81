#!/usr/bin/env python3
"""bench.py -- benchmarks of the render block on MI355X.

Metric (BASELINE.json): ray-steps/s (whole job) + relit faces/s on 256x256 faces, 160 march steps;
ray_steps = B*L*H*W*N nominal, never "steps executed".

--workload render (default; BASELINE configs[1]): a batch of 8 synthetic 256x256 faces per GPU, one light each,
    forward-only normals + shadow + shade.  One "step" = one pass of the hot path over one batch: one
    gcfr_render_from_depth_fwd enqueue -- what RelightNet.forward runs for T8:353-522 -- (prepass: depth repack, statistics,
    depth bounds, light prep; then the ray march with the normals stencil and the shading fused into its epilogue), inputs
    resident in HBM.  The same line carries, measured in the same run: `normals_in_*` (SURVEY 8d's accounting form, normals
    as an input: rounds 1-3's headline step), `config5_*` (configs[4]'s per-GPU shape: 1 face x 18 lights x 512x512 x 320),
    `train_*` (configs[2]: batch 32, the full training step) and `worst_case`.  Steps are independent batches: `--streams S` (default 4) keeps S of them in
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
                  launch from profiles/pmc_summary.json x the guide's SPEC issue cycles per class; `frac_measured_costs`
                  prices the same mix at the sustained costs tools/ubench_valu measured on this chip) / launch duration,
                  against 1024 SIMDs x 2.4 GHz -- a fraction <= 1 of a limit that binds;
                  `hbm`: north_star's accounting kept beside it (17.4 algorithmic B per nominal ray-step vs 8 TB/s; the
                  gathers are cache-served and ~91 % of the nominal ray-steps are provably skipped, so it exceeds 1);
                  `traffic` = HBM bytes per launch from the PMC passes (separate --pmc runs, KiB units, read side x2);
  cpu_baseline -- oracle/materialised.py (op-for-op torch-CPU port of the reference, which cannot travel to the GPU
                  box), T8 form B=3, forward and forward+backward, best of {8, 32} host threads (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

T_START = time.perf_counter()   # process start (before torch pages in): the aux legs of the headline line stop adding work after ~170 s

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H = W = 256
N_SAMPLES = 160
FACES_PER_GPU = 8
ALGO_BYTES_PER_RAY_STEP = 17.4      # 4 f32 depth corners + 1 u8 mask cell + 0.4 B amortised pixel I/O
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8 TB/s spec


sys.path.insert(0, os.path.join(ROOT, "tools"))
from scenes import LIGHTS18, ffhq_faces, synth_faces, synth_faces_sized  # noqa: E402,F401  (tools/scenes.py: shared with the tests)


def cpu_baseline(seed0=0, runs=2, with_backward=True):
    """The CPU baseline of record: oracle/materialised.py -- the op-for-op torch-CPU port
    of T8:352-524, bit-equal to the imported reference (tests/test_oracle_vs_reference.py); the reference's own .py
    cannot travel to the GPU box -- in the reference's training form: a batch of B = 3 faces, normals from depth
    inside the timed region (T8:353), 256 x 256 x 160.  Forward under no_grad at 8 and 32 host threads (64 was never the best in
    rounds 2-3 and is dropped so that the line's other legs fit the default run), median
    of `runs` after a warm-up each, the BEST thread count reported (all 256 hyper-threads of the GPU host are 14x
    slower than 32 for these memory-bound elementwise ops); then forward+backward (autograd through the port, as
    loss.backward() replays the reference's graph, T8:655) at that thread count, median of `runs`.
    Checker code used strictly as the reported baseline; bounded: about one minute of host time."""
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
    #  profiles/r02_bench_render.json; 64 threads: 4.8 M/s against 8.2 M/s at 8, BENCH_r03.json)
    for th in sorted({min(8, cores), min(32, cores)}):
        torch.set_num_threads(th)
        forward()                                                            # warm-up (thread pool, allocator)
        med, ts = median_time(forward, runs)
        by_threads[th] = {"median_s": med, "runs_s": ts, "ray_steps_per_sec": steps / med}
    best = max(by_threads, key=lambda th: by_threads[th]["ray_steps_per_sec"])
    out = {"value": by_threads[best]["ray_steps_per_sec"], "unit": "ray-steps/s", "cores": best, "kind": "port",
           "host_threads_available": cores, "faces_per_s": B / by_threads[best]["median_s"],
           "forward_by_threads": {str(k): v for k, v in by_threads.items()},
           "sample": "T8 form, 3 faces 256x256x160 incl. normals, forward no_grad, oracle/materialised.py, median of %d runs, "
                     "best of %s threads: %.2f s/batch" % (runs, sorted(by_threads), by_threads[best]["median_s"])}
    if with_backward:
        torch.set_num_threads(best)
        med, ts = median_time(forward_backward, 1)       # (one run: ~8 s; round 6 gave its second run's time to the relight_e2e leg)
        out["forward_backward"] = {"value": steps / med, "unit": "ray-steps/s", "cores": best, "median_s": med,
                                   "runs_s": ts, "faces_per_s": B / med,
                                   "sample": "same batch, forward + autograd backward of sum(rendered) + sum(shadow weights), one run"}
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


def measured_copy_bandwidth_gbs(dev, mb=1024, iters=10):
    """Achievable HBM rate (read + write bytes) -- the denominator SURVEY.md 8d asks to report beside the 8 TB/s spec peak:
    the library's float4 grid-stride copy kernel (gcfr_copy_probe: one workgroup per CU, four 16-B loads in flight per lane,
    non-temporal -- the best of tools/copy_probe_sweep.hip's sweep: 6.27 TB/s, MI355X_MICROARCH.md quotes 6.29), 1 GiB in and 1 GiB
    out, HIP events on the launch stream.  Rounds 1-4 timed torch's copy_ here, which reached 4.8 TB/s."""
    import ctypes
    from geomconsistentfr_amd import _lib
    L = _lib.load()
    n = mb * 1024 * 1024
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    a.fill_(1)
    st = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(L.gcfr_copy_probe(a.data_ptr(), b.data_ptr(), ctypes.c_size_t(n), st), "gcfr_copy_probe")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        _lib.check(L.gcfr_copy_probe(a.data_ptr(), b.data_ptr(), ctypes.c_size_t(n), st), "gcfr_copy_probe")
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * n * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9


def parity_spot_check(rig):
    """CHECKER LEG (the only use of oracle/ in the timed process besides cpu_baseline): face 0 of the timed batch 0, through the
    TIMED plan (a replay of its hipGraph), against the C oracle -- minimum distance bit for bit, shadow weight and RGB <= 2e-5
    (gates: 1e-4 / 1e-3, BASELINE.json).  Raises on a difference: a fast library that renders something else must not produce a
    line.  Returns a short description for config.parity_spot_check."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import c_oracle
    from normals_restatement import depth_to_normals
    rig.issue(0, 1)
    torch.cuda.synchronize()
    out = rig.plans[0].out
    dd, mm, al, nr, li, am = [None if t is None else t.detach().cpu().numpy() for t in rig.batches[0]]
    size, Ll, N = rig.size, rig.L, rig.N
    depth, mask, albedo = dd[:1].reshape(1, size, size), mm[:1].reshape(1, size, size), al[:1]
    light, amb = li.reshape(rig.B, Ll, 3)[:1], am.reshape(rig.B, Ll)[:1]
    _, pt = c_oracle.light_prep(light.reshape(-1, 3), clamp_z_min=0.0)
    pt = pt.reshape(1, Ll, 3)
    md, _ = c_oracle.shadow_min_distance(depth, mask, pt, c_oracle.sample_table(rig.prm.t0, rig.prm.dt, N))
    got_md = out["minimum_distance"][:1].cpu().numpy().reshape(md.shape)
    if not np.array_equal(got_md, md):
        raise SystemExit("bench.py: parity spot check FAILED: %d of %d minimum distances differ from the C oracle"
                         % (int((got_md != md).sum()), md.size))
    if rig.from_depth:
        fx, fy, cx, cy, zoff = rig.cam
        K = torch.zeros(1, 3, 3, dtype=torch.float64)
        K[:, 0, 0], K[:, 1, 1], K[:, 2, 2], K[:, 0, 2], K[:, 1, 2] = fx, fy, 1.0, cx, cy
        n = depth_to_normals(torch.from_numpy(depth)[:, None] + zoff, K)
        n[:, 1] = -n[:, 1]
        normals = n.numpy()
    else:
        normals = nr[:1].astype(np.float64)
    ref = c_oracle.shade(normals, depth, albedo, pt, amb, md, intensity=float(rig.prm.directional_intensity))
    e_w = float(np.abs(out["shadow_mask_weights"][:1].cpu().numpy().reshape(ref["shadow_w"].shape) - ref["shadow_w"]).max())
    e_rgb = float(np.abs(out["rendered_images"][:1].cpu().numpy().reshape(ref["rendered"].shape) - ref["rendered"]).max())
    if not (e_w <= 2e-5 and e_rgb <= 2e-5):
        raise SystemExit("bench.py: parity spot check FAILED: max|dw| = %.3g, max|dRGB| = %.3g against the C oracle (2e-5)" % (e_w, e_rgb))
    return "ok: face 0 of the timed batch through the timed plan vs the C oracle -- min_dist bit-equal (%d pixels), max|dw| %.1e, max|dRGB| %.1e" % (md.size, e_w, e_rgb)


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

    def destroy(self, e):
        self.hip.hipEventDestroy(e)


# ------------------------------------------------------------------------------------------------
# process layout: `python bench.py --gpus N` spawns its own N ranks (one per GPU) when it was not started by
# torch.distributed.run; under torch.distributed.run (RANK / WORLD_SIZE set) it is one of the ranks.
# ------------------------------------------------------------------------------------------------
def self_launch(a):
    """Re-exec this script as N ranks under torch.distributed.run on 127.0.0.1 (a free port), rank r -> GPU r.
    Returns the launcher's exit code; rank 0's JSON line passes through on stdout."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    env["GCFR_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class Ranks:
    """torch.distributed plumbing of one rank: `nccl` (= RCCL) when every rank has its own GPU; `gloo` when the ranks
    share GPUs (--oversubscribe: launcher / barrier / max-over-ranks rehearsal on a box with fewer GPUs -- RCCL refuses
    two ranks on one device) or when there is no GPU at all (--dry-run on CPU)."""

    def __init__(self, a):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        n_dev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        self.n_dev = n_dev
        if not a.dry_run:
            assert n_dev > 0, "bench.py needs a GPU (the product has no CPU path)"
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(self.world)))
        self.shared_gpus = n_dev < local_world
        if self.shared_gpus and n_dev > 0 and not (a.oversubscribe or a.dry_run):
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (add --oversubscribe to rehearse the launch "
                             "path with several ranks per GPU over gloo)" % (self.world, n_dev))
        self.dev = torch.device("cuda", self.local_rank % n_dev) if n_dev else torch.device("cpu")
        if n_dev:
            torch.cuda.set_device(self.dev)
        self.dist, self.backend = None, None
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            self.backend = "gloo" if (self.shared_gpus or not n_dev) else "nccl"
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)   # nccl == RCCL on ROCm
            else:
                dist.init_process_group("gloo")
            self.dist = dist
        self.cdev = self.dev if self.backend == "nccl" else torch.device("cpu")   # where collective payloads live

    def fence(self):
        if self.n_dev:
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if self.n_dev:
            torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = torch.tensor([seconds], dtype=torch.float64, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather(self, value):
        """[value of rank 0, ..., value of rank N-1] on every rank"""
        if self.dist is None:
            return [float(value)]
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.cdev)
        out = [torch.zeros_like(t) for _ in range(self.world)]
        self.dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def counted_ranks(self):
        """number of ranks that took part in a SUM all-reduce of ones that really ran on the collective backend"""
        if self.dist is None:
            return 1
        t = torch.ones(1, dtype=torch.float32, device=self.cdev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        if self.n_dev:
            torch.cuda.synchronize()
        return int(round(float(t.item())))

    def describe(self):
        n = self.counted_ranks()
        return {"n_gpus": self.world, "rccl_ranks": n if self.backend == "nccl" else (1 if self.dist is None else None),
                "ranks": n,
                "collective_backend": {"nccl": "nccl (RCCL)", "gloo": "gloo (%d ranks share %d GPU(s): launch-path "
                                       "rehearsal, not a scaling number)" % (self.world, self.n_dev), None: "none (1 rank)"}[self.backend],
                "self_launched": os.environ.get("GCFR_BENCH_SELF_LAUNCHED") == "1"}

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


FORCED_REGIONS = 0       # --regions N: exactly N timed regions (profiling runs want 1); 0 = the rule below


def regions_needed(steps, est_ms_per_step):
    """A fenced region shorter than ~200 ms is dominated by its fixed fence / drain cost and is ONE sample (round 2's
    record rested on a 1.03 ms region): repeat it >= 25 times, bounded to ~2 s in total."""
    if FORCED_REGIONS > 0:
        return FORCED_REGIONS
    region_ms = steps * est_ms_per_step
    if region_ms >= 200.0:
        return 1
    return int(max(25, min(200, 2000.0 / max(region_ms, 1e-3))))


# ------------------------------------------------------------------------------------------------
# workload "render": BASELINE configs[1]
# ------------------------------------------------------------------------------------------------
_RIG_STREAMS = {}


def rig_streams(dev, n):
    """The process's HIP streams for batches in flight: created ONCE per device and shared by every rig / leg of the run.
    torch hands out pooled streams round-robin and HIP maps streams onto a handful of hardware queues; a rig that created its own
    four streams late in a run (dozens of streams created before it: plans' capture streams, earlier rigs) was seen to lose its
    overlap -- the `many_batches` leg read 1.57 T on its own late streams and 1.97 T on these, reproducibly; the relight leg's
    eleven-light pair 28.7 k against 41.3 k images/s.  Streams are plumbing: one set per process."""
    key = (dev.type, dev.index)
    have = _RIG_STREAMS.setdefault(key, [])
    while len(have) < n:
        have.append(torch.cuda.Stream(device=dev))
    return have[:n]


class RenderRig:
    """One render workload on one rank: `streams` batches of B faces resident in HBM, one RenderFwdPlan (own outputs and
    workspace) per batch, each captured into a hipGraph; `timed(steps, streams)` issues `steps` steps round-robin and
    returns the fenced wall time (max over ranks)."""

    def __init__(self, rk, B, size=256, lights=1, samples=160, mask="ellipse", depth_noise=0.0, data="synthetic",
                 streams=4, from_depth=False, want_argmin=False, knobs=None, graph=True, mode="plan", pixels="all",
                 normals_stage="auto", n_batches=None):
        from geomconsistentfr_amd import RenderParams, _lib
        from geomconsistentfr_amd import block as R
        self.rk, self.R, self._lib = rk, R, _lib
        self.B, self.size, self.L, self.N = B, size, lights, samples
        self.mask, self.depth_noise, self.data = mask, depth_noise, data
        self.from_depth, self.want_argmin, self.mode = from_depth, want_argmin, mode
        self.knobs = knobs or {}
        self.normals_stage = normals_stage
        self.base_opt = _lib.options(**self.knobs) if self.knobs else None
        self.n_streams = max(1, streams)
        dev = self.dev = rk.dev
        self.default_shape = size == 256 and lights == 1 and samples == 160
        self.prm = RenderParams(pixels=pixels) if self.default_shape else RenderParams(n_samples=samples, dt=0.8 / samples, pixels=pixels)
        self.cam = (1570.0 * size / 256.0, 1570.0 * size / 256.0, size / 2.0, size / 2.0, 1610.0)
        # distinct batches resident in HBM, each with its own plan (outputs + workspace) and graph: by default one per stream (the
        # headline: 4 x 8 faces, ~154 MB -- inside the 256-MB Infinity Cache); `n_batches` > streams cycles through more of them
        # (the `many_batches` leg: 64 x 8 faces, ~2.4 GB, far outside it)
        self.n_batches = max(self.n_streams, n_batches or self.n_streams)
        self.batches = [self._device_batch(j) for j in range(self.n_batches)]
        self.inputs = [self._inputs_of(bt) for bt in self.batches]
        self.use_plans = mode == "plan"
        self.plans = [self._new_plan() for _ in range(self.n_batches)] if self.use_plans else None
        self.use_graph, self.graph_error = self.use_plans and graph, None
        if self.use_graph:
            try:
                for p_, inp in zip(self.plans, self.inputs):
                    p_.capture(*inp)
            except Exception as e:      # a runtime that cannot capture: same kernels, issued call by call
                self.graph_error, self.use_graph = repr(e), False
                torch.cuda.synchronize()
        self.streams = rig_streams(dev, self.n_streams)

    # -- data ----------------------------------------------------------------------------------
    def _device_batch(self, j):
        """the j-th batch of B faces of this rank.  Every stream renders its OWN faces (geometry, mask, albedo: batches
        in flight are different data, as in serving) under the SAME light assignment as batch 0 (face i takes light i of
        the list), so that every batch is the workload BASELINE.md section 4 defines and not a harder or easier mix of
        grazing and overhead lights."""
        B, rank = self.B, self.rk.rank
        seed0 = rank * 1_000_000 + j * B
        if self.data == "ffhq":
            assert self.default_shape, "--data ffhq: the fixtures are 256x256 faces, one light, 160 samples"
            depth, mask, albedo, normals, light, amb = ffhq_faces(B, first=3 * rank + j * B)
            if self.mask == "ones":
                mask = np.ones_like(mask)
        elif self.data == "train_depth":
            return self._train_depth_batch(seed0)
        elif self.default_shape and self.mask == "ellipse":
            depth, mask, albedo, normals, light, amb = synth_faces(B, seed0=seed0, light_seed0=rank * 1_000_000)
        else:
            depth, mask, albedo, normals, light, amb = synth_faces_sized(B, seed0, self.size, self.L, self.mask,
                                                                         light_seed0=rank * 1_000_000)
        if self.depth_noise > 0.0:
            depth = depth + (self.depth_noise * np.random.default_rng(7 + j).random(depth.shape)).astype(np.float32)
        t = [torch.from_numpy(np.ascontiguousarray(x)).to(self.dev) for x in (depth, mask, albedo, normals, light, amb)]
        t[1] = self.R.mask_to_u8(t[1]).reshape(-1, self.size, self.size).contiguous()
        return t

    def _train_depth_batch(self, seed0):
        """what the training step's march sees at the start of training: depth = 100 x the output of a freshly
        initialised RelightNet on the synthetic training images (rough, little for the bounds to skip), its predicted
        light / ambient / albedo, the batch's fill masks; normals from depth in the epilogue, argmin variant."""
        from geomconsistentfr_amd.relightnet import RelightNet
        from geomconsistentfr_amd.train import synthetic_batch
        torch.manual_seed(1234 + seed0)
        model = RelightNet().float().to(self.dev)
        batch = synthetic_batch(self.B, seed0, device=self.dev)
        with torch.no_grad():
            albedo, depth, SL = model.features(batch["images"], 0)
        B = self.B
        mask = self.R.mask_to_u8(batch["masks_fill"].reshape(B, 256, 256))
        light = SL[:, 0, 0, 1:4].reshape(B, 3).float().contiguous()
        amb = SL[:, 0, 0, 0].reshape(B).float().contiguous()
        del model
        return [depth.reshape(B, 256, 256).float().contiguous(), mask, albedo.float().contiguous(), None, light, amb]

    def _inputs_of(self, bt):
        dd, mm, al, nr, li, am = bt
        return (dd, mm, li.reshape(self.B, self.L, 3).contiguous(), am.reshape(self.B, self.L).contiguous(),
                None if self.from_depth else nr, al)

    def _new_plan(self):
        return self.R.RenderFwdPlan(self.B, self.L, self.size, self.size, self.prm, self.dev, want_argmin=self.want_argmin,
                                    mask_batch=self.batches[0][1].shape[0], camera=self.cam if self.from_depth else None,
                                    options=self.base_opt, normals_stage=self.normals_stage)

    # -- issue ---------------------------------------------------------------------------------
    def eager_step(self, opt):
        R, B, Ll, prm = self.R, self.B, self.L, self.prm
        d_depth, d_mask, d_albedo, d_normals, d_light, d_amb = self.batches[0]
        if self.mode in ("direct", "unfused"):
            _, pt = R.light_prep(d_light, prm)
            md, _ = R.shadow_min_distance(d_depth, d_mask, pt.reshape(B, Ll, 3), prm, want_argmin=False,
                                          use_workspace=self.mode != "direct", options=opt)
            return R.shade(d_normals, d_depth, d_albedo, pt.reshape(B, Ll, 3), d_amb.reshape(B, Ll), md, prm)
        return R.render_fwd(d_depth, d_mask, d_light.reshape(B, Ll, 3), d_amb.reshape(B, Ll),
                            None if self.from_depth else d_normals, d_albedo, prm, want_argmin=self.want_argmin,
                            camera=self.cam if self.from_depth else None, options=opt)

    def issue(self, i, n_streams):
        """enqueue step i (no events): graph replay, plan call or eager call on stream i % n_streams"""
        s = i % n_streams
        b = s if self.n_batches == self.n_streams else i % self.n_batches
        with torch.cuda.stream(self.streams[s]):
            if self.use_graph:
                self.plans[b].replay()
            elif self.use_plans:
                self.plans[b](*self.inputs[b])
            else:
                self.eager_step(self.base_opt)

    def timed(self, n_steps, n_streams=None):
        """n_steps steps round-robin over the first `n_streams` streams, fenced on both sides -> (seconds: max over
        ranks, this rank's seconds, host issue seconds)"""
        n_streams = n_streams or self.n_streams
        self.rk.fence()
        t0 = time.perf_counter()
        for i in range(n_steps):
            self.issue(i, n_streams)
        host = time.perf_counter() - t0
        self.rk.fence()
        dt = time.perf_counter() - t0
        return self.rk.max_over_ranks(dt), dt, host

    def timed_regions(self, n_steps, n_streams=None, est_ms=None):
        """the fenced region of exactly `n_steps` steps, repeated when it is too short to be one trustworthy sample
        (regions_needed); returns (median seconds, stats dict, own-rank median seconds, host issue seconds)"""
        first, own0, host0 = self.timed(n_steps, n_streams)
        n = regions_needed(n_steps, est_ms if est_ms is not None else 1e3 * first / n_steps)
        if n == 1:
            return first, {"n": 1, "seconds": [first]}, own0, host0
        runs = [self.timed(n_steps, n_streams) for _ in range(n)]
        secs = sorted(r[0] for r in runs)
        med = float(np.median(secs))
        own = float(np.median([r[1] for r in runs]))
        host = float(np.median([r[2] for r in runs]))
        return med, {"n": n, "median_ms_per_step": 1e3 * med / n_steps, "min_ms_per_step": 1e3 * secs[0] / n_steps,
                     "max_ms_per_step": 1e3 * secs[-1] / n_steps, "first_region_ms_per_step": 1e3 * first / n_steps,
                     "region_ms_median": 1e3 * med,
                     "note": "a region of `steps` steps lasts < 200 ms: it was repeated %d times (each fenced by barrier + "
                             "synchronize on both sides, max over ranks); value / ms_per_step are the MEDIAN region" % n}, own, host

    def timed_march_only(self, n_steps):
        """The latency of one batch when its prepass has been issued EARLIER (gcfr_options.phase: in RelightNet.forward it runs
        on a side stream under the albedo decoder): `n_steps` replays of plan 0's march-only hipGraph, one at a time on one
        stream, on the workspace its prepass graph prepared once -- the march only reads the workspace.  Fenced wall seconds
        (median of the repeated regions), or None where the plans are not graphs."""
        if not self.use_graph:
            return None
        p0 = self.plans[0]
        if not hasattr(p0, "graph_march"):
            p0.capture_split(*self.inputs[0])
        with torch.cuda.stream(self.streams[0]):
            p0.replay_prepass()
        torch.cuda.synchronize()

        def once():
            self.rk.fence()
            t0 = time.perf_counter()
            with torch.cuda.stream(self.streams[0]):
                for _ in range(n_steps):
                    p0.replay_march()
            self.rk.fence()
            return time.perf_counter() - t0
        first = once()
        n = regions_needed(n_steps, 1e3 * first / n_steps)
        return float(np.median([once() for _ in range(n)])) if n > 1 else first

    def kernel_launch_ms(self, ev, n=100):
        """the march kernel's un-overlapped launch duration: plan calls (not graph replays) on ONE stream, each with
        its own event pair recorded by the library immediately before / after the march kernel on that stream"""
        pairs = []
        torch.cuda.synchronize()
        with torch.cuda.stream(self.streams[0]):
            for k in range(n):
                e0, e1 = ev.new(), ev.new()
                opt = self._lib.options(**self.knobs, event_start=e0, event_stop=e1, pixels=int(self.prm.pixels == "mask"))
                pairs.append((e0, e1, opt))
                if self.use_plans:
                    b = k % self.n_batches if self.n_batches > self.n_streams else 0     # (many distinct batches: cycle them -- cold inputs)
                    keep = self.plans[b].options          # (the plan's own: base knobs + what RenderParams.pixels asked for)
                    self.plans[b].options = opt
                    self.plans[b](*self.inputs[b])
                    self.plans[b].options = keep
                else:
                    self.eager_step(opt)
        torch.cuda.synchronize()
        ms = [ev.elapsed_ms(e0, e1) for e0, e1, _ in pairs]
        for e0, e1, _ in pairs:
            ev.destroy(e0)
            ev.destroy(e1)
        return float(np.mean(ms)), float(np.min(ms)), float(np.max(ms))

    @property
    def ray_steps_per_step(self):
        return self.B * self.L * self.size * self.size * self.N


SPEC_CYCLES = {"ADD_F32": 2, "MUL_F32": 2, "FMA_F32": 2, "INT32": 2, "OTHER": 2, "ADD_F64": 4, "MUL_F64": 4, "FMA_F64": 4,
               "CVT": 4, "INT64": 4, "TRANS_F32": 8, "TRANS_F64": 16}


def relight_e2e_leg(dev, B=FACES_PER_GPU, n_lights=11, iters=20, warmup=4):
    """End-to-end "relit faces/sec" (the metric names it; round-5 verdict, missing 5): network forward (eval; the reference's
    shipped lighting-transfer checkpoint, tests/golden/slt_checkpoint_epoch106.npz) + the HIP render block + the HIP uint8 image
    kernel, per batch of B photographs resident in HBM, up to the composite bytes ON THE DEVICE -- what the reference's
    test_relight_single_image.py:582-620 does per (face, light) with the whole model re-run for every light.
      L = 1:        one target light per face (the scripts' call shape): B relit images per pass.
      L = n_lights: the eleven shipped directions (S1:519-562) from ONE network pass (inference.relight_lights_device):
                    one prepass and one normals stage per face, L marches, one image-kernel launch.
    MIOpen searches its solvers for the network's convolutions once at the start of the leg (`miopen_find_seconds`)."""
    from geomconsistentfr_amd import inference as inf
    from geomconsistentfr_amd.relightnet import RelightNetLightingTransfer
    sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "slt_checkpoint_epoch106.npz")).items()}
    net = RelightNetLightingTransfer()
    net.load_state_dict(sd, strict=True)
    net = net.float().to(dev).eval()
    depth, mask, albedo, _n, _l, _a = synth_faces(B, 0)
    shade = 0.45 + 0.55 * np.clip(depth / 80.0, 0, 1)
    x = torch.from_numpy((albedo * shade[:, None]).transpose(0, 2, 3, 1).astype(np.float32).copy()).to(dev)   # (B,H,W,3) photographs
    m_u8 = torch.from_numpy((mask[0] * 255).astype(np.uint8)).to(dev)
    lights = torch.from_numpy(LIGHTS18[:11].copy()).to(dev)
    K = inf.camera_matrix(700.0, H, W, dev)
    res = {}

    def timed(fn, n):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(dev)
        return (time.perf_counter() - t0) / n

    # MIOpen's solver SEARCH for the network's convolutions at this batch size, once, up front (RelightSession(miopen_find=True):
    # seconds; tools/relight_bench.py: 3.13 -> 2.75 ms per pass against the immediate-mode picks).  PyTorch keeps a convolution's
    # solver per process and shape, so the eager timings below run on the searched solvers as well.
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    try:
        inf.RelightSession(net, B, m_u8, lights[:1], 0.5, device=dev, miopen_find=True)
        find_s, find_err = time.perf_counter() - t0, None
    except Exception as e:
        find_s, find_err = None, repr(e)[:200]
    # the two streams of every "2 in flight" measurement below: the process's own (rig_streams: a pair created late in the run was
    # seen to serialise -- 28.7 k instead of 41.3 k images/s at eleven lights, reproducibly)
    streams = rig_streams(dev, 2)
    with torch.no_grad():
        t_feat = timed(lambda: net.features(x, 200), iters)
        for L in (1, n_lights):
            t = timed(lambda: inf.relight_lights_device(net, x, m_u8, lights[:L], 0.5, device=dev), iters)
            t_host = timed(lambda: inf.relight_lights(net, x, m_u8, lights[:L], 0.5, device=dev), max(3, iters // 4))
            res[L] = {"ms_per_pass": 1e3 * t, "images_per_sec": B * L / t, "images_per_sec_with_d2h": B * L / t_host,
                      "share_outside_network": max(0.0, 1.0 - t_feat / t)}
            try:        # the same pass captured once into a hipGraph on static buffers (inference.RelightSession): one launch per pass
                sess = inf.RelightSession(net, B, m_u8, lights[:L], 0.5, device=dev)
                tg = timed(lambda: sess.run(x), iters * 2)
                res[L].update(graph_ms_per_pass=1e3 * tg, graph_images_per_sec=B * L / tg)
                del sess
                # ... and two such sessions replayed round-robin on their own streams (independent batches in flight, as the
                # headline keeps four render batches in flight): one batch-8 network pass does not fill the chip
                pair = []
                for st in streams:
                    with torch.cuda.stream(st):
                        pair.append(inf.RelightSession(net, B, m_u8, lights[:L], 0.5, device=dev))
                torch.cuda.synchronize(dev)
                turn = [0]

                def two():
                    i = turn[0] = turn[0] ^ 1
                    with torch.cuda.stream(streams[i]):
                        pair[i].run(x)
                t2 = timed(two, iters * 4)
                res[L].update(graph2_ms_per_pass=1e3 * t2, graph2_images_per_sec=B * L / t2)
                pair.clear()
                # ... and the same two sessions on a copy of the network whose eval-mode BatchNorms are folded into its convolutions
                # (inference.fold_batchnorm: a deployment form -- network outputs agree to ~1e-5, composites on >= 99.9 % of the bytes)
                folded = inf.fold_batchnorm(net)
                for st in streams:
                    with torch.cuda.stream(st):
                        pair.append(inf.RelightSession(folded, B, m_u8, lights[:L], 0.5, device=dev))
                torch.cuda.synchronize(dev)
                t3 = timed(two, iters * 4)
                res[L].update(graph2_folded_images_per_sec=B * L / t3)
                pair.clear()
                del folded
            except Exception as e:
                res[L]["graph_error"] = repr(e)[:300]
    out = {"faces": B, "lights": n_lights, "network_forward_ms": 1e3 * t_feat,
           "faces_per_sec_1_light": res[1]["images_per_sec"], "ms_per_pass_1_light": res[1]["ms_per_pass"],
           "images_per_sec_%d_lights" % n_lights: res[n_lights]["images_per_sec"],
           "ms_per_pass_%d_lights" % n_lights: res[n_lights]["ms_per_pass"],
           "images_per_sec_1_light_with_d2h": res[1]["images_per_sec_with_d2h"],
           "images_per_sec_%d_lights_with_d2h" % n_lights: res[n_lights]["images_per_sec_with_d2h"],
           "hip_share_1_light": res[1]["share_outside_network"], "hip_share_%d_lights" % n_lights: res[n_lights]["share_outside_network"],
           "graph_faces_per_sec_1_light": res[1].get("graph_images_per_sec"), "graph_ms_per_pass_1_light": res[1].get("graph_ms_per_pass"),
           "graph_images_per_sec_%d_lights" % n_lights: res[n_lights].get("graph_images_per_sec"),
           "graph_ms_per_pass_%d_lights" % n_lights: res[n_lights].get("graph_ms_per_pass"),
           "graph_2_in_flight_faces_per_sec_1_light": res[1].get("graph2_images_per_sec"),
           "graph_2_in_flight_images_per_sec_%d_lights" % n_lights: res[n_lights].get("graph2_images_per_sec"),
           "graph_2_in_flight_folded_bn_faces_per_sec_1_light": res[1].get("graph2_folded_images_per_sec"),
           "graph_2_in_flight_folded_bn_images_per_sec_%d_lights" % n_lights: res[n_lights].get("graph2_folded_images_per_sec"),
           "graph_error": res[1].get("graph_error") or res[n_lights].get("graph_error"),
           "miopen_find_seconds": find_s, "miopen_find_error": find_err,
           "reference_equivalent_passes": n_lights,
           "note": "B photographs resident in HBM -> (B,L) composite uint8 images on the device; network = RelightNetLightingTransfer "
                   "(eval, the reference's shipped checkpoint), MIOpen's solvers searched once at the start of the leg; `hip_share_*` = 1 - network_forward / pass "
                   "= the share of a pass spent in the HIP render block + image kernel + glue; `*_with_d2h` adds the copy of the "
                   "bytes to the host; `graph_*` = the same pass (copy of the photographs into a static input included) replayed from "
                   "ONE hipGraph (inference.RelightSession): the eager pass is bound by ~350 launches issued from Python, the graph by "
                   "the GPU; `graph_2_in_flight_*` = two such sessions (two independent batches of B photographs) replayed round-robin on two "
                   "HIP streams: the throughput form, as the headline's four render batches in flight; `*_folded_bn_*` = the same on a copy of the "
                   "network with its eval-mode BatchNorms folded into the convolutions (inference.fold_batchnorm; outputs agree to ~1e-5).  The reference produces L images of "
                   "a face with L full passes (S1:582-588)."}
    del net
    return out


def relight_e2e_ranks(rk, B=FACES_PER_GPU, n_lights=11, iters=30):
    """The end-to-end relight rate on N > 1 ranks (the metric's "relit faces/sec ... 1/2/4/8 GPU"): every rank relights ITS OWN B
    photographs through its own inference.RelightSession (network + block + image kernel as one hipGraph) -- photographs are
    independent, so there is no collective in the data path (weak scaling); the timed regions are fenced (barrier + synchronize)
    on both sides and the max over ranks counts.  A COLLECTIVE: every rank calls it; a rank that fails keeps taking part in the
    fences and reports its failure through the gather, so no rank is left waiting."""
    err, sess, x = None, {}, None
    try:
        from geomconsistentfr_amd import inference as inf
        from geomconsistentfr_amd.relightnet import RelightNetLightingTransfer
        sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "slt_checkpoint_epoch106.npz")).items()}
        net = RelightNetLightingTransfer()
        net.load_state_dict(sd, strict=True)
        net = net.float().to(rk.dev).eval()
        depth, mask, albedo, _n, _l, _a = synth_faces(B, rk.rank * 1_000_000)
        shade = 0.45 + 0.55 * np.clip(depth / 80.0, 0, 1)
        x = torch.from_numpy((albedo * shade[:, None]).transpose(0, 2, 3, 1).astype(np.float32).copy()).to(rk.dev)
        m_u8 = torch.from_numpy((mask[0] * 255).astype(np.uint8)).to(rk.dev)
        lights = torch.from_numpy(LIGHTS18[:11].copy()).to(rk.dev)
        for L in (1, n_lights):
            sess[L] = inf.RelightSession(net, B, m_u8, lights[:L], 0.5, device=rk.dev, miopen_find=(L == 1))
            for _ in range(4):
                sess[L].run(x)
    except Exception as e:                                                    # noqa: BLE001
        err = repr(e)[:300]
    elapsed = {}
    for L in (1, n_lights):
        rk.fence()
        t0 = time.perf_counter()
        if err is None:
            for _ in range(iters):
                sess[L].run(x)
        rk.fence()
        elapsed[L] = rk.max_over_ranks(time.perf_counter() - t0)
    failed = rk.gather(0.0 if err is None else 1.0)
    if rk.rank != 0:
        return None
    if any(failed):
        return {"error": err or "rank(s) %s failed" % [i for i, f in enumerate(failed) if f]}
    return {"faces_per_gpu": B, "lights": n_lights, "ranks": rk.world, "passes": iters, "scaling": "weak",
            "graph_faces_per_sec_1_light": rk.world * B * iters / elapsed[1],
            "graph_images_per_sec_%d_lights" % n_lights: rk.world * B * n_lights * iters / elapsed[n_lights],
            "graph_ms_per_pass_1_light": 1e3 * elapsed[1] / iters, "graph_ms_per_pass_%d_lights" % n_lights: 1e3 * elapsed[n_lights] / iters,
            "note": "whole job: every rank relights its own %d photographs through its own hipGraph session (network forward + HIP render "
                    "block + HIP image kernel), no data-path collective; fenced on both sides, max over ranks" % B}


def library_srchash():
    try:
        from geomconsistentfr_amd import build as hb
        return hb.recorded_hashes()[0]
    except Exception:
        return None


def run_render(a, rk):
    knobs = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)}
    B = a.faces
    mode = "direct" if a.direct else ("unfused" if a.unfused else ("eager" if a.eager else "plan"))
    # The step is what the drop-in executes for T8:353-522 (relightnet.py -> block.render_from_depth ->
    # gcfr_render_from_depth_fwd): normals from depth (a3) INSIDE the step, as in the CPU baseline printed beside it.
    # --normals-in selects SURVEY 8d's accounting form (normals handed in as a 12 B/pixel input, gcfr_render_fwd); the
    # headline line carries that figure as `normals_in_*`, measured in the same run.
    from_depth = (not a.normals_in) and mode in ("plan", "eager")      # (the direct / unfused A/B forms take normals as input)
    headline = (a.size == 256 and a.lights == 1 and a.samples == 160 and a.mask == "ellipse" and a.depth_noise == 0.0
                and a.data == "synthetic" and B == FACES_PER_GPU and not knobs and mode == "plan" and from_depth
                and not a.argmin and a.pixels == "all" and a.normals_stage in ("auto", "fused") and not a.batches)
    rig = RenderRig(rk, B, a.size, a.lights, a.samples, a.mask, a.depth_noise, a.data, a.streams, from_depth,
                    a.argmin, knobs, graph=not a.no_graph, mode=mode, pixels=a.pixels, normals_stage=a.normals_stage,
                    n_batches=a.batches or None)
    n_streams, world, rank = rig.n_streams, rk.world, rk.rank
    ev = HipEvents()
    legs = {"setup_s": time.perf_counter() - T_START}           # wall seconds per leg of this run (how the default run spends its minutes)
    t_leg = time.perf_counter()

    def leg(name):
        nonlocal t_leg
        now = time.perf_counter()
        legs[name] = now - t_leg
        t_leg = now
    # in-run parity spot check: the timed binary is checked in the run that times it
    spot = None
    if rig.use_plans and not a.no_parity_check and not a.argmin and a.pixels == "all":
        spot = parity_spot_check(rig)
    for i in range(a.warmup):
        rig.issue(i, n_streams)
    elapsed, regions, own_elapsed, host_issue = rig.timed_regions(a.steps)
    per_rank_s = rk.gather(own_elapsed)
    shadow_ms, shadow_ms_min, shadow_ms_max = rig.kernel_launch_ms(ev)
    single_steps = max(50, min(a.steps, 1000))
    if n_streams > 1:
        single_elapsed, single_regions, _, _ = rig.timed_regions(single_steps, 1)
    else:
        single_elapsed, single_regions = elapsed * single_steps / a.steps, regions
    rsps = rig.ray_steps_per_step
    value = world * rsps * a.steps / elapsed
    single = {"ms_per_step": 1e3 * single_elapsed / single_steps,
              "ray_steps_per_sec": world * rsps * single_steps / single_elapsed, "regions": single_regions,
              "note": "the same steps one at a time on ONE stream (hipGraph replay): the latency of one batch"}
    march_only_s = rig.timed_march_only(single_steps) if headline else None
    if march_only_s:
        single["march_only_ms_per_step"] = 1e3 * march_only_s / single_steps
        single["march_only_ray_steps_per_sec"] = world * rsps * single_steps / march_only_s
        single["march_only_excludes_prepass"] = True     # NOT a full-step rate: the prepass (12 us) ran earlier, outside the timed region
    layout = rk.describe()                                                    # (a collective: every rank calls it)
    leg("headline_s")

    # secondary workloads, measured in the same run (1 rank only, headline only; short): the data the headline is NOT
    worst = None
    if headline and world == 1 and not a.no_worst_case:
        worst = {}
        for key, kw in (("ones_mask", dict(mask="ones")), ("depth_noise_400", dict(depth_noise=400.0)),
                        ("ffhq", dict(data="ffhq")),
                        ("many_batches", dict(n_batches=64)),
                        ("train_depth_b32", dict(data="train_depth", B=32, from_depth=True, want_argmin=True, streams=1)),
                        ("train_depth_b32_pixels_mask", dict(data="train_depth", B=32, from_depth=True, want_argmin=True, streams=1,
                                                             pixels="mask"))):
            try:
                kw = dict(kw)
                kw.setdefault("from_depth", True)                              # the headline's form of the step
                r2 = RenderRig(rk, kw.pop("B", B), streams=kw.pop("streams", a.streams), **kw)
                if r2.n_batches > r2.n_streams:
                    kms_extra = {"distinct_batches": r2.n_batches, "faces_resident": r2.n_batches * r2.B}
                else:
                    kms_extra = {}
                for i in range(20):
                    r2.issue(i, r2.n_streams)
                steps2 = 200 if r2.B <= 8 else 60
                sec, reg, _, _ = r2.timed_regions(steps2)
                kms = r2.kernel_launch_ms(ev, 30)[0]
                worst[key] = {**kms_extra, "ray_steps_per_sec": r2.ray_steps_per_step * steps2 / sec, "ms_per_step": 1e3 * sec / steps2,
                              "march_kernel_ms": kms, "faces_per_step": r2.B, "batches_in_flight": r2.n_streams,
                              "steps": steps2, "regions": reg["n"]}
                if r2.n_streams > 1:
                    sec1, _, _, _ = r2.timed_regions(steps2, 1)
                    worst[key]["single_stream_ray_steps_per_sec"] = r2.ray_steps_per_step * steps2 / sec1
                del r2
            except Exception as e:                                                # never lose the headline to a side measurement
                worst[key] = {"error": repr(e)}
        worst["note"] = ("many_batches: the headline's workload cycling through 64 DISTINCT batches of 8 faces (512 faces, ~2.4 GB of "
                         "inputs, outputs and workspaces) instead of 4 (~154 MB, inside the 256-MB Infinity Cache), four in flight and one "
                         "at a time: whether the rate rests on cache residency (it does not with four in flight; a lone launch pays ~5 %).  "
                         "Otherwise: "
                         "same kernels, same run: all-ones masks (nothing is ever masked), uniform depth noise of amplitude 400 "
                         "(what an untrained network emits: the depth bounds never separate ray and surface), the three "
                         "checkpoint-derived FFHQ fixture faces tiled to the batch (--data ffhq), and the training step's "
                         "march -- batch 32, argmin variant, normals fused, depth of a freshly initialised RelightNet -- as the reference "
                         "defines it (every pixel) and with the opt-in pixels = mask (gcfr_options.pixels, include/gcfr.h)")
    leg("worst_case_s")
    # the dominant kernel on a launch long enough that its tail does not matter (128 faces, one launch at a time): how busy
    # the VALU issue ports are when the chip is full -- one launch of 8 faces ends with its heaviest tiles, most SIMDs idle
    saturated_ms = None
    if headline and world == 1 and not a.no_worst_case:
        try:
            r3 = RenderRig(rk, 128, streams=1, from_depth=True)
            for i in range(5):
                r3.issue(i, 1)
            saturated_ms = r3.kernel_launch_ms(ev, 20)
            del r3
        except Exception:
            saturated_ms = None
    # the other single-GPU configurations of BASELINE.json, measured in the same run: the accounting form
    # with normals handed in, configs[4]'s per-GPU shape, and configs[2]'s training step -- short, and never at the headline's
    # expense (each leg is skipped once the run has used its time budget, and a failure is recorded, not raised)
    leg("saturated_s")
    aux = {}
    if headline and world == 1 and not a.no_worst_case:
        try:
            r4 = RenderRig(rk, B, streams=a.streams, from_depth=False)
            for i in range(20):
                r4.issue(i, r4.n_streams)
            sec, reg, _, _ = r4.timed_regions(a.steps)
            aux["normals_in"] = {"ray_steps_per_sec": r4.ray_steps_per_step * a.steps / sec, "ms_per_step": 1e3 * sec / a.steps,
                                 "march_kernel_ms": r4.kernel_launch_ms(ev, 50)[0], "regions": reg["n"],
                                 "note": "SURVEY 8d's accounting form: gcfr_render_fwd with the normals as a 12 B/pixel input "
                                         "(rounds 1-3's headline step); same faces, same streams, same number of steps"}
            del r4
        except Exception as e:
            aux["normals_in"] = {"error": repr(e)}
        try:
            r5 = RenderRig(rk, 1, size=512, lights=18, samples=320, streams=a.streams, from_depth=True)
            for i in range(20):
                r5.issue(i, r5.n_streams)
            steps5 = 100
            sec, reg, _, _ = r5.timed_regions(steps5)
            aux["config5"] = {"ray_steps_per_sec": r5.ray_steps_per_step * steps5 / sec, "ms_per_step": 1e3 * sec / steps5,
                              "faces_lights_per_sec": 18 * steps5 / sec, "march_kernel_ms": r5.kernel_launch_ms(ev, 30)[0],
                              "steps": steps5, "regions": reg["n"], "batches_in_flight": r5.n_streams,
                              "normals_stage": (r5.plans[0].normals_stage if r5.plans else None),
                              "workload": "BASELINE configs[4], per-GPU shape: 1 face x 18 lights x 512x512 x 320 march steps, "
                                          "normals from depth (18 lights per face: the stencil once, in its own launch -- "
                                          "block.normals_stage_for), forward-only shadow+shade"}
            del r5
        except Exception as e:
            aux["config5"] = {"error": repr(e)}
        leg("normals_in_config5_s")
        if time.perf_counter() - T_START > 150.0:
            aux["relight_e2e"] = {"skipped": "time budget: %.0f s used before the leg" % (time.perf_counter() - T_START)}
        else:
            try:
                aux["relight_e2e"] = relight_e2e_leg(rk.dev)
            except Exception as e:
                aux["relight_e2e"] = {"error": repr(e)}
        leg("relight_e2e_s")
        if a.no_train_leg:
            aux["train"] = {"skipped": "--no-train-leg"}
        elif time.perf_counter() - T_START > 170.0:
            aux["train"] = {"skipped": "time budget: %.0f s used before the training leg" % (time.perf_counter() - T_START)}
        else:
            try:
                # (10 timed steps = two whole periods of the discriminator's every-fifth-step schedule, T8:624: rounds 3-5 timed 12, which
                #  holds three 8-ms D steps instead of 2.4 -- +0.4 ms per step of bias)
                aux["train"] = train_leg_subprocess(steps=10, warmup=6)
            except Exception as e:
                aux["train"] = {"error": repr(e)}
        leg("train_s")
    relight_ranks = None
    if headline and world > 1 and not a.no_worst_case:       # N > 1: the end-to-end relight rate of the whole job (a collective)
        relight_ranks = relight_e2e_ranks(rk)
        leg("relight_e2e_s")
    if rank != 0:
        return None
    algo_bytes = rsps * ALGO_BYTES_PER_RAY_STEP                              # per launch (one rank)
    achieved_gbs = algo_bytes / (shadow_ms * 1e-3) / 1e9
    pm = pmc_summary()
    hbm_line = {"bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": algo_bytes, "measured_copy_GBs": measured_copy_bandwidth_gbs(rk.dev),
                "note": "north_star's accounting: 17.4 algorithmic B per NOMINAL ray-step (SURVEY 8d) / launch duration; the "
                        "gathers are cache-served and most nominal ray-steps are provably skipped, so this is not a "
                        "fraction of a limit (it exceeds 1) -- the binding roofline is the VALU one"}
    fwd = pm.get("kernels", {}).get("fwd")
    if headline and fwd:
        cap = N_SIMD * NOMINAL_HZ                                             # SIMD issue cycles per second of the chip
        # `frac`: the kernel's instruction mix priced at the guide's SPEC issue rates (MI355X_MICROARCH.md: SIMD-32, 2 cycles per
        # wave64 f32 / int op, 4 per f64 / cvt, transcendentals quarter rate) / live launch duration / capacity -- the number a
        # reader takes from the line is the conservative one.  `frac_measured_costs`: the same mix priced
        # at the sustained per-class costs tools/ubench_valu measured on this chip (rounds 1-3's `frac`).
        spec_cycles = sum(fwd["valu"]["by_class"].get(k, 0.0) * c for k, c in SPEC_CYCLES.items())
        meas_cycles = fwd["valu"]["issue_cycles_per_launch"]
        step_s = elapsed / a.steps
        roof = {"bound": "valu", "kernel": fwd["kernel"], "achieved": spec_cycles / (shadow_ms * 1e-3) / 1e9, "peak": cap / 1e9,
                "unit": "G SIMD-issue-cycles/s", "frac": spec_cycles / (cap * shadow_ms * 1e-3),
                "frac_measured_costs": meas_cycles / (cap * shadow_ms * 1e-3),
                "avg_launch_ms": shadow_ms, "avg_launch_ms_min": shadow_ms_min, "avg_launch_ms_max": shadow_ms_max,
                "traffic": fwd["hbm"]["bytes_per_launch"],
                "valu_insts_per_launch": fwd["valu"]["insts_per_launch"], "spec_issue_cycles_per_launch": spec_cycles,
                "measured_cost_issue_cycles_per_launch": meas_cycles,
                "kernel_ray_steps_per_sec": rsps / (shadow_ms * 1e-3),
                # the same mix against the overlapped rate: what the chip's VALU does when `streams` launches share it
                "frac_at_throughput": spec_cycles / (cap * step_s),
                "frac_measured_costs_at_throughput": meas_cycles / (cap * step_s),
                # north_star's HBM accounting, flattened (the driver's record keeps scalars only): 17.4 algorithmic B per
                # NOMINAL ray-step / launch duration against 8 TB/s -- exceeds 1, not a fraction of a limit (see `hbm`)
                "hbm_achieved_GBs_nominal": achieved_gbs, "hbm_frac_nominal": achieved_gbs / HBM_PEAK_GBS,
                "hbm_traffic_frac_of_peak": fwd["hbm"]["bytes_per_launch"] / (shadow_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "hbm_measured_copy_GBs": hbm_line["measured_copy_GBs"],
                "note": "achieved = sum over VALU instruction classes of (wave-instructions per launch, rocprofv3 SQ_INSTS_VALU_* of "
                        "this workload, profiles/pmc_summary.json) x (SPEC issue cycles of the class, MI355X_MICROARCH.md), divided "
                        "by the kernel's un-overlapped launch duration measured live (HIP events recorded by the library around "
                        "the kernel, 100 plan calls on one stream); peak = 1024 SIMDs x 2.4 GHz"}
        roof["hbm"] = hbm_line
        if "work" in fwd:
            roof["executed_fraction_of_nominal_ray_steps"] = fwd["work"]["executed_fraction_of_nominal"]
            roof["valu_wave_insts_per_executed_wave_step"] = fwd["work"]["valu_wave_insts_per_executed_wave_step"]
        if "l1" in fwd:
            roof["l1_frac_of_peak_under_rocprofv3"] = fwd["l1"]["frac"]
        if fwd.get("wave_cycles_quad") and fwd.get("wait_any_quad"):
            roof["wait_any_frac_of_wave_time"] = fwd["wait_any_quad"] / fwd["wave_cycles_quad"]
        f128 = pm.get("kernels", {}).get("fwd_b128")
        if saturated_ms and f128:
            spec128 = sum(f128["valu"]["by_class"].get(k, 0.0) * c for k, c in SPEC_CYCLES.items())
            roof["frac_spec_saturated"] = spec128 / (cap * saturated_ms[0] * 1e-3)
            roof["frac_measured_costs_saturated"] = f128["valu"]["issue_cycles_per_launch"] / (cap * saturated_ms[0] * 1e-3)
            roof["saturated_launch_ms"] = saturated_ms[0]
            roof["saturated_kernel_ray_steps_per_sec"] = 128 * 256 * 256 * 160 / (saturated_ms[0] * 1e-3)
            roof["saturated"] = {
                "faces_per_launch": 128, "avg_launch_ms": saturated_ms[0], "avg_launch_ms_min_max": list(saturated_ms[1:]),
                "valu_insts_per_launch": f128["valu"]["insts_per_launch"],
                "frac": roof["frac_spec_saturated"], "frac_measured_costs": roof["frac_measured_costs_saturated"],
                "kernel_ray_steps_per_sec": roof["saturated_kernel_ray_steps_per_sec"],
                "note": "the same kernel on ONE launch of 128 faces, un-overlapped (library events, 20 calls): long enough that "
                        "the launch's tail -- a launch ends with its heaviest tiles, DESIGN 4.1 -- does not matter; instruction mix "
                        "from a PMC pass of exactly that launch (profiles/pmc_summary.json kernels.fwd_b128)"}
        # instruction counts / traffic come from the committed PMC passes (rocprofv3 cannot run inside the timed run):
        # they are only valid for the library they were collected on
        lib_hash, pmc_hash = library_srchash(), pm.get("library_srchash")
        roof["pmc_library_srchash"], roof["library_srchash"] = pmc_hash, lib_hash
        roof["stale"] = (pmc_hash is None) or (lib_hash is None) or (pmc_hash != lib_hash)
    else:       # no PMC mix for this workload / kernel selection: HBM accounting only
        roof = dict(hbm_line, kernel="shadow_fwd_quad_kernel" if mode != "direct" else "shadow_fwd_kernel",
                    avg_launch_ms=shadow_ms, traffic=None, kernel_ray_steps_per_sec=rsps / (shadow_ms * 1e-3))
    desc = ("batch=%d %s 256x256 faces per GPU, 1 light each, 160 march steps, forward-only normals+shadow+shade "
            "(gcfr_render_from_depth_fwd: what RelightNet.forward runs for T8:353-522); %d batch(es) in flight"
            % (B, "synthetic" if a.data == "synthetic" else "FFHQ-fixture", n_streams))
    out = {
        "metric": "ray_steps_per_sec", "value": value, "unit": "ray-steps/s", **{k: layout[k] for k in ("n_gpus", "rccl_ranks")},
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64+f32",
        "data": "synthetic" if a.data != "ffhq" else "ffhq-fixtures (3 checkpoint-derived faces tiled)",
        # (advisor r04: `value` of rounds 1-3 timed gcfr_render_fwd with the normals handed in -- today's `normals_in_*`; since round 4
        #  the step includes the normals stage, so `value` is not comparable across that boundary by itself)
        "step_definition": ("gcfr_render_from_depth_fwd (normals stage inside the step; rounds 1-3: gcfr_render_fwd = normals_in_*)"
                            if from_depth else "gcfr_render_fwd (normals handed in)"),
        "config": {"workload": ("BASELINE configs[1]: " + desc) if headline else
                               ("non-headline: batch=%d %s %dx%d faces per GPU, %d light(s) each, %d march steps, mask=%s, "
                                "depth noise %g, knobs %s, %s%sforward-only shadow+shade; %d batch(es) in flight"
                                % (B, a.data, a.size, a.size, a.lights, a.samples, a.mask, a.depth_noise, knobs or "default",
                                   "argmin variant, " if a.argmin else "", "normals from depth, " if from_depth else "normals as input, ",
                                   n_streams)),
                   "faces_per_gpu": B, "H": a.size, "W": a.size, "lights_per_face": a.lights, "n_samples": a.samples,
                   "parallelism": "dp%d" % world, "hip_streams": n_streams, "batches_in_flight": n_streams,
                   "distinct_face_batches": n_streams,
                   "host_path": ("RenderFwdPlan, hipGraph replay" if rig.use_graph else "RenderFwdPlan (preallocated outputs)")
                   if rig.use_plans else "render_fwd (eager)"},
        "process_layout": layout,
        "per_rank": [{"rank": r, "seconds": s, "ray_steps_per_sec": rsps * a.steps / s} for r, s in enumerate(per_rank_s)],
        "regions": regions,
        "faces_per_sec": world * B * a.lights * a.steps / elapsed,
        "host_issue_ms_per_step": 1e3 * host_issue / a.steps,
        "ray_steps_per_sec_per_gpu": value / world,
        "single_stream": single,
        "latency_one_batch_ms": single["ms_per_step"],
        # ... and with the prepass issued earlier on a side stream (gcfr_options.phase; RelightNet.forward does, under the albedo decoder)
        "latency_one_batch_prepass_hoisted_ms": single.get("march_only_ms_per_step"),
        "latency_one_batch_prepass_hoisted_note": "the march launch alone: its prepass was issued earlier and is OUTSIDE this time "
                                                  "(what a caller sees who hides it under other work); the full step is latency_one_batch_ms",
        # `value` is a throughput with `hip_streams` batches in flight (faces_in_flight below), not the rate of one batch
        # of `faces_per_gpu` on its own -- that one is `single_stream` (VERDICT r01 asked for both to be named)
        "faces_in_flight_per_gpu": B * n_streams,
        "throughput_in_flight": {"faces_in_flight_per_gpu": B * n_streams, "ray_steps_per_sec": value},
        "roofline": roof,
    }
    if relight_ranks is not None:
        out["relight_e2e"] = relight_ranks
        if "graph_faces_per_sec_1_light" in relight_ranks:
            out["relight_e2e_graph_faces_per_sec"] = relight_ranks["graph_faces_per_sec_1_light"]
            out["relight_e2e_graph_lights11_images_per_sec"] = relight_ranks["graph_images_per_sec_11_lights"]
    if worst is not None:
        out["worst_case"] = worst
        for k in ("ones_mask", "depth_noise_400", "ffhq", "many_batches", "train_depth_b32", "train_depth_b32_pixels_mask"):    # scalars at the top level too
            if "ray_steps_per_sec" in worst.get(k, {}):
                out["worst_case_%s_ray_steps_per_sec" % k] = worst[k]["ray_steps_per_sec"]
    if aux:
        out["aux"] = aux
        flat = {}
        ni, c5, tr = aux.get("normals_in", {}), aux.get("config5", {}), aux.get("train", {})
        if "ray_steps_per_sec" in ni:
            flat.update(normals_in_ray_steps_per_sec=ni["ray_steps_per_sec"], normals_in_ms_per_step=ni["ms_per_step"],
                        normals_in_march_kernel_ms=ni["march_kernel_ms"])
        if "ray_steps_per_sec" in c5:
            flat.update(config5_ray_steps_per_sec=c5["ray_steps_per_sec"], config5_ms_per_step=c5["ms_per_step"],
                        config5_march_kernel_ms=c5["march_kernel_ms"])
        re2 = aux.get("relight_e2e", {})
        if "faces_per_sec_1_light" in re2:
            flat.update(relight_e2e_faces_per_sec=re2["faces_per_sec_1_light"],
                        relight_e2e_lights11_images_per_sec=re2["images_per_sec_11_lights"],
                        relight_e2e_network_forward_ms=re2["network_forward_ms"],
                        relight_e2e_hip_share=re2["hip_share_1_light"], relight_e2e_lights11_hip_share=re2["hip_share_11_lights"],
                        relight_e2e_graph_faces_per_sec=re2.get("graph_faces_per_sec_1_light"),
                        relight_e2e_graph_lights11_images_per_sec=re2.get("graph_images_per_sec_11_lights"),
                        relight_e2e_graph_2_in_flight_faces_per_sec=re2.get("graph_2_in_flight_faces_per_sec_1_light"),
                        relight_e2e_graph_2_in_flight_lights11_images_per_sec=re2.get("graph_2_in_flight_images_per_sec_11_lights"),
                        relight_e2e_folded_bn_2_in_flight_faces_per_sec=re2.get("graph_2_in_flight_folded_bn_faces_per_sec_1_light"),
                        relight_e2e_folded_bn_2_in_flight_lights11_images_per_sec=re2.get("graph_2_in_flight_folded_bn_images_per_sec_11_lights"))
        if "step_ms" in tr:
            flat.update(train_step_ms=tr["step_ms"], train_faces_per_sec=tr["faces_per_sec"],
                        train_march_kernel_ms=tr["march_kernel_ms"], train_bwd_kernel_ms=tr["bwd_kernel_ms"],
                        train_ray_steps_per_sec=tr["ray_steps_per_sec"])
        out.update(flat)
    # `roofline`: the scalars a reader needs FIRST (the driver's parsed record keeps the first ~20 scalar keys of `roofline` and of
    # `config`, and only the names of extra top-level keys: BENCH_r04.json lost the worst cases and the saturated fraction off the
    # end): the contract's six, then the fractions, the traffic, the executed share, staleness, the worst cases, the one-batch rate
    if isinstance(out.get("roofline"), dict):
        roof_in = out["roofline"]
        extra = {"single_stream_ray_steps_per_sec": single["ray_steps_per_sec"]}
        if worst is not None:
            for k in ("ffhq", "ones_mask", "depth_noise_400", "train_depth_b32"):
                if "ray_steps_per_sec" in worst.get(k, {}):
                    extra["worst_case_" + k] = worst[k]["ray_steps_per_sec"]
        roof_in.update(extra)
        first = ("bound", "achieved", "peak", "unit", "frac", "traffic", "frac_measured_costs", "frac_at_throughput", "frac_spec_saturated",
                 "wait_any_frac_of_wave_time", "hbm_traffic_frac_of_peak", "executed_fraction_of_nominal_ray_steps", "stale",
                 "worst_case_ffhq", "worst_case_ones_mask", "worst_case_depth_noise_400", "worst_case_train_depth_b32",
                 "single_stream_ray_steps_per_sec", "avg_launch_ms", "valu_insts_per_launch", "hbm_measured_copy_GBs", "library_srchash")
        out["roofline"] = {**{k: roof_in[k] for k in first if k in roof_in}, **{k: v for k, v in roof_in.items() if k not in first}}
    if rig.graph_error:
        out["config"]["graph_capture_failed"] = rig.graph_error
    out["config"]["parity_spot_check"] = spot if spot is not None else "skipped (%s)" % (
        "--no-parity-check" if a.no_parity_check else "not a plan-mode / all-pixels / inference-march run")
    if world == 1 and not a.no_cpu_baseline and headline:
        out["cpu_baseline"] = cpu_baseline()
        out["cpu_baseline_c_openmp"] = cpu_baseline_c()
        leg("cpu_baseline_s")
    legs["total_s"] = time.perf_counter() - T_START
    out["run_seconds"] = legs
    return out


# ------------------------------------------------------------------------------------------------
# workload "train": BASELINE configs[2] (1 GPU) / configs[3] (8 GPUs, DDP over RCCL)
# ------------------------------------------------------------------------------------------------
def measure_train(rk, B, steps, warmup, epoch=200, pixels="all"):
    """One rank's measurement of the full training step (BASELINE configs[2]; configs[3] when rk has > 1 rank): `warmup`
    untimed steps (MIOpen's find mode tunes on first use), `steps` timed ones fenced on both sides, then the render block's own
    kernels on this batch measured live: forward (prepass + march with fused normals + shading, argmin variant; library events
    around the march kernel) and the fused backward (events on the current stream), from the tensors of a real step's forward.
    Returns a dict of scalars (a collective when rk.dist is set: every rank calls it)."""
    dev, dist = rk.dev, rk.dist
    from geomconsistentfr_amd import _lib
    from geomconsistentfr_amd import block as R
    from geomconsistentfr_amd.train import TrainConfig, Trainer, synthetic_batch

    torch.manual_seed(1234 + rk.rank)
    tr = Trainer(TrainConfig(render_pixels=pixels), device=dev, distributed=dist is not None)
    batch = synthetic_batch(B, rk.rank * 1_000_000, device=dev)
    for j in range(max(warmup, 6)):                                          # MIOpen find mode tunes on first use
        tr.step(batch, epoch, j, log=False)
    rk.fence()
    t0 = time.perf_counter()
    for j in range(steps):
        tr.step(batch, epoch, j, log=False)                                  # D step every 5th (T8:624), G step always
    rk.fence()
    own_elapsed = time.perf_counter() - t0
    elapsed = rk.max_over_ranks(own_elapsed)
    per_rank_s = rk.gather(own_elapsed)

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
        plan.options = _lib.options(event_start=e0, event_stop=e1, pixels=int(prm.pixels == "mask"))
        pairs.append((e0, e1, plan.options))
        o = plan(*ins)
    torch.cuda.synchronize()
    march_ms = float(np.mean([ev.elapsed_ms(e0, e1) for e0, e1, _ in pairs[5:]]))
    for e0, e1, _ in pairs:
        ev.destroy(e0)
        ev.destroy(e1)
    plan.options = _lib.options(pixels=1) if prm.pixels == "mask" else None
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
    del tr, plan
    step_ms = 1e3 * elapsed / steps
    return {"step_ms": step_ms, "faces_per_sec": rk.world * B * steps / elapsed,
            "ray_steps_per_sec": rk.world * B * 256 * 256 * N_SAMPLES * steps / elapsed,
            "march_kernel_ms": march_ms, "bwd_kernel_ms": bwd_ms, "render_block_share_of_step": (march_ms + bwd_ms) / step_ms,
            "faces_per_gpu": B, "steps": steps, "warmup": max(warmup, 6), "epoch": epoch, "elapsed_s": elapsed, "render_pixels": pixels,
            "per_rank_s": per_rank_s,
            "workload": "BASELINE configs[%d]: batch=%d per GPU, full training step (RelightNet forward with the fused HIP render "
                        "block, PatchGAN step every 5th iteration, seven losses, backward through the fused HIP backward, two "
                        "Adam steps)%s" % (2 if rk.world == 1 else 3, B,
                                           "" if rk.world == 1 else ", DistributedDataParallel over RCCL")}


def train_leg_subprocess(steps, warmup, timeout_s=420):
    """The configs[2] leg of the headline line, run as `bench.py --workload train` in its OWN process while this one idles:
    exactly the stand-alone measurement (MIOpen's find mode tunes its convolutions against a quiet, freshly initialised
    device context -- run in-process behind the render legs' graphs, plans and streams the same 12 steps took 131 ms each
    instead of 32 in one of two runs, profiles/r04_report.md).  Returns measure_train()'s scalars."""
    import subprocess
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    cmd = [sys.executable, os.path.abspath(__file__), "--workload", "train", "--steps", str(steps), "--warmup", str(warmup), "--gpus", "1"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    if r.returncode != 0:
        raise RuntimeError("train leg exited with %d: %s" % (r.returncode, r.stderr[-400:]))
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    return {"step_ms": d["ms_per_step"], "faces_per_sec": d["faces_per_sec"], "ray_steps_per_sec": d["value"],
            "march_kernel_ms": d["train_march_kernel_ms"], "bwd_kernel_ms": d["train_bwd_kernel_ms"],
            "render_block_share_of_step": d["render_block_ms"]["share_of_step"], "faces_per_gpu": d["config"]["faces_per_gpu"],
            "steps": d["steps"], "warmup": d["warmup"], "workload": d["config"]["workload"], "process": "subprocess (bench.py --workload train)"}


def run_train(a, rk):
    rank, world = rk.rank, rk.world
    B = a.faces if a.faces != FACES_PER_GPU else 32                          # configs[2]: batch=32 per GPU
    m = measure_train(rk, B, a.steps, a.warmup, pixels=a.pixels)
    layout = rk.describe()
    if rank != 0:
        return None
    bwd_ms, march_ms = m["bwd_kernel_ms"], m["march_kernel_ms"]
    pm = pmc_summary().get("kernels", {})
    px_bytes = 72.0     # per pixel: reads depth 4 + albedo 12 + g_rendered 12 + min_dist 4 + argmin 4 + normals 12 (+ stencil
    #                     neighbours from cache), writes grad_albedo 12 + grad_depth read-modify-write 8 + light partials ~0
    traffic = pm.get("bwd", {}).get("hbm", {}).get("bytes_per_launch")
    bwd_roof = {"bound": "hbm", "kernel": "gcfr::render_bwd_single_light_kernel", "avg_launch_ms": bwd_ms,
                "achieved": B * 65536 * px_bytes / (bwd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": B * 65536 * px_bytes / (bwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "algorithmic_bytes_per_launch": B * 65536 * px_bytes,
                "traffic": traffic,
                "note": "one backward sample per pixel: compulsory I/O 72 B per pixel; `traffic` is that of tools/bwd_bench.py's "
                        "DENSE upstream gradient at batch 32 (profiles/pmc_summary.json kernels.bwd), this step's gradient is "
                        "masked (about half the pixels carry one)"}
    return {
        "metric": "ray_steps_per_sec", "value": m["ray_steps_per_sec"], "unit": "ray-steps/s", "n_gpus": world,
        "rccl_ranks": layout["rccl_ranks"], "process_layout": layout,
        "per_rank": [{"rank": r, "seconds": s_, "faces_per_sec": B * a.steps / s_} for r, s_ in enumerate(m["per_rank_s"])],
        "steps": a.steps, "warmup": m["warmup"], "ms_per_step": m["step_ms"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 network (MIOpen) + f64/f32 render block", "data": "synthetic",
        "config": {"workload": m["workload"],
                   "faces_per_gpu": B, "global_batch": B * world, "H": 256, "W": 256, "n_samples": N_SAMPLES,
                   "parallelism": "dp%d" % world, "epoch": m["epoch"],
                   "ssim_blur": "aten (train._DepthwiseBlur: the SSIM's depthwise blurs on ATen's kernels; rounds 2-5: MIOpen's, +3 ms)"},
        "faces_per_sec": m["faces_per_sec"],
        "train_step_ms": m["step_ms"], "train_faces_per_sec": m["faces_per_sec"], "train_march_kernel_ms": march_ms,
        "train_bwd_kernel_ms": bwd_ms,
        "render_block_ms": {"forward_march_kernel": march_ms, "fused_backward_kernel": bwd_ms,
                            "share_of_step": m["render_block_share_of_step"],
                            "note": "the step is MIOpen's fp32 convolutions and BatchNorm (59 %) plus bandwidth-bound ATen glue "
                                    "(profiles/r06_train_step_breakdown_aten.md); the render block's two big kernels are this share of it"},
        "roofline": bwd_roof,
    }


def compact(obj):
    """The ONE JSON line without its prose: `note` / `sample` / `workload` strings longer than 120 characters are cut (what every
    key means and how it is measured is DESIGN.md section 5's table), per-run lists go.  Keeps the line under ~8 KB so that a
    consumer that stores only the tail of stdout still holds all of it.  `--verbose-json` prints everything."""
    if isinstance(obj, dict):
        out = {}
        for k, v in obj.items():
            if k in ("runs_s", "forward_by_threads"):
                continue
            if k == "note" and isinstance(v, str) and len(v) > 120:
                continue
            if isinstance(v, str) and len(v) > 240:
                v = v[:237] + "..."
            out[k] = compact(v)
        return out
    if isinstance(obj, list):
        return [compact(v) for v in obj]
    return obj


def run_dry(a, rk):
    """--dry-run: the process layout only (rendezvous, one collective, the JSON line) -- what a CPU-only container can
    check of `bench.py --gpus N` (tests/test_bench_launch.py); no kernels, no numbers."""
    layout = rk.describe()
    per_rank = rk.gather(rk.rank)
    faces = (a.faces if a.faces != FACES_PER_GPU else 32) if a.workload == "train" else a.faces       # as run_train / run_render
    seeds = rk.gather(rk.rank * 1_000_000)                       # seed0 of rank r's faces: RenderRig._device_batch / measure_train
    faces_all = rk.gather(faces)
    if rk.rank != 0:
        return None
    return {"dry_run": True, "metric": "ray_steps_per_sec", "value": None, "unit": "ray-steps/s", **layout,
            "ranks_seen": per_rank, "workload": a.workload, "faces_per_rank": faces, "global_batch": int(sum(faces_all)),
            "seed0_per_rank": [int(s_) for s_ in seeds], "parallelism": "dp%d" % rk.world,
            "nominal_ray_steps_per_step": int(sum(faces_all)) * a.lights * a.size * a.size * a.samples,
            # the partition of BASELINE configs[4] / DESIGN.md section 6: whole faces per rank, ALL lights of a face on its rank
            "lights_per_face": a.lights, "face_lights_per_rank": faces * a.lights, "lights_split_across_ranks": False,
            "size": a.size, "samples": a.samples,
            # the end-to-end relight leg (relight_e2e_leg) belongs to rank 0's single-GPU headline line only
            "relight_e2e_leg_runs": bool(rk.world == 1 and a.workload == "render" and not a.no_worst_case),
            # ... and with N > 1 ranks of the headline shape every rank relights its own photographs (relight_e2e_ranks, weak scaling)
            "relight_e2e_ranks_runs": bool(rk.world > 1 and a.workload == "render" and not a.no_worst_case and a.size == 256
                                           and a.lights == 1 and a.samples == 160 and a.faces == FACES_PER_GPU)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1,
                    help="ranks = GPUs of this node; started without torch.distributed.run, bench.py spawns the N ranks itself")
    ap.add_argument("--steps", type=int, default=None, help="default: 3000 (render), 20 (train)")
    ap.add_argument("--warmup", type=int, default=None, help="default: 50 (render), 6 (train)")
    ap.add_argument("--workload", choices=["render", "train"], default="render",
                    help="render = BASELINE configs[1] (batch 8, forward); train = configs[2]/[3] (batch 32, full step)")
    ap.add_argument("--faces", type=int, default=FACES_PER_GPU, help="faces per GPU per step (configs[1]: 8; train: 32)")
    ap.add_argument("--data", choices=["synthetic", "ffhq", "train_depth"], default="synthetic",
                    help="ffhq = the three checkpoint-derived FFHQ fixture faces (tests/golden/inputs.npz) tiled to the batch; "
                         "train_depth = depth / light / albedo of a freshly initialised RelightNet (use with --from-depth --argmin)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-check", action="store_true",
                    help="skip the in-run parity spot check (face 0 of the timed batch through the timed plan vs the C oracle)")
    ap.add_argument("--no-worst-case", action="store_true", help="skip the secondary workloads of the headline line")
    ap.add_argument("--regions", type=int, default=0,
                    help="number of fenced timed regions of `steps` steps (0 = auto: repeated when a region is shorter than 200 ms; "
                         "profiling scripts pass 1)")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than GPUs (ranks share GPUs, collectives over gloo): rehearses the launch path")
    ap.add_argument("--dry-run", action="store_true", help="rendezvous + one collective + JSON only (runs without a GPU)")
    ap.add_argument("--direct", action="store_true", help="A/B: direct-gather kernel (no workspace prepass)")
    ap.add_argument("--unfused", action="store_true", help="A/B: three separate entry points instead of gcfr_render_fwd")
    ap.add_argument("--from-depth", action="store_true",
                    help="(the default since round 4, accepted for old scripts) normals from depth (T8:353-354) inside the march "
                         "epilogue: gcfr_render_from_depth_fwd, what RelightNet.forward runs")
    ap.add_argument("--normals-in", action="store_true",
                    help="SURVEY 8d's accounting form instead: normals handed in as a 12 B/pixel input (gcfr_render_fwd; rounds "
                         "1-3's headline step -- the default line carries it as normals_in_*)")
    ap.add_argument("--normals-stage", choices=["auto", "fused", "kernel"], default="auto",
                    help="'kernel' = the normals stage as its own launch (gcfr_normals_fwd) in front of gcfr_render_fwd instead of "
                         "fused into the march epilogue (three launches per step, the same bits); 'auto' = the product's rule: by "
                         "the number of lights per face (block.normals_stage_for: fused below 16)")
    ap.add_argument("--verbose-json", action="store_true", help="print the line with every note / sample description (default: compact)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the configs[2] training-step leg of the headline line")
    ap.add_argument("--pixels", choices=["all", "mask"], default="all",
                    help="mask = RenderParams(pixels='mask') / gcfr_options.pixels = 1: pixels outside the mask are not marched (opt-in "
                         "deviation, include/gcfr.h; every loss of the training script multiplies them by the mask).  Non-headline.")
    ap.add_argument("--argmin", action="store_true", help="the training-time march (argmin tracked, 5 waves/SIMD)")
    ap.add_argument("--batches", type=int, default=0,
                    help="distinct batches resident in HBM the steps cycle through (default: one per stream -- 4 x 8 faces, inside the "
                         "Infinity Cache; 64: 512 faces = 2.4 GB, the `many_batches` leg's form)")
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
    ap.add_argument("--mask", choices=["ellipse", "ones", "zeros"], default="ellipse",
                    help="'ones' = worst case: no fully masked wave-step exists, nothing is skipped; 'zeros' = nothing to march: "
                         "the kernel's fixed cost per tile (prologue + epilogue)")
    ap.add_argument("--tune", type=str, default="",
                    help="A/B: comma list of gcfr_options knobs, e.g. tile_w=32,ksplit=1,depth_bound_skip=0,group=2 "
                         "(never changes a result bit)")
    a = ap.parse_args()
    if a.steps is None:
        a.steps = 3000 if a.workload == "render" else 20
    if a.warmup is None:
        a.warmup = 50 if a.workload == "render" else 6
    global FORCED_REGIONS
    FORCED_REGIONS = max(0, a.regions)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(a))                     # this process becomes the launcher of N ranks
    rk = Ranks(a)
    if rk.world != a.gpus:
        raise SystemExit("bench.py --gpus %d was started with WORLD_SIZE=%d" % (a.gpus, rk.world))
    out = run_dry(a, rk) if a.dry_run else (run_render if a.workload == "render" else run_train)(a, rk)
    if rk.rank == 0:
        print(json.dumps(compact(out) if not a.verbose_json else out), flush=True)
    rk.close()


if __name__ == "__main__":
    main()
