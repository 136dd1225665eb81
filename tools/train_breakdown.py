#!/usr/bin/env python3
"""What BASELINE configs[2]'s training step (batch 32, ~33 ms) is made of  (round-5 verdict: "the step nobody has opened").

  tools/train_breakdown.py run [--steps 10] [--faces 32] [--levers ...]      (GPU)  warm-up, then `steps` steps between two
        SENTINEL launches (gcfr_copy_probe: a kernel name that appears nowhere else in the step), each step's phases
        bracketed by in-stream events (no synchronisation inside the window): hourglass forward, render block forward, D
        step, G losses (+ PatchGAN forward), G backward, G optimiser.  Prints one JSON line (phase ms per step).
        Under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python tools/train_breakdown.py run` the trace holds
        the kernels of exactly those steps between the sentinels.
  tools/train_breakdown.py classify DIR [--phases phases.json] [--out profiles/r06_train_step_breakdown.md]   (CPU)
        reads DIR/**/*kernel_trace.csv, keeps the dispatches between the two sentinels, groups them by class
        (MIOpen convolution forward / backward-data / backward-weights by kernel name, BatchNorm, elementwise / reduction /
        copy kernels of ATen, the SSIM's depthwise convolutions, optimiser, the HIP render block) and writes the table:
        ms per step and share per class, launches per step, GPU-idle time inside the window.
"""
import argparse
import csv
import glob
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


# ------------------------------------------------------------------------------------------------
# run (GPU)
# ------------------------------------------------------------------------------------------------
def run(a):
    import numpy as np
    import torch
    from geomconsistentfr_amd import _lib
    from geomconsistentfr_amd.train import (TrainConfig, Trainer, discriminator_losses, generator_losses, synthetic_batch)

    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    tr = Trainer(TrainConfig(ssim_stacked=bool(a.ssim_stacked), ssim_blur=a.ssim_blur), device=dev)
    batch = synthetic_batch(a.faces, 0, device=dev)
    for j in range(a.warmup):
        tr.step(batch, a.epoch, j, log=False)
    torch.cuda.synchronize()
    # the plain step, fenced: the number bench.py --workload train reports
    t0 = time.perf_counter()
    for j in range(a.steps):
        tr.step(batch, a.epoch, j, log=False)
    torch.cuda.synchronize()
    plain_ms = 1e3 * (time.perf_counter() - t0) / a.steps

    L_ = _lib.load()
    probe_src = torch.zeros(1 << 16, dtype=torch.uint8, device=dev)
    probe_dst = torch.empty_like(probe_src)

    def sentinel():
        _lib.check(L_.gcfr_copy_probe(probe_src.data_ptr(), probe_dst.data_ptr(), probe_src.numel(),
                                      torch.cuda.current_stream(dev).cuda_stream), "gcfr_copy_probe")

    names = ["hourglass_fwd", "render_block_fwd", "d_step", "g_losses_fwd", "g_backward", "g_optimizer"]
    marks = []

    def mark():
        if a.phase_sentinels:       # a sentinel launch at every phase boundary: the kernel trace can then be cut into phases
            sentinel()
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    import geomconsistentfr_amd.relightnet as RN
    model = tr.model
    torch.cuda.synchronize()
    if not a.phase_sentinels:
        sentinel()
    t0 = time.perf_counter()
    for j in range(a.steps):
        ev = [mark()]
        img = batch["images"].permute(0, 3, 1, 2)
        m3 = batch["masks_fill"].permute(0, 3, 1, 2).expand(-1, 3, -1, -1)
        # RelightNet.forward, opened at the T8:352 seam (same calls, the prepass hook as the model issues it)
        cam = model._camera(tr.K)
        mk = batch["masks_fill"].reshape(a.faces, 256, 256)
        early = []
        albedo, depth, SL = model.features(batch["images"], a.epoch, lambda d, s: early.append(
            RN.render_from_depth_prepass(d, s[:, 0, 0, 1:4], cam, mk, model.render_params)))
        ev.append(mark())
        r = RN.render_from_depth(depth, albedo, SL[:, 0, 0, 1:4], SL[:, 0, 0, 0], cam, model.normal_z_offset, mk,
                                 model.render_params, prepared=early[0])
        out = (albedo, depth, r["shadow_mask_weights"], r["ambient_light"], r["full_shading"], r["rendered_images"],
               r["unit_light_direction"], r["ambient_values"])
        rendered = out[5]
        composite = rendered * m3 + (1.0 - m3) * img
        ev.append(mark())
        if j % tr.cfg.gd_ratio == 0:
            tr.opt_d.zero_grad(set_to_none=True)
            d_fake, d_real = discriminator_losses(tr.disc, composite.detach(), img)
            (d_fake + d_real).backward()
            tr.opt_d.step()
        ev.append(mark())
        tr.opt.zero_grad(set_to_none=True)
        for p in tr.patchgan.parameters():
            p.requires_grad_(False)
        Ls = generator_losses(out, batch, tr.patchgan(composite), tr.cfg.ssim_stacked, tr.cfg.ssim_blur)
        ev.append(mark())
        Ls["total"].backward()
        for p in tr.patchgan.parameters():
            p.requires_grad_(True)
        ev.append(mark())
        tr.opt.step()
        ev.append(mark())
        marks.append(ev)
    if not a.phase_sentinels:
        sentinel()
    torch.cuda.synchronize()
    opened_ms = 1e3 * (time.perf_counter() - t0) / a.steps
    phases = {n: float(np.mean([m[i].elapsed_time(m[i + 1]) for m in marks])) for i, n in enumerate(names)}
    d_steps = sum(1 for j in range(a.steps) if j % tr.cfg.gd_ratio == 0)
    print(json.dumps({"faces": a.faces, "steps": a.steps, "epoch": a.epoch, "step_ms_plain": plain_ms, "step_ms_opened": opened_ms,
                      "phase_ms_per_step": phases, "d_steps_in_window": d_steps, "phase_names": names,
                      "phase_sentinels": bool(a.phase_sentinels),
                      "note": "in-stream events, no synchronisation inside the window; d_step is averaged over ALL steps (it runs "
                              "every %d-th)" % tr.cfg.gd_ratio}))


# ------------------------------------------------------------------------------------------------
# classify (CPU)
# ------------------------------------------------------------------------------------------------
CLASSES = [
    # (class, regex on the kernel name) -- first match wins
    ("HIP render block (gcfr::)", r"gcfr::"),
    ("BatchNorm (MIOpen)", r"[Bb]atch[Nn]orm"),
    ("depthwise conv (ATen: the SSIM's blurs)", r"DepthwiseConv|depthwise|conv_depthwise"),
    ("MIOpen tensor ops (bias add, transposes, casts)", r"SubTensorOp|OpTensor|batched_transpose|transpose_|Transpose"),
    ("MIOpen conv backward-weights", r"[Ww]rw|WrW|bwd_?wei|BwdWei|backward_weights|wrw"),
    ("MIOpen conv backward-data", r"[Bb]wd(?!.*[Ww]ei)|backward_data|Bwd|_bwd_"),
    ("MIOpen conv forward", r"[Ff]wd|Conv|conv|igemm|gemm|Cijk|SubTensorOp|naive_conv|Im2Col|im2col|Col2Im|winograd|Winograd|sp3"),
    ("optimiser (Adam, foreach)", r"multi_tensor_apply|adam|Adam|FusedOptimizer|foreach"),
    ("reductions (sum / mean)", r"reduce_kernel|Reduce"),
    ("upsample / pooling", r"upsample|max_pool|avg_pool|MaxPool|AvgPool|pooling|Pool"),
    ("copies / fills / cat", r"copy|Copy|fill|Fill|CatArray|memcpy|Memcpy|memset"),
    ("elementwise (ATen)", r"elementwise|vectorized|unrolled|index|gather|scatter|where|clamp|sigmoid|leaky"),
]


def classify_name(name):
    for cls, pat in CLASSES:
        if re.search(pat, name):
            return cls
    return "other"


def classify(a):
    files = glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under %s" % a.dir)
    rows = []
    for f in files:
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    sent = [i for i, r in enumerate(rows) if "copy_probe" in r[2]]
    if len(sent) < 2:
        raise SystemExit("sentinels (gcfr copy_probe) not found: %d" % len(sent))
    phases = json.load(open(a.phases)) if a.phases else None
    steps = phases["steps"] if phases else a.steps
    by_phase = None
    if phases and phases.get("phase_sentinels"):
        # 7 sentinels per step (one at every phase boundary): the last 7 * steps of them bracket the window's phases
        names = phases["phase_names"]
        per = len(names) + 1
        sent = sent[-per * steps:]
        assert len(sent) == per * steps, (len(sent), per * steps)
        lo, hi = sent[0], sent[-1]
        by_phase = {n: {} for n in names}
        by_phase_name = {n: {} for n in names}
        for q, i0 in enumerate(sent):
            ph = q % per
            if ph == len(names):
                continue
            i1 = sent[q + 1]
            for s_, e_, n_ in rows[i0 + 1:i1]:
                d = by_phase[names[ph]].setdefault(classify_name(n_), [0, 0])
                d[0] += e_ - s_
                d[1] += 1
                d = by_phase_name[names[ph]].setdefault(re.sub(r"\s+", " ", n_)[:120], [0, 0])
                d[0] += e_ - s_
                d[1] += 1
        win = [r_ for r_ in rows[lo + 1:hi] if "copy_probe" not in r_[2]]
    else:
        lo, hi = sent[-2], sent[-1]
        win = rows[lo + 1:hi]
    t_lo, t_hi = rows[lo][1], rows[hi][0]
    by = {}
    by_name = {}
    short = lambda n: re.sub(r"\s+", " ", n)[:120]
    for s, e, n in win:
        c = classify_name(n)
        d = by.setdefault(c, [0, 0])
        d[0] += e - s
        d[1] += 1
        k = by_name.setdefault((c, short(n)), [0, 0])
        k[0] += e - s
        k[1] += 1
    # busy time = union of the dispatch intervals (kernels of different streams overlap: the prepass under the albedo decoder)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        busy += cur_e - cur_s
    window = t_hi - t_lo
    total_kernel = sum(v[0] for v in by.values())
    ms = lambda ns: ns / 1e6 / steps
    noprof = json.load(open(a.noprof)) if a.noprof else None
    lines = ["# BASELINE configs[2]: what one training step is made of (round 6)", "",
             "`rocprofv3 --kernel-trace` of `tools/train_breakdown.py run --steps %d --faces %d --phase-sentinels`: the dispatches between the"
             % (steps, phases["faces"] if phases else 32),
             "sentinel launches (steady state, after MIOpen's find and the warm-up steps), classified by kernel name.", ""]
    if noprof:
        lines += ["**Without a profiler attached** (the step as `bench.py --workload train` times it; in-stream events between the phases): "
                  "**%.2f ms per step** -- %s.  Under rocprofv3 every launch costs the host ~5 us more, so the traced window below is "
                  "longer than that and shows idle time the untraced step does not have: the untraced step's phases add up to the "
                  "kernel time (the GPU is busy throughout)." % (noprof["step_ms_plain"], ", ".join("%s %.2f" % (k, v) for k, v in noprof["phase_ms_per_step"].items())), ""]
    lines += [
             "| class | ms / step | share of GPU-busy | launches / step |", "|---|---:|---:|---:|"]
    for c, (ns, n) in sorted(by.items(), key=lambda kv: -kv[1][0]):
        lines.append("| %s | %.3f | %.1f %% | %.1f |" % (c, ms(ns), 100.0 * ns / total_kernel, n / steps))
    lines += ["| **sum of kernel durations** | **%.3f** | 100 %% | %.1f |" % (ms(total_kernel), len(win) / steps),
              "", "* window (sentinel to sentinel): **%.3f ms / step**; GPU busy (union of dispatch intervals): %.3f ms / step; "
              "**idle inside the window: %.3f ms / step (%.1f %%)**; kernels overlapping on two streams: %.3f ms / step."
              % (ms(window), ms(busy), ms(window - busy), 100.0 * (window - busy) / window, ms(total_kernel - busy)), ""]
    conv = sum(v[0] for k, v in by.items() if k.startswith("MIOpen conv"))
    lines += ["* MIOpen convolutions (forward + backward-data + backward-weights): %.3f ms / step = **%.1f %%** of the GPU-busy time; "
              "with BatchNorm %.1f %%." % (ms(conv), 100.0 * conv / total_kernel,
                                          100.0 * (conv + by.get("BatchNorm (MIOpen)", [0])[0]) / total_kernel), ""]
    if phases:
        lines += ["## Phases of the step (in-stream events, no synchronisation inside the window)", "",
                  "| phase | ms / step |", "|---|---:|"]
        for k, v in phases["phase_ms_per_step"].items():
            lines.append("| %s | %.3f |" % (k, v))
        lines += ["| **sum** | **%.3f** |" % sum(phases["phase_ms_per_step"].values()), "",
                  "* fenced wall time of the plain `Trainer.step`: %.3f ms / step; of the opened step above: %.3f ms / step."
                  % (phases["step_ms_plain"], phases["step_ms_opened"]), "* " + phases["note"], ""]
    if by_phase:
        classes = [c for c, _ in sorted(by.items(), key=lambda kv: -kv[1][0])]
        lines += ["## Kernel time by phase and class (ms / step; the trace cut at the phase sentinels)", "",
                  "| class | " + " | ".join(by_phase) + " |", "|---|" + "---:|" * len(by_phase)]
        for c in classes:
            lines.append("| %s | " % c + " | ".join("%.3f" % ms(by_phase[p_].get(c, [0])[0]) for p_ in by_phase) + " |")
        lines.append("| **sum** | " + " | ".join("**%.3f**" % ms(sum(v[0] for v in by_phase[p_].values())) for p_ in by_phase) + " |")
        lines.append("| launches / step | " + " | ".join("%.1f" % (sum(v[1] for v in by_phase[p_].values()) / steps) for p_ in by_phase) + " |")
        lines.append("")
    if a.json:
        json.dump({"steps": steps, "window_ms_per_step": ms(window), "busy_ms_per_step": ms(busy),
                   "by_class": {c: {"ms_per_step": ms(v[0]), "launches_per_step": v[1] / steps} for c, v in by.items()},
                   "by_phase": None if not by_phase else {p_: {c: {"ms_per_step": ms(v[0]), "launches_per_step": v[1] / steps} for c, v in d_.items()}
                                                          for p_, d_ in by_phase.items()},
                   "kernels_by_phase": None if not by_phase else {p_: [{"name": n, "ms_per_step": ms(v[0]), "launches_per_step": v[1] / steps}
                                                                           for n, v in sorted(d_.items(), key=lambda kv: -kv[1][0])[:40]]
                                                                      for p_, d_ in by_phase_name.items()},
                   "kernels": [{"class": c, "name": n, "ms_per_step": ms(v[0]), "launches_per_step": v[1] / steps}
                               for (c, n), v in sorted(by_name.items(), key=lambda kv: -kv[1][0])]}, open(a.json, "w"), indent=1)
    lines += ["## The twenty-five kernels with the most time", "", "| class | kernel | ms / step | launches / step |", "|---|---|---:|---:|"]
    for (c, n), (ns, cnt) in sorted(by_name.items(), key=lambda kv: -kv[1][0])[:25]:
        lines.append("| %s | `%s` | %.3f | %.1f |" % (c, n.replace("|", "/"), ms(ns), cnt / steps))
    text = "\n".join(lines) + "\n"
    if a.out:
        open(a.out, "w").write(text)
    print(text)


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    r.add_argument("--steps", type=int, default=10)
    r.add_argument("--warmup", type=int, default=8)
    r.add_argument("--faces", type=int, default=32)
    r.add_argument("--epoch", type=int, default=200)
    r.add_argument("--ssim-blur", choices=["aten", "miopen"], default="aten", help="TrainConfig.ssim_blur")
    r.add_argument("--ssim-stacked", type=int, default=0, help="TrainConfig.ssim_stacked (0: the five separate blurs of rounds 2-5)")
    r.add_argument("--phase-sentinels", action="store_true", help="a sentinel launch at every phase boundary (classify then splits by phase)")
    c = sub.add_parser("classify")
    c.add_argument("dir")
    c.add_argument("--phases", default=None)
    c.add_argument("--steps", type=int, default=10)
    c.add_argument("--out", default=None)
    c.add_argument("--noprof", default=None, help="the JSON line of a `run` WITHOUT a profiler attached (the step as the bench times it)")
    c.add_argument("--json", default=None, help="every kernel name with its time and launches per step, by class / phase")
    a = ap.parse_args()
    run(a) if a.cmd == "run" else classify(a)


if __name__ == "__main__":
    main()
