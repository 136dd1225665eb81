#!/usr/bin/env python3
"""Diagnostic (GPU box): run-to-run jitter of the training forward + backward at epoch 0 (untrained network, rough depth).
Two passes of the same loss on the same shard in one process: max |difference| of the render block's leaf gradients (depth,
albedo, light/ambient head) and of every network parameter gradient, relative to the gradient's own maximum."""
import copy
import json
import sys
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomconsistentfr_amd.train import TrainConfig, Trainer, synthetic_batch  # noqa: E402
from geomconsistentfr_amd.block import render_from_depth  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    if "--deterministic" in sys.argv:            # MIOpen's deterministic algorithms: is the network's own forward / backward the source?
        torch.backends.cudnn.deterministic = True
        torch.backends.cudnn.benchmark = False
    torch.manual_seed(0)
    tr = Trainer(TrainConfig(miopen_find=False), device=dev)
    b = synthetic_batch(2, 0, device=dev)
    out = {}

    def net_grads():
        m2 = copy.deepcopy(tr.model)
        o = m2(b["images"], 0, tr.K, b["masks_fill"])
        (o[5].mean() + o[2].mean() + 1e-3 * o[1].abs().mean()).backward()
        return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).flatten() for p in m2.parameters()])

    with torch.no_grad():
        f1 = tr.model.features(b["images"], 0)
        f2 = tr.model.features(b["images"], 0)
    out["features_twice"] = {n: float((x - y).abs().max()) for n, x, y in zip(("albedo", "depth", "SL"), f1, f2)}
    g1, g2 = net_grads(), net_grads()
    out["network_params"] = {"max_abs_diff": float((g1 - g2).abs().max()), "max_abs": float(g1.abs().max())}

    def leaf_grads(terms):
        with torch.no_grad():
            albedo, depth, SL = tr.model.features(b["images"], 0)
        leaves = [t.detach().clone().requires_grad_() for t in (albedo, depth, SL)]
        r = render_from_depth(leaves[1], leaves[0], leaves[2][:, 0, 0, 1:4], leaves[2][:, 0, 0, 0], tr.K, 1610.0,
                              b["masks_fill"].reshape(2, 256, 256), tr.model.render_params)
        loss = 0.0
        if "rendered" in terms:
            loss = loss + r["rendered_images"].mean()
        if "w" in terms:
            loss = loss + r["shadow_mask_weights"].mean()
        loss.backward()
        return [t.grad.detach().clone() for t in leaves]

    for terms in (("rendered",), ("w",), ("rendered", "w")):
        a, c = leaf_grads(terms), leaf_grads(terms)
        out["leaves_" + "+".join(terms)] = {n: {"max_abs_diff": float((x - y).abs().max()), "max_abs": float(x.abs().max())}
                                            for n, x, y in zip(("albedo", "depth", "SL"), a, c)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
