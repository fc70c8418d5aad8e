"""Generates tests/golden/g14_f64_referee.npz: a FLOAT64 referee for the network-level gradient checks.

The fixtures G5 / G5b (the reference's own KPFCNN run in float32) and G12 (the reference's KPFCNN_featureAggre classes,
float32) bound the HIP path against ANOTHER float32 evaluation of the same network; through ~36 layers of train-mode
BatchNorm and LeakyReLU two float32 evaluation orders differ by 1e-4 .. 1e-3 in some gradients, so those bounds had to be
wide and could not show that the HIP path is no further from the truth than the reference's own float32 run is. This
script runs oracle/torch_port.py (the CPU restatement that test_oracle_vs_golden.py pins to those fixtures) in float64
on the SAME inputs and weights -- taken from the committed fixtures, nothing of /root/reference is needed -- and stores

    g5/..., g5b_deform/..., g5b_deform_mod/...   logits, loss, every gradient the float32 fixture holds (float64)
    g12/<variant>/...                            logits, loss, per parameter: float64 gradient norm + the float64 values
                                                 at the fixture's 64 digest indices
    g13_early_19k/..., g13_late_deform_mod_55k/... the same digest (256 elements) at BASELINE's own sizes

The GPU tests then assert, per parameter, err(HIP, float64) <= 2 x err(reference float32, float64) (+ a floor of one
float32 rounding of the tensor's scale) and log both columns.   Usage:  python tests/golden/make_f64_referee.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

from conftest import load_golden  # noqa: E402
from oracle import torch_port  # noqa: E402
import test_oracle_vs_golden as tog  # noqa: E402


def to64(x):
    if isinstance(x, list):
        return [to64(v) for v in x]
    if torch.is_tensor(x) and x.is_floating_point():
        return x.double()
    return x


def run64(sd, cfg, batch, leaf_names):
    sd = {k: to64(v) for k, v in sd.items()}
    leaf = {k: sd[k].clone().requires_grad_(True) for k in leaf_names}
    sd.update(leaf)
    b = {k: to64(v) for k, v in batch.items()}
    out, reg = torch_port.forward(sd, cfg, b, None, True)
    ce = torch_port.loss_fn(out, b["labels"], [], cfg)
    loss = torch_port.loss_fn(out, b["labels"], reg, cfg)
    loss.backward()
    return out.detach().numpy(), float(ce.item()), float(loss.item()), {k: v.grad.numpy() for k, v in leaf.items() if v.grad is not None}


def main():
    torch.set_num_threads(8)
    arrs = {}
    for tag, name, cfg_of in (("g5", "g5_kpfcnn", lambda g: tog.g5_config()),
                              ("g5b_deform", "g5b_kpfcnn_deform", lambda g: tog.g5b_config(int(g["modulated"]))),
                              ("g5b_deform_mod", "g5b_kpfcnn_deform_mod", lambda g: tog.g5b_config(int(g["modulated"])))):
        g = load_golden(name)
        sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd/")}
        names = [k[5:] for k in g if k.startswith("grad/")]
        logits, ce, loss, grads = run64(sd, cfg_of(g), tog.g5_batch(g), names)
        arrs[tag + "/logits"] = logits
        arrs[tag + "/output_loss"] = np.float64(ce)
        arrs[tag + "/loss"] = np.float64(loss)
        for k in names:
            arrs["%s/grad/%s" % (tag, k)] = grads[k]
        print(tag, "float64 loss", loss, "| fixture", float(g["loss"]), "| worst fixture-vs-f64 gradient error",
              max(np.linalg.norm(g["grad/" + k].astype(np.float64) - grads[k]) / np.linalg.norm(grads[k]) for k in names))
    g = load_golden("g12_fusion_wirings")
    for variant in ("early", "middle", "late"):
        cfg, sd, b = tog.g12_inputs(g, variant)
        sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
        names = [k for k, v in sdt.items() if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var", "kernel_points"))]
        logits, ce, loss, grads = run64(sdt, cfg, b, names)
        arrs["g12/%s/logits" % variant] = logits
        arrs["g12/%s/loss" % variant] = np.float64(loss)
        worst = 0.0
        for n in sorted(k[len(variant) + 7:] for k in g if k.startswith(variant + "/gnorm/")):
            got = grads[n].reshape(-1)
            idx = g["%s/gidx/%s" % (variant, n)]
            arrs["g12/%s/gnorm/%s" % (variant, n)] = np.float64(np.linalg.norm(got))
            arrs["g12/%s/gval/%s" % (variant, n)] = got[idx].astype(np.float64)
            ref_norm = float(g["%s/gnorm/%s" % (variant, n)])
            if np.linalg.norm(got) > 0:
                worst = max(worst, abs(ref_norm / np.linalg.norm(got) - 1.0))
        print("g12", variant, "float64 loss", loss, "| fixture", float(g[variant + "/loss"]),
              "| worst fixture-vs-f64 |norm ratio - 1|", worst)
    # G13: BASELINE's own sizes. The fixture is ONE float32 run of the CPU port; here the same port in float64 on the same
    # batch (rebuilt by make_golden.g13_case_inputs with the fixture's limits and rotations) and weights.
    import importlib
    import time
    import util
    import make_golden
    cases = [("g13_early_19k", "early", False, 1.2)]
    if os.environ.get("MVK_REFEREE_55K", "1") == "1":
        cases.append(("g13_late_deform_mod_55k", "late", True, 1.7))
    for name, variant, deformable, radius in cases:
        t0 = time.time()
        g = load_golden(name)
        cfg, b, limits, rots = make_golden.g13_case_inputs(variant, deformable, deformable, radius,
                                                           limits=[int(v) for v in g["limits"]], rotations=list(g["rotations"]))
        shapes = {str(n): tuple(int(v) for v in str(sh).split(",") if v) for n, sh in zip(g["param_names"], g["param_shapes"])}
        kp = {k[3:]: g[k] for k in g if k.startswith("kp/")}
        sd = util.g13_state(shapes, variant, deformable, kp)
        sdt = {k: torch.from_numpy(v) for k, v in sd.items()}
        names = [k for k, v in sdt.items() if v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var", "kernel_points"))]
        logits, ce, loss, grads = run64(sdt, cfg, b, names)
        arrs[name + "/loss"] = np.float64(loss)
        arrs[name + "/logits"] = logits[g["logit_rows"]]
        worst = 0.0
        for n in sorted(k[6:] for k in g if k.startswith("gnorm/")):
            got = grads[n].reshape(-1)
            arrs["%s/gnorm/%s" % (name, n)] = np.float64(np.linalg.norm(got))
            arrs["%s/gval/%s" % (name, n)] = got[g["gidx/" + n]].astype(np.float64)
            if np.linalg.norm(got) > 0:
                worst = max(worst, abs(float(g["gnorm/" + n]) / np.linalg.norm(got) - 1.0))
        print(name, "float64 loss", loss, "| fixture", float(g["loss"]), "| worst fixture-vs-f64 |norm ratio - 1|", worst,
              "| %.0f s" % (time.time() - t0), flush=True)
    path = os.path.join(HERE, "g14_f64_referee.npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes,", len(arrs), "arrays")


if __name__ == "__main__":
    main()
