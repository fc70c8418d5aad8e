import importlib, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden
from util import nb_d2
from test_oracle_vs_golden import _Cfg
PKG = "enhancing-3d-point-cloud-segmentation-using-multi-modal-fusion-with-2d-images_amd"
common = importlib.import_module(PKG + ".dropin.datasets.common")
g = load_golden("g3_pyramid")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
pyr = common.segmentation_inputs_sphere(_Cfg, T(g["points0"]), g["lens0"], list(g["limits"]), torch.int32, rotations=list(g["rotations"]))
for l in range(5):
    p, ln = g["points%d" % l], g["lengths%d" % l]
    for name, got, q, s in [("neighbors", pyr["neighbors"][l], p, p)] + ([("pools", pyr["pools"][l], g["points%d" % (l+1)], p), ("upsamples", pyr["upsamples"][l], p, g["points%d" % (l+1)])] if l < 4 else []):
        got = got.cpu().numpy(); want = g["%s%d" % (name, l)]
        print(l, name, got.shape, want.shape, "equal", np.array_equal(got, want))
        if got.shape == want.shape and not np.array_equal(got, want):
            dg, dw = nb_d2(q, s, None, None, got), nb_d2(q, s, None, None, want)
            bad = np.nonzero((dg != dw).any(1))[0]
            print("  rows with different d2:", bad[:5], len(bad))
            if len(bad):
                r = bad[0]; print("  got", got[r][:12], dg[r][:12]); print("  want", want[r][:12], dw[r][:12])
