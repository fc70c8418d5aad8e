"""CPU: the oracle (own C / numpy restatement) against golden vectors captured from the
reference itself (tests/golden/make_golden.py). This is what pins the oracle."""
import numpy as np
import pytest

from oracle import cport, npref
from conftest import load_golden
from util import bits_equal, assert_neighbors_equal_mod_ties, rel_err


@pytest.mark.parametrize("name", ["g1_sub_4096", "g1_sub_batch", "g1_sub_maxp"])
def test_subsample_batch_bit_exact(name):
    g = load_golden(name)
    sp, sl = cport.subsample_batch(g["points"], g["lens"], dl=float(g["dl"]), max_p=int(g.get("max_p", 0)))
    assert np.array_equal(sl, g["out_lens"])
    assert bits_equal(sp, g["out_points"])          # values AND unordered_map iteration order


def test_subsample_features_labels_bit_exact():
    g = load_golden("g1_sub_feat_lab")
    p, f, l = cport.subsample(g["points"], g["features"], g["labels"], dl=float(g["dl"]))
    assert bits_equal(p, g["out_points"]) and bits_equal(f, g["out_features"])
    assert np.array_equal(l, g["out_labels"])       # incl. tie-broken majority votes


def test_subsample_against_live_reference_if_built():
    if cport.ref() is None:
        pytest.skip("oracle/_ref/libref.so not built here")
    rng = np.random.default_rng(7)
    for n, dl in [(1, 0.1), (13, 0.5), (14, 0.01), (777, 0.05), (50000, 0.03)]:
        p = (rng.random((n, 3)) * [3, 2, 1]).astype(np.float32)
        a = cport.subsample_batch(p, [n], dl=dl, impl="oracle")
        b = cport.subsample_batch(p, [n], dl=dl, impl="ref")
        assert bits_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_neighbors_conv_exact():
    g = load_golden("g2_nb_conv_b1")
    nb = cport.radius_neighbors_batch(g["queries"], g["supports"], g["q_lens"], g["s_lens"], float(g["radius"]))
    assert nb.dtype == np.int32 and np.array_equal(nb, g["out"])   # tie-free fixture: exact order


def test_neighbors_volumetric_exact():
    g = load_golden("g2_nb_volumetric")
    nb = cport.radius_neighbors_batch(g["queries"], g["supports"], g["q_lens"], g["s_lens"], float(g["radius"]))
    assert np.array_equal(nb, g["out"])


def test_neighbors_ragged_pool_upsample_mod_ties():
    g = load_golden("g2_nb_pool_up_b3")
    pool = cport.radius_neighbors_batch(g["coarse"], g["fine"], g["coarse_lens"], g["fine_lens"], float(g["r_pool"]))
    up = cport.radius_neighbors_batch(g["fine"], g["coarse"], g["fine_lens"], g["coarse_lens"], float(g["r_up"]))
    assert_neighbors_equal_mod_ties(pool, g["out_pool"], g["coarse"], g["fine"], g["coarse_lens"], g["fine_lens"])
    assert_neighbors_equal_mod_ties(up, g["out_up"], g["fine"], g["coarse"], g["fine_lens"], g["coarse_lens"])
    # the isolated query has an all-pad row
    iso = int(g["coarse_lens"][0]) + 3
    assert np.all(pool[iso] == g["fine"].shape[0])


KP_CASES = [("g4_kpconv_config1", "linear", "sum"), ("g4_kpconv_gaussian", "gaussian", "sum"),
            ("g4_kpconv_constant", "constant", "sum"), ("g4_kpconv_closest", "linear", "closest"),
            ("g4_kpconv_cin66", "linear", "sum"), ("g4_kpconv_cin2", "linear", "sum"),
            ("g4_kpconv_strided", "linear", "sum")]


@pytest.mark.parametrize("name,influence,agg", KP_CASES)
def test_kpconv_numpy_vs_reference(name, influence, agg):
    g = load_golden(name)
    idx = g["idx"].astype(np.int64)
    args = [g[k].astype(np.float64) for k in ("q", "s")] + [idx, g["x"].astype(np.float64),
            g["kernel_points"].astype(np.float64), g["weights"].astype(np.float64), float(g["extent"])]
    y = npref.kpconv_forward(*args, influence=influence, aggregation=agg)
    assert rel_err(y, g["y"]) < 1e-4               # fp64 restatement vs fp32 reference: 1e-4 rel (north_star)
    dx, dW = npref.kpconv_backward(*args, g["g"].astype(np.float64), influence=influence, aggregation=agg)
    assert rel_err(dx, g["x_grad"]) < 1e-4
    assert rel_err(dW, g["weights_grad"]) < 1e-4


def test_pool_helpers_exact():
    g = load_golden("g4_pools")
    assert bits_equal(npref.max_pool(g["x"], g["pool_idx"]), g["max_pool"])
    assert bits_equal(npref.closest_pool(g["x"], g["pool_idx"]), g["closest_pool"])
