"""Shared test helpers (comparison rules stated once)."""
import numpy as np


def bits_equal(a, b):
    """Bit-exact equality for float32 arrays (also distinguishes -0/+0 and NaN payloads)."""
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def nb_d2(q, s, q_lens, s_lens, nb):
    """float32 d2 of every entry of a neighbour matrix ((dx*dx+dy*dy)+dz*dz, nanoflann order); pad -> +inf."""
    Ns = s.shape[0]
    sp = np.concatenate([s, np.full((1, 3), np.inf, np.float32)], 0)
    d = q[:, None, :] - sp[nb]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
    return np.where(nb >= Ns, np.float32(np.inf), d2.astype(np.float32))


def canon_ties(nb, d2):
    """Sort indices inside equal-d2 runs (tie groups compared as sets)."""
    order = np.lexsort((nb, d2), axis=1) if False else None
    out = np.empty_like(nb)
    for i in range(nb.shape[0]):
        o = np.lexsort((nb[i], d2[i]))
        out[i] = nb[i][o]
    return out


def assert_neighbors_equal_mod_ties(got, want, q, s, q_lens, s_lens, cropped=False):
    """Same rows up to the order inside equal-d2 groups. cropped=True: the matrices were cut to a
    column limit, which may cut through the LAST tie group of a row -- that group is then only
    compared by its distances."""
    assert got.shape == want.shape and got.dtype == want.dtype
    dg, dw = nb_d2(q, s, q_lens, s_lens, got), nb_d2(q, s, q_lens, s_lens, want)
    # both sorted ascending, identical distance sequences
    assert np.all(np.diff(np.where(np.isinf(dg), np.float32(3e38), dg), axis=1) >= 0)
    assert np.array_equal(dg, dw)
    cg, cw = canon_ties(got, dg), canon_ties(want, dw)
    if cropped and got.shape[1] > 0:
        keep = dg != dg[:, -1:]
        keep |= np.isinf(dg)
        assert np.array_equal(np.where(keep, cg, -1), np.where(keep, cw, -1))
    else:
        assert np.array_equal(cg, cw)


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def check_err(label, err, bound):
    """Assert err < bound and record the MEASURED error next to the bound: printed (pytest -s) and appended to
    $MVK_PARITY_LOG (default gpurun_out/parity_errors.txt) so the bounds can be read against what the run shows."""
    import os
    line = "%-78s measured %.3e  bound %.1e" % (label, float(err), float(bound))
    print(line)
    path = os.environ.get("MVK_PARITY_LOG")
    if path is None:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        path = os.path.join(root, "gpurun_out", "parity_errors.txt")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "a") as f:
            f.write(line + "\n")
    except OSError:
        pass
    assert err < bound, line
