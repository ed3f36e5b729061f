import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def g5c_mesh(g):
    """Golden set G5c stores its seeds and the reference's outputs only; its 300 x 330 mesh is rebuilt here from the stored
    parameters exactly as tests/golden/gen_golden.py::polar_grid built it (synthetic grid + the oracle's inverse projection
    for the T-points' lat/lon) and checked against the stored checksums and probes before it is used."""
    from oracle import oracle as orc
    from sitrack_amd import synthetic as syn
    Nj, Ni, dkm, warp, yc, xc = g["mesh"]
    Nj, Ni = int(Nj), int(Ni)
    m = syn.make_grid(Nj, Ni, dkm=float(dkm), warp=float(warp))
    for k in ("Yt", "Yu", "Yv", "Yf"):
        m[k] = m[k] + float(yc)
    for k in ("Xt", "Xu", "Xv", "Xf"):
        m[k] = m[k] + float(xc)
    ll = orc.CartNPSkm2Geo1D(np.stack([m["Yt"].ravel(), m["Xt"].ravel()], axis=1))
    m["latT"] = np.ascontiguousarray(ll[:, 0].reshape(Nj, Ni))
    m["lonT"] = np.ascontiguousarray(np.mod(ll[:, 1], 360.).reshape(Nj, Ni))
    assert np.array_equal(m["latT"][::37, ::41], g["lat_probe"]) and np.array_equal(m["lonT"][::37, ::41], g["lon_probe"]), \
        "the rebuilt G5c mesh differs from the one the golden outputs were generated on"
    assert m["latT"].sum() == float(g["lat_sum"]) and m["lonT"].sum() == float(g["lon_sum"])
    tmask = m["tmask"].copy()
    for j0, j1, i0, i1 in g["tmask_boxes"]:
        tmask[j0:j1, i0:i1] = 0
    sic = np.ones((Nj, Ni))
    for (j0, j1, i0, i1), v in zip(g["sic_boxes"], g["sic_vals"]):
        sic[j0:j1, i0:i1] = v
    return m, tmask, sic


def g6b_case(g):
    """Golden set G6b (fast flow, 120 x 140 warped grid): grid and fields rebuilt from the stored parameters, guarded by checksums."""
    from sitrack_amd import synthetic as syn
    Nj, Ni, dkm, warp = g["mesh"]
    K, seed, umax, drift, ripple = g["fields"]
    grid = syn.make_grid(int(Nj), int(Ni), dkm=float(dkm), warp=float(warp))
    u, v, sic = syn.make_fields(grid, K=int(K), seed=int(seed), umax=float(umax), drift=float(drift), ripple=float(ripple))
    tmask = grid["tmask"].copy()
    for j0, j1, i0, i1 in g["tmask_boxes"]:
        tmask[j0:j1, i0:i1] = 0
    sic = sic.copy()
    for k in range(int(K)):
        sic[k, 20:30, 90 + 2 * k:105 + 2 * k] = 0.03
    assert u.astype(np.float64).sum() == float(g["u_sum"]) and sic.astype(np.float64).sum() == float(g["sic_sum"]), \
        "the rebuilt G6b fields differ from the ones the golden outputs were generated on"
    grid["tmask"] = tmask
    return grid, u, v, sic


_CUT_CACHE = {}


def baseline_cut_case(g):
    """Golden sets G6c / G6d (the reference's trajectories on the first 10^3 buoys x 100 records of the real C2 / C3 workloads
    of bench.py) and G6e / G6f (the same on bench.py's curvilinear workloads: C3 with `--warp 1.0`, `--config c5shape`): grid, the
    32 resident records and the buoys' positions rebuilt from the stored seeds exactly as bench.py builds them, guarded by
    checksums.  Cached per workload (the 4096^2 fields are 6.4 GB)."""
    from sitrack_amd import synthetic as syn
    kind = str(g["kind"]) if "kind" in g else "regular"
    Nj, Ni, dkm, warp = (g["mesh"][k] for k in range(4))
    Nj, Ni = int(Nj), int(Ni)
    nAll, bseed, nP = (int(x) for x in g["buoys"])
    K, fseed, umax, drift = g["fields"][:4]
    key = (Nj, Ni, kind)
    if key not in _CUT_CACHE:
        _CUT_CACHE.clear()
        grid = syn.make_grid(Nj, Ni, dkm=float(dkm), warp=float(warp))
        fkw = {}
        if kind == "c5shape":
            syn.shift_grid(grid, float(g["mesh"][4]), float(g["mesh"][5]))
            j0, j1, i0, i1 = (int(x) for x in g["island"])
            grid["tmask"][j0:j1, i0:i1] = 0
            rng = np.random.default_rng(bseed)
            yx = np.stack([rng.uniform(grid["Yt"].min() + 30, grid["Yt"].max() - 30, nAll),
                           rng.uniform(grid["Xt"].min() + 30, grid["Xt"].max() - 30, nAll)], axis=1)
            yx0 = np.ascontiguousarray(yx[:int(g["cand_idx"][-1]) + 1].astype(np.float32).astype(np.float64)[g["cand_idx"]])
            fkw = dict(ripple=float(g["fields"][4]))
        else:
            _, yx = syn.make_buoys(grid, nAll, seed=bseed, frac=0.6)
            yx0 = np.ascontiguousarray(yx[g["cand_idx"]] if kind == "c3warp" else yx[:nP])
        del yx
        u, v, sic = syn.make_fields(grid, K=int(K), seed=int(fseed), umax=float(umax), drift=float(drift), **fkw)
        if kind == "c5shape":
            j0, j1, i0, i1 = (int(x) for x in g["polynya"])
            sic[:, j0:j1, i0:i1] = 0.03
        ok = yx0.sum() == float(g["yx0_sum"]) and float(g["sic_sum"]) == float(sic.sum(dtype=np.float64))
        ok = ok and (kind == "c5shape" or (sic == 1).all())
        for q, k in enumerate(g["probe_records"]):
            ok = ok and u[k].astype(np.float64).sum() == g["u_sum"][q] and v[k].astype(np.float64).sum() == g["v_sum"][q]
        assert ok, "the rebuilt workload differs from the one the golden outputs were generated on"
        if kind == "regular":
            assert np.array_equal(syn.regular_host_cell(grid, yx0), g["jiT0"])
        _CUT_CACHE[key] = (grid, u, v, sic, yx0)
    return _CUT_CACHE[key]


def traj_digest_row(pos, msk, jit, alive):
    """one record's digest as tests/golden/gen_golden.py::traj_digest forms it"""
    m = msk == 1
    bits = np.ascontiguousarray(pos[m]).reshape(-1).view(np.uint64)
    return np.array([np.bitwise_xor.reduce(bits) if bits.size else 0, int(m.sum()),
                     int((jit.astype(np.int64) * np.array([100003, 1])).sum() % (1 << 62)), int((alive == 1).sum())], dtype=np.uint64)
