"""Synthetic NEMO-like C-grids, ice-velocity records and buoy seeds.

Host-side numpy only (inputs for tests, the golden generator and bench.py);
nothing here is on the accelerated path.  Shapes and dtypes follow what the
reference driver feeds its hot loop:

* grid arrays `(Nj,Ni)` fp64 in km in the polar-stereographic plane, order
  `[j,i]`, as returned by `GetModelGrid`/`GetModelUVGrid`
  (reference sitrack/ncio.py:22-92);
* `tmask` int8 (ncio.py:40);
* `u_ice`, `v_ice`, `siconc` records `(Nj,Ni)`, stored fp32 like NEMO output and
  promoted to fp64 by the reference on read (si3_part_tracker.py:200-202,372-374).

The layout of SURVEY.md section 8(d): T at (j,i), U at (j,i+1/2), V at (j+1/2,i),
F at (j+1/2,i+1/2) in index space, spacing `dkm`, domain centred on (0,0).
"""
import numpy as np


def _index_to_plane(jj, ii, Nj, Ni, dkm, warp):
    """Smooth map from (fractional) index space to (y,x) [km].

    warp=0 gives the regular axis-aligned grid of SURVEY 8(d); warp>0 shears and
    stretches it so that cells become general (non axis-aligned) quadrangles.
    """
    y = dkm * (jj - 0.5 * (Nj - 1))
    x = dkm * (ii - 0.5 * (Ni - 1))
    if warp:
        y = y + warp * 0.30 * dkm * np.sin(2.0 * np.pi * 1.3 * ii / Ni + 0.3)
        x = x * (1.0 + warp * 0.05 * (jj / Nj)) + warp * 0.20 * dkm * np.sin(2.0 * np.pi * jj / Nj)
    return y, x


def make_grid(Nj, Ni, dkm=4.0, warp=0.0, rim=2, dtype=np.float64):
    """Returns a dict with Yt,Xt,Yf,Xf,Yu,Xu,Yv,Xv (fp64 km), tmask (int8), resol (km)."""
    jj, ii = np.meshgrid(np.arange(Nj, dtype=np.float64), np.arange(Ni, dtype=np.float64), indexing="ij")
    g = {"Nj": Nj, "Ni": Ni, "dkm": dkm, "warp": warp}
    g["Yt"], g["Xt"] = _index_to_plane(jj, ii, Nj, Ni, dkm, warp)
    g["Yu"], g["Xu"] = _index_to_plane(jj, ii + 0.5, Nj, Ni, dkm, warp)
    g["Yv"], g["Xv"] = _index_to_plane(jj + 0.5, ii, Nj, Ni, dkm, warp)
    g["Yf"], g["Xf"] = _index_to_plane(jj + 0.5, ii + 0.5, Nj, Ni, dkm, warp)
    for k in ("Yt", "Xt", "Yu", "Xu", "Yv", "Xv", "Yf", "Xf"):
        g[k] = np.ascontiguousarray(g[k], dtype=dtype)
    tmask = np.ones((Nj, Ni), dtype=np.int8)
    if rim:
        tmask[:rim, :] = 0
        tmask[-rim:, :] = 0
        tmask[:, :rim] = 0
        tmask[:, -rim:] = 0
    g["tmask"] = tmask
    # local resolution like ncio.py:56-57  sqrt(e1t^2+e2t^2) [km]
    g["resol"] = np.full((Nj, Ni), np.sqrt(2.0) * dkm, dtype=np.float64)
    return g


def make_fields(grid, K=8, seed=2024, umax=0.3, drift=0.05, ripple=0.0, dtype=np.float32):
    """K records of (u,v,sic): solid-body rotation + per-record uniform drift.

    u = -Omega (y_U - y_c) + a_k,  v = Omega (x_V - x_c) + b_k,  max|u|,|v| <= umax.
    `ripple` adds a small cell-scale perturbation (so that the nearest-point and
    cell-mean velocity rules differ measurably).  Returned as (K,Nj,Ni) arrays.
    """
    rng = np.random.default_rng(seed)
    ab = rng.uniform(-drift, drift, size=(K, 2))
    Yu, Xv = grid["Yu"], grid["Xv"]
    rmax = max(np.abs(Yu).max(), np.abs(Xv).max())
    omega = (umax - drift - ripple) / rmax        # [m/s per km]
    Nj, Ni = grid["Nj"], grid["Ni"]
    u = np.empty((K, Nj, Ni), dtype=dtype)
    v = np.empty((K, Nj, Ni), dtype=dtype)
    for k in range(K):
        uk = -omega * Yu + ab[k, 0]
        vk = omega * Xv + ab[k, 1]
        if ripple:
            jj, ii = np.meshgrid(np.arange(Nj), np.arange(Ni), indexing="ij")
            uk = uk + ripple * np.sin(0.9 * ii + 0.37 * jj + k)
            vk = vk + ripple * np.cos(0.8 * jj - 0.41 * ii + 2 * k)
        u[k] = uk
        v[k] = vk
    sic = np.ones((K, Nj, Ni), dtype=dtype)
    return u, v, sic


def nearest_t_guess(grid, yx):
    """Index-space guess of the nearest T-point for (y,x) positions (regular part only)."""
    Nj, Ni, dkm = grid["Nj"], grid["Ni"], grid["dkm"]
    j = np.rint(yx[:, 0] / dkm + 0.5 * (Nj - 1)).astype(np.int64)
    i = np.rint(yx[:, 1] / dkm + 0.5 * (Ni - 1)).astype(np.int64)
    return np.stack([np.clip(j, 2, Nj - 3), np.clip(i, 2, Ni - 3)], axis=1)


def nearest_t_plane(grid, yx):
    """Nearest T-point in the (y,x) plane (k-d tree): a good FindContainingCell guess on warped grids."""
    from scipy.spatial import cKDTree
    Nj, Ni = grid["Nj"], grid["Ni"]
    tree = cKDTree(np.stack([grid["Yt"].ravel(), grid["Xt"].ravel()], axis=1))
    _, k = tree.query(yx)
    j, i = np.unravel_index(k, (Nj, Ni))
    return np.stack([np.clip(j, 2, Nj - 3), np.clip(i, 2, Ni - 3)], axis=1).astype(np.int64)


def nearest_t_index(grid, yx, iters=6):
    """Nearest T-point of (y,x) positions on a make_grid() mesh with warp > 0 by inverting _index_to_plane with a few
    fixed-point sweeps (the warp terms move a point by a fraction of a cell and vary over the whole mesh, so the map
    contracts fast): O(nP), no tree -- what bench-size buoy sets (1e7) on warped meshes are seeded with, as the guess
    for the library's own FindContainingCell."""
    Nj, Ni, dkm, warp = grid["Nj"], grid["Ni"], grid["dkm"], grid["warp"]
    y0, x0 = float(grid.get("y_shift", 0.0)), float(grid.get("x_shift", 0.0))
    y, x = yx[:, 0] - y0, yx[:, 1] - x0
    jj = y / dkm + 0.5 * (Nj - 1)
    ii = x / dkm + 0.5 * (Ni - 1)
    for _ in range(iters if warp else 0):
        jj = (y - warp * 0.30 * dkm * np.sin(2.0 * np.pi * 1.3 * ii / Ni + 0.3)) / dkm + 0.5 * (Nj - 1)
        ii = (x - warp * 0.20 * dkm * np.sin(2.0 * np.pi * jj / Nj)) / (dkm * (1.0 + warp * 0.05 * (jj / Nj))) + 0.5 * (Ni - 1)
    j = np.rint(jj).astype(np.int64)
    i = np.rint(ii).astype(np.int64)
    return np.stack([np.clip(j, 2, Nj - 3), np.clip(i, 2, Ni - 3)], axis=1)


def shift_grid(grid, dy, dx):
    """move the whole mesh in the plane (a NANUK4-like mesh does not sit on the pole)"""
    for k in ("Yt", "Yu", "Yv", "Yf"):
        grid[k] = grid[k] + dy
    for k in ("Xt", "Xu", "Xv", "Xf"):
        grid[k] = grid[k] + dx
    grid["y_shift"] = grid.get("y_shift", 0.0) + dy
    grid["x_shift"] = grid.get("x_shift", 0.0) + dx
    return grid


def make_buoys(grid, nP, seed=1234, frac=0.6):
    """Uniform random buoys in the central `frac` of the domain; IDs 1..nP (int64)."""
    rng = np.random.default_rng(seed)
    Nj, Ni, dkm = grid["Nj"], grid["Ni"], grid["dkm"]
    hy = 0.5 * frac * dkm * (Nj - 1)
    hx = 0.5 * frac * dkm * (Ni - 1)
    yx = np.empty((nP, 2), dtype=np.float64)
    yx[:, 0] = rng.uniform(-hy, hy, size=nP)
    yx[:, 1] = rng.uniform(-hx, hx, size=nP)
    ids = np.arange(1, nP + 1, dtype=np.int64)
    return ids, yx


def regular_host_cell(grid, yx):
    """Exact host cell (jT,iT) on the REGULAR grid (warp=0) under the reference's
    IsInsideQuadrangle convention: inside iff y in (ymin,ymax] and x in (xmin,xmax]
    (SURVEY 8a row a6).  Cell (jT,iT) spans F[jT-1,iT-1]..F[jT,iT], i.e.
    y in ((jT-1/2)d,(jT+1/2)d] about the grid centre.  Used to seed bench-size buoy
    sets without an O(nP Nj Ni) search; parity tests use the library's own locate.
    """
    assert grid["warp"] == 0.0
    Nj, Ni = grid["Nj"], grid["Ni"]
    Yf, Xf = grid["Yf"], grid["Xf"]
    yF = np.ascontiguousarray(Yf[:, 0])
    xF = np.ascontiguousarray(Xf[0, :])
    # smallest jf with y <= yF[jf]  -> upper F row of the host cell
    jT = np.searchsorted(yF, yx[:, 0], side="left").astype(np.int64)
    iT = np.searchsorted(xF, yx[:, 1], side="left").astype(np.int64)
    return np.stack([jT, iT], axis=1)
