"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/sitrk.h declares, and fails loudly without a GPU (no CPU fallback)."""
import os
import re
import subprocess

import numpy as np
import pytest

import sitrack_amd as sit
from sitrack_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "sitrk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sitrk_[a-z0-9_]+)\s*\(", txt)))


def test_library_builds_and_exports_every_declared_symbol():
    so = _lib.build()
    assert os.path.exists(so)
    names = declared_symbols()
    assert len(names) >= 20
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (sitrk_[a-z0-9_]+)", out))
    assert set(names) <= exported, sorted(set(names) - exported)
    # nothing but the C ABI is exported
    assert all(s.startswith("sitrk_") for s in re.findall(r" T (\S+)", out))
    # the ctypes binding covers exactly the header
    assert sorted(_lib._SIGNATURES) == names
    L = _lib.lib()
    assert L.sitrk_version() == 100


def _build_c_consumer(tmp_path):
    exe = str(tmp_path / "abi_smoke")
    so_dir = os.path.join(ROOT, "sitrack_amd")
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", so_dir, "-l:libsitrk.so",
                    "-Wl,-rpath," + so_dir], check=True)
    return exe


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/sitrk.h compiles as C11 and a C program links against libsitrk.so; without a GPU it fails cleanly."""
    _lib.build()
    exe = _build_c_consumer(tmp_path)
    import torch
    r = subprocess.run([exe], capture_output=True, text=True)
    if torch.cuda.device_count() > 0:        # (device_count does not initialise the GPU in this process; is_available does)
        assert r.returncode == 0, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "no HIP device" in r.stdout, r.stdout + r.stderr


def test_host_mirror_carries_the_names_the_reference_driver_uses():
    """`import sitrack_amd as sit`: every `sit.<name>` si3_part_tracker.py of the reference refers to exists here
    (all but `PlotMesh`: plotting is out of scope), plus the functions its SeedInit is built from."""
    import inspect
    import sitrack_amd as sit
    for name in ("CartNPSkm2Geo1D", "CrossedEdge", "FillValue", "GetModelGrid", "GetModelUVGrid", "GetTimeSpan",
                 "IsInsideQuadrangle", "LoadNCdata", "ModelFileTimeInfo", "NewHostCell", "SeedFileTimeInfo", "SeedInit",
                 "Survive", "UpdtInd4NewCell", "intersect2Seg", "ncSaveCloudBuoys",
                 "NearestPoint", "Haversine", "FindContainingCell", "Geo2CartNPSkm1D", "ConvertGeo2CartesianNPSkm",
                 "ConvertCartesianNPSkm2Geo", "_ccw_"):
        assert hasattr(sit, name), name
    # positional parameters in the reference's order (extras only after them, keyword `ctx`)
    want = {"SeedInit": ["pIDs", "pSG", "pSC", "platT", "plonT", "pYf", "pXf", "pResolKM", "maskT", "xIceConc", "iverbose"],
            "Survive": ["kID", "kjiT", "pmskT", "pIceC", "iverbose"],
            "CrossedEdge": ["pP1", "pP2", "ji4vert", "pY", "pX", "iverbose"],
            "NewHostCell": ["kcross", "pP1", "pP2", "ji4vert", "pY", "pX", "iverbose"],
            "UpdtInd4NewCell": ["knhc", "ji4vert", "kjiT", "iverbose"],
            "IsInsideQuadrangle": ["y", "x", "quad"],
            "intersect2Seg": ["pcA", "pcB", "pcC", "pcD"],
            "NearestPoint": ["pntGcoor", "pLat", "pLon", "rd_found_km", "resolkm", "ji_prv", "np_box_r", "max_itr"],
            "GetTimeSpan": ["dt", "vtime_mod", "iSdA", "iMdA", "iMdB", "iStop", "iverbose"],
            "CartNPSkm2Geo1D": ["pcoorC", "lat0", "lon0"], "GetModelGrid": ["fNCmeshmask", "alsoF"], "GetModelUVGrid": ["fNCmeshmask"]}
    for name, params in want.items():
        got = list(inspect.signature(getattr(sit, name)).parameters)
        assert got[:len(params)] == params, (name, got)


def test_constant_division_identity_on_cpu(tmp_path):
    """The hot loop evaluates `/1000.` as mul + 2 fma (sitrk_geom.h::div1000).  Same IEEE operations on the host:
    random values, the values closest to rounding midpoints, binade edges - all equal to the true quotient."""
    exe = str(tmp_path / "div1000_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "c", "div1000_check.c"), "-lm"],
                   check=True)
    r = subprocess.run([exe, "4000000"], capture_output=True, text=True)
    assert r.returncode == 0 and " 0 mismatches" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_c_consumer_steps_three_buoys(tmp_path):
    r = subprocess.run([_build_c_consumer(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "ok (3 alive)" in r.stdout, r.stdout + r.stderr


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.device_count() > 0:        # (device_count does not initialise the GPU in this process; is_available does)
        pytest.skip("a GPU is present")
    with pytest.raises(sit.SitrkError, match="no HIP device"):
        sit.Context(0)
    with pytest.raises(sit.SitrkError):
        sit.CartNPSkm2Geo1D(np.zeros((3, 2)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "sitrack_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower(), os.path.join(dirpath, f)


def test_only_the_allowed_places_touch_the_oracle():
    """oracle/ is test infrastructure: besides tests/, only __graft_entry__ (build of the checker + smoke) and bench.py's
    cpu_baseline leg may import it."""
    import ast
    offenders = []
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "tests", "oracle", "gpurun_out", "__pycache__", ".pytest_cache")]
        for f in files:
            if not f.endswith(".py"):
                continue
            path = os.path.join(dirpath, f)
            tree = ast.parse(open(path).read())
            for node in ast.walk(tree):
                mods = []
                if isinstance(node, ast.ImportFrom) and node.module:
                    mods = [node.module]
                elif isinstance(node, ast.Import):
                    mods = [a.name for a in node.names]
                if any(m == "oracle" or m.startswith("oracle.") for m in mods):
                    offenders.append(os.path.relpath(path, ROOT))
    assert sorted(set(offenders)) == ["__graft_entry__.py", "bench.py"], offenders
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        uses = any(isinstance(n, ast.ImportFrom) and n.module == "oracle" for n in ast.walk(fn))
        assert uses == fn.name.startswith("cpu_baseline"), fn.name


def test_get_time_span_matches_golden(golden):
    g = golden("g8_timespan.npz")
    vt = g["vtime"]
    for c in g["cases"]:
        sd, stop = int(c[0]), (None if c[1] < 0 else int(c[1]))
        got = sit.GetTimeSpan(3600., vt, sd, vt[0], vt[-1], iStop=stop)
        assert tuple(int(x) for x in got) == tuple(int(x) for x in c[2:])
    with pytest.raises(ValueError):
        sit.GetTimeSpan(3600., vt, int(vt[0]) - 7200, vt[0], vt[-1])


def test_vertices_are_a_function_of_the_cell(golden):
    g = golden("g5_seedinit.npz")
    assert np.array_equal(sit.vertices_of(g["ojiT"]), g["overt"])


@pytest.mark.gpu
def test_nemoseed_matches_reference_golden(golden):
    """G10 (made by the reference's own nemoSeed) through the C ABI on the GPU: which seeds, in which order -- bit for bit --
    plus the fused projection against the projection entry point; sub-sampling that does not divide the mesh, the
    restriction mask, F-points, NaN inputs and non-0/1 mask values like numpy treats them."""
    from sitrack_amd.seeding import nemoSeed
    g = golden("g10_nemoseed.npz")
    cases = {"a": dict(khss=1), "b": dict(khss=3), "c": dict(khss=2, fmsk_rstrct=g["rmask"]),
             "d": dict(khss=1, platF=g["latF"], plonF=g["lonF"]),
             "e": dict(khss=4, fmsk_rstrct=g["rmask"], platF=g["latF"], plonF=g["lonF"])}
    ctx = sit.Context(0)
    try:
        for tag, kw in cases.items():
            got, yx = nemoSeed(g["tmask"], g["lat"], g["lon"], g["sic"], ctx=ctx, return_yx=True, **kw)
            assert np.array_equal(got, g["seed_" + tag]), tag
            assert len(got) > 0 and np.array_equal(yx, ctx.geo2cart(got))
        # what numpy does with odd inputs: a NaN latitude / concentration does not satisfy `<`, so the point stays;
        # mask values other than 0/1 only seed where the product is exactly 1; F-points need 4 <= sum < 8
        rng = np.random.default_rng(3)
        Nj, Ni = 29, 23
        lat = 50. + 40. * rng.random((Nj, Ni)); lon = 360. * rng.random((Nj, Ni))
        sic = rng.choice([0.5, 0.95, np.nan], size=(Nj, Ni)); lat[rng.random((Nj, Ni)) < 0.1] = np.nan
        tm = rng.choice([0, 1, 1, 1, 2], size=(Nj, Ni)).astype('i1'); rm = rng.choice([0, 1, 1, -1], size=(Nj, Ni)).astype('i1')
        for khss in (1, 2, 5):
            m = (tm[::khss, ::khss] * rm[::khss, ::khss]).astype('i1')
            with np.errstate(invalid='ignore'):
                m[lat[::khss, ::khss] < 55.] = 0
                m[sic[::khss, ::khss] < 0.9] = 0
            wantT = np.stack([lat[::khss, ::khss][m == 1], lon[::khss, ::khss][m == 1]], axis=1)
            mf = np.zeros_like(m)
            mf[1:-1, 1:-1] = (m[2:, 1:-1] + m[1:-1, 2:] + m[:-2, 1:-1] + m[1:-1, :-2]) / 4
            wantF = np.stack([(lat + 0.01)[::khss, ::khss][mf == 1], (lon + 0.02)[::khss, ::khss][mf == 1]], axis=1)
            ll, yx, nT, nF = ctx.nemo_seed(tm, lat, lon, sic, khss=khss, rmask=rm, latF=lat + 0.01, lonF=lon + 0.02)
            assert (nT, nF) == (len(wantT), len(wantF)) and np.array_equal(ll, np.concatenate([wantT, wantF]), equal_nan=True)
    finally:
        ctx.close()


def test_roofline_goes_stale_when_the_kernel_changes():
    """the roofline's constants come from counter passes of ONE binary: another fingerprint (an edited kernel) flips `stale`"""
    import bench
    shipped = bench.isa_fingerprint()
    assert shipped and len(shipped["sha256"]) == 16                   # written by the build next to the library
    assert bench.roofline_is_stale(dict(shipped), shipped) is False
    edited = dict(shipped, sha256="0" * 16)
    assert bench.roofline_is_stale(edited, shipped) is True and bench.roofline_is_stale(None, shipped) is True
    assert bench.roofline_is_stale(shipped, None) is True


def test_box_arithmetic_of_the_binding():
    """`Context.box_of`: the box a record's next steps can touch from the buoys' extreme host cells -- [jmin-2-age, jmax+3+age) x
    [imin-2-age, imax+3+age), clipped to the mesh, columns widened to 16-byte lines (host arithmetic only: no device needed)."""
    class Mesh:
        Nj, Ni = 100, 64
    box_of = _lib.Context.box_of
    assert box_of(Mesh, 10, 20, 9, 30, 0) == (8, 23, 4, 36)            # columns [7, 33) -> [4, 36)
    assert box_of(Mesh, 10, 20, 10, 29, 0, align=1) == (8, 23, 8, 32)
    assert box_of(Mesh, 10, 20, 9, 30, 5) == (3, 28, 0, 40)
    assert box_of(Mesh, 1, 98, 1, 62, 0) == (0, 100, 0, 64)            # clipped
    assert box_of(Mesh, 50, 50, 61, 62, 7) == (41, 60, 52, 64)
    assert box_of(Mesh, 1, 0, 1, 0, 3) == (0, 0, 0, 0)                 # no live buoy


def test_kernel_fingerprints_are_written_next_to_the_library():
    """the build leaves the fingerprint of the kernels as shipped next to libsitrk.so (tools/kernel_fingerprint.py): what bench.py
    ties its roofline constants to"""
    import json
    _lib.build()
    d = json.load(open(os.path.splitext(_lib.SO_PATH)[0] + ".isa.json"))
    assert d["arch"] == "gfx950" and len(d["kernels"]) == 4
    for name, k in d["kernels"].items():
        assert k is not None and len(k["sha256"]) == 16 and k["static"]["valu"] > 50 and k["code_bytes"] > 1000, name
    fused = d["kernels"]["advect_run_kernel<float,1,false>"]["static"]
    assert fused["valu64"] > 0.4 * fused["valu"] and fused["lds"] > 10 and fused["vmem"] > 10


def test_nearest_t_index_agrees_with_the_tree_on_warped_meshes():
    """`synthetic.nearest_t_index` (what bench.py seeds 1e7 buoys on curvilinear meshes with, as the guess of FindContainingCell) against
    the k-d tree: the same T-point, or a direct neighbour where two are nearly equidistant"""
    from sitrack_amd import synthetic as syn
    for g in (syn.make_grid(300, 260, dkm=4.0, warp=1.0), syn.shift_grid(syn.make_grid(566, 492, dkm=12.5, warp=1.0), -250., 150.)):
        rng = np.random.default_rng(2)
        yx = np.stack([rng.uniform(g["Yt"].min() + 30, g["Yt"].max() - 30, 20000), rng.uniform(g["Xt"].min() + 30, g["Xt"].max() - 30, 20000)], axis=1)
        a, b = syn.nearest_t_plane(g, yx), syn.nearest_t_index(g, yx)
        assert np.abs(a - b).max() <= 1 and (a != b).any(axis=1).mean() < 0.02
