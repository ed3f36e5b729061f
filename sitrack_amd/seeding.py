"""Idealised seeding on the model grid -- the surface of the reference's seeding generators (`nemoSeed`, `SidfexSeeding`,
`ReadFromSidfexDatFile`; sitrack/tracking.py:331-442) as used by tools/generate_idealized_seeding.py.

`nemoSeed` runs on the GPU (`sitrk_nemo_seed`, sitrack_amd/csrc/sitrk_seed.h): the decision which points of the sub-sampled
mesh carry a seed, their compaction in the reference's output order and their projection to the polar-stereographic plane
happen in three kernels; only the mesh arrays go in and the seed list comes out.  There is no host version."""
import numpy as np


def _same_shape(a, ref):
    return np.shape(a) == np.shape(ref)


def nemoSeed(pmskT, platT, plonT, pIC, khss=1, fmsk_rstrct=[], platF=[], plonF=[], ctx=None, return_yx=False):
    """Seeds on every `khss`-th T-point (reference tracking.py:365-442, same arguments): where the land-sea mask --
    times the optional restriction mask `fmsk_rstrct` -- is 1, north of 55N, under an ice concentration of at least 0.9.
    When F-point coordinates of the mesh's shape are passed, F-points whose four sub-sampled T-neighbours all carry a
    seed are appended.  Returns the (n,2) [lat,lon] array in the reference's order; with `return_yx` also their (n,2)
    [y,x] km in the polar-stereographic plane (what the seeding tool computes next, util.py:394-410)."""
    from .tracking import default_context
    restricted = _same_shape(fmsk_rstrct, pmskT)
    with_f = _same_shape(platF, pmskT) and _same_shape(plonF, pmskT)
    if not (_same_shape(platT, pmskT) and _same_shape(plonT, pmskT) and _same_shape(pIC, pmskT)):
        raise ValueError('ERROR [nemoSeed()]: mask, coordinates and ice concentration must share one shape')
    latlon, yx, _, _ = (ctx or default_context()).nemo_seed(
        pmskT, platT, plonT, pIC, khss=khss, rmask=fmsk_rstrct if restricted else None,
        latF=platF if with_f else None, lonF=plonF if with_f else None)
    return (latlon, yx) if return_yx else latlon


def ReadFromSidfexDatFile(filepath='./sidfexloc.dat'):
    """Text file `id lon lat` per line (reference sitrack/tracking.py:344-361)."""
    import os
    if not os.path.exists(filepath):
        raise FileNotFoundError("'%s' file is missing." % filepath)
    return np.atleast_2d(np.genfromtxt(filepath))


def SidfexSeeding(filepath='./sidfexloc.dat'):
    """Reference sitrack/tracking.py:331-341 -> (n,2) [lat,lon] and the buoy IDs (int)."""
    dat = ReadFromSidfexDatFile(filepath)
    return dat[:, [2, 1]], dat[:, 0].astype(int)
