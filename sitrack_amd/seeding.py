"""Idealised seeding on the model grid -- vectorised mirror of reference `nemoSeed`
(sitrack/tracking.py:365-442), used by tools/generate_idealized_seeding.py.  Host-side numpy
(the reference fills its output in a Python loop over the kept points); the forward projection
of the seeds goes through libsitrk (`Geo2CartNPSkm1D`)."""
import numpy as np


def nemoSeed(pmskT, platT, plonT, pIC, khss=1, fmsk_rstrct=[], platF=[], plonF=[]):
    """Every `khss`-th T-point north of 55N with ice concentration >= 0.9 inside the (optional)
    restriction mask, in C order; optionally followed by the F-points whose four neighbouring
    (sub-sampled) T-points are all kept.  Returns (n,2) [lat,lon]."""
    lAddF = (np.shape(platF) == np.shape(pmskT) and np.shape(plonF) == np.shape(pmskT))
    zmsk = pmskT[::khss, ::khss]
    zlat = platT[::khss, ::khss]
    zlon = plonT[::khss, ::khss]
    (Nj, Ni) = np.shape(zmsk)
    msk_T = np.zeros((Nj, Ni), dtype='i1')
    msk_T[:, :] = zmsk[:, :]
    if np.shape(fmsk_rstrct) == np.shape(pmskT):
        maskR = fmsk_rstrct[::khss, ::khss]
        if np.shape(maskR) != (Nj, Ni):
            raise ValueError('ERROR [nemoSeed()]: restricted area mask does not agree in shape with model output!')
        msk_T[:, :] = msk_T[:, :] * maskR[:, :]
    msk_T[zlat < 55.] = 0                                   # only north of 55N (:406)
    ztmp = np.zeros((Nj, Ni))
    ztmp[:, :] = pIC[::khss, ::khss]
    msk_T[ztmp < 0.9] = 0                                   # only over a decent concentration of ice (:411-412)
    keep = msk_T == 1
    zLatLon = np.stack([zlat[keep], zlon[keep]], axis=1).astype(np.float64)
    if lAddF:
        zlatF = platF[::khss, ::khss]
        zlonF = plonF[::khss, ::khss]
        msk_F = np.zeros(np.shape(msk_T), dtype='i1')
        msk_F[1:-1, 1:-1] = (msk_T[2:, 1:-1] + msk_T[1:-1, 2:] + msk_T[:-2, 1:-1] + msk_T[1:-1, :-2]) / 4
        keepF = msk_F == 1
        zLatLon = np.concatenate([zLatLon, np.stack([zlatF[keepF], zlonF[keepF]], axis=1).astype(np.float64)])
    return zLatLon


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
