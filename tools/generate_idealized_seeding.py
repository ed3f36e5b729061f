#!/usr/bin/env python3
"""Seeding file on the model grid -- the flags of the reference tools generate_idealized_seeding.py /
generate_sidfex_seeding.py (-d -m -i -v -k -S -f -N, --lsidfex) minus the coarsening path (-C needs
`gudhi`/`mojito`).  Writes ./nc/sitrack_seeding_<nemoTsi3|nemoTmm|sidfex>_<YYYYMMDD_hh>[_HSS<S>].nc with the schema
of reference ncio.py:131-197.  `--lsidfex 1` seeds from a text file `id lon lat` (reference tools/sidfexloc.dat)."""
import argparse
import os
import sys
from datetime import datetime, timezone

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import sitrack_amd as sit                      # noqa: E402
from sitrack_amd import driver, ncio           # noqa: E402
from sitrack_amd.seeding import nemoSeed, SidfexSeeding       # noqa: E402


def main(argv=None):
    ap = argparse.ArgumentParser(description='SITRACK idealised seeding (MI355X build)')
    ap.add_argument('-d', '--dat0', required=True, help='initial date in the form <YYYY-MM-DD_hh:mm:ss>')
    ap.add_argument('-m', '--fmmm', default=None, help='model `mesh_mask` file of NEMO config used in SI3 run')
    ap.add_argument('--lsidfex', type=int, default=0, help='Switch to 1 for SIDFEX seeding.')
    ap.add_argument('--sidfex-file', default='./sidfexloc.dat', help='text file `id lon lat` (extra; the reference hard-codes ./sidfexloc.dat)')
    ap.add_argument('-i', '--fsi3', default=None, help='output file of SI3 containing sea-ice concentration')
    ap.add_argument('-v', '--nsic', default='siconc', help='name of sea-ice concentration in SI3 file')
    ap.add_argument('-k', '--krec', type=int, default=0, help='use sea-ice concentration at this record')
    ap.add_argument('-S', '--ihss', type=int, default=1, help='horizontal subsampling factor to apply')
    ap.add_argument('-f', '--fmsk', default=None, help='mask (on SI3 model domain) to control seeding region')
    ap.add_argument('-C', '--crsn', type=int, default=0, help='coarsening in km (not supported here)')
    ap.add_argument('-N', '--ncnf', default='NANUK4', help='name of the horizontak NEMO config used')
    ap.add_argument('--device', type=int, default=0)
    a = ap.parse_args(argv)
    if a.crsn >= 1:
        raise SystemExit('-C/--crsn needs the gudhi-based SubSampCloud of `mojito`: out of scope of this build')
    if a.ihss < 1 or a.ihss > 20:
        raise SystemExit('ERROR: chosen horizontal subsampling makes no sense iHSS=%d' % a.ihss)
    seeding_type = 'sidfex' if a.lsidfex == 1 else ('nemoTsi3' if a.fsi3 else 'nemoTmm')
    ctx = sit.Context(a.device)
    if seeding_type == 'sidfex':
        XseedGC, zIDs = SidfexSeeding(a.sidfex_file)
        return write_seeding(ctx, a, seeding_type, XseedGC, zIDs)
    if not a.fmmm:
        raise SystemExit('ERROR: you have to specify a MeshMask file with `-m`')
    imaskt, xlatT, xlonT, xYt, xXt, xYf, xXf, xResKM = ncio.GetModelGrid(a.fmmm, ctx=ctx)
    if a.fsi3:
        rec = ncio.ModelRecords(a.fsi3)
        (xIC,) = rec.fields(a.krec, (a.nsic,))
        rec.close()
        if np.shape(xIC) != np.shape(imaskt):
            raise SystemExit('ERROR: wrong shape for sea-ice concentration read')
    else:
        xIC = np.ones(np.shape(imaskt))
    FSmask = []
    if a.fmsk:
        with ncio._Reader(a.fmsk) as f:
            FSmask = np.array(f.var('tmask'), dtype='i1')
        if np.shape(FSmask) != np.shape(imaskt):
            raise SystemExit('ERROR: `shape(FSmask) != shape(imaskt)`')
    # which points carry a seed, their order and their projection: one call into the library (sitrk_nemo_seed)
    XseedGC, XseedYX = nemoSeed(imaskt, xlatT, xlonT, xIC, khss=a.ihss, fmsk_rstrct=FSmask, ctx=ctx, return_yx=True)
    zIDs = np.arange(1, XseedGC.shape[0] + 1, dtype=int)
    return write_seeding(ctx, a, seeding_type, XseedGC, zIDs, XseedYX)


def write_seeding(ctx, a, seeding_type, XseedGC, zIDs, XseedYX=None):
    nP = XseedGC.shape[0]
    t0 = driver.clock2epoch(a.dat0)
    if XseedYX is None:
        XseedYX = sit.Geo2CartNPSkm1D(XseedGC, ctx=ctx)
    cdate = datetime.fromtimestamp(t0, timezone.utc).strftime("%Y%m%d_%H")
    cextra = '_HSS' + str(a.ihss) if a.ihss > 1 else ''
    foutnc = './nc/sitrack_seeding_' + seeding_type + '_' + cdate + cextra + '.nc'
    ncio.ncSaveCloudBuoys(foutnc, np.array([t0], dtype='i4'), zIDs, XseedYX[None, :, 0], XseedYX[None, :, 1],
                          XseedGC[None, :, 0], XseedGC[None, :, 1], corigin='idealized_seeding', cauthor='generate_idealized_seeding.py')
    print(' *** %d buoys seeded => %s' % (nP, foutnc))
    return foutnc


if __name__ == '__main__':
    main()
