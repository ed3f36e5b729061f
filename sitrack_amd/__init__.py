"""sitrack_amd -- MI355X-native per-buoy advection for the `sitrack` sea-ice tracker.

Import as `import sitrack_amd as sit`: the hot-path functions keep the reference's
names (`sit.SeedInit`, `sit.CartNPSkm2Geo1D`, `sit.GetTimeSpan`, ...).
"""
from ._lib import Context, SitrkError, FillValue, build, lib, SO_PATH      # noqa: F401
from .tracking import (SeedInit, FindContainingCell, CartNPSkm2Geo1D, Geo2CartNPSkm1D, GetTimeSpan,  # noqa: F401
                       ConvertGeo2CartesianNPSkm, ConvertCartesianNPSkm2Geo,
                       IceTracker, vertices_of, default_context, rmin_conc, rFoundKM)
from .predicates import (_ccw_, intersect2Seg, IsInsideQuadrangle, CrossedEdge, NewHostCell, UpdtInd4NewCell,  # noqa: F401
                         Survive, Haversine, NearestPoint)
from .ncio import (GetModelGrid, GetModelUVGrid, LoadNCtime, LoadNCdata, SeedFileTimeInfo, ModelFileTimeInfo,  # noqa: F401
                   ncSaveCloudBuoys, chck4f)
from . import synthetic                                                      # noqa: F401
