"""Minimal stand-in for the `netCDF4` package (absent from this image), just enough of its API for
sitrack_amd/ncio.py's netCDF4 branch to be exercised by tests: Dataset(path[, 'w', format=]), createDimension,
createVariable(name, type, dims, fill_value=, zlib=, complevel=), variables[...] with slicing, attributes and
set_auto_mask, dimensions[name].size, close.  Storage is NetCDF-3 through scipy (int64 is kept as float64)."""
import numpy as np
from scipy.io import netcdf_file


class _Dim:
    def __init__(self, size):
        self.size = size


class _Var:
    def __init__(self, v, ds, name):
        object.__setattr__(self, "_v", v)
        object.__setattr__(self, "_ds", ds)
        object.__setattr__(self, "_name", name)

    def set_auto_mask(self, flag):
        pass

    def __getitem__(self, idx):
        return np.array(self._v[idx])

    def __setitem__(self, idx, val):
        self._v[idx] = val

    def __getattr__(self, k):
        a = getattr(self._v, k)
        return a.decode() if isinstance(a, bytes) else a

    def __setattr__(self, k, val):
        setattr(self._v, k, val)


class Dataset:
    def __init__(self, path, mode="r", format=None):
        self._f = netcdf_file(path, "w", version=2) if mode == "w" else netcdf_file(path, "r", mmap=False, maskandscale=False)
        self.created = []

    @property
    def variables(self):
        return {k: _Var(v, self, k) for k, v in self._f.variables.items()}

    @property
    def dimensions(self):
        return {k: _Dim(self._f._recs if n is None else n) for k, n in self._f.dimensions.items()}

    def createDimension(self, name, size):
        self._f.createDimension(name, size)

    def createVariable(self, name, typ, dims, fill_value=None, zlib=False, complevel=4):
        self.created.append((name, typ, dims, fill_value, zlib, complevel))
        v = self._f.createVariable(name, 'f8' if typ == 'i8' else typ, dims)
        if fill_value is not None:
            v._FillValue = np.array(fill_value, dtype=typ)
        return _Var(v, self, name)

    def __setattr__(self, k, val):
        if k in ("_f", "created"):
            object.__setattr__(self, k, val)
        else:
            setattr(self._f, k, val)

    def close(self):
        self._f.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
