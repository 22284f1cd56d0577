"""Result files in the shape the reference's plotting scripts consume.

The reference saves `results` (one Dict(:time, :iter, :err_traj, :err_x0) per sweep point,
random_linear_problem.jl:188) and `Ns` with JLD2 (`@save "horizon_comp.jld2" results Ns`,
run_random_linear.jl:125) and `benchmarks/plotting.jl:53-110::comparison_plot(results, Ns, ...)` reads
`res[:time][:, i]` per solver column.  JLD2 files are HDF5 files; what is written here is plain HDF5 with the
same content, one group per sweep point:

    /Ns                         int64 [P]            the sweep values (N_mpc, n or m)
    /results/<i>/time           float64 [steps, C]   ms per solve, column 1 = this library (per instance: the
                                                     batch's step time / batch), further columns optional
    /results/<i>/iter           int64   [steps, C]   iterations (median over the batch in column 1)
    /results/<i>/err_traj, err_x0                    optional

HDF5 stores arrays row-major, Julia reads them column-major, so a Julia reader sees the dimensions reversed:
the arrays are written transposed ([C, steps] in HDF5) -- exactly what JLD2 does (tests/golden/
extract_ref_fixtures.py reads the reference's own files that way).  Five lines of Julia rebuild the Dicts:

    using HDF5
    h = h5open("horizon_comp.h5"); Ns = read(h["Ns"])
    results = [Dict(Symbol(k) => read(h["results/$i/$k"]) for k in keys(h["results/$i"])) for i in 1:length(Ns)]
    comparison_plot(results, Ns, "knot points (N)", legend=("ALTRO-HIP",))      # benchmarks/plotting.jl:53

The writer drives the HDF5 C library through ctypes (libhdf5 ships with this image's conda); there is no
pure-Python fallback: without the library write_results raises.
"""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

_H5 = None
H5F_ACC_TRUNC, H5F_ACC_RDONLY, H5P_DEFAULT, H5S_ALL = 2, 0, 0, 0


def _lib():
    global _H5
    if _H5 is not None:
        return _H5
    cands = [os.environ.get("ALTRO_LIBHDF5"), ctypes.util.find_library("hdf5")]
    cands += sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    for c in cands:
        if not c:
            continue
        try:
            L = C.CDLL(c)
            L.H5open()
            break
        except OSError:
            continue
    else:
        raise RuntimeError("libhdf5 not found (set ALTRO_LIBHDF5); the result writer needs the HDF5 C library")
    hid = C.c_int64
    for name, res, args in (
            ("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Fopen", hid, [C.c_char_p, C.c_uint, hid]),
            ("H5Fclose", C.c_int, [hid]), ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]), ("H5Gclose", C.c_int, [hid]),
            ("H5Screate_simple", hid, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]), ("H5Sclose", C.c_int, [hid]),
            ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ("H5Dopen2", hid, [hid, C.c_char_p, hid]),
            ("H5Dwrite", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]), ("H5Dread", C.c_int, [hid, hid, hid, hid, hid, C.c_void_p]),
            ("H5Dget_space", hid, [hid]), ("H5Sget_simple_extent_ndims", C.c_int, [hid]),
            ("H5Sget_simple_extent_dims", C.c_int, [hid, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
            ("H5Dget_type", hid, [hid]), ("H5Tget_class", C.c_int, [hid]), ("H5Tclose", C.c_int, [hid]), ("H5Dclose", C.c_int, [hid])):
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    L.t_f64 = hid.in_dll(L, "H5T_NATIVE_DOUBLE_g").value
    L.t_i64 = hid.in_dll(L, "H5T_NATIVE_INT64_g").value
    _H5 = L
    return L


def _write(L, loc, name, arr):
    a = np.ascontiguousarray(arr)
    a = a.astype(np.int64) if a.dtype.kind in "iub" else a.astype(np.float64)
    dims = (C.c_uint64 * max(1, a.ndim))(*a.shape)
    sp = L.H5Screate_simple(a.ndim, dims, None)
    ty = L.t_i64 if a.dtype.kind == "i" else L.t_f64
    ds = L.H5Dcreate2(loc, name.encode(), ty, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
    if ds < 0 or L.H5Dwrite(ds, ty, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data_as(C.c_void_p)) < 0:
        raise RuntimeError("HDF5 write of %s failed" % name)
    L.H5Dclose(ds)
    L.H5Sclose(sp)


def write_results(path, sweep_values, results):
    """results: one dict per sweep point as benchmarks.run_* return them ("time": ms per MPC step for the whole
    batch [steps], "iter": [steps, batch], "batch"), or ready-made matrices {"time": [steps, C], "iter": [steps, C]}."""
    L = _lib()
    f = L.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
    if f < 0:
        raise RuntimeError("cannot create " + path)
    try:
        _write(L, f, "Ns", np.asarray(sweep_values, dtype=np.int64))
        g = L.H5Gcreate2(f, b"results", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        for i, r in enumerate(results, start=1):          # 1-based group names: the Julia reader indexes results[i]
            gi = L.H5Gcreate2(g, str(i).encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
            t = np.asarray(r["time"], dtype=np.float64)
            it = np.asarray(r["iter"])
            if t.ndim == 1:                                # batched run: per-solve time = batch step time / batch
                t = (t / float(r.get("batch", 1)))[:, None]
                it = np.median(it.reshape(it.shape[0], -1), axis=1).astype(np.int64)[:, None]
            _write(L, gi, "time", t.T)                     # transposed: Julia sees [steps, C]
            _write(L, gi, "iter", np.asarray(it).T)
            for k in ("err_traj", "err_x0"):
                if k in r:
                    _write(L, gi, k, np.asarray(r[k], dtype=np.float64).T)
            L.H5Gclose(gi)
        L.H5Gclose(g)
    finally:
        L.H5Fclose(f)
    return path


def read_dataset(path, name):
    """Read one dataset back through the HDF5 library (as numpy, HDF5's row-major view)."""
    L = _lib()
    f = L.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise RuntimeError("cannot open " + path)
    try:
        ds = L.H5Dopen2(f, name.encode(), H5P_DEFAULT)
        if ds < 0:
            raise KeyError(name)
        sp = L.H5Dget_space(ds)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (C.c_uint64 * max(1, nd))()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        ty = L.H5Dget_type(ds)
        is_int = L.H5Tget_class(ty) == 0
        out = np.empty(tuple(dims[:nd]), dtype=np.int64 if is_int else np.float64)
        L.H5Dread(ds, L.t_i64 if is_int else L.t_f64, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
        L.H5Tclose(ty)
        L.H5Sclose(sp)
        L.H5Dclose(ds)
        return out
    finally:
        L.H5Fclose(f)
