#!/usr/bin/env python3
"""Write the HDF5 fixtures of tests/test_h5_io.py with libhdf5 ITSELF (the C library h5py wraps), driven through ctypes:

    python tests/golden/make_h5_fixtures.py          # needs a libhdf5.so (here: /opt/conda/lib/libhdf5.so.103, HDF5 1.10.4)

so that `mridc_amd.collections.common.parts.h5lite` is pinned against files it did not write.  Layout follows what h5py produces for a
fastMRI-style volume (tests/collections/reconstruction/fastmri/create_temp_data.py:10-104 of the reference writes such files with
h5py): `kspace` / `sensitivity_map` as the {r, i} compound, a chunked + shuffled + deflated dataset, a boolean enum mask, float32
targets, scalar float64 and variable-length UTF-8 string attributes, the `ismrmrd_header` scalar variable-length string dataset, a
subgroup.  `h5_latest.h5` is the same content written with libver = latest (superblock 3, version-2 object headers, link messages).
The expected arrays are regenerated from the same seed by the test; only the .h5 files are committed."""
import ctypes as C
import glob
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
XML = ("<?xml version=\"1.0\"?><ismrmrdHeader xmlns=\"http://www.ismrm.org/ISMRMRD\"><encoding><encodedSpace><matrixSize><x>12</x><y>10</y>"
       "<z>1</z></matrixSize></encodedSpace><reconSpace><matrixSize><x>8</x><y>6</y><z>1</z></matrixSize></reconSpace><encodingLimits>"
       "<kspace_encoding_step_1><minimum>0</minimum><maximum>8</maximum><center>5</center></kspace_encoding_step_1></encodingLimits>"
       "</encoding></ismrmrdHeader>")


def arrays(seed=7):
    rng = np.random.default_rng(seed)
    S, Cc, H, W = 3, 4, 12, 10
    d = {}
    d["kspace"] = (rng.standard_normal((S, Cc, H, W)) + 1j * rng.standard_normal((S, Cc, H, W))).astype(np.complex64)
    d["sensitivity_map"] = (rng.standard_normal((S, Cc, H, W)) + 1j * rng.standard_normal((S, Cc, H, W))).astype(np.complex64)
    d["mask"] = rng.random((H, W)) < 0.35
    d["reconstruction_rss"] = rng.standard_normal((S, 8, 6)).astype(np.float32)
    d["eta"] = (rng.standard_normal((S, H, W)) + 1j * rng.standard_normal((S, H, W))).astype(np.complex128)
    d["grp/counts"] = rng.integers(-50, 50, (5, 3)).astype(np.int16)
    return d


def main():
    cands = [os.environ.get("LIBHDF5")] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    lib = None
    for c in cands:
        if c and os.path.exists(c):
            lib = C.CDLL(c)
            break
    if lib is None:
        sys.exit("no libhdf5 found (set LIBHDF5)")
    hid, hsz, i = C.c_int64, C.c_uint64, C.c_int
    lib.H5open()
    g = lambda n: hid.in_dll(lib, n).value                                  # noqa: E731   H5T_NATIVE_* are variables behind macros
    for name, res, args in [("H5Fcreate", hid, [C.c_char_p, C.c_uint, hid, hid]), ("H5Pcreate", hid, [hid]), ("H5Screate_simple", hid, [i, C.POINTER(hsz), C.POINTER(hsz)]),
                            ("H5Screate", hid, [i]), ("H5Tcreate", hid, [i, C.c_size_t]), ("H5Tinsert", i, [hid, C.c_char_p, C.c_size_t, hid]),
                            ("H5Tcopy", hid, [hid]), ("H5Tset_size", i, [hid, C.c_size_t]), ("H5Tset_cset", i, [hid, i]), ("H5Tenum_create", hid, [hid]),
                            ("H5Tenum_insert", i, [hid, C.c_char_p, C.c_void_p]),
                            ("H5Dcreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid, hid]), ("H5Dwrite", i, [hid, hid, hid, hid, hid, C.c_void_p]),
                            ("H5Acreate2", hid, [hid, C.c_char_p, hid, hid, hid, hid]), ("H5Awrite", i, [hid, hid, C.c_void_p]),
                            ("H5Gcreate2", hid, [hid, C.c_char_p, hid, hid, hid]), ("H5Pset_chunk", i, [hid, i, C.POINTER(hsz)]),
                            ("H5Pset_deflate", i, [hid, C.c_uint]), ("H5Pset_shuffle", i, [hid]), ("H5Pset_libver_bounds", i, [hid, i, i]),
                            ("H5Fclose", i, [hid]), ("H5Dclose", i, [hid]), ("H5Aclose", i, [hid]), ("H5Gclose", i, [hid]), ("H5Sclose", i, [hid])]:
        f = getattr(lib, name)
        f.restype, f.argtypes = res, args
    F32, F64, I16, I8, CS1 = g("H5T_NATIVE_FLOAT_g"), g("H5T_NATIVE_DOUBLE_g"), g("H5T_NATIVE_SHORT_g"), g("H5T_NATIVE_SCHAR_g"), g("H5T_C_S1_g")
    P_FA, P_DC = g("H5P_CLS_FILE_ACCESS_ID_g"), g("H5P_CLS_DATASET_CREATE_ID_g")

    def ctype(base, size):                      # h5py's complex: compound {r, i}
        t = lib.H5Tcreate(6, 2 * size)
        lib.H5Tinsert(t, b"r", 0, base)
        lib.H5Tinsert(t, b"i", size, base)
        return t

    vstr = lib.H5Tcopy(CS1)
    lib.H5Tset_size(vstr, C.c_size_t(-1).value)          # H5T_VARIABLE
    lib.H5Tset_cset(vstr, 1)                             # UTF-8
    boolt = lib.H5Tenum_create(I8)                       # h5py's bool
    for nm, v in ((b"FALSE", 0), (b"TRUE", 1)):
        lib.H5Tenum_insert(boolt, nm, C.byref(C.c_int8(v)))

    def dset(loc, name, a, typ, dcpl=0):
        dims = (hsz * a.ndim)(*a.shape)
        sp = lib.H5Screate_simple(a.ndim, dims, None)
        d_ = lib.H5Dcreate2(loc, name.encode(), typ, sp, 0, dcpl, 0)
        assert d_ >= 0, name
        a = np.ascontiguousarray(a)
        assert lib.H5Dwrite(d_, typ, 0, 0, 0, a.ctypes.data_as(C.c_void_p)) >= 0
        lib.H5Dclose(d_)
        lib.H5Sclose(sp)

    def attr(loc, name, typ, buf):
        sp = lib.H5Screate(0)
        a_ = lib.H5Acreate2(loc, name.encode(), typ, sp, 0, 0)
        assert lib.H5Awrite(a_, typ, buf) >= 0
        lib.H5Aclose(a_)
        lib.H5Sclose(sp)

    d = arrays()
    for fname, latest in (("h5_fastmri.h5", False), ("h5_latest.h5", True)):
        fapl = lib.H5Pcreate(P_FA)
        if latest:
            lib.H5Pset_libver_bounds(fapl, 2, 2)         # H5F_LIBVER_V110 .. latest in 1.10: enum {EARLIEST 0, V18 1, V110 2}
        path = os.path.join(HERE, fname)
        fid = lib.H5Fcreate(path.encode(), 2, 0, fapl)   # H5F_ACC_TRUNC
        assert fid >= 0
        dset(fid, "kspace", d["kspace"], ctype(F32, 4))
        dc = lib.H5Pcreate(P_DC)
        lib.H5Pset_chunk(dc, 4, (hsz * 4)(1, 2, 5, 10))
        lib.H5Pset_shuffle(dc)
        lib.H5Pset_deflate(dc, 4)
        if not latest:                                   # (version-4 chunk indices of libver latest are not read by h5lite)
            dset(fid, "sensitivity_map", d["sensitivity_map"], ctype(F32, 4), dc)
        else:
            dset(fid, "sensitivity_map", d["sensitivity_map"], ctype(F32, 4))
        dset(fid, "mask", d["mask"].astype(np.int8), boolt)
        dc2 = lib.H5Pcreate(P_DC)
        lib.H5Pset_chunk(dc2, 3, (hsz * 3)(2, 8, 4))
        dset(fid, "reconstruction_rss", d["reconstruction_rss"], F32, 0 if latest else dc2)
        dset(fid, "eta", d["eta"], ctype(F64, 8))
        gid = lib.H5Gcreate2(fid, b"grp", 0, 0, 0)
        dset(gid, "counts", d["grp/counts"], I16)
        lib.H5Gclose(gid)
        # scalar variable-length string dataset (fastMRI's ismrmrd_header) and attributes
        sp = lib.H5Screate(0)
        d_ = lib.H5Dcreate2(fid, b"ismrmrd_header", vstr, sp, 0, 0, 0)
        s = C.c_char_p(XML.encode())
        assert lib.H5Dwrite(d_, vstr, 0, 0, 0, C.byref(s)) >= 0
        lib.H5Dclose(d_)
        lib.H5Sclose(sp)
        attr(fid, "max", F64, C.byref(C.c_double(0.00123)))
        attr(fid, "norm", F64, C.byref(C.c_double(0.0456)))
        for k, v in (("patient_id", "0beefc0ffeeé"), ("acquisition", "AXT2")):
            s = C.c_char_p(v.encode("utf-8"))
            attr(fid, k, vstr, C.byref(s))
        lib.H5Fclose(fid)
        print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
