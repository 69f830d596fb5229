"""A small HDF5 reader / writer (numpy + zlib only) for the on-disk side of the reconstruction path: fastMRI-style volumes in
(`kspace`, `sensitivity_map`, `mask`, `eta`, `reconstruction_*` datasets, file attributes, the `ismrmrd_header` string) and the
`reconstruction` dataset out.  It stands where the reference calls h5py (data/mri_data.py:247-318, common/parts/utils.py:275-290,
models/base.py:521-587); h5py is not a dependency of this package.

The interface is the slice of h5py those call sites use: `File(path, "r" | "w")` as a context manager, `key in f`, `f.keys()`,
`f[key]` -> `Dataset` (`.shape`, `.dtype`, `.ndim`, `ds[...]` with numpy basic indexing, `np.asarray(ds)`), `f.attrs` (a dict), groups
nest, `f.create_dataset(name, data=...)`, `f.attrs[name] = value` before close.

Reads: superblock versions 0-3, version-1 and version-2 object headers (with continuation blocks), old-style groups (symbol table:
B-tree v1 + local heap) and new-style groups with compact link storage, contiguous / compact / chunked (B-tree v1 index) layouts with
the deflate and shuffle filters, fixed-point, floating-point, fixed and variable-length strings (global heap), enums (h5py booleans)
and the {r, i} compound h5py / PyTables use for complex numbers, attributes of those types.  Anything else raises NotImplementedError
naming the feature.  Writes: superblock 0, one root group, contiguous datasets, scalar / 1-D numeric and string attributes -- what
`save_reconstructions` and the test fixtures need -- in the layout libhdf5 itself produces, so h5py, h5dump and MATLAB read the files."""
import mmap
import os
import struct
import zlib

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(OSError):
    pass


def _u(buf, off, n):
    return int.from_bytes(buf[off:off + n], "little")


def _zero(buf, off):
    """Offset of the first zero byte at or after `off`."""
    while buf[off] != 0:
        off += 1
    return off


# ---- datatypes ------------------------------------------------------------------------------------------------------------------------
class _Type:
    """Parsed datatype message: numpy dtype for fixed-size elements, or a variable-length string marker."""

    def __init__(self, dtype=None, vlen_str=False, size=0, complex_of=None):
        self.dtype, self.vlen_str, self.size, self.complex_of = dtype, vlen_str, size, complex_of


def _parse_type(buf, off):
    """Returns (_Type, bytes consumed)."""
    cv = buf[off]
    cls, ver = cv & 0x0F, cv >> 4
    b0, b1 = buf[off + 1], buf[off + 2]
    size = _u(buf, off + 4, 4)
    p = off + 8
    if cls == 0:                                     # fixed point
        order = ">" if b0 & 1 else "<"
        kind = "i" if b0 & 8 else "u"
        return _Type(np.dtype(f"{order}{kind}{size}"), size=size), 8 + 4
    if cls == 1:                                     # floating point (IEEE layouts only)
        order = ">" if b0 & 1 else "<"
        if size not in (2, 4, 8):
            raise NotImplementedError(f"h5lite: {size}-byte floating-point type")
        return _Type(np.dtype(f"{order}f{size}"), size=size), 8 + 12
    if cls == 3:                                     # fixed-length string
        return _Type(np.dtype(f"S{size}"), size=size), 8
    if cls == 6:                                     # compound: only {r, i} of two equal floats (complex numbers)
        n = b0 | (b1 << 8)
        members = []
        for _ in range(n):
            e = _zero(buf, p)
            name = bytes(buf[p:e]).decode()
            if ver < 3:
                p += (e - p + 8) // 8 * 8            # null-terminated, padded to a multiple of 8
                moff = _u(buf, p, 4)
                p += 4
                if ver == 1:
                    p += 1 + 3 + 4 + 4 + 16          # dimensionality, reserved, permutation, reserved, 4 dimension sizes
            else:
                p = e + 1
                nb = 1 if size < 256 else (2 if size < 65536 else (3 if size < 16777216 else 4))
                moff = _u(buf, p, nb)
                p += nb
            mt, used = _parse_type(buf, p)
            p += used
            members.append((name, moff, mt))
        names = [m[0] for m in members]
        if (sorted(names) == ["i", "r"] and members[0][2].dtype is not None and members[0][2].dtype.kind == "f"
                and members[0][2].dtype == members[1][2].dtype and size == 2 * members[0][2].size):
            r = [m for m in members if m[0] == "r"][0]
            if r[1] != 0:
                raise NotImplementedError("h5lite: complex compound with the imaginary part first")
            base = r[2].dtype
            return _Type(np.dtype(f"{base.byteorder if base.byteorder != '=' else '<'}c{size}"), size=size, complex_of=base), p - off
        dt = np.dtype({"names": names, "formats": [m[2].dtype for m in members], "offsets": [m[1] for m in members], "itemsize": size})
        return _Type(dt, size=size), p - off
    if cls == 8:                                     # enum (h5py stores bool as an int8 enum): the base integer
        n = b0 | (b1 << 8)
        base, used = _parse_type(buf, p)
        p += used
        for _ in range(n):
            e = _zero(buf, p)
            p = p + (e - p + 8) // 8 * 8 if ver < 3 else e + 1
        p += n * base.size
        return _Type(base.dtype, size=size), p - off
    if cls == 9:                                     # variable length: strings only
        base, used = _parse_type(buf, p)
        if (b0 & 0x0F) != 1:
            raise NotImplementedError("h5lite: variable-length sequences (only variable-length strings are read)")
        return _Type(None, vlen_str=True, size=size), 8 + used
    raise NotImplementedError(f"h5lite: datatype class {cls}")


def _parse_space(buf, off):
    ver, rank, flags = buf[off], buf[off + 1], buf[off + 2]
    if ver == 1:
        p = off + 8
    elif ver == 2:
        if buf[off + 3] == 2:                        # null dataspace
            return None
        p = off + 4
    else:
        raise NotImplementedError(f"h5lite: dataspace message version {ver}")
    return tuple(_u(buf, p + 8 * i, 8) for i in range(rank))


# ---- reading --------------------------------------------------------------------------------------------------------------------------
class _Reader:
    def __init__(self, path):
        # the file image is memory-mapped, never read: metadata walks touch a few pages, a slice read touches that slice's bytes (the mapping
        # lives as long as an array sliced out of it does; it is not closed explicitly)
        with open(path, "rb") as f:
            size = os.fstat(f.fileno()).st_size
            self.buf = memoryview(mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)) if size else memoryview(b"")
        b = self.buf
        base = 0
        while bytes(b[base:base + 8]) != _SIG:       # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(b):
                raise H5Error(f"{path}: not an HDF5 file (no superblock signature)")
        ver = b[base + 8]
        if ver in (0, 1):
            if b[base + 13] != 8 or b[base + 14] != 8:
                raise NotImplementedError("h5lite: files with offsets / lengths other than 8 bytes")
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = _u(b, p, 8)
            self.root = _u(b, p + 32 + 8, 8)         # root symbol-table entry: link name offset, object header address, ...
        elif ver in (2, 3):
            if b[base + 9] != 8 or b[base + 10] != 8:
                raise NotImplementedError("h5lite: files with offsets / lengths other than 8 bytes")
            self.base = _u(b, base + 12, 8)
            self.root = _u(b, base + 12 + 24, 8)
        else:
            raise NotImplementedError(f"h5lite: superblock version {ver}")
        self._gcol = {}

    # -- object headers ------------------------------------------------------------------------------------------------------------
    def messages(self, addr):
        """[(type, data offset, size, flags)] of the object header at `addr` (absolute offsets into the file image)."""
        b, a = self.buf, self.base + addr
        out = []
        if bytes(b[a:a + 4]) == b"OHDR":             # version 2
            flags = b[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            nb = 1 << (flags & 3)
            size0 = _u(b, p, nb)
            p += nb
            blocks = [(p, p + size0)]
            track = bool(flags & 4)
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 <= end:
                    t, sz, fl = b[p], _u(b, p + 1, 2), b[p + 3]
                    p += 4 + (2 if track else 0)
                    if t == 0x10:
                        ca, cl = self.base + _u(b, p, 8), _u(b, p + 8, 8)
                        blocks.append((ca + 4, ca + cl - 4))     # "OCHK" + messages + checksum
                    elif t != 0:
                        out.append((t, p, sz, fl))
                    p += sz
            return out
        if b[a] != 1:
            raise H5Error(f"h5lite: unknown object header at {addr:#x}")
        nmsg, size0 = _u(b, a + 2, 2), _u(b, a + 8, 4)
        blocks = [(a + 16, a + 16 + size0)]
        while blocks and len(out) < nmsg + 64:
            p, end = blocks.pop(0)
            while p + 8 <= end:
                t, sz, fl = _u(b, p, 2), _u(b, p + 2, 2), b[p + 4]
                p += 8
                if t == 0x10:
                    blocks.append((self.base + _u(b, p, 8), self.base + _u(b, p, 8) + _u(b, p + 8, 8)))
                elif t != 0:
                    out.append((t, p, sz, fl))
                p += sz
        return out

    # -- groups ------------------------------------------------------------------------------------------------------------------------
    def links(self, addr):
        """{name: object header address} of the group at `addr`."""
        b = self.buf
        out = {}
        for t, p, sz, fl in self.messages(addr):
            if t == 0x11:                            # symbol table: B-tree v1 + local heap
                btree, heap = _u(b, p, 8), _u(b, p + 8, 8)
                h = self.base + heap
                if bytes(b[h:h + 4]) != b"HEAP":
                    raise H5Error("h5lite: bad local heap")
                data = self.base + _u(b, h + 24, 8)
                self._walk_group_btree(btree, data, out)
            elif t == 0x06:                          # link message
                ver, lf = b[p], b[p + 1]
                q = p + 2
                ltype = 0
                if lf & 8:
                    ltype = b[q]
                    q += 1
                if lf & 4:
                    q += 8
                if lf & 16:
                    q += 1
                nb = 1 << (lf & 3)
                ln = _u(b, q, nb)
                q += nb
                name = bytes(b[q:q + ln]).decode()
                q += ln
                if ltype != 0:
                    continue                         # soft / external links are not followed
                out[name] = _u(b, q, 8)
            elif t == 0x02:                          # link info: dense storage when the fractal heap address is defined
                ver, lf = b[p], b[p + 1]
                q = p + 2 + (8 if lf & 1 else 0)
                if _u(b, q, 8) != _UNDEF:
                    raise NotImplementedError("h5lite: groups with dense link storage (fractal heap)")
        return out

    def _walk_group_btree(self, addr, heap_data, out):
        b, a = self.buf, self.base + addr
        if bytes(b[a:a + 4]) != b"TREE" or b[a + 4] != 0:
            raise H5Error("h5lite: bad group B-tree node")
        level, n = b[a + 5], _u(b, a + 6, 2)
        p = a + 24
        for i in range(n):
            child = _u(b, p + 8, 8)                  # key i (8), child i (8), ...
            p += 16
            if level > 0:
                self._walk_group_btree(child, heap_data, out)
                continue
            s = self.base + child
            if bytes(b[s:s + 4]) != b"SNOD":
                raise H5Error("h5lite: bad symbol table node")
            for k in range(_u(b, s + 6, 2)):
                e = s + 8 + 40 * k
                no = heap_data + _u(b, e, 8)
                out[bytes(b[no:no + self._strlen(no)]).decode()] = _u(b, e + 8, 8)

    def _strlen(self, off):
        b, n = self.buf, 0
        while b[off + n] != 0:
            n += 1
        return n

    # -- global heap (variable-length strings) -----------------------------------------------------------------------------------------
    def gheap(self, addr, index):
        col = self._gcol.get(addr)
        if col is None:
            b, a = self.buf, self.base + addr
            if bytes(b[a:a + 4]) != b"GCOL":
                raise H5Error("h5lite: bad global heap collection")
            end = a + _u(b, a + 8, 8)
            p, col = a + 16, {}
            while p + 16 <= end:
                idx, sz = _u(b, p, 2), _u(b, p + 8, 8)
                if idx == 0:
                    break
                col[idx] = bytes(b[p + 16:p + 16 + sz])
                p += 16 + (sz + 7) // 8 * 8
            self._gcol[addr] = col
        return col[index]

    def vlen_strings(self, raw, count):
        out = []
        for i in range(count):
            ln, addr, idx = _u(raw, 16 * i, 4), _u(raw, 16 * i + 4, 8), _u(raw, 16 * i + 12, 4)
            out.append(self.gheap(addr, idx)[:ln].decode("utf-8", "replace") if ln and addr not in (0, _UNDEF) else "")
        return out

    # -- attributes --------------------------------------------------------------------------------------------------------------------
    def attributes(self, addr):
        b = self.buf
        out = {}
        for t, p, sz, fl in self.messages(addr):
            if t != 0x0C:
                continue
            ver = b[p]
            nsz, tsz, ssz = _u(b, p + 2, 2), _u(b, p + 4, 2), _u(b, p + 6, 2)
            q = p + 8 + (1 if ver == 3 else 0)
            pad = (lambda v: (v + 7) // 8 * 8) if ver == 1 else (lambda v: v)
            name = bytes(b[q:q + nsz]).split(b"\x00")[0].decode()
            q += pad(nsz)
            typ, _ = _parse_type(b, q)
            q += pad(tsz)
            shape = _parse_space(b, q)
            q += pad(ssz)
            out[name] = self._decode(typ, shape, b[q:p + sz], scalar_ok=True)
        return out

    def _decode(self, typ, shape, raw, scalar_ok=False):
        if shape is None:
            return None
        count = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if typ.vlen_str:
            vals = self.vlen_strings(raw, count)
            return vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape)
        arr = np.frombuffer(raw, dtype=typ.dtype, count=count).reshape(shape)
        if typ.dtype.kind == "S":
            arr = np.array([v.split(b"\x00")[0] for v in arr.reshape(-1)], dtype=object).reshape(shape)
            return arr[()].decode("utf-8", "replace") if shape == () else arr
        if arr.dtype.byteorder == ">":
            arr = arr.astype(arr.dtype.newbyteorder("="))
        return arr[()] if (shape == () and scalar_ok) else arr


class Dataset:
    """A dataset opened for reading.  `ds[i]` / `ds[i:j]` (and tuples starting that way) decode only the rows of the leading axis they ask
    for: a byte range of the memory-mapped image for contiguous storage, the chunks that intersect the rows for chunked storage -- one
    slice of a multi-GB fastMRI volume costs one slice of I/O, as with h5py.  Any other key reads the whole array once (cached) and
    indexes it.  Contiguous storage is sliced from the file image without a copy until `astype` / arithmetic."""

    def __init__(self, reader, addr, name):
        self._r, self._addr, self.name = reader, addr, name
        b = reader.buf
        self._layout = self._filters = None
        self.shape, self._type = (), None
        for t, p, sz, fl in reader.messages(addr):
            if t == 0x01:
                self.shape = _parse_space(b, p)
            elif t == 0x03:
                self._type, _ = _parse_type(b, p)
            elif t == 0x08:
                self._layout = (p, sz)
            elif t == 0x0B:
                self._filters = p
        if self._type is None or self._layout is None:
            raise H5Error(f"h5lite: {name} is not a dataset")
        self._cache = None

    @property
    def dtype(self):
        return np.dtype(object) if self._type.vlen_str else self._type.dtype.newbyteorder("=")

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def attrs(self):
        return self._r.attributes(self._addr)

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of a scalar dataset")
        return self.shape[0]

    def __iter__(self):
        return iter(self[()])

    def __array__(self, dtype=None, copy=None):
        a = self[()]
        return a.astype(dtype) if dtype is not None else a

    def __getitem__(self, key):
        rows = self._leading_rows(key)
        if rows is not None:
            lo, hi, rest = rows
            part = self._read(lo, hi)
            return part[rest] if isinstance(part, np.ndarray) else part
        a = self._read()
        return a[key] if isinstance(a, np.ndarray) else a

    def _leading_rows(self, key):
        """(lo, hi, key to apply to rows lo .. hi) when `key` selects a unit-stride run of the leading axis of an uncached numeric
        dataset of rank >= 1, else None."""
        if self._cache is not None or not self.shape or self._type.vlen_str or self._type.dtype.kind == "S":
            return None
        lead, tail = (key[0], tuple(key[1:])) if isinstance(key, tuple) and key else (key, ())
        n = self.shape[0]
        if isinstance(lead, (int, np.integer)) and not isinstance(lead, (bool, np.bool_)):
            i = int(lead) + (n if lead < 0 else 0)
            if not 0 <= i < n:
                raise IndexError(f"index {int(lead)} is out of bounds for axis 0 with size {n}")
            return i, i + 1, (0,) + tail
        if isinstance(lead, slice):
            lo, hi, step = lead.indices(n)
            if step != 1 or hi <= lo or (lo == 0 and hi == n):
                return None
            return lo, hi, (slice(None),) + tail
        return None

    def _filter_list(self):
        b, p = self._r.buf, self._filters
        if p is None:
            return []
        ver, n = b[p], b[p + 1]
        q = p + (8 if ver == 1 else 2)
        out = []
        for _ in range(n):
            fid = _u(b, q, 2)
            if ver == 1 or fid >= 256:
                nlen = _u(b, q + 2, 2)
                q += 2
            else:
                nlen = 0
            nvals = _u(b, q + 4, 2)
            q += 6
            q += (nlen + 7) // 8 * 8 if ver == 1 else nlen
            vals = [_u(b, q + 4 * i, 4) for i in range(nvals)]
            q += 4 * nvals
            if ver == 1 and nvals % 2:
                q += 4
            out.append((fid, vals))
        return out

    def _read(self, lo=None, hi=None):
        """The whole array (cached), or -- lo / hi given -- rows lo .. hi of the leading axis (not cached)."""
        if self._cache is not None:
            return self._cache if lo is None else self._cache[lo:hi]
        r, b = self._r, self._r.buf
        p, sz = self._layout
        ver = b[p]
        typ, shape = self._type, self.shape
        if shape is None:
            return None
        esize = typ.size
        partial = lo is not None
        rowbytes = esize * int(np.prod(shape[1:], dtype=np.int64)) if shape else esize
        skip = lo * rowbytes if partial else 0
        if partial:
            shape = (hi - lo,) + tuple(shape[1:])
        nbytes = esize * (int(np.prod(shape, dtype=np.int64)) if shape else 1)
        if ver in (3, 4):
            cls = b[p + 1]
            if ver == 4 and cls == 2:
                raise NotImplementedError("h5lite: version-4 chunk indices (chunked datasets written with libver='latest')")
            if cls == 0:
                raw = b[p + 4 + skip:p + 4 + _u(b, p + 2, 2)]
            elif cls == 1:
                addr = _u(b, p + 2, 8)
                raw = bytes(nbytes) if addr == _UNDEF else b[r.base + addr + skip:r.base + addr + skip + nbytes]
            elif cls == 2:
                nd = b[p + 2]
                btree = _u(b, p + 3, 8)
                chunk = tuple(_u(b, p + 11 + 4 * i, 4) for i in range(nd - 1))
                raw = self._read_chunked(btree, chunk, esize, lo, hi)
            else:
                raise NotImplementedError(f"h5lite: data layout class {cls}")
        elif ver in (1, 2):
            nd, cls = b[p + 1], b[p + 2]
            if cls != 1:
                raise NotImplementedError("h5lite: version-1/2 layout messages other than contiguous")
            addr = _u(b, p + 8, 8)
            raw = b[r.base + addr + skip:r.base + addr + skip + nbytes]
        else:
            raise NotImplementedError(f"h5lite: data layout message version {ver}")
        out = r._decode(typ, shape, raw)
        if not partial:
            self._cache = out
        return out

    def _read_chunked(self, btree, chunk, esize, lo=None, hi=None):
        """Raw bytes of the array -- of rows lo .. hi of its leading axis when given: only the chunks that intersect them are inflated."""
        r, b = self._r, self._r.buf
        row0, row1 = (0, self.shape[0]) if lo is None else (lo, hi)
        shape = (row1 - row0,) + tuple(self.shape[1:])
        filters = self._filter_list()
        for fid, _ in filters:
            if fid not in (1, 2):
                raise NotImplementedError(f"h5lite: filter id {fid} (deflate and shuffle are read)")
        out = np.zeros(shape, dtype=np.dtype(f"V{esize}"))
        if btree == _UNDEF:
            return out.tobytes()
        nd = len(shape)
        cbytes = esize * int(np.prod(chunk, dtype=np.int64))

        def walk(addr):
            a = r.base + addr
            if bytes(b[a:a + 4]) != b"TREE" or b[a + 4] != 1:
                raise H5Error("h5lite: bad chunk B-tree node")
            level, n = b[a + 5], _u(b, a + 6, 2)
            ksz = 8 + 8 * (nd + 1)
            p = a + 24
            for _ in range(n):
                csize, fmask = _u(b, p, 4), _u(b, p + 4, 4)
                offs = tuple(_u(b, p + 8 + 8 * i, 8) for i in range(nd))
                child = _u(b, p + ksz, 8)
                p += ksz + 8
                if level > 0:
                    walk(child)
                    continue
                if offs[0] >= row1 or offs[0] + chunk[0] <= row0:
                    continue                                     # no row of this chunk is wanted
                data = bytes(b[r.base + child:r.base + child + csize])
                for k in range(len(filters) - 1, -1, -1):        # undo the pipeline in reverse
                    if fmask & (1 << k):
                        continue
                    fid = filters[k][0]
                    if fid == 1:
                        data = zlib.decompress(data)
                    elif fid == 2:
                        n_el = len(data) // esize
                        data = np.frombuffer(data[:n_el * esize], np.uint8).reshape(esize, n_el).T.tobytes() + data[n_el * esize:]
                blk = np.frombuffer(data[:cbytes], dtype=out.dtype).reshape(chunk)
                first, last = max(offs[0], row0), min(offs[0] + chunk[0], row1)
                sl_out = (slice(first - row0, last - row0),) + tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs[1:], chunk[1:], shape[1:]))
                sl_in = (slice(first - offs[0], last - offs[0]),) + tuple(slice(0, s.stop - s.start) for s in sl_out[1:])
                out[sl_out] = blk[sl_in]
        walk(btree)
        return out.tobytes()


class Group:
    def __init__(self, reader, addr, name="/"):
        self._r, self._addr, self.name = reader, addr, name
        self._links = None

    def _l(self):
        if self._links is None:
            self._links = self._r.links(self._addr)
        return self._links

    def keys(self):
        return list(self._l().keys())

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._l())

    def __contains__(self, key):
        try:
            self._resolve(key)
            return True
        except KeyError:
            return False

    def _resolve(self, key):
        node = self
        parts = [p for p in key.split("/") if p]
        for i, part in enumerate(parts):
            links = node._l() if isinstance(node, Group) else {}
            if part not in links:
                raise KeyError(f"Unable to open object (object '{part}' doesn't exist)")
            addr = links[part]
            is_group = any(t in (0x11, 0x02, 0x06, 0x0A) for t, *_ in self._r.messages(addr)) and not any(
                t == 0x08 for t, *_ in self._r.messages(addr))
            node = Group(self._r, addr, "/" + "/".join(parts[:i + 1])) if is_group else Dataset(self._r, addr, "/" + "/".join(parts[:i + 1]))
        return node

    def __getitem__(self, key):
        return self._resolve(key)

    @property
    def attrs(self):
        return self._r.attributes(self._addr)


# ---- writing --------------------------------------------------------------------------------------------------------------------------
def _type_msg(dt):
    """Datatype message body for a numpy dtype (little-endian numerics, complex as the {r, i} compound, fixed strings)."""
    dt = np.dtype(dt)
    if dt.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize) + struct.pack("<HH", 0, 8 * dt.itemsize)
    if dt.kind == "b":
        return _type_msg(np.int8)
    if dt.kind == "f":
        if dt.itemsize == 4:
            prop = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            b0, b1 = 0x20, 31
        elif dt.itemsize == 8:
            prop = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            b0, b1 = 0x20, 63
        elif dt.itemsize == 2:
            prop = struct.pack("<HHBBBBI", 0, 16, 10, 5, 0, 10, 15)
            b0, b1 = 0x20, 15
        else:
            raise NotImplementedError(f"h5lite: writing {dt}")
        return struct.pack("<BBBBI", 0x11, b0, b1, 0, dt.itemsize) + prop      # mantissa normalisation: implied msb; sign bit position
    if dt.kind == "c":
        base = _type_msg(np.dtype(f"f{dt.itemsize // 2}"))
        body = b""
        for i, name in enumerate((b"r", b"i")):
            body += name + b"\x00" * 7 + struct.pack("<IBBBBII4I", i * (dt.itemsize // 2), 0, 0, 0, 0, 0, 0, 0, 0, 0, 0) + base
        return struct.pack("<BBBBI", 0x16, 2, 0, 0, dt.itemsize) + body
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x00, 0, 0, max(dt.itemsize, 1))          # null-terminated, ASCII
    raise NotImplementedError(f"h5lite: writing dtype {dt}")


def _space_msg(shape):
    if shape == ():
        return struct.pack("<BBBB4x", 1, 0, 0, 0)
    return struct.pack("<BBBB4x", 1, len(shape), 1, 0) + b"".join(struct.pack("<Q", s) for s in shape) * 2      # dims, then max dims


def _pad8(b):
    return b + b"\x00" * (-len(b) % 8)


def _msg(t, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", t, len(body), flags) + body


def _as_array(value):
    if isinstance(value, str):
        value = value.encode("utf-8")
    if isinstance(value, bytes):
        return np.array(value + b"\x00", dtype=f"S{len(value) + 1}")
    a = np.asarray(value)
    if a.dtype.kind == "U":
        a = np.char.encode(a, "utf-8")
    if a.dtype == np.bool_:
        a = a.astype(np.int8)
    if a.dtype.kind in "iufc" and a.dtype.byteorder == ">":
        a = a.astype(a.dtype.newbyteorder("<"))
    return a


def _attr_msg(name, value):
    a = _as_array(value)
    nm = name.encode() + b"\x00"
    t, s = _type_msg(a.dtype), _space_msg(a.shape)
    return _msg(0x0C, struct.pack("<BBHHH", 1, 0, len(nm), len(t), len(s)) + _pad8(nm) + _pad8(t) + _pad8(s) + a.tobytes())


class _Writer:
    _LEAF_K, _INT_K = 64, 16                         # up to 128 datasets in the root group's single symbol-table node

    def __init__(self, path):
        self.path = path
        self.datasets = {}
        self.attrs = {}

    def create_dataset(self, name, data=None, shape=None, dtype=None, **kwargs):
        name = name.strip("/")
        if "/" in name or not name:
            raise NotImplementedError("h5lite: datasets are written into the root group only")
        if name in self.datasets:
            raise ValueError(f"Unable to create dataset (name already exists): {name}")
        if data is None:
            data = np.zeros(shape, dtype=dtype or np.float32)
        a = _as_array(data)
        if dtype is not None:
            a = a.astype(dtype)
        if len(self.datasets) >= 2 * self._LEAF_K:
            raise NotImplementedError("h5lite: more than 128 datasets per file")
        self.datasets[name] = np.array(a, order="C", copy=True)      # (np.ascontiguousarray would promote a scalar to 1-D)
        return a

    def close(self):
        names = sorted(self.datasets, key=lambda s: s.encode())
        # local heap data: the empty string at offset 0, then the names, then one free block
        heap = bytearray(8)
        noff = {}
        for n in names:
            noff[n] = len(heap)
            heap += _pad8(n.encode() + b"\x00")
        free_off = len(heap)
        heap += struct.pack("<QQ", 1, 32) + bytes(16)                   # free block: next = 1 (none), size 32
        sb_size = 24 + 4 * 8 + 40
        snod_size = 8 + 2 * self._LEAF_K * 40
        tree_size = 24 + (2 * self._INT_K + 1) * 8 + 2 * self._INT_K * 8
        root_msgs = _msg(0x11, struct.pack("<QQ", 0, 0)) + b"".join(_attr_msg(k, v) for k, v in self.attrs.items())
        root_hdr_size = 16 + len(root_msgs)
        a_root = sb_size
        a_tree = a_root + root_hdr_size
        a_heap = a_tree + tree_size
        a_heap_data = a_heap + 32
        a_snod = a_heap_data + len(heap)
        pos = a_snod + snod_size
        hdrs, data_addr = {}, {}
        for n in names:
            a = self.datasets[n]
            msgs = (_msg(0x01, _space_msg(a.shape)) + _msg(0x03, _type_msg(a.dtype), flags=1) + _msg(0x05, struct.pack("<BBBB", 2, 2, 2, 0))
                    + _msg(0x08, struct.pack("<BBQQ", 3, 1, 0, a.nbytes)))
            hdrs[n] = (pos, msgs)
            pos += 16 + len(msgs)
        for n in names:
            pos = (pos + 7) // 8 * 8
            data_addr[n] = pos
            pos += self.datasets[n].nbytes
        eof = pos
        out = bytearray(eof)
        # superblock 0
        out[0:8] = _SIG
        out[8:24] = struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, self._LEAF_K, self._INT_K, 0)
        out[24:56] = struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
        out[56:96] = struct.pack("<QQII", 0, a_root, 1, 0) + struct.pack("<QQ", a_tree, a_heap)
        # root object header (version 1)
        root_msgs = _msg(0x11, struct.pack("<QQ", a_tree, a_heap)) + b"".join(_attr_msg(k, v) for k, v in self.attrs.items())
        out[a_root:a_root + 16] = struct.pack("<BBHII4x", 1, 0, 1 + len(self.attrs), 1, len(root_msgs))
        out[a_root + 16:a_root + 16 + len(root_msgs)] = root_msgs
        # group B-tree: one leaf entry pointing at the symbol-table node
        out[a_tree:a_tree + 24] = b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if names else 0, _UNDEF, _UNDEF)
        if names:
            out[a_tree + 24:a_tree + 48] = struct.pack("<QQQ", 0, a_snod, noff[names[-1]])
        # local heap
        out[a_heap:a_heap + 32] = b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, a_heap_data)
        out[a_heap_data:a_heap_data + len(heap)] = heap
        # symbol table node
        out[a_snod:a_snod + 8] = b"SNOD" + struct.pack("<BBH", 1, 0, len(names))
        for i, n in enumerate(names):
            out[a_snod + 8 + 40 * i:a_snod + 48 + 40 * i] = struct.pack("<QQII16x", noff[n], hdrs[n][0], 0, 0)
        # datasets
        for n in names:
            a = self.datasets[n]
            at = hdrs[n][0]
            body = (_msg(0x01, _space_msg(a.shape)) + _msg(0x03, _type_msg(a.dtype), flags=1) + _msg(0x05, struct.pack("<BBBB", 2, 2, 2, 0))
                    + _msg(0x08, struct.pack("<BBQQ", 3, 1, data_addr[n], a.nbytes)))
            out[at:at + 16] = struct.pack("<BBHII4x", 1, 0, 4, 1, len(body))
            out[at + 16:at + 16 + len(body)] = body
            out[data_addr[n]:data_addr[n] + a.nbytes] = a.tobytes()
        tmp = str(self.path) + ".tmp"
        with open(tmp, "wb") as f:
            f.write(out)
        os.replace(tmp, self.path)


class File(Group):
    """`h5lite.File(path, mode)`: "r" reads an existing file, "w" creates one (written on close)."""

    def __init__(self, path, mode="r"):
        self.filename, self.mode = str(path), mode
        self._w = None
        if mode == "r":
            super().__init__(_Reader(self.filename), None)
            self._addr = self._r.root
        elif mode == "w":
            self._w = _Writer(self.filename)
        else:
            raise ValueError(f"h5lite: mode {mode!r} (only 'r' and 'w')")

    @property
    def attrs(self):
        return self._w.attrs if self._w is not None else self._r.attributes(self._addr)

    def create_dataset(self, name, data=None, **kwargs):
        if self._w is None:
            raise OSError("Unable to create dataset (file is open read-only)")
        return self._w.create_dataset(name, data=data, **kwargs)

    def keys(self):
        return sorted(self._w.datasets) if self._w is not None else super().keys()

    def __contains__(self, key):
        return key.strip("/") in self._w.datasets if self._w is not None else super().__contains__(key)

    def __getitem__(self, key):
        return self._w.datasets[key.strip("/")] if self._w is not None else super().__getitem__(key)

    def close(self):
        if self._w is not None:
            self._w.close()
            self._w = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        if exc[0] is None:
            self.close()
        return False
