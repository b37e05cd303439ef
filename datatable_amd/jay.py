"""Jay -> numpy (zero-copy, memory-mapped) -> HBM: device-resident ingestion for the group-by path
(SURVEY.md 8(f) row 4).

Jay is datatable's native binary format (src/core/jay/README.md): "JAY1\\0\\0\\0\\0", a data section of
raw, 8-byte-aligned column buffers, a FlatBuffers meta section (schema src/core/jay/jay.fbs), its size
as int64, "\\0\\0\\0\\01JAY".  The fixed-width columns the path uses are stored exactly in the layout
libdthip consumes (SentinelFw: T[nrows], NA = INT*_MIN / NaN), so a column is one contiguous byte range
of the file: it is mapped (numpy.memmap, no copy) and copied host->device once, with no parsing of data.

What is restated from the reference: the container checks of check_jay_signature / open_jay_from_mbuf
(src/core/jay/open_jay.cc:47-118), both column encodings (column_from_jay1 :141-193 -- `stype` + `data`;
column_from_jay2 :219-287 -- `type` + `buffers` = [validity, data]), `nkeys`.  The FlatBuffers reader
below implements just the wire format needed for that schema (vtables, scalars, structs, strings,
vectors of tables/structs); the flatbuffers package is not a dependency.  String / array columns are
outside the accelerated path: `strings="skip"` leaves them out, the default raises.
"""
import mmap
import struct

import numpy as np

from . import _lib as L

# jay::SType (jay.fbs) -> dthip stype; None = not a fixed-width column of the path
_JAY2ST = {0: L.BOOL, 1: L.INT8, 2: L.INT16, 3: L.INT32, 4: L.INT64, 5: L.FLOAT32, 6: L.FLOAT64}
_JAY_NAMES = {7: "str32", 8: "str64", 9: "date32", 10: "time64", 11: "void", 12: "arr32", 13: "arr64"}
_NP = {L.BOOL: np.int8, L.INT8: np.int8, L.INT16: np.int16, L.INT32: np.int32, L.INT64: np.int64,
       L.FLOAT32: np.float32, L.FLOAT64: np.float64}


class _Table:
    """a FlatBuffers table inside `buf` at absolute position `pos`"""

    def __init__(self, buf, pos):
        self.buf, self.pos = buf, pos
        soff = struct.unpack_from("<i", buf, pos)[0]
        self.vt = pos - soff
        self.vtsize = struct.unpack_from("<H", buf, self.vt)[0]

    def _field(self, fid):
        o = 4 + 2 * fid
        if o + 2 > self.vtsize:
            return 0
        off = struct.unpack_from("<H", self.buf, self.vt + o)[0]
        return self.pos + off if off else 0

    def scalar(self, fid, fmt, default=0):
        p = self._field(fid)
        return struct.unpack_from(fmt, self.buf, p)[0] if p else default

    def struct_at(self, fid, fmt):
        p = self._field(fid)
        return struct.unpack_from(fmt, self.buf, p) if p else None

    def _indirect(self, fid):
        p = self._field(fid)
        return p + struct.unpack_from("<I", self.buf, p)[0] if p else 0

    def table(self, fid):
        p = self._indirect(fid)
        return _Table(self.buf, p) if p else None

    def string(self, fid):
        p = self._indirect(fid)
        if not p:
            return None
        n = struct.unpack_from("<I", self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n]).decode("utf-8")

    def vector_of_tables(self, fid):
        p = self._indirect(fid)
        if not p:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        out = []
        for i in range(n):
            e = p + 4 + 4 * i
            out.append(_Table(self.buf, e + struct.unpack_from("<I", self.buf, e)[0]))
        return out

    def vector_of_structs(self, fid, fmt):
        p = self._indirect(fid)
        if not p:
            return None
        n = struct.unpack_from("<I", self.buf, p)[0]
        sz = struct.calcsize(fmt)
        return [struct.unpack_from(fmt, self.buf, p + 4 + sz * i) for i in range(n)]


# field ids, in schema order (a union takes two: its type tag, then its value)
_F_NROWS, _F_NCOLS, _F_NKEYS, _F_COLUMNS = 0, 1, 2, 3
_C_STYPE, _C_DATA, _C_STRDATA, _C_NAME, _C_NULLCOUNT, _C_STATS_TYPE, _C_STATS, _C_TYPE, _C_NROWS, _C_BUFFERS, _C_CHILDREN = range(11)
_T_STYPE = 0


class JayError(IOError):
    pass


def read_meta(buf):
    """buf: the whole file (bytes / mmap / memoryview).  Returns (nrows, nkeys, [column dicts]) where a
    column dict has name, jay_stype, stype (dthip code or None), offset (absolute, bytes), nbytes, nullcount."""
    size = len(buf)
    if size < 24:
        raise JayError("Invalid Jay file of size %d" % size)
    if bytes(buf[:3]) != b"JAY":
        raise JayError("Invalid signature for a Jay file: first 4 bytes are `%s`" % bytes(buf[:4]).decode("latin1"))
    if bytes(buf[size - 3:size]) != b"JAY" and bytes(buf[size - 4:size]) != b"JAY1":
        raise JayError("Invalid signature for a Jay file: last 4 bytes are `%s`" % bytes(buf[size - 4:size]).decode("latin1"))
    if bytes(buf[:8]) != b"JAY1\0\0\0\0":
        raise JayError("Unsupported Jay file version: %s" % bytes(buf[3:8]).decode("latin1"))
    meta_size = struct.unpack_from("<q", buf, size - 16)[0]
    if meta_size < 0 or meta_size > size - 24 or meta_size % 8:
        raise JayError("Invalid meta record size in a Jay file: %d" % meta_size)
    meta0 = size - 16 - meta_size
    root = _Table(buf, meta0 + struct.unpack_from("<I", buf, meta0)[0])
    nrows = root.scalar(_F_NROWS, "<Q")
    ncols = root.scalar(_F_NCOLS, "<Q")
    nkeys = root.scalar(_F_NKEYS, "<i")
    cols = []
    for c in root.vector_of_tables(_F_COLUMNS):
        typ = c.table(_C_TYPE)
        if typ is not None:                                  # column_from_jay2: type + buffers [validity, data]
            jst = typ.scalar(_T_STYPE, "<B")
            bufs = c.vector_of_structs(_C_BUFFERS, "<QQ") or []
            cn = c.scalar(_C_NROWS, "<Q")
            if cn != nrows:
                raise JayError("Length of column %d is %d, however the Frame contains %d rows" % (len(cols), cn, nrows))
            data = bufs[1] if len(bufs) >= 2 else None
            if len(bufs) >= 1 and bufs[0][1] > 0 and jst in _JAY2ST:
                raise NotImplementedError("Jay column `%s` has a validity bitmap (Arrow layout)" % c.string(_C_NAME))
        else:                                                # column_from_jay1: stype + data
            jst = c.scalar(_C_STYPE, "<B")
            data = c.struct_at(_C_DATA, "<QQ")
        st = _JAY2ST.get(jst)
        off, nbytes = (data[0] + 8, data[1]) if data is not None else (0, 0)     # offsets count from the data section (+8)
        if st is not None:
            want = nrows * np.dtype(_NP[st]).itemsize
            if nbytes != want or off + nbytes > meta0:
                raise JayError("Column `%s`: data buffer of %d bytes at %d does not hold %d rows" % (c.string(_C_NAME), nbytes, off, nrows))
        cols.append({"name": c.string(_C_NAME), "jay_stype": jst, "stype": st, "offset": off, "nbytes": nbytes,
                     "nullcount": c.scalar(_C_NULLCOUNT, "<Q")})
    if len(cols) != ncols:
        raise JayError("Jay meta lists %d columns, the Frame has %d" % (len(cols), ncols))
    return nrows, nkeys, cols


def _map(path):
    with open(path, "rb") as fh:
        return mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)


def jay_columns(path, strings="error"):
    """[(name, stype, numpy array)] of the fixed-width columns; the arrays are views of the mapped file"""
    mm = _map(path)
    nrows, nkeys, cols = read_meta(mm)
    out = []
    for i, c in enumerate(cols):
        if c["stype"] is None:
            if strings == "skip":
                if i < nkeys:
                    nkeys = 0          # a key column is left out: the rest is no longer a keyed frame
                continue
            raise NotImplementedError("Jay column `%s` of type %s is outside the accelerated path"
                                      % (c["name"], _JAY_NAMES.get(c["jay_stype"], c["jay_stype"])))
        a = np.frombuffer(mm, dtype=_NP[c["stype"]], count=nrows, offset=c["offset"]) if nrows else np.zeros(0, _NP[c["stype"]])
        out.append((c["name"], c["stype"], a))
    return out, nkeys


def open_jay(path, strings="error"):
    """datatable.fread / dt.open_jay for a .jay file: a Frame over the mapped column buffers (no copy)"""
    from .frame import Frame
    cols, nkeys = jay_columns(path, strings)
    fr = Frame._from_columns([a for _, _, a in cols], [st for _, st, _ in cols], [nm for nm, _, _ in cols])
    fr._nkeys = nkeys
    return fr


def to_device(path, ctx=None, strings="error"):
    """Jay -> HBM without a parsed copy and without torch: ({name: DevCol}, nrows) -- one dthip_malloc + one host->device
    copy per column straight from the mapped file (the pages are read-only and only read).  The DevCols own their
    buffers and go into any DTHIP_DEVICE call of `ctx` (default: the calling thread's default context)."""
    from .engine import default_context
    ctx = ctx or default_context()
    cols, _ = jay_columns(path, strings)
    return {nm: ctx.upload(a, st) for nm, st, a in cols}, (len(cols[0][2]) if cols else 0)
