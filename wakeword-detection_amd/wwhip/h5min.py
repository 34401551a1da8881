"""Minimal HDF5 reader / writer for the reference's feature files (no h5py in this image).

The reference stores its pre-computed log-mel features as one HDF5 dataset per clip
(``utils/filter_dataset_to_h5.py:136-145``: float32 ``[T, 40]`` plus the attributes
``is_hotword``, ``speaker``, ``speech_start_ts``, ``speech_end_ts``) and reads them back with
``h5py`` (``utils/evaluate_tf_lite_opts.py:35-47``).  h5py writes such files with the library's
default "earliest" format: superblock version 0, version-1 object headers, old-style groups
(symbol-table B-tree + local heap), contiguous dataset storage, variable-length strings in a
global heap.  This module implements that subset of the published HDF5 file format
(https://docs.hdfgroup.org/hdf5/develop/_f_m_t3.html) and nothing else: anything outside it - superblock
version 2/3 and version-2 object headers (``libver='latest'``), link-message groups, chunked or filtered
(deflate / shuffle) dataset storage, compound / array / bitfield / reference datatypes - raises
``H5UnsupportedError`` (an ``H5FormatError`` and a ``NotImplementedError``) naming the feature, so that a file written with other h5py options fails loudly
instead of being half-read (install h5py for those: ``wwhip.evaluate.open_h5`` prefers it when present).
Every line of the reader is exercised by tests/test_h5min.py on h5py-written files.

    with h5min.File(path) as f:                 # read-only
        for key in f.keys():
            feats = f[key][()]                  # numpy array
            label = f[key].attrs["is_hotword"]

    h5min.write_datasets(path, {name: (array, {"is_hotword": 1, ...})})

The reader is pinned by the reference's own h5py-written files (the Keras checkpoints under
``wwdetect/CRNN/models/*/``, committed as fixtures under ``tests/golden/keras_h5``): every weight
it returns equals the weight the TFLite reader extracts from the converted ``.tflite`` of the same
model (tests/test_h5min.py).  The writer emits the same structures h5py emits for the feature
files and is checked by reading its output back.
"""
from __future__ import annotations

import struct
from typing import Any, Dict, Iterator, List, Optional, Tuple

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5FormatError(ValueError):
    pass


class H5UnsupportedError(H5FormatError, NotImplementedError):
    """A valid HDF5 feature this reader does not implement (chunked / compact / filtered storage, superblock v2+, v2 object
    headers, new-style groups, compound / array types).  A subclass of :class:`H5FormatError`, so callers that fall back on
    that (``open_h5``, the checkpoint readers) catch it too; also a ``NotImplementedError``.  Files the reader handles are
    what h5py writes with its defaults (``libver='earliest'``, contiguous, no compression): INTEGRATION.md."""


def _pad8(n: int) -> int:
    return (n + 7) & ~7


# ------------------------------------------------------------------------------------------------
# datatypes
# ------------------------------------------------------------------------------------------------
class _DType:
    """Parsed datatype message: ``np`` is the numpy dtype of one element as stored;
    ``kind`` is 'plain', 'vlen_str', 'vlen_seq' or 'enum_bool'."""

    def __init__(self, np_dtype, kind: str = "plain", size: int = 0, base: Optional["_DType"] = None,
                 enum: Optional[Dict[str, int]] = None, strpad: int = 0) -> None:
        self.np = np_dtype
        self.kind = kind
        self.size = size
        self.base = base
        self.enum = enum
        self.strpad = strpad


def _parse_dtype(buf: bytes, off: int) -> Tuple[_DType, int]:
    """Returns (dtype, bytes consumed)."""
    cv, b0, b1, b2, size = struct.unpack_from("<BBBBI", buf, off)
    cls, ver = cv & 0x0F, cv >> 4
    p = off + 8
    if cls == 0:  # fixed point
        order = ">" if (b0 & 1) else "<"
        signed = bool(b0 & 8)
        dt = np.dtype(f"{order}{'i' if signed else 'u'}{size}")
        return _DType(dt, size=size), 8 + 4
    if cls == 1:  # floating point
        order = ">" if (b0 & 1) else "<"
        if size not in (2, 4, 8):
            raise H5FormatError(f"unsupported float size {size}")
        return _DType(np.dtype(f"{order}f{size}"), size=size), 8 + 12
    if cls == 3:  # fixed-length string
        return _DType(np.dtype(f"S{size}"), size=size, strpad=b0 & 0x0F), 8
    if cls in (4, 6, 7):  # bitfield, compound, reference: the feature files and Keras checkpoints hold none
        raise H5UnsupportedError(f"HDF5 datatype class {cls} ({ {4: 'bitfield', 6: 'compound', 7: 'reference'}[cls] })")
    if cls == 8:  # enumeration
        n = b0 | (b1 << 8)
        base, used = _parse_dtype(buf, p)
        q = p + used
        names = []
        for _ in range(n):
            end = buf.index(b"\0", q)
            names.append(buf[q:end].decode("utf-8"))
            q = q + _pad8(end - q + 1) if ver < 3 else end + 1
        vals = np.frombuffer(buf, base.np, n, q)
        q += n * base.size
        enum = {k: int(v) for k, v in zip(names, vals)}
        kind = "enum_bool" if set(enum) == {"FALSE", "TRUE"} else "plain"
        return _DType(base.np, kind=kind, size=size, base=base, enum=enum), q - off
    if cls == 9:  # variable length
        vtype = b0 & 0x0F
        base, used = _parse_dtype(buf, p)
        kind = "vlen_str" if vtype == 1 else "vlen_seq"
        return _DType(np.dtype("V%d" % size), kind=kind, size=size, base=base), 8 + used
    if cls == 10:
        raise H5UnsupportedError("HDF5 array datatype")
    raise H5FormatError(f"unsupported datatype class {cls}")


def _parse_dataspace(buf: bytes, off: int) -> Tuple[Optional[Tuple[int, ...]], int]:
    ver, rank, flags = buf[off], buf[off + 1], buf[off + 2]
    if ver == 1:
        p = off + 8
    elif ver == 2:  # written with libver='latest'; h5py's default ('earliest') and this module's writer use version 1
        raise H5UnsupportedError("HDF5 dataspace message version 2")
    else:
        raise H5FormatError(f"dataspace version {ver}")
    dims = struct.unpack_from("<%dQ" % rank, buf, p)
    p += 8 * rank
    if flags & 1:
        p += 8 * rank
    if ver == 1 and flags & 2:
        p += 8 * rank
    return tuple(int(d) for d in dims), p - off


# ------------------------------------------------------------------------------------------------
# reader
# ------------------------------------------------------------------------------------------------
class _Message:
    __slots__ = ("type", "flags", "data")

    def __init__(self, mtype: int, flags: int, data: bytes) -> None:
        self.type, self.flags, self.data = mtype, flags, data


class _Object:
    """An object header (group or dataset) with its messages parsed lazily."""

    def __init__(self, f: "File", addr: int, name: str) -> None:
        self._f = f
        self._addr = addr
        self.name = name
        self._msgs = f._read_header(addr)
        self._attrs: Optional[Dict[str, Any]] = None

    def _find(self, mtype: int) -> Optional[_Message]:
        for m in self._msgs:
            if m.type == mtype:
                return m
        return None

    @property
    def attrs(self) -> Dict[str, Any]:
        if self._attrs is None:
            out: Dict[str, Any] = {}
            for m in self._msgs:
                if m.type == 0x000C:
                    k, v = self._f._parse_attribute(m.data)
                    out[k] = v
            self._attrs = out
        return self._attrs


class Dataset(_Object):
    def __init__(self, f: "File", addr: int, name: str) -> None:
        super().__init__(f, addr, name)
        ds, dt, lay = self._find(0x0001), self._find(0x0003), self._find(0x0008)
        if ds is None or dt is None or lay is None:
            raise H5FormatError(f"{name}: not a dataset")
        self._shape, _ = _parse_dataspace(ds.data, 0)
        self._dt, _ = _parse_dtype(dt.data, 0)
        self._layout = lay.data
        if self._find(0x000B) is not None:
            raise H5UnsupportedError(f"{name}: filtered (deflate / shuffle / fletcher32) dataset storage")

    @property
    def shape(self) -> Tuple[int, ...]:
        return self._shape or ()

    @property
    def dtype(self) -> np.dtype:
        if self._dt.kind == "vlen_str":
            return np.dtype(object)
        if self._dt.kind == "enum_bool":
            return np.dtype(bool)
        return self._dt.np.newbyteorder("=") if self._dt.np.kind in "iuf" else self._dt.np

    def __len__(self) -> int:
        return self.shape[0]

    def _raw(self) -> bytes:
        f, b = self._f, self._layout
        shape = self.shape
        n_elem = int(np.prod(shape, dtype=np.int64)) if self._shape is not None else 0
        nbytes = n_elem * self._dt.size
        ver = b[0]
        if ver == 3:
            cls = b[1]
            if cls == 0:
                raise H5UnsupportedError(f"{self.name}: compact dataset storage")
            if cls == 1:
                addr, size = struct.unpack_from("<QQ", b, 2)
                return b"\0" * nbytes if addr == UNDEF else f._buf[addr:addr + nbytes]
            if cls == 2:
                raise H5UnsupportedError(f"{self.name}: chunked dataset storage")
            raise H5FormatError(f"layout class {cls}")
        if ver in (1, 2):  # libhdf5 older than 1.6
            raise H5UnsupportedError(f"{self.name}: data layout message version {ver}")
        raise H5FormatError(f"layout version {ver}")

    def __getitem__(self, key) -> Any:
        arr = self._f._decode(self._raw(), self._dt, self._shape)
        if key is Ellipsis or key == ():
            return arr
        return arr[key]

    def __array__(self, dtype=None, copy=None):
        a = self[()]
        return a if dtype is None else a.astype(dtype)


class Group(_Object):
    def __init__(self, f: "File", addr: int, name: str) -> None:
        super().__init__(f, addr, name)
        self._links: Optional[Dict[str, int]] = None

    def _load_links(self) -> Dict[str, int]:
        if self._links is None:
            links: Dict[str, int] = {}
            st = self._find(0x0011)
            if st is not None:
                btree, heap = struct.unpack_from("<QQ", st.data, 0)
                for name, addr in self._f._iter_symbols(btree, heap):
                    links[name] = addr
            if st is None and (self._find(0x0002) is not None or any(m.type == 0x0006 for m in self._msgs)):
                # link-info / link messages: groups written with libver='latest'
                raise H5UnsupportedError(f"{self.name}: new-style HDF5 groups (link messages / fractal-heap storage)")
            self._links = links
        return self._links

    def keys(self) -> List[str]:
        return sorted(self._load_links())  # h5py iterates in name order for old-style groups

    def __iter__(self) -> Iterator[str]:
        return iter(self.keys())

    def __len__(self) -> int:
        return len(self._load_links())

    def __contains__(self, name: str) -> bool:
        try:
            self[name]
            return True
        except KeyError:
            return False

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def __getitem__(self, path: str):
        node: _Object = self
        base = self.name.rstrip("/")
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group):
                raise KeyError(path)
            links = node._load_links()
            if part not in links:
                raise KeyError(f"{path!r}: no object {part!r} in {node.name!r}")
            base = base + "/" + part
            node = self._f._open(links[part], base)
        return node


class File(Group):
    """Read-only HDF5 file (whole file held in memory via mmap-free read; feature files are small)."""

    def __init__(self, path: str, mode: str = "r") -> None:
        if mode != "r":
            raise ValueError("h5min.File is read-only; use h5min.write_datasets to create files")
        import mmap
        self._fh = open(path, "rb")
        try:
            self._buf = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        except (ValueError, OSError):
            self._buf = self._fh.read()
        b = self._buf
        base = 0
        while b[base:base + 8] != SIGNATURE:  # the superblock may sit at 0, 512, 1024, ...
            base = 512 if base == 0 else base * 2
            if base + 8 > len(b):
                raise H5FormatError(f"{path}: not an HDF5 file")
        ver = b[base + 8]
        if ver in (0, 1):
            if b[base + 13] != 8 or b[base + 14] != 8:
                raise H5FormatError("only 8-byte offsets and lengths are supported")
            p = base + 24 + (4 if ver == 1 else 0)
            self._base, = struct.unpack_from("<Q", b, p)
            root = p + 32  # root group symbol table entry
            _, addr, cache = struct.unpack_from("<QQI", b, root)
        elif ver in (2, 3):  # written with libver='latest' (v2 object headers, link messages): not what the reference writes
            raise H5UnsupportedError(f"{path}: HDF5 superblock version {ver}")
        else:
            raise H5FormatError(f"superblock version {ver}")
        self._cache: Dict[int, _Object] = {}
        self._gcol: Dict[int, Dict[int, bytes]] = {}
        self.filename = path
        super().__init__(self, addr, "/")

    # -- context manager ---------------------------------------------------------------------
    def close(self) -> None:
        try:
            if hasattr(self._buf, "close"):
                self._buf.close()
        finally:
            self._fh.close()

    def __enter__(self) -> "File":
        return self

    def __exit__(self, *exc) -> None:
        self.close()

    # -- object headers ----------------------------------------------------------------------
    def _read_header(self, addr: int) -> List[_Message]:
        b = self._buf
        addr += self._base if addr != UNDEF else 0
        msgs: List[_Message] = []
        if b[addr:addr + 4] == b"OHDR":
            raise H5UnsupportedError("HDF5 version-2 object headers")
        ver, _, nmsg, _, hsize = struct.unpack_from("<BBHII", b, addr)
        if ver != 1:
            raise H5FormatError(f"object header version {ver} at {addr}")
        blocks = [(addr + 16, hsize)]
        while blocks and len(msgs) < nmsg + 64:
            q, sz = blocks.pop(0)
            end = q + sz
            while q + 8 <= end:
                mtype, msize, mflags = struct.unpack_from("<HHB", b, q)
                q += 8
                data = bytes(b[q:q + msize])
                q += msize
                if mtype == 0x0010:
                    coff, clen = struct.unpack_from("<QQ", data, 0)
                    blocks.append((coff + self._base, clen))
                elif mtype != 0:
                    msgs.append(_Message(mtype, mflags, data))
        return msgs

    def _open(self, addr: int, name: str) -> _Object:
        obj = self._cache.get(addr)
        if obj is None:
            msgs = self._read_header(addr)
            is_ds = any(m.type == 0x0008 for m in msgs)
            obj = Dataset(self, addr, name) if is_ds else Group(self, addr, name)
            self._cache[addr] = obj
        return obj

    # -- old-style groups ----------------------------------------------------------------------
    def _heap_data(self, heap: int) -> int:
        b = self._buf
        heap += self._base
        if b[heap:heap + 4] != b"HEAP":
            raise H5FormatError("bad local heap signature")
        return struct.unpack_from("<Q", b, heap + 24)[0] + self._base

    def _iter_symbols(self, btree: int, heap: int) -> Iterator[Tuple[str, int]]:
        b = self._buf
        data = self._heap_data(heap)
        stack = [btree + self._base]
        while stack:
            node = stack.pop()
            if b[node:node + 4] == b"SNOD":
                n, = struct.unpack_from("<H", b, node + 6)
                for i in range(n):
                    e = node + 8 + 40 * i
                    noff, oaddr = struct.unpack_from("<QQ", b, e)
                    end = b.find(b"\0", data + noff)
                    yield bytes(b[data + noff:end]).decode("utf-8"), oaddr
                continue
            if b[node:node + 4] != b"TREE":
                raise H5FormatError("bad B-tree signature")
            ntype, level, used = struct.unpack_from("<BBH", b, node + 4)
            if ntype != 0:
                raise H5FormatError("expected a group B-tree node")
            p = node + 24
            children = []
            for i in range(used):
                p += 8  # key i
                children.append(struct.unpack_from("<Q", b, p)[0] + self._base)
                p += 8
            stack.extend(reversed(children))

    # -- values --------------------------------------------------------------------------------
    def _global_heap(self, addr: int) -> Dict[int, bytes]:
        col = self._gcol.get(addr)
        if col is None:
            b = self._buf
            a = addr + self._base
            if b[a:a + 4] != b"GCOL":
                raise H5FormatError("bad global heap signature")
            size, = struct.unpack_from("<Q", b, a + 8)
            col = {}
            p = a + 16
            while p + 16 <= a + size:
                idx, _, _, osz = struct.unpack_from("<HHIQ", b, p)
                if idx == 0:
                    break
                col[idx] = bytes(b[p + 16:p + 16 + osz])
                p += 16 + _pad8(osz)
            self._gcol[addr] = col
        return col

    def _decode(self, raw: bytes, dt: _DType, shape: Optional[Tuple[int, ...]]):
        if shape is None:
            return None
        n = int(np.prod(shape, dtype=np.int64))
        if dt.kind in ("vlen_str", "vlen_seq"):
            out = np.empty(n, dtype=object)
            for i in range(n):
                length, addr, idx = struct.unpack_from("<IQI", raw, i * dt.size)
                data = self._global_heap(addr).get(idx, b"")[: length * (dt.base.size if dt.kind == "vlen_seq" else 1)] \
                    if addr not in (0, UNDEF) else b""
                if dt.kind == "vlen_str":
                    out[i] = data.decode("utf-8", "replace")
                else:
                    out[i] = np.frombuffer(data, dt.base.np).astype(dt.base.np.newbyteorder("="))
            out = out.reshape(shape)
            return out[()] if shape == () else out
        arr = np.frombuffer(raw, dtype=dt.np, count=n).reshape(shape)
        if dt.np.kind in "iuf":
            arr = arr.astype(dt.np.newbyteorder("="))
        else:
            arr = arr.copy()
        if dt.kind == "enum_bool":
            arr = arr.astype(bool)
        return arr[()] if shape == () else arr

    def _parse_attribute(self, d: bytes) -> Tuple[str, Any]:
        ver = d[0]
        nsz, tsz, ssz = struct.unpack_from("<HHH", d, 2)
        p = 8
        if ver == 3:
            p += 1
        pad = _pad8 if ver == 1 else (lambda x: x)
        name = d[p:p + nsz].split(b"\0", 1)[0].decode("utf-8")
        p += pad(nsz)
        if ver >= 2 and d[1] & 1:
            raise H5FormatError(f"attribute {name!r}: shared datatypes are not supported")
        dt, _ = _parse_dtype(d, p)
        p += pad(tsz)
        shape, _ = _parse_dataspace(d, p)
        p += pad(ssz)
        val = self._decode(d[p:], dt, shape)
        if isinstance(val, np.ndarray) and val.dtype.kind == "S" and dt.kind == "plain" and val.shape == ():
            val = val[()]
        return name, val


# ------------------------------------------------------------------------------------------------
# writer (the structures h5py emits for the reference's feature files)
# ------------------------------------------------------------------------------------------------
def _dtype_msg(dt: np.dtype) -> bytes:
    dt = np.dtype(dt)
    if dt.kind == "f":
        props = {2: (0, 16, 10, 5, 0, 10, 15), 4: (0, 32, 23, 8, 0, 23, 127), 8: (0, 64, 52, 11, 0, 52, 1023)}[dt.itemsize]
        sign_pos = dt.itemsize * 8 - 1
        return struct.pack("<BBBBI", 0x11, 0x20, sign_pos, 0, dt.itemsize) + \
            struct.pack("<HHBBBBI", props[0], props[1], props[2], props[3], props[4], props[5], props[6])
    if dt.kind in "iu":
        return struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0, 0, 0, dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
    if dt.kind == "S":
        return struct.pack("<BBBBI", 0x13, 0x01, 0, 0, dt.itemsize)  # null-padded ASCII, as h5py writes numpy bytes
    raise TypeError(f"h5min cannot store dtype {dt}")


_BOOL_ENUM = (struct.pack("<BBBBI", 0x18, 2, 0, 0, 1) + _dtype_msg(np.dtype("i1")) +
              b"FALSE\0\0\0" + b"TRUE\0\0\0\0" + bytes([0, 1]))
_VLEN_STR = struct.pack("<BBBBI", 0x19, 0x01, 0x01, 0, 16) + struct.pack("<BBBBI", 0x13, 0x00, 0, 0, 1)  # vlen UTF-8 of 1-byte chars


def _dataspace_msg(shape: Tuple[int, ...]) -> bytes:
    rank = len(shape)
    flags = 1 if rank else 0
    b = struct.pack("<BBBBI", 1, rank, flags, 0, 0)
    b += struct.pack("<%dQ" % rank, *shape)
    if rank:
        b += struct.pack("<%dQ" % rank, *shape)  # max dims = dims
    return b


def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = data + b"\0" * (_pad8(len(data)) - len(data))
    return struct.pack("<HHBBBB", mtype, len(data), flags, 0, 0, 0) + data


class _Writer:
    def __init__(self) -> None:
        self.buf = bytearray()
        self.gheap: List[bytes] = []

    def alloc(self, data: bytes, align: int = 8) -> int:
        while len(self.buf) % align:
            self.buf.append(0)
        addr = len(self.buf)
        self.buf += data
        return addr

    def attr_msg(self, name: str, value: Any) -> bytes:
        nb = name.encode("utf-8") + b"\0"
        if isinstance(value, (bool, np.bool_)):
            dt, ds, data = _BOOL_ENUM, _dataspace_msg(()), bytes([1 if value else 0])
        elif isinstance(value, str):
            self.gheap.append(value.encode("utf-8"))
            dt, ds = _VLEN_STR, _dataspace_msg(())
            data = ("GHEAPREF", len(self.gheap), len(self.gheap[-1]))  # patched once the heap address is known
        else:
            arr = np.asarray(value)
            if arr.dtype.kind == "U":
                arr = arr.astype("S")
            if arr.dtype == np.float16:
                arr = arr.astype(np.float32)
            arr = np.asarray(arr.astype(arr.dtype.newbyteorder("<")) if arr.dtype.kind in "iuf" else arr, order="C")
            dt, ds, data = _dtype_msg(arr.dtype), _dataspace_msg(arr.shape), arr.tobytes()
        head = struct.pack("<BBHHH", 1, 0, len(nb), len(dt), len(ds))
        body = nb + b"\0" * (_pad8(len(nb)) - len(nb)) + dt + b"\0" * (_pad8(len(dt)) - len(dt)) + \
            ds + b"\0" * (_pad8(len(ds)) - len(ds))
        return head, body, data


def write_datasets(path: str, datasets: Dict[str, Tuple[np.ndarray, Dict[str, Any]]],
                   root_attrs: Optional[Dict[str, Any]] = None) -> None:
    """Write a flat HDF5 file: one contiguous dataset per entry of ``datasets`` (name ->
    (array, attributes)), all in the root group - the layout of the reference's
    ``train/dev/test.h5`` (``filter_dataset_to_h5.py:136-145``).  ``bool`` attributes become
    h5py's FALSE/TRUE enum, ``str`` attributes variable-length UTF-8 strings, numbers scalars."""
    w = _Writer()
    names = sorted(datasets)  # B-tree keys must be in name order
    for n in names:
        if "/" in n or not n:
            raise ValueError(f"dataset name {n!r}: nested groups are not supported by the writer")
    w.buf += b"\0" * 96  # superblock (56 bytes + 40-byte root symbol table entry), patched at the end

    # ---- global heap for variable-length strings: collect first, so that its address is known
    pending: List[Tuple[int, bytes, bytes, Any]] = []  # per object: header pieces
    objs: List[Tuple[str, np.ndarray, List]] = []
    for n in names:
        arr, attrs = datasets[n]
        arr = np.asarray(arr)
        if arr.dtype.kind in "iuf":
            arr = arr.astype(arr.dtype.newbyteorder("<"))
        arr = np.asarray(arr, order="C")
        objs.append((n, arr, [w.attr_msg(k, v) for k, v in (attrs or {}).items()]))
    root_attr_msgs = [w.attr_msg(k, v) for k, v in (root_attrs or {}).items()]
    gheap_addr = UNDEF
    if w.gheap:
        body = bytearray()
        for i, s in enumerate(w.gheap, 1):
            body += struct.pack("<HHIQ", i, 1, 0, len(s)) + s + b"\0" * (_pad8(len(s)) - len(s))
        size = max(4096, _pad8(16 + len(body) + 16))
        free = size - 16 - len(body)
        body += struct.pack("<HHIQ", 0, 0, 0, free) + b"\0" * (free - 16)
        gheap_addr = w.alloc(b"GCOL" + struct.pack("<BBBBQ", 1, 0, 0, 0, size) + bytes(body))

    def finish_attr(piece) -> bytes:
        head, body, data = piece
        if isinstance(data, tuple):
            _, idx, length = data
            data = struct.pack("<IQI", length, gheap_addr, idx)
        return _msg(0x000C, head + body + data)

    # ---- datasets: raw data, then object header
    entries: List[Tuple[str, int]] = []
    for n, arr, attr_pieces in objs:
        daddr = w.alloc(arr.tobytes()) if arr.size else UNDEF
        msgs = _msg(0x0001, _dataspace_msg(arr.shape))
        msgs += _msg(0x0003, _dtype_msg(arr.dtype), flags=1)
        msgs += _msg(0x0005, struct.pack("<BBBB", 2, 2, 2, 0))  # fill value v2: late alloc, never written, undefined
        msgs += _msg(0x0008, struct.pack("<BBQQ", 3, 1, daddr, arr.nbytes))
        for piece in attr_pieces:
            msgs += finish_attr(piece)
        nmsg = 4 + len(attr_pieces)
        hdr = struct.pack("<BBHII", 1, 0, nmsg, 1, len(msgs)) + b"\0" * 4 + msgs
        entries.append((n, w.alloc(hdr)))

    # ---- local heap with the link names (offset 0 holds the empty string, as the library does)
    heap_data = bytearray(b"\0" * 8)
    name_off: Dict[str, int] = {}
    for n, _ in entries:
        name_off[n] = len(heap_data)
        nb = n.encode("utf-8") + b"\0"
        heap_data += nb + b"\0" * (_pad8(len(nb)) - len(nb))
    free_off = len(heap_data)
    heap_data += struct.pack("<QQ", 1, 16)  # one free block: next = 1 (none), size 16
    heap_data_addr = w.alloc(bytes(heap_data))
    heap_addr = w.alloc(b"HEAP" + struct.pack("<BBBBQQQ", 0, 0, 0, 0, len(heap_data), free_off, heap_data_addr))

    # ---- symbol table nodes (2*leafK = 8 entries each) under a B-tree (2*internalK = 32 children per node)
    LEAF, INTERNAL = 4, 16
    level: List[Tuple[int, int, int]] = []  # (address, first key offset (exclusive lower), last key offset)
    for i in range(0, max(len(entries), 1), 2 * LEAF):
        grp = entries[i:i + 2 * LEAF]
        snod = bytearray(b"SNOD" + struct.pack("<BBH", 1, 0, len(grp)))
        for n, addr in grp:
            snod += struct.pack("<QQII", name_off[n], addr, 0, 0) + b"\0" * 16
        snod += b"\0" * (8 + 40 * 2 * LEAF - len(snod))
        level.append((w.alloc(bytes(snod)), 0, name_off[grp[-1][0]] if grp else 0))
    depth = 0
    while True:
        nxt: List[Tuple[int, int, int]] = []
        groups = [level[i:i + 2 * INTERNAL] for i in range(0, len(level), 2 * INTERNAL)]
        addrs = []
        for g in groups:
            addrs.append(w.alloc(b"\0" * (24 + (2 * INTERNAL) * 16 + 8)))
        for gi, g in enumerate(groups):
            node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, depth, len(g),
                                                   addrs[gi - 1] if gi > 0 else UNDEF,
                                                   addrs[gi + 1] if gi + 1 < len(groups) else UNDEF))
            # key[0] = largest name of the left neighbour ("" for the first); key[i+1] = largest name in child i
            prev_key = 0 if gi == 0 else groups[gi - 1][-1][2]
            node += struct.pack("<Q", prev_key)
            for child, _, last in g:
                node += struct.pack("<QQ", child, last)
            node += b"\0" * (24 + (2 * INTERNAL) * 16 + 8 - len(node))
            w.buf[addrs[gi]:addrs[gi] + len(node)] = node
            nxt.append((addrs[gi], 0, g[-1][2]))
        level = nxt
        depth += 1
        if len(level) == 1:
            break
    btree_addr = level[0][0]

    # ---- root group object header
    msgs = _msg(0x0011, struct.pack("<QQ", btree_addr, heap_addr))
    for piece in root_attr_msgs:
        msgs += finish_attr(piece)
    root_hdr = struct.pack("<BBHII", 1, 0, 1 + len(root_attr_msgs), 1, len(msgs)) + b"\0" * 4 + msgs
    root_addr = w.alloc(root_hdr)
    eof = len(w.buf)
    sb = SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF, INTERNAL, 0)
    sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, root_addr, 1, 0) + struct.pack("<QQ", btree_addr, heap_addr)
    assert len(sb) == 96
    w.buf[0:96] = sb
    with open(path, "wb") as fh:
        fh.write(bytes(w.buf))
