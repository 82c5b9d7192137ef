"""Minimal read-only HDF5 reader (test infrastructure; h5py is not in this image).

Covers exactly what the reference's test fixtures under ``tests/test_data/*.h5`` use (written by
h5py with its default ``libver='earliest'``): superblock version 0, version-1 object headers with
continuation blocks, old-style groups (symbol-table message -> v1 B-tree of ``SNOD`` nodes +
local heap), contiguous or compact dataset layouts (layout message version 3), little-endian
fixed-point and IEEE float datatypes, simple dataspaces, and version-1 attribute messages with
scalar / simple dataspaces.  The structure follows the public "HDF5 File Format Specification
Version 1.1/2.0" (The HDF Group); nothing here comes from the reference repository, which only
*writes* these files through h5py (``tests/produce_integration_test_data.py:432-446,529-560``).

    f = H5File(path)
    f["coeval/power_density"]  -> numpy array
    f.attrs                    -> dict of root attributes
    f.keys("coeval")           -> names in a group
"""

from __future__ import annotations

import struct
from pathlib import Path

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(RuntimeError):
    pass


def _u(buf, off, n):
    return int.from_bytes(buf[off:off + n], "little")


class _Obj:
    """One parsed object header: its messages by type."""

    def __init__(self):
        self.msgs = []  # (type, bytes)

    def first(self, t):
        for mt, body in self.msgs:
            if mt == t:
                return body
        return None

    def all(self, t):
        return [b for mt, b in self.msgs if mt == t]


class H5File:
    def __init__(self, path):
        self.buf = Path(path).read_bytes()
        b = self.buf
        if b[:8] != b"\x89HDF\r\n\x1a\n":
            raise H5Error("not an HDF5 file")
        if b[8] != 0:
            raise H5Error(f"superblock version {b[8]} not supported")
        self.so = b[13]  # size of offsets
        self.sl = b[14]  # size of lengths
        if self.so != 8 or self.sl != 8:
            raise H5Error("only 8-byte offsets/lengths supported")
        # 24: base address, free-space, end-of-file, driver info; then the root symbol table entry
        self.base = _u(b, 24, 8)
        root_entry = 24 + 4 * 8
        self.root_header = _u(b, root_entry + 8, 8)
        self._tree = {}
        self.attrs = self._attrs(self._header(self.root_header))

    # ---- object headers -----------------------------------------------------------------
    def _header(self, addr) -> _Obj:
        b = self.buf
        addr += self.base
        if b[addr] != 1:
            raise H5Error(f"object header version {b[addr]} not supported")
        nmsg = _u(b, addr + 2, 2)
        size = _u(b, addr + 8, 4)
        obj = _Obj()
        blocks = [(addr + 16, size)]
        while blocks and len(obj.msgs) < nmsg:
            off, left = blocks.pop(0)
            end = off + left
            while off + 8 <= end and len(obj.msgs) < nmsg:
                mtype = _u(b, off, 2)
                msize = _u(b, off + 2, 2)
                body = b[off + 8: off + 8 + msize]
                off += 8 + msize
                if mtype == 0x10:  # continuation
                    blocks.append((_u(body, 0, 8) + self.base, _u(body, 8, 8)))
                obj.msgs.append((mtype, body))
        return obj

    # ---- groups -------------------------------------------------------------------------
    def _group_entries(self, obj: _Obj) -> dict:
        st = obj.first(0x11)
        if st is None:
            raise H5Error("object is not an old-style group")
        btree = _u(st, 0, 8)
        heap = _u(st, 8, 8)
        hb = self.buf
        h = heap + self.base
        if hb[h:h + 4] != b"HEAP":
            raise H5Error("bad local heap")
        heap_data = _u(hb, h + 8 + 16, 8) + self.base
        out = {}
        self._walk_btree(btree, heap_data, out)
        return out

    def _walk_btree(self, addr, heap_data, out):
        b = self.buf
        a = addr + self.base
        if b[a:a + 4] == b"SNOD":
            n = _u(b, a + 6, 2)
            e = a + 8
            for _ in range(n):
                name_off = _u(b, e, 8)
                hdr = _u(b, e + 8, 8)
                s = heap_data + name_off
                name = b[s:b.index(b"\0", s)].decode()
                out[name] = hdr
                e += 40
            return
        if b[a:a + 4] != b"TREE":
            raise H5Error("bad group B-tree node")
        if b[a + 4] != 0:
            raise H5Error("not a group B-tree")
        used = _u(b, a + 6, 2)
        p = a + 8 + 16  # skip left/right siblings
        # key0, child0, key1, child1, ..., key_n
        for i in range(used):
            child = _u(b, p + 8 + i * 16, 8)
            self._walk_btree(child, heap_data, out)

    def _resolve(self, path):
        addr = self.root_header
        for part in [p for p in path.split("/") if p]:
            key = addr
            if key not in self._tree:
                self._tree[key] = self._group_entries(self._header(addr))
            try:
                addr = self._tree[key][part]
            except KeyError as e:
                raise KeyError(path) from e
        return addr

    def keys(self, path=""):
        addr = self._resolve(path)
        return sorted(self._group_entries(self._header(addr)))

    def is_group(self, path):
        return self._header(self._resolve(path)).first(0x11) is not None

    def __contains__(self, path):
        try:
            self._resolve(path)
            return True
        except KeyError:
            return False

    # ---- datatypes / dataspaces ---------------------------------------------------------
    @staticmethod
    def _dtype(body):
        cls = body[0] & 0x0F
        ver = body[0] >> 4
        bits0 = body[1]
        size = _u(body, 4, 4)
        if ver not in (1, 2, 3):
            raise H5Error("datatype version")
        if bits0 & 1:
            raise H5Error("big-endian data not supported")
        if cls == 0:
            signed = bool(bits0 & 0x08)
            return np.dtype(f"<{'i' if signed else 'u'}{size}")
        if cls == 1:
            return np.dtype(f"<f{size}")
        if cls == 3:  # fixed-length string
            return np.dtype(f"S{size}")
        if cls == 8:  # enum (h5py bool): base type follows
            return H5File._dtype(body[8:])
        raise H5Error(f"datatype class {cls} not supported")

    @staticmethod
    def _dtype_len(body):
        """Length in bytes of a datatype message (needed inside attribute messages)."""
        cls = body[0] & 0x0F
        if cls == 0:
            return 8 + 4
        if cls == 1:
            return 8 + 12
        if cls == 3:
            return 8
        raise H5Error(f"datatype class {cls} not supported in attribute")

    @staticmethod
    def _shape(body):
        ver = body[0]
        rank = body[1]
        if ver == 1:
            off = 8
        elif ver == 2:
            off = 4
        else:
            raise H5Error("dataspace version")
        return tuple(_u(body, off + 8 * i, 8) for i in range(rank))

    # ---- datasets -----------------------------------------------------------------------
    def __getitem__(self, path):
        obj = self._header(self._resolve(path))
        dt_b, sp_b, lay = obj.first(0x03), obj.first(0x01), obj.first(0x08)
        if dt_b is None or sp_b is None or lay is None:
            raise H5Error(f"{path} is not a dataset")
        dt = self._dtype(dt_b)
        shape = self._shape(sp_b)
        n = int(np.prod(shape)) if shape else 1
        if lay[0] != 3:
            raise H5Error("layout message version")
        cls = lay[1]
        if cls == 0:  # compact
            sz = _u(lay, 2, 2)
            raw = lay[4:4 + sz]
        elif cls == 1:  # contiguous
            addr = _u(lay, 2, 8)
            sz = _u(lay, 10, 8)
            if addr == UNDEF:
                return np.zeros(shape, dt)
            raw = self.buf[addr + self.base: addr + self.base + sz]
        else:
            raise H5Error("chunked layout not supported")
        return np.frombuffer(raw, dt, count=n).reshape(shape).copy()

    # ---- attributes ---------------------------------------------------------------------
    def _attrs(self, obj: _Obj) -> dict:
        out = {}
        for body in obj.all(0x0C):
            if body[0] != 1:
                continue
            nlen, dlen, slen = _u(body, 2, 2), _u(body, 4, 2), _u(body, 6, 2)
            pad = lambda x: (x + 7) & ~7
            p = 8
            name = body[p:p + nlen].split(b"\0")[0].decode()
            p += pad(nlen)
            dt_b = body[p:p + dlen]
            p += pad(dlen)
            sp_b = body[p:p + slen]
            p += pad(slen)
            try:
                dt = self._dtype(dt_b)
            except H5Error:
                out[name] = None  # variable-length strings etc.: not needed by the tests
                continue
            shape = self._shape(sp_b) if sp_b[1] else ()
            n = int(np.prod(shape)) if shape else 1
            val = np.frombuffer(body[p:p + n * dt.itemsize], dt, count=n)
            if dt.kind == "S":
                out[name] = val[0].split(b"\0")[0].decode() if not shape else val
            else:
                out[name] = val.reshape(shape) if shape else val[0].item()
        return out

    def attrs_of(self, path):
        return self._attrs(self._header(self._resolve(path)))
