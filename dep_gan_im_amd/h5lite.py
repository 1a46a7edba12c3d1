"""Read-only reader for the subset of HDF5 that Keras weight files use, in pure Python / NumPy.

Why: the authors' checkpoints (`netG_*.h5`, `trained_depuresnet_*.h5`; GT:892, GE:383, UE:402, models/readme.txt) are
Keras 2.x HDF5 files, and h5py is not part of this image.  `Model.load_weights("….h5")` uses h5py when it is installed
and this module when it is not.  What h5py (libver "earliest", Keras' default) writes for `model.save` /
`save_weights` is a small, stable corner of the format, and that corner is what is implemented:

  * superblock version 0 / 1 (and 2 / 3: root object header address only);
  * object headers version 1 with continuation blocks (version 2 "OHDR" headers are recognised and refused);
  * old-style groups: symbol-table message -> B-tree v1 ("TREE") -> symbol nodes ("SNOD") -> local heap ("HEAP");
    new-style compact groups (link messages) as well;
  * datasets with contiguous or compact layout (no chunking / filters: Keras never asks for them), little- or
    big-endian IEEE floats and integers;
  * attributes (message versions 1-3) holding numbers, fixed-length strings (h5py 2.x, the reference's era) or
    variable-length strings in the global heap (h5py 3.x): `layer_names`, `weight_names`, `backend`, `keras_version`.

The object model mirrors the part of h5py that `dep_gan_im_amd.models.weights_from_keras_h5` touches: `File` / `Group`
are mappings (`in`, `[]`, `keys()`), datasets have `.shape`, `.dtype` and `[()]`, every object has `.attrs`.
Format reference: "HDF5 File Format Specification Version 3.0" (The HDF Group) -- sections III.A-E, IV.A.
tests/test_data_cpu.py reads files written by the real HDF5 library through h5py with it.
"""
from __future__ import annotations

import numpy as np

SIGNATURE = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class _Buf:
    def __init__(self, data, so, sl):
        self.d, self.so, self.sl = data, so, sl

    def u(self, off, n):
        return int.from_bytes(self.d[off:off + n], "little")

    def O(self, off):   # noqa: E743 - an offset ("O" in the specification)
        return self.u(off, self.so)

    def L(self, off):
        return self.u(off, self.sl)


def _pad8(n):
    return (n + 7) & ~7


def _parse_dataspace(b):
    ver = b[0]
    rank = b[1]
    if ver == 1:
        off = 8
    elif ver == 2:
        off = 4
    else:
        raise H5Error("dataspace message version %d" % ver)
    return tuple(int.from_bytes(b[off + 8 * i:off + 8 * i + 8], "little") for i in range(rank))


def _parse_datatype(b):
    """-> (numpy dtype or ('S', n), total bytes of the message consumed is not needed: sizes are explicit upstream)."""
    cls, bits0 = b[0] & 0x0F, b[1]
    size = int.from_bytes(b[4:8], "little")
    order = ">" if (bits0 & 1) else "<"
    if cls == 0:   # fixed point
        signed = bool(bits0 & 0x08)
        return np.dtype("%s%s%d" % (order, "i" if signed else "u", size))
    if cls == 1:   # floating point
        return np.dtype("%sf%d" % (order, size))
    if cls == 3:   # fixed-length string
        return np.dtype("S%d" % size)
    if cls == 9 and (bits0 & 0x0F) == 1:   # variable-length string (h5py >= 3 writes lists of bytes this way)
        return "vlen-str"
    raise H5Error("datatype class %d is outside the Keras-weights subset (variable-length strings and compound types "
                  "are not written by Keras 2.x weight files)" % cls)


class _Object:
    """One object header: its messages, lazily interpreted as a group or a dataset."""

    def __init__(self, f, addr, name="/"):
        self._f, self._addr, self.name = f, addr, name
        self._msgs = f._read_header(addr)
        self.attrs = {}
        for t, body in self._msgs:
            if t == 0x000C:
                k, v = f._parse_attribute(body)
                self.attrs[k] = v

    def _first(self, t):
        for tt, body in self._msgs:
            if tt == t:
                return body
        return None

    @property
    def is_group(self):
        return self._first(0x0011) is not None or self._first(0x0002) is not None or self._first(0x0006) is not None \
            or self._first(0x0008) is None


class Dataset(_Object):
    @property
    def shape(self):
        return _parse_dataspace(self._first(0x0001))

    @property
    def dtype(self):
        return _parse_datatype(self._first(0x0003))

    def __getitem__(self, key):
        if key != () and key is not Ellipsis:
            raise H5Error("h5lite datasets are read whole: use ds[()]")
        f, lay = self._f, self._first(0x0008)
        shape, dt = self.shape, self.dtype
        n = int(np.prod(shape, dtype=np.int64)) * dt.itemsize
        ver = lay[0]
        if ver == 3:
            cls = lay[1]
            if cls == 1:
                addr = int.from_bytes(lay[2:2 + f._so], "little")
                raw = b"" if addr == UNDEF else f._data[f._base + addr:f._base + addr + n]
            elif cls == 0:
                sz = int.from_bytes(lay[2:4], "little")
                raw = lay[4:4 + sz]
            else:
                raise H5Error("%s: chunked storage is outside the Keras-weights subset" % self.name)
        elif ver in (1, 2):
            rank, cls = lay[1], lay[2]
            if cls == 1:
                addr = int.from_bytes(lay[8:8 + f._so], "little")
                raw = f._data[f._base + addr:f._base + addr + n]
            else:
                raise H5Error("%s: layout class %d (message version %d) is not supported" % (self.name, cls, ver))
        else:
            raise H5Error("%s: data layout message version %d" % (self.name, ver))
        if len(raw) < n:
            raw = bytes(raw) + b"\0" * (n - len(raw))      # never-written dataset: the fill value (zeros)
        a = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape, dtype=np.int64))).reshape(shape)
        return a.astype(dt.newbyteorder("=")) if dt.kind in "fiu" else a.copy()


class Group(_Object):
    def _links(self):
        if getattr(self, "_cache", None) is not None:
            return self._cache
        f, out = self._f, {}
        st = self._first(0x0011)
        if st is not None:
            btree, heap = int.from_bytes(st[:f._so], "little"), int.from_bytes(st[f._so:2 * f._so], "little")
            f._walk_btree(btree, f._heap_data(heap), out)
        for t, body in self._msgs:
            if t == 0x0006:
                k, addr = f._parse_link(body)
                if addr is not None:
                    out[k] = addr
        if self._first(0x0002) is not None and st is None and not out:
            li = self._first(0x0002)
            flags = li[1]
            off = 2 + (8 if flags & 1 else 0)
            fheap = int.from_bytes(li[off:off + f._so], "little")
            if fheap != UNDEF:
                raise H5Error("%s: densely stored links (fractal heap) are outside the Keras-weights subset" % self.name)
        self._cache = out
        return out

    def keys(self):
        return list(self._links().keys())

    def __contains__(self, k):
        return self._resolve(k) is not None

    def __iter__(self):
        return iter(self.keys())

    def _resolve(self, path):
        node = self
        for part in [p for p in path.split("/") if p]:
            if not isinstance(node, Group) or part not in node._links():
                return None
            node = node._f._open(node._links()[part], (node.name.rstrip("/") + "/" + part))
        return node

    def __getitem__(self, path):
        node = self._resolve(path)
        if node is None:
            raise KeyError("%s has no member %r" % (self.name, path))
        return node


class File(Group):
    def __init__(self, path, mode="r"):
        if mode != "r":
            raise H5Error("h5lite is read-only")
        with open(path, "rb") as fh:
            self._data = fh.read()
        d = self._data
        sb = -1
        off = 0
        while off < len(d):                      # the superblock sits at 0, 512, 1024, ...
            if d[off:off + 8] == SIGNATURE:
                sb = off
                break
            off = 512 if off == 0 else off * 2
        if sb < 0:
            raise H5Error("%s is not an HDF5 file (no superblock signature)" % path)
        ver = d[sb + 8]
        if ver in (0, 1):
            self._so, self._sl = d[sb + 13], d[sb + 14]
            p = sb + 24 + (4 if ver == 1 else 0)
            b = _Buf(d, self._so, self._sl)
            self._base = b.O(p)
            root_entry = p + 4 * self._so
            root = b.O(root_entry + self._so)
        elif ver in (2, 3):
            self._so, self._sl = d[sb + 9], d[sb + 10]
            b = _Buf(d, self._so, self._sl)
            self._base = b.O(sb + 12)
            root = b.O(sb + 12 + 3 * self._so)
        else:
            raise H5Error("superblock version %d" % ver)
        if self._so != 8 or self._sl != 8:
            raise H5Error("only 8-byte offsets / lengths are supported (file has %d / %d)" % (self._so, self._sl))
        self._b = b
        self._objs = {}
        self.filename = path
        _Object.__init__(self, self, root, "/")

    # h5py-style context manager
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self._data = b""
        return False

    def close(self):
        self._data = b""

    # ---- object headers ----
    def _read_header(self, addr):
        d, a = self._data, self._base + addr
        if d[a:a + 4] == b"OHDR":
            raise H5Error("version-2 object headers (libver='latest' files) are outside the Keras-weights subset; "
                          "re-save the weights with Keras / h5py defaults, or install h5py")
        if d[a] != 1:
            raise H5Error("object header version %d at %d" % (d[a], addr))
        nmsg = int.from_bytes(d[a + 2:a + 4], "little")
        size = int.from_bytes(d[a + 8:a + 12], "little")
        blocks = [(a + 16, size)]
        msgs = []
        while blocks and len(msgs) < nmsg:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(msgs) < nmsg:
                t = int.from_bytes(d[p:p + 2], "little")
                sz = int.from_bytes(d[p + 2:p + 4], "little")
                body = d[p + 8:p + 8 + sz]
                p += 8 + sz
                if t == 0x0010:
                    blocks.append((self._base + int.from_bytes(body[:8], "little"), int.from_bytes(body[8:16], "little")))
                msgs.append((t, body))
        return msgs

    def _open(self, addr, name):
        if addr in self._objs:
            return self._objs[addr]
        probe = _Object(self, addr, name)
        obj = (Group if probe.is_group else Dataset).__new__(Group if probe.is_group else Dataset)
        obj.__dict__.update(probe.__dict__)
        obj._cache = None
        self._objs[addr] = obj
        return obj

    # ---- groups ----
    def _heap_data(self, addr):
        d, a = self._data, self._base + addr
        if d[a:a + 4] != b"HEAP":
            raise H5Error("local heap signature missing at %d" % addr)
        seg = int.from_bytes(d[a + 8 + 2 * self._sl:a + 8 + 2 * self._sl + self._so], "little")
        return self._base + seg

    def _walk_btree(self, addr, heap, out):
        d, a = self._data, self._base + addr
        if d[a:a + 4] != b"TREE":
            raise H5Error("B-tree node signature missing at %d" % addr)
        level, used = d[a + 5], int.from_bytes(d[a + 6:a + 8], "little")
        p = a + 8 + 2 * self._so
        for i in range(used):
            child = int.from_bytes(d[p + self._sl:p + self._sl + self._so], "little")
            p += self._sl + self._so
            if level > 0:
                self._walk_btree(child, heap, out)
            else:
                self._read_snod(child, heap, out)

    def _read_snod(self, addr, heap, out):
        d, a = self._data, self._base + addr
        if d[a:a + 4] != b"SNOD":
            raise H5Error("symbol node signature missing at %d" % addr)
        n = int.from_bytes(d[a + 6:a + 8], "little")
        p = a + 8
        for _ in range(n):
            noff = int.from_bytes(d[p:p + self._so], "little")
            oaddr = int.from_bytes(d[p + self._so:p + 2 * self._so], "little")
            end = d.index(b"\0", heap + noff)
            out[d[heap + noff:end].decode("utf8")] = oaddr
            p += 2 * self._so + 24

    def _parse_link(self, b):
        flags = b[1]
        p = 2
        ltype = 0
        if flags & 0x08:
            ltype = b[p]
            p += 1
        if flags & 0x04:
            p += 8
        if flags & 0x10:
            p += 1
        nl = 1 << (flags & 3)
        n = int.from_bytes(b[p:p + nl], "little")
        p += nl
        name = b[p:p + n].decode("utf8")
        p += n
        if ltype != 0:
            return name, None                     # soft / external links: not followed
        return name, int.from_bytes(b[p:p + self._so], "little")

    def _global_heap_object(self, addr, index):
        d, a = self._data, self._base + addr
        if d[a:a + 4] != b"GCOL":
            raise H5Error("global heap collection signature missing at %d" % addr)
        size = int.from_bytes(d[a + 8:a + 8 + self._sl], "little")
        p, end = a + 8 + self._sl, a + size
        while p + 8 + self._sl <= end:
            idx = int.from_bytes(d[p:p + 2], "little")
            osz = int.from_bytes(d[p + 8:p + 8 + self._sl], "little")
            if idx == index:
                return d[p + 8 + self._sl:p + 8 + self._sl + osz]
            if idx == 0:
                break
            p += 8 + self._sl + _pad8(osz)
        raise H5Error("global heap object %d not found in the collection at %d" % (index, addr))

    # ---- attributes ----
    def _parse_attribute(self, b):
        ver = b[0]
        nsz = int.from_bytes(b[2:4], "little")
        tsz = int.from_bytes(b[4:6], "little")
        ssz = int.from_bytes(b[6:8], "little")
        p = 8 + (1 if ver == 3 else 0)
        pad = _pad8 if ver == 1 else (lambda n: n)
        name = b[p:p + nsz].split(b"\0", 1)[0].decode("utf8")
        p += pad(nsz)
        tb = b[p:p + tsz]
        p += pad(tsz)
        sb = b[p:p + ssz]
        p += pad(ssz)
        try:
            dt = _parse_datatype(tb)
        except H5Error:
            return name, None                     # a datatype outside the subset: the attribute is not needed here
        shape = _parse_dataspace(sb) if ssz and sb[0] in (1, 2) and not (sb[0] == 2 and sb[3] == 2) else ()
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if isinstance(dt, str):                   # variable-length strings: (length, collection address, index) each
            vals = []
            for i in range(n):
                q = p + i * (8 + self._so)
                ln = int.from_bytes(b[q:q + 4], "little")
                coll = int.from_bytes(b[q + 4:q + 4 + self._so], "little")
                idx = int.from_bytes(b[q + 4 + self._so:q + 8 + self._so], "little")
                vals.append(self._global_heap_object(coll, idx)[:ln] if ln else b"")
            a = np.array(vals, dtype=object)
            return name, (a.reshape(shape) if shape else a[0])
        a = np.frombuffer(b[p:p + n * dt.itemsize], dtype=dt, count=n)
        if dt.kind in "fiu":
            a = a.astype(dt.newbyteorder("="))
        a = a.reshape(shape) if shape else a[0]
        return name, a
