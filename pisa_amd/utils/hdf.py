"""Reading PISA's HDF5 event files without an HDF5 library: `from_hdf(path)` gives the nested mapping of arrays that
`pisa.utils.hdf.from_hdf` (pisa/utils/hdf.py:56-170) gives through h5py.  h5py / PyTables are not part of this image,
and the files PISA writes (`to_hdf`, hdf.py:172-330: one group per flavour / interaction, one array per variable,
chunked with the byte-shuffle filter, no compression by default, scalars contiguous) use a small part of the format:

  superblock version 0 or 1, version-1 object headers (with continuation blocks), "old style" groups (symbol table
  message -> version-1 B-tree of symbol nodes + local heap), dataspaces of version 1 / 2, fixed-point and IEEE
  floating-point datatypes, fixed-length strings, data layouts of version 3 (compact, contiguous, chunked with a
  version-1 chunk B-tree), the shuffle, deflate and fletcher32 filters, attributes of version 1 - 3.

That part is read here, following the HDF5 file format specification (version 2.0, sections II-IV); anything else
(new-style groups with fractal heaps, variable-length data, external links, ...) raises `NotImplementedError` with the
name of the feature.  Host-side file parsing with numpy; nothing here is on the device path.
"""
import struct
import zlib
from collections import OrderedDict

import numpy as np

__all__ = ["from_hdf", "HDF5File"]

_SIGNATURE = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


class HDF5File:
    def __init__(self, path):
        with open(path, "rb") as f:
            self.buf = f.read()
        self.path = path
        b = self.buf
        start = b.find(_SIGNATURE)
        if start != 0:          # (a user block may precede the superblock at 512, 1024, ...: not written by PISA)
            raise ValueError("%s is not an HDF5 file" % path)
        version = b[8]
        if version not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d (only 0 and 1: files written with the library's"
                                      " default, earliest format)" % version)
        self.size_offsets, self.size_lengths = b[13], b[14]
        if self.size_offsets != 8 or self.size_lengths != 8:
            raise NotImplementedError("HDF5 offsets / lengths of %d / %d bytes" % (self.size_offsets, self.size_lengths))
        pos = 24 + (4 if version == 1 else 0)       # v1 adds indexed-storage K and two reserved bytes
        self.base = self._u64(pos)
        root_entry = pos + 32                       # base, free-space, end-of-file, driver-info addresses
        self.root = self._symbol_entry(root_entry)

    # ---- primitives
    def _u16(self, p):
        return struct.unpack_from("<H", self.buf, p)[0]

    def _u32(self, p):
        return struct.unpack_from("<I", self.buf, p)[0]

    def _u64(self, p):
        return struct.unpack_from("<Q", self.buf, p)[0]

    def _symbol_entry(self, p):
        """(link name offset, object header address, cache type, B-tree address, heap address)"""
        name_off, header, cache = self._u64(p), self._u64(p + 8), self._u32(p + 16)
        btree = heap = None
        if cache == 1:
            btree, heap = self._u64(p + 24), self._u64(p + 32)
        return dict(name_offset=name_off, header=header + self.base, btree=btree, heap=heap)

    # ---- object headers
    def messages(self, address):
        """[(type, flags, payload position, payload size)] of a version-1 object header, continuations followed"""
        b = self.buf
        if b[address:address + 4] == b"OHDR":
            raise NotImplementedError("version-2 object headers (file written with libver='latest')")
        if b[address] != 1:
            raise ValueError("object header version %d at %d" % (b[address], address))
        n_msgs = self._u16(address + 2)
        size = self._u32(address + 8)
        blocks = [(address + 16, size)]             # the prefix is 12 bytes, padded to 16
        out = []
        while blocks and len(out) < n_msgs:
            p, left = blocks.pop(0)
            end = p + left
            while p + 8 <= end and len(out) < n_msgs:
                mtype, msize, flags = self._u16(p), self._u16(p + 2), b[p + 4]
                data = p + 8
                if mtype == 0x10:                   # continuation: (offset, length) of another block of messages
                    blocks.append((self._u64(data) + self.base, self._u64(data + 8)))
                out.append((mtype, flags, data, msize))
                p = data + msize
        return out

    # ---- groups
    def _heap_string(self, heap_address, offset):
        if self.buf[heap_address:heap_address + 4] != b"HEAP":
            raise ValueError("no local heap at %d" % heap_address)
        data = self._u64(heap_address + 24) + self.base
        end = self.buf.index(b"\x00", data + offset)
        return self.buf[data + offset:end].decode("utf-8")

    def _group_entries(self, btree, heap):
        """{name: object header address} of an old-style group"""
        out = OrderedDict()
        b = self.buf
        if b[btree:btree + 4] != b"TREE":
            raise ValueError("no B-tree node at %d" % btree)
        node_type, level, used = b[btree + 4], b[btree + 5], self._u16(btree + 6)
        assert node_type == 0
        p = btree + 24                              # signature, type, level, entries, two sibling addresses
        for i in range(used):
            child = self._u64(p + 8 + 16 * i) + self.base      # key_i (8), child_i (8), ..., key_n
            if level > 0:
                out.update(self._group_entries(child, heap))
                continue
            if b[child:child + 4] != b"SNOD":
                raise ValueError("no symbol node at %d" % child)
            n_sym = self._u16(child + 6)
            for k in range(n_sym):
                e = self._symbol_entry(child + 8 + 40 * k)
                out[self._heap_string(heap, e["name_offset"])] = e["header"]
        return out

    def children(self, header):
        """{name: header address} if the object is a group, else None"""
        for mtype, _, data, _ in self.messages(header):
            if mtype == 0x11:
                return self._group_entries(self._u64(data) + self.base, self._u64(data + 8) + self.base)
            if mtype in (0x02, 0x06):
                raise NotImplementedError("new-style groups (link info / link messages; libver='latest')")
        return None

    # ---- datasets
    def _dataspace(self, p):
        version, rank, flags = self.buf[p], self.buf[p + 1], self.buf[p + 2]
        if version == 1:
            q = p + 8
        elif version == 2:
            if self.buf[p + 3] == 2:                # null dataspace
                return None
            q = p + 4
        else:
            raise NotImplementedError("dataspace message version %d" % version)
        return tuple(self._u64(q + 8 * i) for i in range(rank))

    def _datatype(self, p):
        """numpy dtype of a datatype message"""
        cls, version = self.buf[p] & 0x0F, self.buf[p] >> 4
        bits0 = self.buf[p + 1]
        size = self._u32(p + 4)
        order = ">" if bits0 & 1 else "<"
        if cls == 0:                                # fixed point
            return np.dtype("%s%s%d" % (order, "i" if bits0 & 0x08 else "u", size))
        if cls == 1:                                # floating point (IEEE layouts of 2 / 4 / 8 bytes)
            if size not in (2, 4, 8):
                raise NotImplementedError("%d-byte floating-point type" % size)
            return np.dtype("%sf%d" % (order, size))
        if cls == 3:                                # fixed-length string
            return np.dtype("S%d" % size)
        if cls == 8:                                # enumeration (h5py stores numpy bools as an int8 enum): its base type
            return self._datatype(p + 8)
        names = {2: "time", 4: "bit field", 5: "opaque", 6: "compound", 7: "reference", 9: "variable-length", 10: "array"}
        raise NotImplementedError("HDF5 datatype class '%s' (version %d)" % (names.get(cls, cls), version))

    def _filters(self, p):
        version, n = self.buf[p], self.buf[p + 1]
        q = p + (8 if version == 1 else 2)
        out = []
        for _ in range(n):
            fid = self._u16(q)
            if version == 1 or fid >= 256:
                name_len = self._u16(q + 2)
                flags, n_cd = self._u16(q + 4), self._u16(q + 6)
                q += 8 + (((name_len + 7) // 8) * 8 if version == 1 else name_len)
            else:
                flags, n_cd = self._u16(q + 2), self._u16(q + 4)
                q += 6
            cd = [self._u32(q + 4 * i) for i in range(n_cd)]
            q += 4 * n_cd + (4 if (version == 1 and n_cd % 2) else 0)
            out.append((fid, cd))
        return out

    def _unfilter(self, raw, filters, mask, itemsize):
        for i in reversed(range(len(filters))):
            if mask & (1 << i):
                continue                            # the filter was skipped for this chunk
            fid, _ = filters[i]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:                          # shuffle: byte k of every element stored together
                n = len(raw) // itemsize
                body = np.frombuffer(raw, np.uint8, n * itemsize).reshape(itemsize, n).T.tobytes()
                raw = body + raw[n * itemsize:]
            elif fid == 3:                          # fletcher32: a 4-byte checksum behind the data
                raw = raw[:-4]
            else:
                raise NotImplementedError("HDF5 filter %d" % fid)
        return raw

    def _chunks(self, btree, rank):
        """[(chunk offsets, stored size, filter mask, address)] of a version-1 chunk B-tree"""
        b = self.buf
        if b[btree:btree + 4] != b"TREE":
            raise ValueError("no B-tree node at %d" % btree)
        node_type, level, used = b[btree + 4], b[btree + 5], self._u16(btree + 6)
        assert node_type == 1
        key_size = 8 + 8 * (rank + 1)
        p = btree + 24
        out = []
        for i in range(used):
            k = p + i * (key_size + 8)
            size, mask = self._u32(k), self._u32(k + 4)
            offsets = tuple(self._u64(k + 8 + 8 * d) for d in range(rank))
            child = self._u64(k + key_size) + self.base
            if level > 0:
                out.extend(self._chunks(child, rank))
            else:
                out.append((offsets, size, mask, child))
        return out

    def dataset(self, header):
        """the array of a dataset object (a numpy scalar for a scalar dataspace)"""
        shape = dtype = layout = None
        filters = []
        for mtype, _, data, size in self.messages(header):
            if mtype == 0x01:
                shape = self._dataspace(data)
            elif mtype == 0x03:
                dtype = self._datatype(data)
            elif mtype == 0x0B:
                filters = self._filters(data)
            elif mtype == 0x08:
                layout = (data, size)
        if dtype is None or layout is None:
            raise ValueError("object at %d is not a dataset" % header)
        if shape is None:
            return np.zeros(0, dtype=dtype.newbyteorder("="))
        p = layout[0]
        version, cls = self.buf[p], self.buf[p + 1]
        if version != 3:
            raise NotImplementedError("data layout message version %d" % version)
        count = int(np.prod(shape)) if shape else 1
        if cls == 0:                                # compact: the data sit in the message
            n = self._u16(p + 2)
            arr = np.frombuffer(self.buf, dtype, count, p + 4) if n else np.zeros(count, dtype)
        elif cls == 1:                              # contiguous
            address = self._u64(p + 2)
            if address == _UNDEF:
                arr = np.zeros(count, dtype)
            else:
                arr = np.frombuffer(self.buf, dtype, count, address + self.base)
        elif cls == 2:                              # chunked
            rank = self.buf[p + 2] - 1
            btree = self._u64(p + 3)
            chunk = tuple(self._u32(p + 11 + 4 * d) for d in range(rank))
            arr = np.zeros(shape, dtype)
            if btree != _UNDEF:
                for offsets, size, mask, address in self._chunks(btree + self.base, rank):
                    raw = self._unfilter(self.buf[address:address + size], filters, mask, dtype.itemsize)
                    block = np.frombuffer(raw, dtype, int(np.prod(chunk))).reshape(chunk)
                    sel = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offsets, chunk, shape))
                    arr[sel] = block[tuple(slice(0, s.stop - s.start) for s in sel)]
            arr = arr.reshape(-1)
        else:
            raise NotImplementedError("data layout class %d" % cls)
        arr = arr.astype(dtype.newbyteorder("="), copy=True).reshape(shape)
        return arr[()] if shape == () else arr

    # ---- attributes
    def attributes(self, header):
        out = OrderedDict()
        for mtype, _, data, _ in self.messages(header):
            if mtype != 0x0C:
                continue
            version = self.buf[data]
            name_size, type_size, space_size = self._u16(data + 2), self._u16(data + 4), self._u16(data + 6)
            if version == 1:
                pad = lambda n: (n + 7) // 8 * 8  # noqa: E731
                q = data + 8
            elif version in (2, 3):
                pad = lambda n: n  # noqa: E731
                q = data + 8 + (1 if version == 3 else 0)
            else:
                raise NotImplementedError("attribute message version %d" % version)
            name = self.buf[q:q + name_size].split(b"\x00")[0].decode("utf-8")
            q += pad(name_size)
            try:
                dtype = self._datatype(q)
            except NotImplementedError:
                continue                            # (variable-length strings and the like: not needed by the loaders)
            shape = self._dataspace(q + pad(type_size))
            q += pad(type_size) + pad(space_size)
            if shape is None:
                continue
            count = int(np.prod(shape)) if shape else 1
            val = np.frombuffer(self.buf, dtype, count, q).astype(dtype.newbyteorder("=")).reshape(shape)
            out[name] = val[()] if shape == () else val
        return out

    # ---- the whole file
    def read(self, header=None, choose=None, path=""):
        header = self.root["header"] if header is None else header
        kids = self.children(header)
        if kids is None:
            val = self.dataset(header)
            if isinstance(val, (bytes, np.bytes_)):
                val = val.decode("utf-8")
            return val
        out = OrderedDict()
        for name in sorted(kids):                   # (h5py lists the members of a group in name order)
            child_kids = self.children(kids[name])
            if child_kids is None and choose is not None and name not in choose:
                continue
            out[name] = self.read(kids[name], choose, path + "/" + name)
        return out


def from_hdf(val, return_node=None, choose=None, return_attrs=False):
    """the contents of an HDF5 file as nested OrderedDicts of numpy arrays (pisa/utils/hdf.py:56-170).  `choose`: the
    dataset names to read (others are skipped); `return_node`: only the sub-tree at that path; `return_attrs`: also the
    attributes of the root group"""
    from pisa_amd.utils.resources import find_resource

    f = HDF5File(find_resource(val) if isinstance(val, str) else val)
    header = f.root["header"]
    if return_node:
        for part in [p for p in return_node.split("/") if p]:
            kids = f.children(header)
            if kids is None or part not in kids:
                raise KeyError("no node '%s' in %s" % (return_node, f.path))
            header = kids[part]
    data = f.read(header, choose)
    if return_attrs:
        return data, f.attributes(f.root["header"])
    return data
