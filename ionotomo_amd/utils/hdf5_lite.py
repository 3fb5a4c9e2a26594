"""A small pure-numpy HDF5 reader / writer for the containers the reference keeps on disk.

The reference stores its grids and datapacks with h5py: ``TCI/{xvec,yvec,zvec,M}`` (+ frame attributes,
geometry/tri_cubic.py:81-99, inversion/solution.py:26-47) and ``datapack/*`` (astro/real_data.py:43-117): nested groups,
contiguous float64 datasets, 1-D datasets of variable-length strings (labels, patch names, timestamps) and a few scalar /
small-array attributes.  h5py is not importable in this build's interpreter, so this module speaks exactly that subset of the
HDF5 file format (version-0 superblock, symbol-table groups: local heap + version-1 B-tree + symbol nodes, version-1 object
headers, contiguous / compact layouts, global-heap collections for variable-length strings) -- what ``h5py.File(name, 'w')``
with default settings produces and what every libhdf5 reads.  Not supported (raises): chunked or filtered datasets, new-style
(version-2 object header) groups, compound / reference / enum types.

    tree = {"TCI": {"xvec": xv, "M": M, "@attrs": {"obstime": 1.2e9}}}
    write("model.hdf5", tree)
    read("model.hdf5")["TCI"]["M"]

Cross-checked against real HDF5 in the build container (tests/test_hdf5_lite.py: files written by h5py 3.3 / libhdf5 1.10.6
through the reference's own TriCubic.save are committed as fixtures and read here; files written here are read back by h5py and
h5dump where those exist).  Host-side I/O, nowhere near the hot path.
"""
import struct

import numpy as np

SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF
LEAF_K, INTERNAL_K = 16, 16           # symbol nodes of up to 32 entries, one B-tree leaf per group


def _pad8(b):
    return b + b"\0" * (-len(b) % 8)


# ================================================================================================ writer
def _dt_f64():
    return struct.pack("<B3BI", 0x11, 0x20, 0x3F, 0x00, 8) + struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)


def _dt_vlen_str():
    base = struct.pack("<B3BI", 0x13, 0x00, 0x00, 0x00, 1)                  # H5T_C_S1: 1-byte string, null-terminated
    return struct.pack("<B3BI", 0x19, 0x01, 0x01, 0x00, 16) + base           # class 9 (vlen), type = string, UTF-8


def _dataspace(shape):
    return struct.pack("<BBB5x", 1, len(shape), 0) + b"".join(struct.pack("<Q", int(n)) for n in shape)


def _message(mtype, body, flags=0):
    body = _pad8(body)
    return struct.pack("<HHB3x", mtype, len(body), flags) + body


def _object_header(messages):
    data = b"".join(messages)
    return struct.pack("<BxHII4x", 1, len(messages), 1, len(data)) + data


class _Writer(object):
    def __init__(self):
        self.buf = bytearray(96)                              # superblock, filled in last
        self.gheap = []                                       # strings of the one global-heap collection
        self.gheap_addr = None

    def alloc(self, data):
        while len(self.buf) % 8:
            self.buf += b"\0"
        addr = len(self.buf)
        self.buf += data
        return addr

    # -- variable-length strings: every string of the file lives in one global-heap collection -----------------------
    def _vlen_refs(self, strings):
        out = []
        for s in strings:
            b = s.encode("utf-8") if isinstance(s, str) else bytes(s)
            self.gheap.append(b)
            out.append((len(b), len(self.gheap)))                 # (length, object index 1..)
        return out

    def _value(self, v):
        """(datatype message body, dataspace body, raw data builder) of a dataset / attribute value."""
        if isinstance(v, str) or (isinstance(v, np.ndarray) and v.dtype.kind in "US" and v.ndim == 0):
            refs = self._vlen_refs([str(v)])
            return _dt_vlen_str(), _dataspace(()), ("vlen", refs)
        a = np.asarray(v)
        if a.dtype.kind in "USO":
            refs = self._vlen_refs([str(x) for x in a.ravel()])
            return _dt_vlen_str(), _dataspace(a.shape), ("vlen", refs)
        a = np.asarray(a, dtype=np.float64)
        return _dt_f64(), _dataspace(a.shape), ("raw", a.tobytes(order="C"))

    def _raw(self, data):
        kind, payload = data
        if kind == "raw":
            return payload
        return b"".join(struct.pack("<IQI", n, 0, idx) for n, idx in payload)      # collection address patched at the end

    def _attr_message(self, name, v):
        dt, ds, data = self._value(v)
        nm = name.encode() + b"\0"
        raw = self._raw(data)
        body = struct.pack("<BxHHH", 1, len(nm), len(dt), len(ds)) + _pad8(nm) + _pad8(dt) + _pad8(ds) + raw
        return _message(0x000C, body), (data[0] == "vlen", len(_pad8(nm)) + len(_pad8(dt)) + len(_pad8(ds)) + 8, len(raw))

    def dataset(self, v, attrs):
        dt, ds, data = self._value(v)
        raw = self._raw(data)
        addr = self.alloc(raw) if raw else UNDEF
        if data[0] == "vlen":
            self._patch.append((addr, len(raw)))
        msgs = [_message(0x0001, ds), _message(0x0003, dt, flags=1), _message(0x0005, struct.pack("<BBBB", 2, 2, 0, 0)),
                _message(0x0008, struct.pack("<BBQQ", 3, 1, addr, len(raw)))]
        return self._with_attrs(msgs, attrs)

    def _with_attrs(self, msgs, attrs):
        spots = []
        for k, v in (attrs or {}).items():
            m, (is_vlen, off, n) = self._attr_message(k, v)
            spots.append((len(b"".join(msgs)) + 8 + off, n, is_vlen))
            msgs.append(m)
        addr = self.alloc(_object_header(msgs))
        for off, n, is_vlen in spots:
            if is_vlen:
                self._patch.append((addr + 16 + off, n))
        return addr

    def group(self, tree):
        attrs = tree.get("@attrs")
        children = []
        for name in sorted(k for k in tree if not k.endswith("@attrs")):          # symbol nodes are searched by name: strcmp order
            v = tree[name]
            if isinstance(v, dict):
                children.append((name, self.group(v), True))
            else:
                children.append((name, self.dataset(v, tree.get(name + "@attrs")), False))
        if len(children) > 2 * LEAF_K:
            raise ValueError("more than %d links in one group" % (2 * LEAF_K))
        heap = bytearray(8)
        offs = []
        for name, _, _ in children:
            offs.append(len(heap))
            heap += _pad8(name.encode() + b"\0")
        heap_data = self.alloc(bytes(heap))
        heap_addr = self.alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, heap_data))      # (free list: 1 = H5HL_FREE_NULL)
        snod = bytearray(b"SNOD" + struct.pack("<BxH", 1, len(children)))
        for (name, addr, is_group), off in zip(children, offs):
            if is_group:
                snod += struct.pack("<QQII", off, addr[0], 1, 0) + struct.pack("<QQ", addr[1], addr[2])
            else:
                snod += struct.pack("<QQII16x", off, addr, 0, 0)
        snod += b"\0" * (8 + 2 * LEAF_K * 40 - len(snod))
        snod_addr = self.alloc(bytes(snod))
        tree_node = bytearray(b"TREE" + struct.pack("<BBHQQ", 0, 0, 1 if children else 0, UNDEF, UNDEF))
        tree_node += struct.pack("<QQQ", 0, snod_addr, offs[-1] if offs else 0)
        tree_node += b"\0" * (24 + (2 * INTERNAL_K) * 8 + (2 * INTERNAL_K + 1) * 8 - len(tree_node))
        btree_addr = self.alloc(bytes(tree_node))
        hdr = self._with_attrs([_message(0x0011, struct.pack("<QQ", btree_addr, heap_addr))], attrs)
        return hdr, btree_addr, heap_addr

    def finish(self, root):
        hdr, btree, heap = root
        if self.gheap:
            objs = bytearray()
            for i, b in enumerate(self.gheap):
                objs += struct.pack("<HH4xQ", i + 1, 1, len(b)) + _pad8(b)
            size = max(4096, (16 + len(objs) + 16 + 4095) // 4096 * 4096)
            free = size - 16 - len(objs)
            col = b"GCOL" + struct.pack("<B3xQ", 1, size) + bytes(objs) + struct.pack("<HH4xQ", 0, 0, free) + b"\0" * (free - 16)
            self.gheap_addr = self.alloc(col)
            for addr, n in self._patch:                      # vlen elements: {length u32, collection address u64, index u32}
                for e in range(n // 16):
                    struct.pack_into("<Q", self.buf, addr + 16 * e + 4, self.gheap_addr)
        sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, len(self.buf), UNDEF)
        sb += struct.pack("<QQII", 0, hdr, 1, 0) + struct.pack("<QQ", btree, heap)
        self.buf[:96] = sb
        return bytes(self.buf)


def write(filename, tree):
    """Write the nested dict ``tree`` (groups = dicts, datasets = float arrays or arrays / lists of str, attributes under the
    key "@attrs" of a group -- "<name>@attrs" beside a dataset -- : floats, float arrays or str) as an HDF5 file."""
    w = _Writer()
    w._patch = []
    data = w.finish(w.group(tree))
    with open(filename, "wb") as f:
        f.write(data)


# ================================================================================================ reader
class _Reader(object):
    def __init__(self, data):
        self.d = data
        if data[:8] != SIG:
            raise ValueError("not an HDF5 file")
        ver = data[8]
        if ver not in (0, 1):
            raise NotImplementedError("HDF5 superblock version %d (only the default 'earliest' format is supported)" % ver)
        if data[13] != 8 or data[14] != 8:
            raise NotImplementedError("only 8-byte offsets and lengths")
        off = 24 + (4 if ver == 1 else 0)
        self.base = struct.unpack_from("<Q", data, off)[0]
        root = off + 32
        self.root_header = struct.unpack_from("<Q", data, root + 8)[0]

    # -- object headers (version 1) ---------------------------------------------------------------------------------------
    def messages(self, addr):
        d = self.d
        ver, nmsg, _, size = struct.unpack_from("<BxHII", d, addr)
        if ver != 1:
            raise NotImplementedError("object header version %d (new-style groups are not supported)" % ver)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            p, n = blocks.pop(0)
            end = p + n
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, flags = struct.unpack_from("<HHB", d, p)
                body = d[p + 8:p + 8 + msize]
                p += 8 + msize
                if mtype == 0x0010:
                    o, l = struct.unpack_from("<QQ", body)
                    blocks.append((o + self.base, l))
                out.append((mtype, body))
        return out

    def datatype(self, b):
        cls, ver = b[0] & 0x0F, b[0] >> 4
        bits = b[1] | (b[2] << 8) | (b[3] << 16)
        size = struct.unpack_from("<I", b, 4)[0]
        order = ">" if bits & 1 else "<"
        if cls == 0:
            return ("num", np.dtype("%s%s%d" % (order, "i" if bits & 8 else "u", size)))
        if cls == 1:
            return ("num", np.dtype("%sf%d" % (order, size)))
        if cls == 3:
            return ("str", size)
        if cls == 9:
            if (bits & 0x0F) != 1:
                raise NotImplementedError("variable-length sequences")
            return ("vlen_str", 16)
        raise NotImplementedError("HDF5 datatype class %d" % cls)

    def dataspace(self, b):
        ver, rank, flags = b[0], b[1], b[2]
        if ver == 1:
            return tuple(struct.unpack_from("<%dQ" % rank, b, 8)) if rank else ()
        if ver == 2:
            if b[3] == 2:
                return None                                   # null dataspace
            return tuple(struct.unpack_from("<%dQ" % rank, b, 4)) if rank else ()
        raise NotImplementedError("dataspace version %d" % ver)

    def gheap_object(self, col, index):
        d = self.d
        if d[col:col + 4] != b"GCOL":
            raise ValueError("bad global heap collection")
        size = struct.unpack_from("<Q", d, col + 8)[0]
        p, end = col + 16, col + size
        while p + 16 <= end:
            idx, _, n = struct.unpack_from("<HH4xQ", d, p)
            if idx == index:
                return d[p + 16:p + 16 + n]
            if idx == 0:
                break
            p += 16 + (n + 7) // 8 * 8
        raise ValueError("global heap object %d not found" % index)

    def decode(self, dt, shape, raw):
        kind, info = dt
        n = int(np.prod(shape)) if shape else 1
        if kind == "num":
            a = np.frombuffer(raw, dtype=info, count=n).astype(info.newbyteorder("="))
            return a.reshape(shape) if shape else a.reshape(()).item()
        if kind == "str":
            vals = [raw[i * info:(i + 1) * info].split(b"\0")[0].decode("utf-8") for i in range(n)]
        else:
            vals = []
            for i in range(n):
                ln, col, idx = struct.unpack_from("<IQI", raw, 16 * i)
                vals.append(self.gheap_object(col + self.base, idx)[:ln].decode("utf-8") if ln else "")
        return np.array(vals, dtype=object).reshape(shape) if shape else vals[0]

    def attributes(self, msgs):
        out = {}
        for mtype, b in msgs:
            if mtype != 0x000C:
                continue
            ver = b[0]
            nsz, dsz, ssz = struct.unpack_from("<HHH", b, 2)
            p = 8 + (1 if ver == 3 else 0)
            al = (lambda x: (x + 7) // 8 * 8) if ver == 1 else (lambda x: x)
            name = b[p:p + nsz].split(b"\0")[0].decode()
            p += al(nsz)
            dt = self.datatype(b[p:p + dsz])
            p += al(dsz)
            shape = self.dataspace(b[p:p + ssz])
            p += al(ssz)
            out[name] = self.decode(dt, shape, b[p:])
        return out

    def dataset(self, msgs):
        dt = shape = layout = None
        for mtype, b in msgs:
            if mtype == 0x0003:
                dt = self.datatype(b)
            elif mtype == 0x0001:
                shape = self.dataspace(b)
            elif mtype == 0x0008:
                layout = b
        ver, cls = layout[0], layout[1]
        if ver != 3:
            raise NotImplementedError("data layout message version %d" % ver)
        n = int(np.prod(shape)) if shape else 1
        nbytes = n * (dt[1].itemsize if dt[0] == "num" else dt[1])
        if cls == 0:
            size = struct.unpack_from("<H", layout, 2)[0]
            raw = layout[4:4 + size]
        elif cls == 1:
            addr, size = struct.unpack_from("<QQ", layout, 2)
            raw = b"\0" * nbytes if addr == UNDEF else self.d[addr + self.base:addr + self.base + nbytes]
        else:
            raise NotImplementedError("chunked datasets")
        return self.decode(dt, shape, raw)

    # -- old-style groups: B-tree of symbol nodes, names in a local heap ----------------------------------------------------
    def group_links(self, btree, heap):
        d = self.d
        if d[heap:heap + 4] != b"HEAP":
            raise ValueError("bad local heap")
        heap_data = struct.unpack_from("<Q", d, heap + 24)[0] + self.base
        links = []

        def walk(node):
            if d[node:node + 4] == b"SNOD":
                nsym = struct.unpack_from("<H", d, node + 6)[0]
                for i in range(nsym):
                    noff, hdr = struct.unpack_from("<QQ", d, node + 8 + 40 * i)
                    name = d[heap_data + noff:d.index(b"\0", heap_data + noff)].decode()
                    links.append((name, hdr + self.base))
                return
            if d[node:node + 4] != b"TREE":
                raise ValueError("bad group B-tree node")
            used = struct.unpack_from("<H", d, node + 6)[0]
            for i in range(used):
                walk(struct.unpack_from("<Q", d, node + 24 + 16 * i + 8)[0] + self.base)
        walk(btree)
        return links

    def read_object(self, hdr):
        msgs = self.messages(hdr)
        st = [b for t, b in msgs if t == 0x0011]
        if st:
            btree, heap = struct.unpack_from("<QQ", st[0])
            out = {}
            for name, child in self.group_links(btree + self.base, heap + self.base):
                val = self.read_object(child)
                if isinstance(val, tuple):
                    out[name], out[name + "@attrs"] = val
                else:
                    out[name] = val
            attrs = self.attributes(msgs)
            if attrs:
                out["@attrs"] = attrs
            return out
        if any(t == 0x0002 for t, _ in msgs):
            raise NotImplementedError("new-style (link-message) groups: write the file with h5py's default libver")
        attrs = self.attributes(msgs)
        return (self.dataset(msgs), attrs) if attrs else self.dataset(msgs)


def read(filename):
    """The file as a nested dict: groups = dicts (attributes under "@attrs"), datasets = numpy arrays (strings: object arrays;
    attributes of dataset "x" under "x@attrs" beside it)."""
    with open(filename, "rb") as f:
        r = _Reader(f.read())
    return r.read_object(r.root_header + r.base)


def get(tree, path, default=None):
    """tree["a"]["b"] for path "a/b"; ``default`` when absent."""
    node = tree
    for part in path.strip("/").split("/"):
        if not isinstance(node, dict) or part not in node:
            return default
        node = node[part]
    return node


def put(tree, path, value):
    parts = path.strip("/").split("/")
    node = tree
    for part in parts[:-1]:
        node = node.setdefault(part, {})
    node[parts[-1]] = value
