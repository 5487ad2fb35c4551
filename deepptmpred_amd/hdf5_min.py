"""A small, dependency-free reader and writer for the HDF5 subset Keras-2 weight files use.

The reference checkpoints are Keras "legacy" HDF5 files written through h5py with its
default (earliest) format (pred/train_PTM/trainer.py:321-326 ``ModelCheckpoint``;
loaded at predict.py:569-577).  h5py is not importable in the build or run
environment, so this module parses the file format directly:

read  superblock v0/v1, object header v1 (+ continuation blocks), old-style groups
      (symbol-table message -> B-tree v1 -> symbol-table nodes -> local heap),
      new-style compact link messages, contiguous and compact dataset layouts,
      little-endian float/integer data, fixed-length and variable-length strings
      (global heap), attribute messages v1-v3.
write superblock v0, old-style groups, contiguous little-endian datasets, fixed-length
      string / numeric attributes -- the same structures libhdf5 emits for such files.

Chunked or compressed datasets, dense attribute storage and the v2+ superblocks
(libver='latest') are out of scope and raise ``H5FormatError``.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Optional, Tuple, Union

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIGNATURE = b"\x89HDF\r\n\x1a\n"


class H5FormatError(ValueError):
    pass


# =============================================================================== reader

class _Datatype:
    def __init__(self, cls: int, size: int, np_dtype=None, is_vlen_str: bool = False):
        self.cls, self.size, self.np_dtype, self.is_vlen_str = cls, size, np_dtype, is_vlen_str


class H5Object:
    """A group or a dataset."""

    def __init__(self, f: "H5File", addr: int, name: str):
        self._f, self._addr, self.name = f, addr, name
        self._msgs = f._read_object_header(addr)
        self.attrs: Dict[str, Union[np.ndarray, str, bytes]] = {}
        for t, body in self._msgs:
            if t == 0x000C:
                k, v = f._parse_attribute(body)
                self.attrs[k] = v
            elif t == 0x0015:
                raise H5FormatError("dense attribute storage is not supported")
        self._links: Optional[Dict[str, int]] = None

    # ---- group interface
    @property
    def is_group(self) -> bool:
        return any(t in (0x0011, 0x0006, 0x0002) for t, _ in self._msgs) and not self.is_dataset

    @property
    def is_dataset(self) -> bool:
        return any(t == 0x0008 for t, _ in self._msgs)

    def _load_links(self) -> Dict[str, int]:
        if self._links is None:
            links: Dict[str, int] = {}
            for t, body in self._msgs:
                if t == 0x0011:  # symbol table message: B-tree address, local heap address
                    btree, heap = struct.unpack_from("<QQ", body, 0)
                    links.update(self._f._read_group_btree(btree, heap))
                elif t == 0x0006:  # link message (compact new-style group)
                    name, addr = self._f._parse_link(body)
                    if addr is not None:
                        links[name] = addr
            self._links = links
        return self._links

    def keys(self) -> List[str]:
        return sorted(self._load_links())

    def __contains__(self, key: str) -> bool:
        try:
            self[key]
            return True
        except KeyError:
            return False

    def __getitem__(self, path: str) -> "H5Object":
        obj = self
        for part in [p for p in path.split("/") if p]:
            links = obj._load_links()
            if part not in links:
                raise KeyError(f"{obj.name}: no member {part!r}")
            obj = H5Object(self._f, links[part], obj.name.rstrip("/") + "/" + part)
        return obj

    # ---- dataset interface
    def read(self) -> np.ndarray:
        if not self.is_dataset:
            raise H5FormatError(f"{self.name} is not a dataset")
        shape = dtype = layout = None
        for t, body in self._msgs:
            if t == 0x0001:
                shape = self._f._parse_dataspace(body)
            elif t == 0x0003:
                dtype = self._f._parse_datatype(body)
            elif t == 0x0008:
                layout = body
            elif t == 0x000B:
                raise H5FormatError(f"{self.name}: filtered (compressed) datasets are not supported")
        if shape is None or dtype is None or layout is None:
            raise H5FormatError(f"{self.name}: incomplete dataset header")
        n = int(np.prod(shape)) if shape else 1
        raw = self._f._read_layout(layout, n * dtype.size)
        return self._f._decode(raw, dtype, shape)


class H5File(H5Object):
    def __init__(self, path: str):
        with open(path, "rb") as fh:
            self._buf = fh.read()
        b = self._buf
        if b[:8] != SIGNATURE:
            raise H5FormatError(f"{path}: not an HDF5 file")
        version = b[8]
        if version not in (0, 1):
            raise H5FormatError(f"{path}: superblock version {version} (libver='latest' files) is not supported")
        self._so, self._sl = b[13], b[14]
        if (self._so, self._sl) != (8, 8):
            raise H5FormatError("only 8-byte offsets/lengths are supported")
        off = 24 if version == 0 else 28
        self._base, _, _, _ = struct.unpack_from("<QQQQ", b, off)
        root_entry = off + 32
        _, root_addr = struct.unpack_from("<QQ", b, root_entry)
        super().__init__(self, root_addr, "/")

    # ---- low level
    def _read_object_header(self, addr: int) -> List[Tuple[int, bytes]]:
        b = self._buf
        addr += self._base
        if b[addr:addr + 4] == b"OHDR":
            raise H5FormatError("object header v2 (libver='latest') is not supported")
        version, _, nmsgs, _, hsize = struct.unpack_from("<BBHIi", b, addr)
        if version != 1:
            raise H5FormatError(f"object header version {version} is not supported")
        msgs: List[Tuple[int, bytes]] = []
        blocks = [(addr + 16, hsize)]
        while blocks and len(msgs) < nmsgs:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(msgs) < nmsgs:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, pos)
                body = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:  # continuation
                    coff, clen = struct.unpack_from("<QQ", body, 0)
                    blocks.append((coff + self._base, clen))
                msgs.append((mtype, body))
        return msgs

    def _heap_string(self, heap_addr: int, offset: int) -> str:
        b = self._buf
        h = heap_addr + self._base
        if b[h:h + 4] != b"HEAP":
            raise H5FormatError("bad local heap signature")
        data_addr = struct.unpack_from("<Q", b, h + 24)[0] + self._base
        s = data_addr + offset
        e = b.index(b"\x00", s)
        return b[s:e].decode("utf-8")

    def _read_group_btree(self, btree: int, heap: int) -> Dict[str, int]:
        b = self._buf
        out: Dict[str, int] = {}
        stack = [btree]
        while stack:
            a = stack.pop() + self._base
            if b[a:a + 4] == b"TREE":
                ntype, level, used = struct.unpack_from("<BBH", b, a + 4)
                if ntype != 0:
                    raise H5FormatError("unexpected B-tree node type in a group")
                pos = a + 8 + 16  # skip siblings
                for i in range(used):
                    child = struct.unpack_from("<Q", b, pos + 8 + i * 16)[0]
                    stack.append(child)
            elif b[a:a + 4] == b"SNOD":
                nsym = struct.unpack_from("<H", b, a + 6)[0]
                for i in range(nsym):
                    e = a + 8 + i * 40
                    name_off, obj_addr = struct.unpack_from("<QQ", b, e)
                    out[self._heap_string(heap, name_off)] = obj_addr
            else:
                raise H5FormatError("bad group B-tree / symbol node signature")
        return out

    def _parse_link(self, body: bytes) -> Tuple[str, Optional[int]]:
        version, flags = body[0], body[1]
        pos = 2
        ltype = 0
        if flags & 0x08:
            ltype = body[pos]; pos += 1
        if flags & 0x04:
            pos += 8
        if flags & 0x10:
            pos += 1
        lsz = 1 << (flags & 3)
        nlen = int.from_bytes(body[pos:pos + lsz], "little"); pos += lsz
        name = body[pos:pos + nlen].decode("utf-8"); pos += nlen
        if ltype != 0:
            return name, None
        return name, struct.unpack_from("<Q", body, pos)[0]

    def _parse_dataspace(self, body: bytes) -> Tuple[int, ...]:
        version, rank, flags = body[0], body[1], body[2]
        if version == 1:
            pos = 8
        elif version == 2:
            pos = 4
        else:
            raise H5FormatError(f"dataspace version {version}")
        return tuple(struct.unpack_from("<Q", body, pos + 8 * i)[0] for i in range(rank))

    def _parse_datatype(self, body: bytes) -> _Datatype:
        cls = body[0] & 0x0F
        bits0 = body[1]
        size = struct.unpack_from("<I", body, 4)[0]
        if cls == 0:  # fixed point
            if bits0 & 1:
                raise H5FormatError("big-endian integers are not supported")
            signed = bool(bits0 & 0x08)
            return _Datatype(cls, size, np.dtype(f"<{'i' if signed else 'u'}{size}"))
        if cls == 1:  # floating point
            if bits0 & 1:
                raise H5FormatError("big-endian floats are not supported")
            return _Datatype(cls, size, np.dtype(f"<f{size}"))
        if cls == 3:  # fixed-length string
            return _Datatype(cls, size, np.dtype(f"S{size}"))
        if cls == 9:  # variable length
            is_str = (bits0 & 0x0F) == 1
            if not is_str:
                raise H5FormatError("variable-length sequences are not supported")
            return _Datatype(cls, size, None, True)
        raise H5FormatError(f"datatype class {cls} is not supported")

    def _read_layout(self, body: bytes, nbytes: int) -> bytes:
        version = body[0]
        if version != 3:
            raise H5FormatError(f"data layout message version {version} is not supported")
        lclass = body[1]
        if lclass == 0:  # compact
            size = struct.unpack_from("<H", body, 2)[0]
            return body[4:4 + size][:nbytes]
        if lclass == 1:  # contiguous
            addr, size = struct.unpack_from("<QQ", body, 2)
            if addr == UNDEF:
                return b"\x00" * nbytes
            a = addr + self._base
            return self._buf[a:a + nbytes]
        raise H5FormatError("chunked datasets are not supported")

    def _global_heap_object(self, coll_addr: int, index: int) -> bytes:
        b = self._buf
        a = coll_addr + self._base
        if b[a:a + 4] != b"GCOL":
            raise H5FormatError("bad global heap signature")
        csize = struct.unpack_from("<Q", b, a + 8)[0]
        pos, end = a + 16, a + csize
        while pos + 16 <= end:
            idx, _, _, osize = struct.unpack_from("<HHIQ", b, pos)
            if idx == 0:
                break
            if idx == index:
                return b[pos + 16:pos + 16 + osize]
            pos += 16 + (osize + 7) // 8 * 8
        raise H5FormatError("global heap object not found")

    def _decode(self, raw: bytes, dt: _Datatype, shape: Tuple[int, ...]):
        n = int(np.prod(shape)) if shape else 1
        if dt.is_vlen_str:
            items = []
            for i in range(n):
                _, addr, idx = struct.unpack_from("<IQI", raw, i * 16)
                items.append(self._global_heap_object(addr, idx).decode("utf-8") if addr not in (0, UNDEF) else "")
            return items[0] if not shape else np.array(items, dtype=object).reshape(shape)
        arr = np.frombuffer(raw[: n * dt.size], dtype=dt.np_dtype).reshape(shape)
        if dt.cls == 3:
            if not shape:
                return bytes(arr.item()).rstrip(b"\x00")
            return arr
        return arr.copy()

    def _parse_attribute(self, body: bytes):
        version = body[0]
        if version == 1:
            nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
            pos = 8
            pad = lambda x: (x + 7) // 8 * 8  # noqa: E731
        elif version in (2, 3):
            nsz, tsz, ssz = struct.unpack_from("<HHH", body, 2)
            pos = 8 if version == 2 else 9
            pad = lambda x: x  # noqa: E731
        else:
            raise H5FormatError(f"attribute message version {version}")
        name = body[pos:pos + nsz].split(b"\x00")[0].decode("utf-8"); pos += pad(nsz)
        dt = self._parse_datatype(body[pos:pos + tsz]); pos += pad(tsz)
        shape = self._parse_dataspace(body[pos:pos + ssz]) if ssz else (); pos += pad(ssz)
        return name, self._decode(body[pos:], dt, shape)


# =============================================================================== writer

class _Node:
    def __init__(self):
        self.children: Dict[str, "_Node"] = {}
        self.data: Optional[np.ndarray] = None
        self.attrs: Dict[str, Union[np.ndarray, bytes, str, float, int]] = {}


class H5Writer:
    """Builds a tree in memory, then serialises it.  ``create_dataset('a/b/c', array)``
    creates intermediate groups; ``set_attr(path, name, value)`` attaches an attribute
    (numpy array, list of byte strings -> fixed-length strings, ``str``/``bytes`` ->
    scalar fixed-length string, or a Python number)."""

    LEAF_K = 32       # symbols per symbol-table node = 2K
    INTERNAL_K = 16

    def __init__(self):
        self.root = _Node()

    def _node(self, path: str, create: bool = True) -> _Node:
        n = self.root
        for part in [p for p in path.split("/") if p]:
            if part not in n.children:
                if not create:
                    raise KeyError(path)
                n.children[part] = _Node()
            n = n.children[part]
        return n

    def create_group(self, path: str) -> None:
        self._node(path)

    def create_dataset(self, path: str, data) -> None:
        self._node(path).data = np.ascontiguousarray(data)

    def set_attr(self, path: str, name: str, value) -> None:
        self._node(path).attrs[name] = value

    # ---- message encoders
    @staticmethod
    def _dtype_msg(dt: np.dtype) -> bytes:
        if dt.kind == "f":
            size = dt.itemsize
            if size == 4:
                props = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
                bits = bytes([0x20, 31, 0])
            elif size == 8:
                props = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
                bits = bytes([0x20, 63, 0])
            else:
                raise H5FormatError("only float32/float64 can be written")
            return bytes([0x11]) + bits + struct.pack("<I", size) + props
        if dt.kind in "iu":
            bits = bytes([0x08 if dt.kind == "i" else 0x00, 0, 0])
            return bytes([0x10]) + bits + struct.pack("<I", dt.itemsize) + struct.pack("<HH", 0, dt.itemsize * 8)
        if dt.kind == "S":
            return bytes([0x13, 0x01, 0, 0]) + struct.pack("<I", dt.itemsize)   # null-padded ASCII (what numpy S arrays map to)
        raise H5FormatError(f"cannot write dtype {dt}")

    @staticmethod
    def _space_msg(shape: Tuple[int, ...]) -> bytes:
        return struct.pack("<BBBB4x", 1, len(shape), 0, 0) + b"".join(struct.pack("<Q", d) for d in shape)

    @classmethod
    def _attr_msg(cls, name: str, value) -> bytes:
        if isinstance(value, str):
            value = value.encode("utf-8")
        if isinstance(value, bytes):
            arr = np.array(value, dtype=f"S{max(len(value), 1)}")
        elif isinstance(value, (list, tuple)) and value and isinstance(value[0], (bytes, str)):
            bs = [v.encode("utf-8") if isinstance(v, str) else v for v in value]
            arr = np.array(bs, dtype=f"S{max(max(len(v) for v in bs), 1)}")
        else:
            arr = np.asarray(value)
            if arr.dtype.kind == "U":
                arr = arr.astype("S")
            if arr.dtype == np.float16:
                arr = arr.astype(np.float32)
        nm = name.encode("utf-8") + b"\x00"
        dtm, spm = cls._dtype_msg(arr.dtype), cls._space_msg(arr.shape)
        pad = lambda x: x + b"\x00" * (-len(x) % 8)  # noqa: E731
        return struct.pack("<BxHHH", 1, len(nm), len(dtm), len(spm)) + pad(nm) + pad(dtm) + pad(spm) + arr.tobytes()

    @staticmethod
    def _ohdr(msgs: List[Tuple[int, bytes]]) -> bytes:
        body = b""
        for t, m in msgs:
            m = m + b"\x00" * (-len(m) % 8)
            body += struct.pack("<HHB3x", t, len(m), 0) + m
        return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body

    # ---- serialisation
    def tobytes(self) -> bytes:
        out = bytearray(b"\x00" * 96)   # superblock v0 (56 bytes + 40-byte root symbol-table entry)

        def alloc(data: bytes, align: int = 8) -> int:
            out.extend(b"\x00" * (-len(out) % align))
            addr = len(out)
            out.extend(data)
            return addr

        def write_node(node: _Node) -> Tuple[int, int, int]:
            """returns (object header address, btree address, heap address); the last two are 0 for datasets"""
            attr_msgs = [(0x000C, self._attr_msg(k, v)) for k, v in node.attrs.items()]
            if node.data is not None:
                arr = node.data
                if arr.dtype == np.float64 or arr.dtype.kind not in "fiuS":
                    arr = arr.astype(np.float32) if arr.dtype.kind == "f" else arr
                raw = arr.tobytes()
                daddr = alloc(raw, 8) if raw else UNDEF
                layout = struct.pack("<BBQQ", 3, 1, daddr, len(raw))
                msgs = [(0x0001, self._space_msg(arr.shape)), (0x0003, self._dtype_msg(arr.dtype)),
                        (0x0005, bytes([2, 2, 2, 0])),   # fill value v2: late alloc, write-if-set, undefined
                        (0x0008, layout)] + attr_msgs
                return alloc(self._ohdr(msgs)), 0, 0
            # group: children first
            entries = []
            for name in sorted(node.children):
                oh, bt, hp = write_node(node.children[name])
                entries.append((name, oh, bt, hp))
            # local heap: offset 0 holds the empty string
            heap_data = bytearray(b"\x00" * 8)
            offs = {}
            for name, *_ in entries:
                offs[name] = len(heap_data)
                nb = name.encode("utf-8") + b"\x00"
                heap_data.extend(nb + b"\x00" * (-len(nb) % 8))
            free_off = len(heap_data)
            heap_data.extend(struct.pack("<QQ", 1, 16))   # one free block: next = 1 (none), size 16
            data_addr = alloc(bytes(heap_data))
            heap_addr = alloc(b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap_data), free_off, data_addr))
            # symbol-table nodes
            per = 2 * self.LEAF_K
            chunks = [entries[i:i + per] for i in range(0, len(entries), per)] or [[]]
            if len(chunks) > 2 * self.INTERNAL_K:
                raise H5FormatError("group too large for the single-level B-tree this writer emits")
            snods, keys = [], [0]
            for ch in chunks:
                body = b"SNOD" + struct.pack("<BxH", 1, len(ch))
                for name, oh, bt, hp in ch:
                    if bt:
                        body += struct.pack("<QQI4xQQ", offs[name], oh, 1, bt, hp)
                    else:
                        body += struct.pack("<QQI4x16x", offs[name], oh, 0)
                body += b"\x00" * (40 * (per - len(ch)))
                snods.append(alloc(body))
                keys.append(offs[ch[-1][0]] if ch else 0)
            used = len(chunks) if entries else 0
            tree = b"TREE" + struct.pack("<BBHQQ", 0, 0, used, UNDEF, UNDEF)
            for i in range(used):      # key_i, child_i, ..., then the closing key directly after the last child
                tree += struct.pack("<QQ", keys[i], snods[i])
            tree += struct.pack("<Q", keys[used] if used else 0)
            tree += b"\x00" * (24 + 2 * self.INTERNAL_K * 16 + 8 - len(tree))   # nodes have a fixed size
            bt_addr = alloc(tree)
            msgs = [(0x0011, struct.pack("<QQ", bt_addr, heap_addr))] + attr_msgs
            return alloc(self._ohdr(msgs)), bt_addr, heap_addr

        root_oh, root_bt, root_hp = write_node(self.root)
        eof = len(out)
        sb = SIGNATURE + struct.pack("<BBBBBBBxHHI", 0, 0, 0, 0, 0, 8, 8, self.LEAF_K, self.INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
        sb += struct.pack("<QQI4xQQ", 0, root_oh, 1, root_bt, root_hp)
        out[:len(sb)] = sb
        return bytes(out)

    def save(self, path: str) -> None:
        with open(path, "wb") as fh:
            fh.write(self.tobytes())
