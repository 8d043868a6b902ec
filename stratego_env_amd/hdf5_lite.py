"""A small pure-Python reader for the HDF5 files the reference's curriculum loader opens (game/util.py:322-387: datasets 'state'
[n,34,R,C] and 'winner' [n] at the root of the file), for boxes without h5py -- the MI355X image has none.

Not a general HDF5 library.  It reads what h5py writes for plain numeric datasets:

  * superblock versions 0 / 1 (h5py's default `libver='earliest'`) and 2 / 3 (`libver='latest'`);
  * root group as a symbol table (B-tree v1 + local heap) or as link messages in a version-2 object header (compact groups);
  * object headers version 1 and 2, continuation blocks included;
  * dataspace versions 1 / 2 (simple), datatypes fixed-point (8 .. 64 bit, signed / unsigned, either byte order) and IEEE float
    (16 / 32 / 64 bit);
  * data layout versions 1 - 3: compact, contiguous, chunked (B-tree v1 chunk index); layout version 4: compact, contiguous, and the
    "single chunk" index; filter pipeline: deflate, shuffle, fletcher32.

Anything else (dense groups, layout-4 chunk indexes other than a single chunk, compound / string / variable-length types, other
filters) raises Hdf5LiteError with the advice to install h5py or convert the table to .npz.  Pinned by real files:
tools/oracle/gen_golden_curriculum_h5.py writes the fixtures under tests/golden/ with the real h5py (contiguous; chunked + gzip +
shuffle + fletcher32, resizable; libver='latest') and records what the reference reads from them; tests/test_hdf5_lite.py compares.
Format: "HDF5 File Format Specification Version 3.0" (The HDF Group), the sections named in the comments below.
"""
import zlib

import numpy as np

SIGNATURE = b'\x89HDF\r\n\x1a\n'


class Hdf5LiteError(ValueError):
    pass


_MAX_NODES = 1 << 20              # object header blocks / B-tree nodes one call may visit
_MAX_DATASET_BYTES = 1 << 31      # a dataset is read whole


def _malformed(what):
    raise Hdf5LiteError("malformed HDF5 file: %s" % what)


def _guarded(fn):
    """Entry points raise Hdf5LiteError (a ValueError) on anything a corrupt file can provoke -- never an IndexError from the parser."""
    def wrapper(self, *a, **k):
        try:
            return fn(self, *a, **k)
        except (Hdf5LiteError, KeyError, OSError):
            raise
        except (IndexError, TypeError, ValueError, ArithmeticError, RecursionError, MemoryError, UnicodeDecodeError, zlib.error) as e:
            raise Hdf5LiteError("malformed HDF5 file (%s: %s)" % (type(e).__name__, e)) from e
    wrapper.__doc__ = fn.__doc__
    return wrapper


def _unsupported(what):
    raise Hdf5LiteError("hdf5_lite cannot read this file (%s): install h5py, or convert the table with "
                        "np.savez(path, state=..., winner=...)" % what)


class _Reader:
    def __init__(self, data):
        self.d = data
        self.O = self.L = 8          # size of offsets / lengths (superblock)
        self.base = 0

    def u(self, at, n):
        return int.from_bytes(self.d[at:at + n], 'little')

    def off(self, at):
        v = self.u(at, self.O)
        return None if v == (1 << (8 * self.O)) - 1 else v + self.base

    def ln(self, at):
        return self.u(at, self.L)


class File:
    """`File(path)['state']` -> numpy array (the whole dataset; these tables are small)."""

    @_guarded
    def __init__(self, path):
        with open(path, 'rb') as fh:
            data = fh.read()
        r = self.r = _Reader(data)
        self._nodes = 0
        at = 0
        while data[at:at + 8] != SIGNATURE:                  # the superblock may sit at 0, 512, 1024, ... (III.A)
            at = 512 if at == 0 else at * 2
            if at + 8 > len(data):
                raise Hdf5LiteError("%s is not an HDF5 file" % path)
        ver = data[at + 8]
        if ver in (0, 1):
            r.O, r.L = data[at + 13], data[at + 14]
            p = at + 24 + (4 if ver == 1 else 0)
            r.base = r.u(p, r.O)
            p += 4 * r.O                                     # base, free-space info, end of file, driver info
            root_header = r.off(p + r.O)                     # root symbol table entry: link name offset, object header address
        elif ver in (2, 3):
            r.O, r.L = data[at + 9], data[at + 10]
            p = at + 12
            r.base = r.u(p, r.O)
            root_header = r.off(p + 3 * r.O)                 # base, superblock extension, end of file, root object header
        else:
            _unsupported("superblock version %d" % ver)
        if r.O not in (2, 4, 8) or r.L not in (2, 4, 8) or root_header is None:
            _malformed("superblock")
        self._links = self._group_links(root_header)

    # ---- object headers (IV.A.1) ---------------------------------------------------------------------------------
    def _messages(self, addr):
        """[(type, flags, body bytes)] of the object header at `addr`, continuation blocks followed."""
        r, d = self.r, self.r.d
        out = []
        if d[addr:addr + 4] == b'OHDR':                      # version 2
            flags = d[addr + 5]
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            n = 1 << (flags & 3)
            size = r.u(p, n)
            p += n
            blocks = [(p, p + size)]
            order = bool(flags & 0x04)
            seen = 0
            while blocks:
                seen += 1
                if seen > _MAX_NODES:
                    _malformed("object header continuation blocks do not end")
                p, end = blocks.pop(0)
                while p + 4 <= end:
                    mtype, msize, mflags = d[p], r.u(p + 1, 2), d[p + 3]
                    p += 4 + (2 if order else 0)
                    body = d[p:p + msize]
                    if mtype == 0x10:                        # continuation: offset, length -> an 'OCHK' block, checksum at its end
                        caddr, clen = r.off(p), r.ln(p + r.O)
                        if caddr is None:
                            _malformed("undefined continuation address")
                        blocks.append((caddr + 4, caddr + clen - 4))
                    elif mtype != 0:
                        out.append((mtype, mflags, body))
                    p += msize
        else:                                                # version 1
            if d[addr] != 1:
                _unsupported("object header version %d" % d[addr])
            nmsg, size = r.u(addr + 2, 2), r.u(addr + 8, 4)
            blocks = [(addr + 16, addr + 16 + size)]
            seen = 0
            while blocks and len(out) < nmsg + 64:
                seen += 1
                if seen > _MAX_NODES:
                    _malformed("object header continuation blocks do not end")
                p, end = blocks.pop(0)
                while p + 8 <= end:
                    mtype, msize, mflags = r.u(p, 2), r.u(p + 2, 2), d[p + 4]
                    p += 8
                    body = d[p:p + msize]
                    if mtype == 0x10:
                        if r.off(p) is None:
                            _malformed("undefined continuation address")
                        blocks.append((r.off(p), r.off(p) + r.ln(p + r.O)))
                    elif mtype != 0:
                        out.append((mtype, mflags, body))
                    p += msize
        return out

    # ---- groups ---------------------------------------------------------------------------------------------------
    def _group_links(self, addr):
        r, d = self.r, self.r.d
        links = {}
        for mtype, _, body in self._messages(addr):
            if mtype == 0x11:                                # symbol table message: B-tree v1 + local heap (IV.A.2.r)
                btree, heap = self._off_in(body, 0), self._off_in(body, r.O)
                if d[heap:heap + 4] != b'HEAP':
                    _unsupported("local heap signature")
                heap_data = self._off_abs(heap + 8 + 2 * r.L)
                self._walk_group_btree(btree, heap_data, links)
            elif mtype == 0x06:                              # link message (IV.A.2.g)
                flags = body[1]
                p = 2
                ltype = 0
                if flags & 0x08:
                    ltype = body[p]; p += 1
                if flags & 0x04:
                    p += 8
                if flags & 0x10:
                    p += 1
                n = 1 << (flags & 3)
                nlen = int.from_bytes(body[p:p + n], 'little'); p += n
                name = bytes(body[p:p + nlen]).decode('utf-8'); p += nlen
                if ltype == 0:
                    links[name] = self._off_in(body, p)
            elif mtype == 0x02:                              # link info: dense link storage (fractal heap) is out of scope
                flags = body[1]
                p = 2 + (8 if flags & 1 else 0)
                if self._off_in(body, p) is not None:
                    _unsupported("a group with dense link storage")
        return links

    def _visit(self, addr, depth):
        """Bounds a B-tree walk: a corrupt file may link a node to itself."""
        self._nodes += 1
        if addr is None or depth > 64 or self._nodes > _MAX_NODES:
            _malformed("B-tree does not end")

    def _off_in(self, body, p):
        v = int.from_bytes(body[p:p + self.r.O], 'little')
        return None if v == (1 << (8 * self.r.O)) - 1 else v + self.r.base

    def _off_abs(self, at):
        return self.r.off(at)

    def _walk_group_btree(self, addr, heap_data, links, depth=0):
        r, d = self.r, self.r.d
        self._visit(addr, depth)
        if d[addr:addr + 4] != b'TREE' or d[addr + 4] != 0:
            _unsupported("group B-tree node")
        level, used = d[addr + 5], r.u(addr + 6, 2)
        self._nodes += used
        p = addr + 8 + 2 * r.O
        for i in range(used):
            child = r.off(p + r.L)                           # key (length), child (offset), key, child, ..., key
            p += r.L + r.O
            if level > 0:
                self._walk_group_btree(child, heap_data, links, depth + 1)
                continue
            if d[child:child + 4] != b'SNOD':
                _unsupported("symbol table node")
            nsym = r.u(child + 6, 2)
            q = child + 8
            for _ in range(nsym):
                name_off, header = r.u(q, r.O), r.off(q + r.O)   # symbol table entry: link name offset, object header address, ...
                end = d.index(b'\0', heap_data + name_off)
                links[bytes(d[heap_data + name_off:end]).decode('utf-8')] = header
                q += 2 * r.O + 24

    def keys(self):
        return list(self._links.keys())

    def __contains__(self, name):
        return name in self._links

    # ---- datasets -------------------------------------------------------------------------------------------------
    @_guarded
    def __getitem__(self, name):
        if name not in self._links:
            raise KeyError(name)
        r, d = self.r, self.r.d
        self._nodes = 0
        shape = dtype = layout = None
        filters = []
        for mtype, _, body in self._messages(self._links[name]):
            if mtype == 0x01:                                # dataspace (IV.A.2.b)
                ver, rank = body[0], body[1]
                p = 8 if ver == 1 else 4
                shape = tuple(int.from_bytes(body[p + i * r.L:p + (i + 1) * r.L], 'little') for i in range(rank))
            elif mtype == 0x03:                              # datatype (IV.A.2.d)
                cls, bits0, size = body[0] & 15, body[1], int.from_bytes(body[4:8], 'little')
                order = '>' if (bits0 & 1) else '<'
                if cls == 0:
                    dtype = np.dtype('%s%s%d' % (order, 'i' if (bits0 & 8) else 'u', size))
                elif cls == 1 and size in (2, 4, 8):
                    dtype = np.dtype('%sf%d' % (order, size))
                else:
                    _unsupported("datatype class %d of %d bytes" % (cls, size))
            elif mtype == 0x08:                              # data layout (IV.A.2.i)
                layout = bytes(body)
            elif mtype == 0x0B:                              # filter pipeline (IV.A.2.l)
                ver, nf = body[0], body[1]
                p = 8 if ver == 1 else 2
                for _ in range(nf):
                    fid = int.from_bytes(body[p:p + 2], 'little'); p += 2
                    nlen = 0
                    if ver == 1 or fid >= 256:
                        nlen = int.from_bytes(body[p:p + 2], 'little'); p += 2
                    p += 2                                   # flags
                    ncv = int.from_bytes(body[p:p + 2], 'little'); p += 2
                    p += (nlen + 7) & ~7 if ver == 1 else nlen
                    cv = [int.from_bytes(body[p + 4 * i:p + 4 * i + 4], 'little') for i in range(ncv)]
                    p += 4 * ncv + (4 if (ver == 1 and ncv % 2) else 0)
                    filters.append((fid, cv))
        if shape is None or dtype is None or layout is None:
            _unsupported("dataset %r without dataspace / datatype / layout" % name)
        n = 1
        for extent in shape:
            n *= int(extent)
        if n * dtype.itemsize > _MAX_DATASET_BYTES:
            _unsupported("dataset %r of %d bytes (curriculum tables are small; this reader holds a dataset in memory)" % (name, n * dtype.itemsize))
        lver = layout[0]
        if lver in (1, 2):                                   # (HDF5 1.6 and older) version, dimensionality, class, 5 reserved, address, sizes
            ndim, lcls = layout[1], layout[2]
            if lcls == 1:
                addr = self._off_in(layout, 8)
                return np.zeros(shape, dtype=dtype) if addr is None else np.frombuffer(d, dtype=dtype, count=n, offset=addr).reshape(shape).copy()
            if lcls == 2:
                btree = self._off_in(layout, 8)
                cdims = tuple(int.from_bytes(layout[8 + r.O + 4 * i:8 + r.O + 4 * i + 4], 'little') for i in range(ndim - 1))
                out = np.zeros(shape, dtype=dtype)
                if btree is not None:
                    self._walk_chunk_btree(btree, ndim - 1, cdims, dtype, filters, out)
                return out
            _unsupported("data layout version %d class %d" % (lver, lcls))
        lcls = layout[1]
        if lver not in (3, 4):
            _unsupported("data layout version %d" % lver)
        if lcls == 0:                                        # compact: the data sits in the message
            size = int.from_bytes(layout[2:4], 'little')
            return np.frombuffer(layout[4:4 + size], dtype=dtype, count=n).reshape(shape).copy()
        if lcls == 1:                                        # contiguous
            addr = self._off_in(layout, 2)
            if addr is None:
                return np.zeros(shape, dtype=dtype)
            return np.frombuffer(d, dtype=dtype, count=n, offset=addr).reshape(shape).copy()
        if lcls != 2:
            _unsupported("data layout class %d" % lcls)
        out = np.zeros(shape, dtype=dtype)
        if lver == 3:
            rank = layout[2] - 1
            btree = self._off_in(layout, 3)
            cdims = tuple(int.from_bytes(layout[3 + r.O + 4 * i:3 + r.O + 4 * i + 4], 'little') for i in range(rank))
            if btree is not None:
                self._walk_chunk_btree(btree, rank, cdims, dtype, filters, out)
            return out
        # version 4: flags, dimensionality, encoded size of a dimension, dimensions, chunk index type
        flags, rank, enc = layout[2], layout[3] - 1, layout[4]
        cdims = tuple(int.from_bytes(layout[5 + enc * i:5 + enc * (i + 1)], 'little') for i in range(rank))
        p = 5 + enc * (rank + 1)
        itype = layout[p]; p += 1
        if itype != 1:
            _unsupported("chunk index type %d of a version-4 layout (written with libver='latest')" % itype)
        size, mask = n * dtype.itemsize, 0
        if flags & 2:                                        # single chunk with filters: its stored size and filter mask come first
            size, mask = int.from_bytes(layout[p:p + r.L], 'little'), int.from_bytes(layout[p + r.L:p + r.L + 4], 'little')
            p += r.L + 4
        addr = self._off_in(layout, p)
        if addr is not None:
            self._place_chunk(d[addr:addr + size], mask, (0,) * rank, cdims, dtype, filters, out)
        return out

    def _walk_chunk_btree(self, addr, rank, cdims, dtype, filters, out, depth=0):
        r, d = self.r, self.r.d
        self._visit(addr, depth)
        if d[addr:addr + 4] != b'TREE' or d[addr + 4] != 1:
            _unsupported("chunk B-tree node")
        level, used = d[addr + 5], r.u(addr + 6, 2)
        self._nodes += used
        p = addr + 8 + 2 * r.O
        key = 8 + 8 * (rank + 1)                             # chunk size, filter mask, offsets (one more for the element size)
        for i in range(used):
            size, mask = r.u(p, 4), r.u(p + 4, 4)
            offs = tuple(r.u(p + 8 + 8 * j, 8) for j in range(rank))
            child = r.off(p + key)
            p += key + r.O
            if level > 0:
                self._walk_chunk_btree(child, rank, cdims, dtype, filters, out, depth + 1)
            else:
                self._place_chunk(d[child:child + size], mask, offs, cdims, dtype, filters, out)

    @staticmethod
    def _place_chunk(raw, mask, offs, cdims, dtype, filters, out):
        raw = bytes(raw)
        for i in reversed(range(len(filters))):              # the pipeline is undone from its last filter to its first
            fid, cv = filters[i]
            if mask & (1 << i):
                continue
            if fid == 1:                                     # deflate, bounded by what a chunk can hold (+ its checksum)
                limit = 1
                for c in cdims:
                    limit *= int(c)
                limit = limit * dtype.itemsize + 4
                if limit > _MAX_DATASET_BYTES:
                    _malformed("chunk of %d bytes" % limit)
                z = zlib.decompressobj()
                raw = z.decompress(raw, limit + 1)
                if len(raw) > limit:
                    _malformed("a chunk inflates beyond its dimensions")
            elif fid == 2:                                   # shuffle: byte k of every element was stored together
                es = int(cv[0]) if cv else dtype.itemsize
                if not 1 <= es <= 16:                        # (h5py writes the element size; 0 would divide by zero below)
                    _malformed("shuffle filter with element size %d" % es)
                a = np.frombuffer(raw, dtype=np.uint8)
                m = len(a) // es
                raw = a[:m * es].reshape(es, m).T.tobytes() + a[m * es:].tobytes()
            elif fid == 3:                                   # fletcher32: the checksum follows the data
                raw = raw[:-4]
            else:
                _unsupported("filter %d" % fid)
        chunk = np.frombuffer(raw, dtype=dtype, count=int(np.prod(cdims))).reshape(cdims)
        sel_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, cdims, out.shape))
        sel_in = tuple(slice(0, s.stop - s.start) for s in sel_out)
        if all(s.stop > s.start for s in sel_out):
            out[sel_out] = chunk[sel_in]

    def close(self):
        self.r = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
