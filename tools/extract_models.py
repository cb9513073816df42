#!/usr/bin/env python3
"""Extracts the checked-in WNN models into small fixtures (run in the build container only):

    python tools/extract_models.py /root/reference/models tests/golden/models

The reference loads `models/*.hdf5` with the hdf5 crate (/root/reference/src/io.rs:36-92).  No HDF5
library exists in this image, so this is a minimal reader for exactly what those files use: version-0
superblock, version-1 object headers with continuation blocks, one symbol-table group, contiguous
unfiltered datasets, version-1 attributes.  Output per model: an .npz holding the seven attributes, the
bloom filter bits (packed), the f32 binarization thresholds and the input permutation -- data only;
the quantisation and everything else the loader does is restated in harness/wnn_model.py.
The 28x28 test image (benches/example_image_7.png, 254 B) is decoded with zlib and stored as raw u8.
"""
import os
import struct
import sys
import zlib

import numpy as np


class H5:
    def __init__(self, path):
        self.d = open(path, "rb").read()
        d = self.d
        assert d[:8] == b"\x89HDF\r\n\x1a\n" and d[8] == 0, "version-0 superblock expected"
        assert d[13] == 8 and d[14] == 8, "8-byte offsets and lengths expected"
        root = 24 + 4 * 8  # base, free-space, eof, driver addresses precede the root symbol-table entry
        _, self.root_header, cache, _ = struct.unpack_from("<QQII", d, root)
        assert cache == 1
        self.btree, self.heap = struct.unpack_from("<QQ", d, root + 24)

    def heap_name(self, off):
        d = self.d
        assert d[self.heap:self.heap + 4] == b"HEAP"
        seg = struct.unpack_from("<Q", d, self.heap + 24)[0]
        end = d.index(b"\0", seg + off)
        return d[seg + off:end].decode()

    def children(self):
        """name -> object header address of every link in the root group"""
        d, out = self.d, {}

        def node(addr):
            assert d[addr:addr + 4] == b"TREE"
            ntype, level, used = struct.unpack_from("<BBH", d, addr + 4)
            assert ntype == 0
            p = addr + 24
            for i in range(used):
                child = struct.unpack_from("<Q", d, p + 8)[0]  # key_i (8) then child_i (8)
                p += 16
                if level:
                    node(child)
                else:
                    assert d[child:child + 4] == b"SNOD"
                    nsym = struct.unpack_from("<H", d, child + 6)[0]
                    for s in range(nsym):
                        name_off, hdr = struct.unpack_from("<QQ", d, child + 8 + 40 * s)
                        out[self.heap_name(name_off)] = hdr

        node(self.btree)
        return out

    def messages(self, addr):
        """(type, payload) of every header message of a version-1 object header, continuations included"""
        d = self.d
        ver, _, nmsg, _, size = struct.unpack_from("<BBHII", d, addr)
        assert ver == 1
        blocks, out = [(addr + 16, size)], []
        while blocks:
            p, ln = blocks.pop(0)
            end = p + ln
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", d, p)
                body = d[p + 8:p + 8 + msize]
                p += 8 + msize
                out.append((mtype, body))
                if mtype == 0x10:
                    blocks.append(struct.unpack_from("<QQ", body))
        return out

    @staticmethod
    def dataspace(b):
        ver, rank, flags = b[0], b[1], b[2]
        assert ver == 1
        return struct.unpack_from("<%dQ" % rank, b, 8) if rank else ()

    @staticmethod
    def dtype(b):
        cls, size = b[0] & 0x0F, struct.unpack_from("<I", b, 4)[0]
        if cls == 0:  # fixed point
            signed = bool(b[1] & 0x08)
            return np.dtype("<%s%d" % ("i" if signed else "u", size))
        if cls == 1:
            return np.dtype("<f%d" % size)
        if cls == 8:  # enum (h5py bool): base type follows the member count
            return np.dtype("<i%d" % size)
        raise ValueError("datatype class %d" % cls)

    def attributes(self, addr):
        out = {}
        for mtype, b in self.messages(addr):
            if mtype != 0x0C:
                continue
            ver, _, nsz, tsz, ssz = struct.unpack_from("<BBHHH", b)
            assert ver == 1
            pad = lambda v: (v + 7) & ~7
            p = 8
            name = b[p:p + nsz].split(b"\0")[0].decode()
            p += pad(nsz)
            dt = self.dtype(b[p:p + tsz])
            p += pad(tsz)
            shape = self.dataspace(b[p:p + ssz])
            p += pad(ssz)
            cnt = int(np.prod(shape)) if shape else 1
            val = np.frombuffer(b, dt, cnt, p)
            out[name] = val.reshape(shape) if shape else val[0]
        return out

    def dataset(self, addr):
        shape = dt = data_addr = size = None
        for mtype, b in self.messages(addr):
            if mtype == 0x01:
                shape = self.dataspace(b)
            elif mtype == 0x03:
                dt = self.dtype(b)
            elif mtype == 0x08:
                assert b[0] == 3 and b[1] == 1, "contiguous layout expected"
                data_addr, size = struct.unpack_from("<QQ", b, 2)
            elif mtype == 0x0B:
                raise ValueError("filtered dataset")
        cnt = int(np.prod(shape))
        assert size == cnt * dt.itemsize
        return np.frombuffer(self.d, dt, cnt, data_addr).reshape(shape)


def png_gray_first_channel(path):
    """Minimal PNG decode (non-interlaced, 8-bit gray / RGB / RGBA / palette) -> first channel of the RGB8
    conversion, which is what io.rs:24-33 keeps."""
    d = open(path, "rb").read()
    assert d[:8] == b"\x89PNG\r\n\x1a\n"
    p, idat, plte = 8, b"", None
    while p < len(d):
        ln, typ = struct.unpack_from(">I4s", d, p)
        body = d[p + 8:p + 8 + ln]
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", body)
            assert depth == 8 and interlace == 0
        elif typ == b"PLTE":
            plte = np.frombuffer(body, np.uint8).reshape(-1, 3)
        elif typ == b"IDAT":
            idat += body
        p += 12 + ln
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    raw = zlib.decompress(idat)
    stride = w * ch
    img = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    q = 0
    for r in range(h):
        f = raw[q]
        line = np.frombuffer(raw, np.uint8, stride, q + 1).astype(np.int32)
        q += 1 + stride
        cur = np.zeros(stride, np.int32)
        for i in range(stride):
            a = cur[i - ch] if i >= ch else 0
            b = prev[i]
            c = prev[i - ch] if i >= ch else 0
            if f == 0:
                pr = 0
            elif f == 1:
                pr = a
            elif f == 2:
                pr = b
            elif f == 3:
                pr = (a + b) // 2
            else:
                pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                pr = a if pa <= pb and pa <= pc else b if pb <= pc else c
            cur[i] = (line[i] + pr) & 255
        img[r] = cur
        prev = cur
    img = img.reshape(h, w, ch)
    if ctype == 3:
        return plte[img[:, :, 0]][:, :, 0].copy()
    return img[:, :, 0].copy()  # gray: R = G = B = gray; RGB(A): R


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    for fn in sorted(os.listdir(src)):
        if not fn.endswith(".hdf5"):
            continue
        h = H5(os.path.join(src, fn))
        attrs = h.attributes(h.root_header)
        kids = h.children()
        bloom = h.dataset(kids["bloom_filters"])
        thr = h.dataset(kids["binarization_thresholds"])
        order = h.dataset(kids["input_order"])
        assert set(np.unique(bloom)) <= {0, 1}
        names = ["num_classes", "num_inputs", "bits_per_input", "num_filter_inputs", "num_filter_entries",
                 "num_filter_hashes", "p"]
        out = os.path.join(dst, fn.replace(".hdf5", ".npz"))
        np.savez_compressed(out, attrs=np.array([int(attrs[n]) for n in names], np.int64),
                            bloom_shape=np.array(bloom.shape, np.int64),
                            bloom_bits=np.packbits(bloom.astype(np.uint8).reshape(-1)),
                            thresholds_f32=thr.astype("<f4"), input_order=order.astype("<u4"))
        print(fn, {n: int(attrs[n]) for n in names}, bloom.shape, thr.shape, order.shape, os.path.getsize(out), "B")
    img = png_gray_first_channel(os.path.join(os.path.dirname(src.rstrip("/")), "benches", "example_image_7.png"))
    np.save(os.path.join(dst, "example_image_7.npy"), img)
    print("image", img.shape, int(img.sum()))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
