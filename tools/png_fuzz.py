"""Byte-flip / truncation / length-field fuzz of the PNG decoder (demfi_png_info / demfi_png_decode, png_codec.cpp): meant to run under the
sanitizer build (tools/asan_check.sh); every call must return a status -- never read or write out of bounds, whatever the bytes say.

    python tools/png_fuzz.py [iterations]
"""
import ctypes as C
import os
import struct
import sys
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                       # noqa: E402

from demfi_amd import _lib as L                          # noqa: E402


def chunk(tag, data):
    return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)


def make_png(h, w, color_type, depth, rng, palette=False):
    ch = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    bpp = ch * depth // 8 if depth >= 8 else 1
    row = (w * ch * depth + 7) // 8
    raw = b''.join(bytes([rng.randint(0, 5)]) + rng.bytes(row) for _ in range(h))
    out = b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, color_type, 0, 0, 0))
    if color_type == 3:
        out += chunk(b'PLTE', rng.bytes(3 * (1 << depth)))
    return out + chunk(b'IDAT', zlib.compress(raw, 1)) + chunk(b'IEND', b'')


def main():
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    lib = L.load()
    rng = np.random.RandomState(1234)
    seeds = []
    for (ct, d) in ((0, 8), (0, 16), (2, 8), (2, 16), (3, 8), (3, 4), (3, 2), (3, 1), (4, 8), (6, 8), (6, 16), (0, 1), (0, 2), (0, 4)):
        for (h, w) in ((1, 1), (5, 7), (16, 33)):
            seeds.append(make_png(h, w, ct, d, rng))
    # the library's own encoder output as well
    img = rng.randint(0, 256, size=(24, 40, 3)).astype(np.uint8)
    cap = lib.demfi_png_encode_bound(24, 40)
    buf = np.zeros(cap, np.uint8)
    nb = C.c_int64(0)
    assert lib.demfi_png_encode(img.ctypes.data, 24, 40, 40 * 3, 1, -1, -1, buf.ctypes.data, cap, C.byref(nb)) == 0
    seeds.append(bytes(buf[:nb.value]))
    ok = bad = 0
    for it in range(n_iter):
        b = bytearray(seeds[it % len(seeds)])
        mode = rng.randint(0, 6)
        if mode == 0:                                       # flip a few bytes anywhere
            for _ in range(rng.randint(1, 6)):
                b[rng.randint(0, len(b))] ^= 1 << rng.randint(0, 8)
        elif mode == 1:                                     # truncate
            b = b[:rng.randint(0, len(b) + 1)]
        elif mode == 2 and len(b) > 33:                     # corrupt a length field
            struct.pack_into('>I', b, 8 if rng.randint(0, 2) else 33, int(rng.choice([0, 1, 13, 2 ** 31 - 1, 2 ** 32 - 1, rng.randint(0, 2 ** 32)])))
        elif mode == 3 and len(b) > 24:                     # rewrite the header dimensions / type / depth (CRC recomputed: the parser goes on)
            w, h = int(rng.choice([0, 1, 7, 33, 65535, 2 ** 31 - 1])), int(rng.choice([0, 1, 5, 16, 2 ** 24]))
            ihdr = struct.pack('>IIBBBBB', w, h, int(rng.choice([1, 2, 4, 8, 16, 3])), int(rng.choice([0, 2, 3, 4, 6, 5])), 0, 0, int(rng.randint(0, 2)))
            b[8:33] = chunk(b'IHDR', ihdr)
        elif mode == 4:                                     # garbage appended / chunks duplicated
            b += bytes(rng.bytes(rng.randint(1, 64)))
        else:                                               # corrupt the compressed stream only
            i = bytes(b).find(b'IDAT')
            if i > 0 and i + 12 < len(b):
                b[i + 4 + rng.randint(0, max(1, len(b) - i - 16))] ^= 0xff
        data = np.frombuffer(bytes(b), np.uint8)
        h, w = C.c_int(0), C.c_int(0)
        st = lib.demfi_png_info(data.ctypes.data if data.size else None, data.size, C.byref(h), C.byref(w))
        if st == 0 and 0 < h.value <= 4096 and 0 < w.value <= 4096:
            out = np.zeros((h.value, w.value, 3), np.uint8)
            st = lib.demfi_png_decode(data.ctypes.data, data.size, out.ctypes.data, w.value * 3, h.value, w.value)
        ok += st == 0
        bad += st != 0
    print('png fuzz: %d inputs, %d decoded, %d rejected, no sanitizer report' % (n_iter, ok, bad))


if __name__ == '__main__':
    main()
