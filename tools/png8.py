"""Minimal PNG codec for 8-bit images (the image has no OpenCV / PIL): enough for HPatches' grayscale patch strips.
read(path) -> H x W uint8 (colour types 0 gray, 2 RGB, 4 gray+alpha, 6 RGBA; 8 bit; non-interlaced; colour is reduced
to gray with the BGR2GRAY integers of spec S11, alpha dropped).  write(path, img) writes a gray 8-bit PNG."""
import struct
import zlib

import numpy as np


def read(path):
    data = open(path, "rb").read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    ch = {0: 1, 2: 3, 4: 2, 6: 4}.get(ctype)
    if depth != 8 or ch is None or interlace:
        raise ValueError("unsupported PNG (need 8-bit, non-interlaced, gray/RGB[A])")
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    prev = np.zeros(w * ch, np.uint8)
    for y in range(h):
        f, line = raw[y, 0], raw[y, 1:].copy()
        if f == 1:
            for i in range(ch, w * ch):
                line[i] = (int(line[i]) + int(line[i - ch])) & 0xff
        elif f == 2:
            line = (line.astype(np.int16) + prev).astype(np.uint8)
        elif f == 3:
            for i in range(w * ch):
                left = int(line[i - ch]) if i >= ch else 0
                line[i] = (int(line[i]) + ((left + int(prev[i])) >> 1)) & 0xff
        elif f == 4:
            for i in range(w * ch):
                a = int(line[i - ch]) if i >= ch else 0
                b = int(prev[i])
                c = int(prev[i - ch]) if i >= ch else 0
                p = a + b - c
                pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                pr = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                line[i] = (int(line[i]) + pr) & 0xff
        out[y] = line
        prev = line
    img = out.reshape(h, w, ch)
    if ch <= 2:
        return np.ascontiguousarray(img[:, :, 0])
    r, g, b = (img[:, :, k].astype(np.uint32) for k in range(3))
    return ((3735 * b + 19235 * g + 9798 * r + 16384) >> 15).astype(np.uint8)


def write(path, img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(h))

    def chunk(typ, body):
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))
