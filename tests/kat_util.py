"""Helpers that restate the reference tests' fixtures: PCG32 (core/rng.h:61-144), UniformSampleSphere
(core/sampling.cpp:98-103) and the jittered triangulated sphere of Triangle.Watertight (tests/shapes.cpp:28-93)."""
import numpy as np

M64 = (1 << 64) - 1


class PCG32:
    def __init__(self, seq=None):
        if seq is None:
            self.state, self.inc = 0x853c49e6748fea9b, 0xda3e39cb94b95bdb
        else:  # SetSequence, rng.h:129-135
            self.state = 0
            self.inc = ((seq << 1) | 1) & M64
            self.uniform_uint32()
            self.state = (self.state + 0x853c49e6748fea9b) & M64
            self.uniform_uint32()

    def uniform_uint32(self):
        old = self.state
        self.state = (old * 0x5851f42d4c957f2d + self.inc) & M64
        xorshifted = (((old >> 18) ^ old) >> 27) & 0xffffffff
        rot = old >> 59
        return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xffffffff

    def uniform_uint32_bounded(self, b):
        threshold = ((~b + 1) & 0xffffffff) % b
        while True:
            r = self.uniform_uint32()
            if r >= threshold:
                return r % b

    def uniform_float(self):
        return min(np.float32(float.fromhex('0x1.fffffep-1')), np.float32(np.float32(self.uniform_uint32()) * np.float32(2.0 ** -32)))


def uniform_sample_sphere(u):
    f = np.float32
    z = f(1) - f(2) * f(u[0])
    r = f(np.sqrt(max(f(0), f(1) - z * z)))
    phi = f(2) * f(np.pi) * f(u[1])
    return np.array([r * f(np.cos(phi)), r * f(np.sin(phi)), z], np.float32)


def jittered_sphere(n_theta=16, n_phi=16, seed=12111):
    rng = PCG32(seed)
    f = np.float32
    verts = []
    for t in range(n_theta):
        theta = f(np.pi) * f(t) / f(n_theta - 1)
        ct, st = f(np.cos(theta)), f(np.sin(theta))
        for p in range(n_phi):
            phi = f(2) * f(np.pi) * f(p) / f(n_phi - 1)
            radius = f(1)
            if t == 0:
                verts.append((0.0, 0.0, float(radius)))
            elif t == n_theta - 1:
                verts.append((0.0, 0.0, float(-radius)))
            elif p == n_phi - 1:
                verts.append(verts[len(verts) - (n_phi - 1)])
            else:
                radius = radius + f(5) * rng.uniform_float()
                verts.append((float(radius * (st * f(np.cos(phi)))), float(radius * (st * f(np.sin(phi)))), float(radius * ct)))
    off = lambda t, p: t * n_phi + p
    idx = []
    for p in range(n_phi - 1):
        idx += [off(0, 0), off(1, p), off(1, p + 1)]
    for t in range(1, n_theta - 2):
        for p in range(n_phi - 1):
            idx += [off(t, p), off(t + 1, p), off(t + 1, p + 1), off(t, p), off(t + 1, p + 1), off(t, p + 1)]
    for p in range(n_phi - 1):
        idx += [off(n_theta - 1, 0), off(n_theta - 2, p), off(n_theta - 2, p + 1)]
    return verts, idx


def read_png_rgb8(path):
    """Minimal PNG decoder for 8-bit RGB / RGBA, non-interlaced (what lodepng_encode24_file and this repository's writer
    produce): returns (h, w, 3) uint8."""
    import struct
    import zlib
    import numpy as np
    b = open(path, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert zlib.crc32(typ + data) == struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0]
        if typ == b"IHDR":
            w, h, depth, ctype, _, _, interlace = struct.unpack(">IIBBBBB", data)
            assert depth == 8 and ctype in (2, 6) and interlace == 0
        elif typ == b"IDAT": idat += data
        pos += 12 + n
    bpp = 3 if ctype == 2 else 4
    raw = zlib.decompress(idat)
    stride = w * bpp
    out = np.zeros((h, stride), np.uint8)
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        ft = raw[y * (stride + 1)]
        line = np.frombuffer(raw, np.uint8, stride, y * (stride + 1) + 1).astype(np.int32)
        cur = np.zeros(stride, np.int32)
        if ft == 0: cur = line
        elif ft == 2: cur = (line + prev) & 255
        else:
            for i in range(stride):
                a = cur[i - bpp] if i >= bpp else 0
                bb = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if ft == 1: p = a
                elif ft == 3: p = (a + bb) >> 1
                else:
                    pa, pb, pc = abs(bb - c), abs(a - c), abs(a + bb - 2 * c)
                    p = a if pa <= pb and pa <= pc else (bb if pb <= pc else c)
                cur[i] = (line[i] + p) & 255
        out[y] = cur
        prev = cur
    return out.reshape(h, w, bpp)[:, :, :3]


def read_tga_rgb8(path):
    """Uncompressed or RLE 24-bit true-colour TGA -> (h, w, 3) uint8, top row first."""
    import numpy as np
    b = open(path, "rb").read()
    idlen, cmap, typ = b[0], b[1], b[2]
    w, h, bpp, desc = b[12] | b[13] << 8, b[14] | b[15] << 8, b[16], b[17]
    assert cmap == 0 and typ in (2, 10) and bpp == 24
    pos = 18 + idlen
    if typ == 2: px = np.frombuffer(b, np.uint8, 3 * w * h, pos).reshape(h, w, 3)
    else:
        out = bytearray()
        while len(out) < 3 * w * h:
            c = b[pos]; pos += 1
            n = (c & 127) + 1
            if c & 128: out += b[pos:pos + 3] * n; pos += 3
            else: out += b[pos:pos + 3 * n]; pos += 3 * n
        px = np.frombuffer(bytes(out), np.uint8).reshape(h, w, 3)
    if not (desc & 0x20): px = px[::-1]
    return px[:, :, ::-1].copy()
