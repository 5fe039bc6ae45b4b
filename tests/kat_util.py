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
