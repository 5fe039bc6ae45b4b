#!/usr/bin/env python3
"""Stand-ins for BASELINE.json configs 4 and 5 (crown, San Miguel: not in the reference repository) that exercise what such
scenes exercise, instead of the one-material heightfield of config 3 scaled up:

  * binary PLY meshes with per-vertex normals and uv (shapes/plymesh.cpp), placed by >= 100 ObjectInstances over a dozen
    object definitions, each with its own BVH (TransformedPrimitive, core/primitive.cpp:76-103): two-level traversal;
  * image-map Kd textures (MIPMap + ray differentials), a bump map, and alpha-masked foliage cards
    (Triangle::Intersect[P]'s alpha test, shapes/triangle.cpp:333-338, 531-569) under a translucent material;
  * a palette of matte / plastic / glass / metal / uber / mix / substrate / translucent materials (divergent shading);
  * an environment-mapped infinite light plus one diffuse area light;
  * config 5: the same inside a HomogeneousMedium that also surrounds the camera, VolPathIntegrator.

Everything is closed-form or drawn from a seeded numpy generator, so the same call writes the same files.  The scene is
plain pbrt-v3 input: the same file feeds the CPU reference and the MI355X path.

usage: gen_divergent.py OUT.pbrt [--tris 5000000] [--xres 1920] [--yres 1080] [--spp 256] [--volumetric]
"""
import argparse
import os
import struct
import zlib

import numpy as np

N_DEFS = 40        # object definitions of displaced spheres ("blobs"), + 1 of foliage cards: with >= 100 instances sharing the
                   # instanced total, 40 definitions keep ~ 37 % of it unique geometry (crown ~ 100 %, San Miguel ~ 25 %)
MIN_INSTANCES = 100


def write_ply(path, P, N, UV, faces):
    """binary_little_endian PLY: x y z nx ny nz u v per vertex, `uchar int vertex_indices` per face."""
    n, m = P.shape[0], faces.shape[0]
    with open(path, "wb") as f:
        f.write((f"ply\nformat binary_little_endian 1.0\nelement vertex {n}\nproperty float x\nproperty float y\nproperty float z\n"
                 f"property float nx\nproperty float ny\nproperty float nz\nproperty float u\nproperty float v\n"
                 f"element face {m}\nproperty list uchar int vertex_indices\nend_header\n").encode())
        f.write(np.ascontiguousarray(np.concatenate([P, N, UV], 1), "<f4").tobytes())
        rec = np.zeros(m, dtype=[("n", "u1"), ("v", "<i4", (3,))])
        rec["n"], rec["v"] = 3, faces
        f.write(rec.tobytes())


def write_png(path, img):
    """8-bit RGB or grey PNG (zlib only)."""
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape[:2]
    ctype = 2 if img.ndim == 3 else 0
    raw = b"".join(b"\0" + img[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
                           chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_pfm(path, img):
    img = np.ascontiguousarray(img[::-1], "<f4")
    with open(path, "wb") as f:
        f.write(f"PF\n{img.shape[1]} {img.shape[0]}\n-1.0\n".encode())
        f.write(img.tobytes())


def blob(k, n_theta, n_phi):
    """Object definition k: a unit sphere displaced by a few closed-form waves (different per k); normals from the
    undisplaced sphere mixed with the displacement gradient's direction is unnecessary -- the analytic sphere normal
    perturbed by the same waves gives smooth, non-geometric shading normals (the reference's shading-geometry path)."""
    t = np.linspace(0.0, np.pi, n_theta)[:, None]
    p = np.linspace(0.0, 2 * np.pi, n_phi)[None, :]
    a, b, c = 3 + k % 5, 2 + (k * 7) % 6, 0.08 + 0.02 * (k % 4)
    r = 1.0 + c * np.sin(a * t + 0.3 * k) * np.cos(b * p) + 0.03 * np.sin((9 + k) * t) * np.sin((11 + 2 * k) * p)
    st, ct, sp, cp = np.sin(t), np.cos(t), np.sin(p), np.cos(p)
    P = np.stack([r * st * cp, r * st * sp, r * ct * np.ones_like(p)], -1)
    wob = 0.25 * np.cos(a * t + 0.3 * k) * np.cos(b * p)
    N = np.stack([st * cp + wob * ct * cp, st * sp + wob * ct * sp, ct - wob * st], -1) * np.ones_like(r)[..., None]
    N /= np.maximum(1e-6, np.linalg.norm(N, axis=-1, keepdims=True))
    UV = np.stack([np.broadcast_to(p / (2 * np.pi), r.shape), np.broadcast_to(t / np.pi, r.shape)], -1)
    idx = (np.arange(n_theta)[:, None] * n_phi + np.arange(n_phi)[None, :])
    q0, q1, q2, q3 = idx[:-1, :-1], idx[1:, :-1], idx[1:, 1:], idx[:-1, 1:]
    faces = np.stack([np.stack([q0, q1, q2], -1), np.stack([q0, q2, q3], -1)], 2).reshape(-1, 3)
    return P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), faces.astype(np.int32)


def foliage(n_cards, rng):
    """Object definition of alpha-masked cards: unit quads at random orientations inside a unit ball, uv 0..1 per card."""
    c = rng.uniform(-1, 1, (n_cards, 3)) * np.array([1.0, 1.0, 0.8])
    u = rng.normal(size=(n_cards, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    w = np.cross(u, rng.normal(size=(n_cards, 3)))
    w /= np.maximum(1e-6, np.linalg.norm(w, axis=1, keepdims=True))
    s = rng.uniform(0.08, 0.2, (n_cards, 1))
    P = np.stack([c - s * u - s * w, c + s * u - s * w, c + s * u + s * w, c - s * u + s * w], 1).reshape(-1, 3)
    n = np.repeat(np.cross(u, w), 4, 0)
    UV = np.tile(np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float64), (n_cards, 1))
    base = 4 * np.arange(n_cards)[:, None]
    faces = np.concatenate([base + np.array([0, 1, 2]), base + np.array([0, 2, 3])], 1).reshape(-1, 3)
    return P, n, UV, faces.astype(np.int32)


def ground(g):
    i = np.arange(g, dtype=np.float64)
    ii, jj = np.meshgrid(i, i, indexing="ij")
    x, y = -12.0 + 24.0 * ii / (g - 1), -12.0 + 24.0 * jj / (g - 1)
    z = 0.25 * np.sin(0.9 * x) * np.cos(0.7 * y) + 0.04 * np.sin(5.1 * x + 3.3 * y)
    dzdx = 0.225 * np.cos(0.9 * x) * np.cos(0.7 * y) + 0.204 * np.cos(5.1 * x + 3.3 * y)
    dzdy = -0.175 * np.sin(0.9 * x) * np.sin(0.7 * y) + 0.132 * np.cos(5.1 * x + 3.3 * y)
    N = np.stack([-dzdx, -dzdy, np.ones_like(z)], -1)
    N /= np.linalg.norm(N, axis=-1, keepdims=True)
    P = np.stack([x, y, z], -1)
    UV = np.stack([ii / (g - 1) * 12, jj / (g - 1) * 12], -1)
    vid = (ii * g + jj).astype(np.int64)
    a, b, c, d = vid[:-1, :-1], vid[1:, :-1], vid[1:, 1:], vid[:-1, 1:]
    faces = np.stack([np.stack([a, b, c], -1), np.stack([a, c, d], -1)], 2).reshape(-1, 3)
    return P.reshape(-1, 3), N.reshape(-1, 3), UV.reshape(-1, 2), faces.astype(np.int32)


def textures(d, rng, res=256, prefix=""):
    """kd_*.png colour maps, leaf_alpha.png (holes: value 0), bump.png, sky.pfm (lat-long environment)."""
    y, x = np.mgrid[0:res, 0:res] / float(res)
    for name, (f1, f2, base) in {"kd_a": (6, 9, (0.7, 0.45, 0.3)), "kd_b": (13, 4, (0.35, 0.6, 0.4)), "kd_c": (3, 17, (0.5, 0.5, 0.75))}.items():
        v = 0.5 + 0.25 * np.sin(2 * np.pi * f1 * x) * np.sin(2 * np.pi * f2 * y) + 0.25 * (((x * 16).astype(int) + (y * 16).astype(int)) % 2)
        write_png(os.path.join(d, prefix + name + ".png"), (np.clip(v[..., None] * np.array(base) * 1.2, 0, 1) * 255).astype(np.uint8))
    r = np.hypot(x - 0.5, y - 0.5)
    leaf = ((r < 0.46) & (np.abs(np.sin(14 * np.arctan2(y - 0.5, x - 0.5))) * 0.25 + 0.2 < 0.5 - r * 0.2) | (r < 0.2)).astype(np.uint8) * 255
    write_png(os.path.join(d, prefix + "leaf_alpha.png"), leaf)
    write_png(os.path.join(d, prefix + "bump.png"), ((0.5 + 0.5 * np.sin(2 * np.pi * 24 * x) * np.sin(2 * np.pi * 24 * y)) * 255).astype(np.uint8))
    th, ph = np.mgrid[0:64, 0:128]
    up = np.cos(th / 64.0 * np.pi)
    sky = np.stack([0.35 + 0.25 * up, 0.45 + 0.3 * up, 0.6 + 0.4 * up], -1) * np.where(up[..., None] > 0, 1.0, 0.15)
    sun = np.exp(-(((th - 14) / 3.0) ** 2 + ((ph - 40) / 3.0) ** 2))
    write_pfm(os.path.join(d, prefix + "sky.pfm"), (sky + 30.0 * sun[..., None] * np.array([1.0, 0.9, 0.7])).astype(np.float32))


MATERIALS = [
    'Material "matte" "texture Kd" "kd_a"',
    'Material "plastic" "texture Kd" "kd_b" "rgb Ks" [ 0.4 0.4 0.4 ] "float roughness" [ 0.05 ]',
    'Material "glass" "float eta" [ 1.5 ]',
    'Material "metal" "float roughness" [ 0.02 ]',
    'Material "uber" "texture Kd" "kd_c" "rgb Ks" [ 0.3 0.3 0.3 ] "rgb Kr" [ 0.1 0.1 0.1 ] "float roughness" [ 0.1 ]',
    'Material "mix" "string namedmaterial1" "m_gold" "string namedmaterial2" "m_clay" "texture amount" "mixamt"',
    'Material "substrate" "rgb Kd" [ 0.4 0.1 0.1 ] "rgb Ks" [ 0.3 0.3 0.3 ] "float uroughness" [ 0.05 ] "float vroughness" [ 0.2 ]',
    'Material "matte" "texture Kd" "kd_a" "texture bumpmap" "bump"',
]


def write_scene(path, tris=5000000, xres=1920, yres=1080, spp=256, volumetric=False, seed=1, filename="divergent.pfm", n_defs=N_DEFS,
                tex_res=256, prefix=""):
    """Writes OUT.pbrt and its assets (PLY meshes, PNG / PFM textures, named prefix + ...) next to it; returns (triangles after
    instancing, object instances)."""
    d = os.path.dirname(os.path.abspath(path))
    rng = np.random.RandomState(seed)
    textures(d, rng, tex_res, prefix)
    # geometry budget: the ground (un-instanced) 8 % of the instanced total; the rest is shared by MIN_INSTANCES or more instances
    # of n_defs blob definitions (about a quarter of the instances are bushes of alpha-masked cards of a third of a blob's size)
    g = max(8, int(np.sqrt(0.08 * tris / 2)) + 1)
    per_def = max(128, int(0.92 * tris / MIN_INSTANCES / 0.83))
    n_theta = max(9, int(np.sqrt(per_def / 4)) + 1)
    n_phi = 2 * (n_theta - 1) + 1
    n_cards = max(64, per_def // 6)
    sizes = []
    for k in range(n_defs):
        P, N, UV, F = blob(k, n_theta, n_phi)
        write_ply(os.path.join(d, f"{prefix}blob{k}.ply"), P, N, UV, F)
        sizes.append(F.shape[0])
    P, N, UV, F = foliage(n_cards, rng)
    write_ply(os.path.join(d, prefix + "foliage.ply"), P, N, UV, F)
    P, N, UV, Fg = ground(g)
    write_ply(os.path.join(d, prefix + "ground.ply"), P, N, UV, Fg)
    # instances until the instanced total reaches `tris`: blobs on a jittered grid, one foliage instance above every third
    placed, n_inst, lines = Fg.shape[0], 0, []
    side = int(np.ceil(np.sqrt(max(MIN_INSTANCES, (tris - placed) / (np.mean(sizes) + F.shape[0] / 3.0)))))
    cells = [(i, j) for i in range(side) for j in range(side)]
    rng.shuffle(cells)
    for n, (i, j) in enumerate(cells):
        if placed >= tris and n_inst >= MIN_INSTANCES:
            break
        k = n % n_defs
        x, y = -10.5 + 21.0 * (i + rng.uniform(0.2, 0.8)) / side, -10.5 + 21.0 * (j + rng.uniform(0.2, 0.8)) / side
        s = 10.5 / side * rng.uniform(0.7, 1.0)
        z = 0.25 * np.sin(0.9 * x) * np.cos(0.7 * y) + s * 0.9
        lines.append(f'AttributeBegin\n  Translate {x:.6g} {y:.6g} {z:.6g}\n  Rotate {rng.uniform(0, 360):.5g} 0 0 1\n  Rotate {rng.uniform(-25, 25):.5g} 1 0 0\n'
                     f'  Scale {s:.6g} {s:.6g} {s * rng.uniform(0.8, 1.3):.6g}\n  ObjectInstance "{prefix}blob{k}"\nAttributeEnd\n')
        placed += sizes[k]
        n_inst += 1
        if n % 3 == 0:  # a bush of alpha-masked cards beside it
            bx, by, bs = x + 1.1 * s * np.cos(0.7 * n), y + 1.1 * s * np.sin(0.7 * n), 0.55 * s
            bz = 0.25 * np.sin(0.9 * bx) * np.cos(0.7 * by) + 0.7 * bs
            lines.append(f'AttributeBegin\n  Translate {bx:.6g} {by:.6g} {bz:.6g}\n  Rotate {rng.uniform(0, 360):.5g} 0 0 1\n  Scale {bs:.6g} {bs:.6g} {bs:.6g}\n'
                         f'  ObjectInstance "{prefix}foliage"\nAttributeEnd\n')
            placed += F.shape[0]
            n_inst += 1
    fog = ('MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [ 0.004 0.005 0.006 ] "rgb sigma_s" [ 0.03 0.025 0.02 ] "float g" [ 0.3 ]\n'
           'MediumInterface "" "fog"\n') if volumetric else ""
    with open(path, "w") as f:
        f.write(f"""# {placed} triangles after instancing, {n_inst} object instances over {n_defs + 1} definitions (+ the ground), scenes/gen_divergent.py --tris {tris}{' --volumetric' if volumetric else ''}
LookAt 0 -17 9  0 -1.5 0.3  0 0 1
{fog}Camera "perspective" "float fov" [ 42 ]
Film "image" "integer xresolution" [ {xres} ] "integer yresolution" [ {yres} ] "string filename" "{filename}"
Sampler "halton" "integer pixelsamples" [ {spp} ]
PixelFilter "box"
Integrator "{'volpath' if volumetric else 'path'}" "integer maxdepth" [ 5 ]
Accelerator "bvh"
WorldBegin
{'MediumInterface "fog" "fog"' if volumetric else ''}
AttributeBegin
  Rotate 35 0 0 1
  LightSource "infinite" "string mapname" "{prefix}sky.pfm" "rgb L" [ 1 1 1 ]
AttributeEnd
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [ 40 34 26 ]
  Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ -2 -2 9  -2 2 9  2 2 9  2 -2 9 ]
AttributeEnd
Texture "kd_a" "spectrum" "imagemap" "string filename" "{prefix}kd_a.png"
Texture "kd_b" "spectrum" "imagemap" "string filename" "{prefix}kd_b.png" "float uscale" [ 3 ] "float vscale" [ 3 ]
Texture "kd_c" "spectrum" "imagemap" "string filename" "{prefix}kd_c.png" "bool trilinear" "true"
Texture "leafalpha" "float" "imagemap" "string filename" "{prefix}leaf_alpha.png" "bool gamma" "false"
Texture "bump" "float" "imagemap" "string filename" "{prefix}bump.png" "bool gamma" "false" "float scale" [ 0.02 ] "float uscale" [ 4 ] "float vscale" [ 4 ]
Texture "mixamt" "spectrum" "checkerboard" "float uscale" [ 6 ] "float vscale" [ 6 ] "rgb tex1" [ 0.15 0.15 0.15 ] "rgb tex2" [ 0.85 0.85 0.85 ]
MakeNamedMaterial "m_gold" "string type" "metal" "float roughness" [ 0.08 ]
MakeNamedMaterial "m_clay" "string type" "matte" "rgb Kd" [ 0.55 0.35 0.25 ] "float sigma" [ 20 ]
""")
        for k in range(n_defs):
            f.write(f'AttributeBegin\n  {MATERIALS[k % len(MATERIALS)]}\n  ObjectBegin "{prefix}blob{k}"\n    Shape "plymesh" "string filename" "{prefix}blob{k}.ply"\n  ObjectEnd\nAttributeEnd\n')
        f.write('AttributeBegin\n  Material "translucent" "rgb Kd" [ 0.25 0.5 0.2 ] "rgb Ks" [ 0.1 0.1 0.1 ] "rgb reflect" [ 0.6 0.6 0.6 ] "rgb transmit" [ 0.4 0.4 0.4 ]\n'
                f'  ObjectBegin "{prefix}foliage"\n    Shape "plymesh" "string filename" "{prefix}foliage.ply" "texture alpha" "leafalpha"\n  ObjectEnd\nAttributeEnd\n')
        f.write(f'AttributeBegin\n  Material "uber" "texture Kd" "kd_b" "rgb Ks" [ 0.05 0.05 0.05 ] "float roughness" [ 0.3 ]\n  Shape "plymesh" "string filename" "{prefix}ground.ply"\nAttributeEnd\n')
        f.writelines(lines)
        f.write("WorldEnd\n")
    return placed, n_inst


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--tris", type=int, default=5000000)
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--volumetric", action="store_true")
    a = ap.parse_args()
    print(write_scene(a.out, a.tris, a.xres, a.yres, a.spp, a.volumetric))
