#!/usr/bin/env python3
"""Generates tests/golden/: small scenes rendered by the UNMODIFIED reference binary
(oracle/_ref/pbrt_oracle, built from /root/reference by Makefile.ref) plus its own ray statistics.

Run in the build container (needs /root/reference); the outputs are committed so the GPU box,
which has no /root/reference, can still check against real reference output.
"""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, os.path.join(ROOT, "scenes"))
import gen_synthetic  # noqa: E402
import gen_divergent  # noqa: E402

CORNELL = open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read()


def cornell(xres, yres, spp, extra_film="", integrator='Integrator "path" "integer maxdepth" [ 5 ]', world_edit=None):
    s = CORNELL
    s = s.replace('"integer xresolution" [ 512 ] "integer yresolution" [ 512 ]',
                  f'"integer xresolution" [ {xres} ] "integer yresolution" [ {yres} ] {extra_film}')
    s = s.replace('"integer pixelsamples" [ 256 ]', f'"integer pixelsamples" [ {spp} ]')
    s = s.replace('Integrator "path" "integer maxdepth" [ 5 ]', integrator)
    if world_edit:
        s = world_edit(s)
    return s


DELTA_POINT = 'AttributeBegin\n  Translate 0 -60 0\n  LightSource "point" "point from" [ 150 500 150 ] "rgb I" [ 40000 60000 90000 ] "rgb scale" [ 2 1 0.5 ]\nAttributeEnd\n'
DELTA_SPOT = ('AttributeBegin\n  Rotate 10 0 0 1\n  LightSource "spot" "point from" [ 400 480 100 ] "point to" [ 200 0 300 ] "rgb I" [ 300000 250000 200000 ] '
              '"float coneangle" [ 35 ] "float conedeltaangle" [ 12 ]\nAttributeEnd\n')
DELTA_DISTANT = 'LightSource "distant" "point from" [ 0.3 1 -0.6 ] "point to" [ 0 0 0 ] "rgb L" [ 0.6 0.7 0.9 ]\n'


SPHERES = ('AttributeBegin\n  Translate 420 60 120\n  Material "matte" "rgb Kd" [ 0.2 0.3 0.7 ]\n  Shape "sphere" "float radius" [ 60 ]\nAttributeEnd\n'
           'AttributeBegin\n  Translate 130 380 330\n  Material "mirror"\n  Shape "sphere" "float radius" [ 70 ]\nAttributeEnd\n'
           'AttributeBegin\n  Translate 300 250 80\n  Material "glass" "float index" [ 1.5 ]\n  Shape "sphere" "float radius" [ 45 ]\nAttributeEnd\n')
SPHERES_PARTIAL = ('AttributeBegin\n  Translate 400 120 150\n  Rotate 40 1 0.2 0.1\n  Scale 1 1.4 0.7\n  Material "plastic" "rgb Kd" [ 0.6 0.3 0.1 ]\n'
                   '  Shape "sphere" "float radius" [ 80 ] "float zmin" [ -50 ] "float zmax" [ 60 ] "float phimax" [ 250 ]\nAttributeEnd\n'
                   'AttributeBegin\n  Translate 150 420 300\n  Rotate -70 1 0 0\n  ReverseOrientation\n  AreaLightSource "diffuse" "rgb L" [ 9 9 12 ] "bool twosided" "true"\n'
                   '  Shape "sphere" "float radius" [ 50 ] "float zmin" [ -20 ] "float phimax" [ 300 ]\nAttributeEnd\n'
                   'AttributeBegin\n  Translate 500 500 500\n  AreaLightSource "area" "rgb L" [ 4000 3000 2000 ]\n  Shape "sphere" "float radius" [ 2 ]\nAttributeEnd\n')
QUADRICS = ('AttributeBegin\n  Translate 420 0 130\n  Rotate -90 1 0 0\n  Material "plastic" "rgb Kd" [ 0.2 0.3 0.7 ]\n  Shape "cylinder" "float radius" [ 50 ] "float zmin" [ 0 ] "float zmax" [ 160 ] "float phimax" [ 270 ]\nAttributeEnd\n'
            'AttributeBegin\n  Translate 130 300 330\n  Rotate 35 1 0.4 0\n  Scale 1 0.6 1.3\n  ReverseOrientation\n  Material "mirror"\n  Shape "cylinder" "float radius" [ 40 ] "float zmin" [ 60 ] "float zmax" [ -60 ]\nAttributeEnd\n'
            'AttributeBegin\n  Translate 300 120 100\n  Rotate 60 1 0 1\n  Material "glass"\n  Shape "disk" "float radius" [ 60 ] "float height" [ 15 ]\nAttributeEnd\n'
            'AttributeBegin\n  Translate 100 1 150\n  Rotate -90 1 0 0\n  Material "matte" "rgb Kd" [ 0.8 0.7 0.1 ]\n  Shape "disk" "float radius" [ 90 ] "float innerradius" [ 40 ] "float phimax" [ 200 ]\nAttributeEnd\n')
QUADRIC_LIGHTS = ('AttributeBegin\n  Translate 100 250 300\n  Rotate 70 0 1 0.3\n  AreaLightSource "diffuse" "rgb L" [ 6 2 2 ] "bool twosided" "true"\n  Shape "cylinder" "float radius" [ 15 ] "float zmin" [ -80 ] "float zmax" [ 80 ] "float phimax" [ 180 ]\nAttributeEnd\n'
                  'AttributeBegin\n  Translate 450 200 150\n  Rotate 110 1 0 0\n  ReverseOrientation\n  AreaLightSource "diffuse" "rgb L" [ 2 6 3 ]\n  Shape "disk" "float radius" [ 40 ]\nAttributeEnd\n'
                  'AttributeBegin\n  Translate 300 400 400\n  Scale 1 0.5 2\n  AreaLightSource "diffuse" "rgb L" [ 3 3 8 ]\n  Shape "cylinder" "float radius" [ 20 ] "float zmin" [ -30 ] "float zmax" [ 30 ]\nAttributeEnd\n')
MIX_MATERIALS = ('MakeNamedMaterial "m_matte" "string type" "matte" "rgb Kd" [ 0.7 0.2 0.2 ]\n'
                 'MakeNamedMaterial "m_metal" "string type" "metal" "float roughness" [ 0.05 ]\n'
                 'MakeNamedMaterial "m_plastic" "string type" "plastic" "rgb Kd" [ 0.1 0.4 0.1 ]\n'
                 'MakeNamedMaterial "m_mirror" "string type" "mirror"\n'
                 'MakeNamedMaterial "m_glass" "string type" "glass"\n'
                 'MakeNamedMaterial "mixA" "string type" "mix" "string namedmaterial1" "m_matte" "string namedmaterial2" "m_metal" "rgb amount" [ 0.3 0.5 0.7 ]\n'
                 'MakeNamedMaterial "mixNested" "string type" "mix" "string namedmaterial1" "mixA" "string namedmaterial2" "m_plastic"\n'
                 'MakeNamedMaterial "mixSpec" "string type" "mix" "string namedmaterial1" "m_mirror" "string namedmaterial2" "m_glass" "rgb amount" [ 0.4 0.4 0.4 ]\n')
SPHERE_ENCLOSING = ('AttributeBegin\n  Translate 278 273 100\n  ReverseOrientation\n  AreaLightSource "diffuse" "rgb L" [ 0.5 0.6 0.8 ]\n'
                    '  Shape "sphere" "float radius" [ 1500 ]\nAttributeEnd\n')


TEXTURES = ('Texture "chk" "spectrum" "checkerboard" "float uscale" [ 6 ] "float vscale" [ 4 ] "rgb tex1" [ 0.8 0.8 0.8 ] "rgb tex2" [ 0.1 0.2 0.5 ]\n'
            'Texture "chkpt" "spectrum" "checkerboard" "string aamode" "none" "float uscale" [ 3 ] "float vscale" [ 5 ] "float udelta" [ 0.3 ] "texture tex1" "chk" "rgb tex2" [ 0.7 0.1 0.1 ]\n'
            'TransformBegin\n  Scale 0.016 0.022 0.0125\n  Texture "chk3" "spectrum" "checkerboard" "integer dimension" [ 3 ] "rgb tex1" [ 0.9 0.9 0.2 ] "rgb tex2" [ 0.2 0.6 0.2 ]\nTransformEnd\n'
            'Texture "famt" "float" "checkerboard" "float uscale" [ 2 ] "float vscale" [ 2 ] "float tex1" [ 0.2 ] "float tex2" [ 0.9 ]\n'
            'Texture "mixed" "spectrum" "mix" "texture tex1" "chk" "rgb tex2" [ 0.9 0.4 0.1 ] "texture amount" "famt"\n'
            'Texture "scaled" "spectrum" "scale" "texture tex1" "chk3" "rgb tex2" [ 0.5 1 1 ]\n'
            'Texture "fsc" "float" "scale" "texture tex1" "famt" "float tex2" [ 0.5 ]\n'
            'Texture "fmix" "float" "mix" "float tex1" [ 0.02 ] "float tex2" [ 0.4 ] "texture amount" "famt"\n'
            'Texture "kconst" "spectrum" "constant" "rgb value" [ 0.3 0.3 0.3 ]\n'
            'Texture "bil" "spectrum" "bilerp" "rgb v00" [ 1 0 0 ] "rgb v01" [ 0 1 0 ] "rgb v10" [ 0 0 1 ] "rgb v11" [ 1 1 0 ]\n'
            'Texture "fbil" "float" "bilerp" "float v00" [ 0.1 ] "float v01" [ 0.9 ] "float v10" [ 0.5 ] "float v11" [ 0.3 ]\n'
            'Texture "uvt" "spectrum" "uv" "float uscale" [ 2.5 ] "float vscale" [ 1.5 ]\n')


def write_test_spds(outdir):
    """Small .spd files (wavelength in nm, value) in the layouts ReadFloatFile accepts: comments, exponents, a missing final
    newline (whose last number the reference drops), an odd value count."""
    open(os.path.join(outdir, "test_kd.spd"), "w").write("# reflectance\n380 0.05\n450 0.1\n500 0.6 # green\n550 0.7\n600 2.5e-1\n780 0.1\n")
    open(os.path.join(outdir, "test_eta.spd"), "w").write("400 1.2\n500 1.0\n600 0.4\n700 0.2\n800 0.25")
    open(os.path.join(outdir, "test_k.spd"), "w").write("# extinction\n400 2.1 500 2.6\n600 3.0 700 3.9\n12\n")


def write_test_images(outdir):
    """Small synthetic images in every container the front end reads: PNG (RGB8 / RGBA8 / palette / gray16), TGA (24-bit
    RLE bottom-up, 8-bit mono top-down), PFM (colour little-endian, mono big-endian); non-power-of-two sizes exercise the
    Lanczos resampling of the MIPMap constructor."""
    import struct, zlib
    import numpy as np
    rng = np.random.default_rng(5)
    def pattern(w, h, seed):
        y, x = np.mgrid[0:h, 0:w]
        r = 0.5 + 0.5 * np.sin(x * 0.7 + seed) * np.cos(y * 0.45)
        g = ((x // 3 + y // 2 + seed) % 2) * 0.8 + 0.1
        b = (x + 2 * y) / float(w + 2 * h)
        return np.stack([r, g, b], axis=2).astype(np.float32)
    def png(path, w, h, ctype, depth, rows, plte=None):
        def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
        raw = b""
        prev = None
        for i, row in enumerate(rows):  # cycle through the five filter types
            ft = i % 5
            bpp = max(1, {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype] * depth // 8)
            cur = np.frombuffer(row, np.uint8).astype(np.int32)
            up = np.zeros_like(cur) if prev is None else prev
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
            ul = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
            if ft == 0: f = cur
            elif ft == 1: f = cur - left
            elif ft == 2: f = cur - up
            elif ft == 3: f = cur - (left + up) // 2
            else:
                p = left + up - ul
                pa, pb, pc = np.abs(p - left), np.abs(p - up), np.abs(p - ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, up, ul))
                f = cur - pred
            raw += bytes([ft]) + (f & 255).astype(np.uint8).tobytes()
            prev = cur
        data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0))
        if plte is not None: data += chunk(b"PLTE", plte)
        data += chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")
        open(path, "wb").write(data)
    img = (pattern(37, 23, 1) * 255).astype(np.uint8)
    png(os.path.join(outdir, "img_rgb.png"), 37, 23, 2, 8, [img[y].tobytes() for y in range(23)])
    img = (pattern(16, 32, 2) * 255).astype(np.uint8)
    rgba = np.concatenate([img, np.full((32, 16, 1), 200, np.uint8)], axis=2)
    png(os.path.join(outdir, "img_rgba.png"), 16, 32, 6, 8, [rgba[y].tobytes() for y in range(32)])
    pal = rng.integers(0, 256, (16, 3)).astype(np.uint8)
    idx = ((np.mgrid[0:20, 0:24][0] * 3 + np.mgrid[0:20, 0:24][1]) % 16).astype(np.uint8)
    png(os.path.join(outdir, "img_pal.png"), 24, 20, 3, 8, [idx[y].tobytes() for y in range(20)], pal.tobytes())
    g16 = (pattern(19, 11, 3)[:, :, 0] * 65535).astype(">u2")
    png(os.path.join(outdir, "img_gray16.png"), 19, 11, 0, 16, [g16[y].tobytes() for y in range(11)])
    # TGA 24-bit RLE, bottom-up
    t = (pattern(21, 13, 4) * 255).astype(np.uint8)[::-1, :, ::-1]  # rows bottom-up, BGR
    body = b""
    for row in t:
        x = 0
        while x < 21:
            run = 1
            while x + run < 21 and run < 128 and (row[x + run] == row[x]).all(): run += 1
            if run > 1: body += bytes([0x80 | (run - 1)]) + row[x].tobytes(); x += run
            else:
                n = 1
                while x + n < 21 and n < 128 and not (x + n + 1 < 21 and (row[x + n] == row[x + n + 1]).all()): n += 1
                body += bytes([n - 1]) + row[x:x + n].tobytes(); x += n
    open(os.path.join(outdir, "img_rle.tga"), "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 10, 0, 0, 0, 0, 0, 21, 13, 24, 0) + body)
    m = (pattern(8, 8, 5)[:, :, 1] * 255).astype(np.uint8)
    open(os.path.join(outdir, "img_mono.tga"), "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 3, 0, 0, 0, 0, 0, 8, 8, 8, 0x20) + m.tobytes())
    # PFM: colour little-endian (scale -2 => values doubled), mono big-endian
    f = pattern(30, 17, 6)
    open(os.path.join(outdir, "img_color.pfm"), "wb").write(b"PF\n30 17\n-2.0\n" + f[::-1].astype("<f4").tobytes())
    f1 = pattern(9, 14, 7)[:, :, 2]
    open(os.path.join(outdir, "img_mono.pfm"), "wb").write(b"Pf\n9 14\n1.0\n" + f1[::-1].astype(">f4").tobytes())


IMAGE_TEXTURES = ('Texture "i_rgb" "spectrum" "imagemap" "string filename" "img_rgb.png" "float uscale" [ 3 ] "float vscale" [ 2 ]\n'
                  'Texture "i_rgba" "spectrum" "imagemap" "string filename" "img_rgba.png" "bool trilinear" "true" "string wrap" "clamp" "float uscale" [ 2.5 ] "float vscale" [ 2.5 ] "float udelta" [ -0.7 ]\n'
                  'Texture "i_pal" "spectrum" "imagemap" "string filename" "img_pal.png" "string wrap" "black" "float uscale" [ 1.7 ] "float vscale" [ 1.7 ] "float udelta" [ -0.3 ] "float maxanisotropy" [ 2 ]\n'
                  'Texture "i_g16" "float" "imagemap" "string filename" "img_gray16.png" "float scale" [ 0.4 ] "bool gamma" "false"\n'
                  'Texture "i_rle" "spectrum" "imagemap" "string filename" "img_rle.tga" "float scale" [ 0.8 ] "float uscale" [ 4 ] "float vscale" [ 4 ]\n'
                  'Texture "i_mono" "float" "imagemap" "string filename" "img_mono.tga" "bool trilinear" "true" "float uscale" [ 2 ]\n'
                  'Texture "i_pfm" "spectrum" "imagemap" "string filename" "img_color.pfm" "float scale" [ 0.3 ]\n'
                  'Texture "i_pfm1" "spectrum" "imagemap" "string filename" "img_mono.pfm" "string mapping" "spherical"\n'
                  'Texture "i_missing" "spectrum" "imagemap" "string filename" "does_not_exist.png"\n'
                  'Texture "i_scaled" "spectrum" "scale" "texture tex1" "i_rgb" "texture tex2" "i_pfm"\n')


def with_image_textures(s):
    s = with_normals(s, uv=True)
    s = s.replace("WorldBegin\n", "WorldBegin\n" + IMAGE_TEXTURES, 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "matte" "texture Kd" "i_rgb"', 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "matte" "texture Kd" "i_rgba"')
    s = s.replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "plastic" "texture Kd" "i_pal" "texture roughness" "i_g16"')
    s = s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "uber" "texture Kd" "i_rle" "texture opacity" "i_pfm1" "texture roughness" "i_mono"')
    s = s.replace("# tall box", 'Material "matte" "texture Kd" "i_scaled"\n# tall box')
    s = s.replace("# short box", 'AttributeBegin\n  Translate 420 70 120\n  Material "matte" "texture Kd" "i_missing"\n  Shape "sphere" "float radius" [ 60 ]\n'
                  '  Translate -270 0 180\n  Material "matte" "texture Kd" "i_pfm"\n  Shape "sphere" "float radius" [ 50 ]\nAttributeEnd\n# short box', 1)
    return s


def with_bump(s):
    s = with_normals(s, uv=True)
    tex = ('Texture "b_chk" "float" "checkerboard" "float uscale" [ 6 ] "float vscale" [ 6 ] "float tex1" [ 0 ] "float tex2" [ 4 ]\n'
           'Texture "b_img" "float" "imagemap" "string filename" "img_gray16.png" "float scale" [ 12 ] "float uscale" [ 3 ] "float vscale" [ 3 ]\n'
           'Texture "b_bil" "float" "bilerp" "float v00" [ 0 ] "float v01" [ 9 ] "float v10" [ 3 ] "float v11" [ -5 ]\n'
           'Texture "kd_chk" "spectrum" "checkerboard" "float uscale" [ 3 ] "float vscale" [ 3 ] "rgb tex1" [ 0.8 0.8 0.8 ] "rgb tex2" [ 0.3 0.3 0.6 ]\n'
           'MakeNamedMaterial "bm1" "string type" "plastic" "texture bumpmap" "b_chk" "rgb Kd" [ 0.6 0.2 0.2 ]\n'
           'MakeNamedMaterial "bm2" "string type" "matte" "texture bumpmap" "b_bil"\n'
           'MakeNamedMaterial "bmix" "string type" "mix" "string namedmaterial1" "bm1" "string namedmaterial2" "bm2"\n')
    s = s.replace("WorldBegin\n", "WorldBegin\n" + tex, 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "matte" "texture Kd" "kd_chk" "texture bumpmap" "b_img"', 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "plastic" "rgb Kd" [ 0.12 0.45 0.15 ] "texture bumpmap" "b_bil"')
    s = s.replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ] "float bumpmap" [ 3 ]')
    s = s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "uber" "texture bumpmap" "b_chk"')
    s = s.replace("# tall box", 'NamedMaterial "bmix"\n# tall box')
    s = s.replace("# short box", 'AttributeBegin\n  Translate 420 70 120\n  Rotate 30 1 0 1\n  Material "metal" "texture bumpmap" "b_chk" "float roughness" [ 0.1 ]\n  Shape "sphere" "float radius" [ 60 ]\n'
                  '  Translate -260 10 170\n  Scale 1 1.3 0.8\n  Material "substrate" "texture bumpmap" "b_img"\n  Shape "cylinder" "float radius" [ 40 ] "float zmin" [ -50 ] "float zmax" [ 50 ]\nAttributeEnd\n'
                  'ObjectBegin "bumpy"\n  Material "plastic" "texture bumpmap" "b_chk"\n  Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ -40 0 -40  40 0 -40  40 0 40  -40 0 40 ] "float uv" [ 0 0 1 0 1 1 0 1 ] "normal N" [ -0.3 1 -0.3  0.3 1 -0.3  0.3 1 0.3  -0.3 1 0.3 ]\nObjectEnd\n'
                  'AttributeBegin\n  Translate 278 200 100\n  Rotate 50 1 0 0\n  Scale 1.5 1 -1\n  ObjectInstance "bumpy"\nAttributeEnd\n# short box', 1)
    return s


FOG = 'MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [ 0.0004 0.0006 0.0008 ] "rgb sigma_s" [ 0.0015 0.0012 0.001 ] "float g" [ 0.4 ]\n'
SMOKE = ('MakeNamedMedium "smoke" "string type" "homogeneous" "rgb sigma_a" [ 0.2 0.5 1.0 ] "rgb sigma_s" [ 3 3 3 ] "float scale" [ 0.01 ] "float g" [ -0.3 ]\n'
         'MakeNamedMedium "thin" "string type" "homogeneous" "rgb sigma_a" [ 0.001 0.001 0.001 ] "rgb sigma_s" [ 0.004 0.002 0.001 ] "float g" [ 0.0005 ]\n')


def with_fog(s, camera_in_fog=True, world_interface='MediumInterface "fog" "fog"\n'):
    """A homogeneous medium filling the whole scene, the camera inside it (MediumInterface before Camera, api.cpp:785-790)."""
    s = s.replace("Camera ", FOG + ('MediumInterface "" "fog"\n' if camera_in_fog else "") + "Camera ", 1)
    return s.replace("WorldBegin\n", "WorldBegin\n" + world_interface, 1)


def with_smoke(s):
    """Media bounded by "none"-material surfaces: a sphere and a triangle-mesh box of smoke, a thin medium inside an
    instanced object, next to the glass/mirror boxes."""
    s = s.replace("Camera ", SMOKE + "Camera ", 1)
    s = s.replace("# short box", 'AttributeBegin\n  Translate 400 330 160\n  MediumInterface "smoke" ""\n  Material "none"\n  Shape "sphere" "float radius" [ 90 ]\nAttributeEnd\n'
                  'AttributeBegin\n  MediumInterface "thin" ""\n  Material ""\n  Shape "trianglemesh" "integer indices" [ 0 2 1 0 3 2  4 5 6 4 6 7  0 1 5 0 5 4  2 3 7 2 7 6  1 2 6 1 6 5  3 0 4 3 4 7 ]\n'
                  '    "point P" [ 20 20 20  300 20 20  300 500 20  20 500 20  20 20 300  300 20 300  300 500 300  20 500 300 ]\nAttributeEnd\n# short box', 1)
    return s


def with_mesh_shapes(s):
    """Shape "heightfield" (6 x 5 samples) and Shape "nurbs": a bicubic patch given by "P", and a rational quadratic x cubic
    surface given by "Pw" with a restricted parameter range, under transforms; plus ReverseOrientation on the patch."""
    import math
    hz = " ".join(f"{0.15 * math.sin(0.9 * i) * math.cos(0.7 * j) + 0.1:.6g}" for j in range(5) for i in range(6))
    cps = " ".join(f"{x} {y} {40 * math.sin(1.3 * x / 100 + y / 70.0):.6g}" for y in (0, 60, 120, 180) for x in (0, 70, 140, 210))
    pw = []
    for j in range(4):
        for (x, z, w) in ((60, 0, 1), (60, 60, 0.7071), (0, 60, 1)):
            pw += [x * w, 50.0 * j * w, z * w, w]
    pw = " ".join(f"{v:.6g}" for v in pw)
    shapes = ('AttributeBegin\n  Translate 60 166 60\n  Scale 220 200 160\n  Rotate -90 1 0 0\n  Material "plastic" "rgb Kd" [ 0.2 0.5 0.7 ]\n'
              '  Shape "heightfield" "integer nu" [ 6 ] "integer nv" [ 5 ] "float Pz" [ %s ]\nAttributeEnd\n'
              'AttributeBegin\n  Translate 300 340 150\n  Rotate 25 0 1 0\n  ReverseOrientation\n  Material "matte" "rgb Kd" [ 0.7 0.6 0.2 ]\n'
              '  Shape "nurbs" "integer nu" [ 4 ] "integer nv" [ 4 ] "integer uorder" [ 4 ] "integer vorder" [ 4 ]\n'
              '    "float uknots" [ 0 0 0 0 1 1 1 1 ] "float vknots" [ 0 0 0 0 2 2 2 2 ] "point P" [ %s ]\nAttributeEnd\n'
              'AttributeBegin\n  Translate 120 20 380\n  Material "mirror"\n'
              '  Shape "nurbs" "integer nu" [ 3 ] "integer nv" [ 4 ] "integer uorder" [ 3 ] "integer vorder" [ 4 ] "float u0" [ 0.1 ] "float v1" [ 0.9 ]\n'
              '    "float uknots" [ 0 0 0 1 1 1 ] "float vknots" [ 0 0 0 0 1 1 1 1 ] "float Pw" [ %s ]\nAttributeEnd\n' % (hz, cps, pw))
    return s.replace("# short box", shapes + "# short box", 1)


QUADRICS3 = ('AttributeBegin\n  Translate 420 0 150\n  Rotate -90 1 0 0\n  Material "plastic" "rgb Kd" [ 0.7 0.3 0.2 ]\n  Shape "cone" "float radius" [ 70 ] "float height" [ 190 ] "float phimax" [ 300 ]\nAttributeEnd\n'
             'AttributeBegin\n  Translate 150 330 330\n  Rotate 140 1 0.3 0\n  Scale 1 0.7 1.2\n  Material "glass"\n  Shape "paraboloid" "float radius" [ 80 ] "float zmin" [ 20 ] "float zmax" [ 150 ]\nAttributeEnd\n'
             'AttributeBegin\n  Translate 300 20 90\n  Rotate -90 1 0 0\n  ReverseOrientation\n  Material "mirror"\n'
             '  Shape "hyperboloid" "point p1" [ 60 10 0 ] "point p2" [ 30 -50 160 ] "float phimax" [ 270 ]\nAttributeEnd\n'
             'AttributeBegin\n  Translate 110 170 120\n  Material "matte" "rgb Kd" [ 0.2 0.6 0.3 ]\n  Shape "hyperboloid" "point p1" [ 40 0 -30 ] "point p2" [ 10 40 0 ]\n'
             '  Translate 0 0 60\n  Shape "cone" "float radius" [ 40 ] "float height" [ 80 ]\n  Translate 0 0 150\n  Rotate 180 1 0 0\n  Shape "paraboloid" "float radius" [ 50 ] "float zmax" [ 70 ]\nAttributeEnd\n'
             'ObjectBegin "q3"\n  Material "uber" "rgb Kd" [ 0.3 0.3 0.8 ]\n  Shape "cone" "float radius" [ 30 ] "float height" [ 60 ]\n  Shape "paraboloid" "float radius" [ 30 ] "float zmax" [ 40 ] "float phimax" [ 200 ]\nObjectEnd\n'
             'AttributeBegin\n  Translate 470 330 330\n  Rotate 60 0 1 1\n  Scale 1.5 1 -1\n  ObjectInstance "q3"\nAttributeEnd\n')

NOISE_TEXTURES = ('TransformBegin\n  Scale 0.02 0.02 0.02\n  Texture "n_fbm" "float" "fbm" "integer octaves" [ 5 ] "float roughness" [ 0.6 ]\n'
                  '  Texture "n_fbms" "spectrum" "fbm"\n  Texture "n_wr" "spectrum" "wrinkled" "integer octaves" [ 6 ] "float roughness" [ 0.4 ]\n'
                  '  Texture "n_wrf" "float" "wrinkled"\n  Texture "n_windy" "float" "windy"\nTransformEnd\n'
                  'TransformBegin\n  Rotate 30 0 1 0\n  Scale 0.03 0.05 0.03\n  Texture "n_marble" "spectrum" "marble" "float scale" [ 1.5 ] "float variation" [ 0.35 ] "integer octaves" [ 4 ]\nTransformEnd\n'
                  'Texture "n_dots" "spectrum" "dots" "float uscale" [ 5 ] "float vscale" [ 7 ] "rgb inside" [ 0.8 0.1 0.1 ] "texture outside" "n_marble"\n'
                  'Texture "n_dotsf" "float" "dots" "float uscale" [ 4 ] "float vscale" [ 4 ] "float inside" [ 0.02 ] "float outside" [ 0.5 ]\n'
                  'Texture "n_abs" "spectrum" "scale" "texture tex1" "n_wr" "rgb tex2" [ 0.9 0.7 0.5 ]\n'
                  'Texture "n_mix" "spectrum" "mix" "texture tex1" "n_marble" "rgb tex2" [ 0.1 0.4 0.1 ] "texture amount" "n_windy"\n')


def with_noise_textures(s):
    s = with_normals(s, uv=True)
    s = s.replace("WorldBegin\n", "WorldBegin\n" + NOISE_TEXTURES, 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "matte" "texture Kd" "n_marble"', 1)        # floor / ceiling / back wall
    s = s.replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "matte" "texture Kd" "n_abs" "texture sigma" "n_windy"')
    s = s.replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "matte" "texture Kd" "n_mix"')
    s = s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "plastic" "texture Kd" "n_dots" "texture roughness" "n_dotsf" "texture bumpmap" "n_fbm"')
    s = s.replace("# tall box", 'Material "uber" "texture Kd" "n_fbms" "texture opacity" "n_wr" "texture bumpmap" "n_wrf"\n# tall box')
    return s


IMAGE_LIGHTS = ('AttributeBegin\n  Translate 278 500 100\n  Rotate 80 1 0 0\n  Rotate 20 0 0 1\n  LightSource "projection" "rgb I" [ 400000 400000 400000 ] "float fov" [ 70 ] "string mapname" "img_rgb.png"\nAttributeEnd\n'
                'AttributeBegin\n  Translate 100 300 400\n  Rotate -100 1 0.2 0\n  LightSource "projection" "rgb I" [ 90000 60000 30000 ] "rgb scale" [ 1 2 3 ] "float fov" [ 40 ]\nAttributeEnd\n'
                'AttributeBegin\n  Translate 450 250 250\n  Rotate 30 0 1 1\n  LightSource "goniometric" "rgb I" [ 150000 150000 200000 ] "string mapname" "img_color.pfm"\nAttributeEnd\n'
                'AttributeBegin\n  Translate 278 100 450\n  LightSource "goniometric" "rgb I" [ 20000 30000 20000 ]\nAttributeEnd\n')

def with_alpha(s):
    s = with_normals(s, uv=True)
    tex = ('Texture "a_chk" "float" "checkerboard" "float uscale" [ 4 ] "float vscale" [ 4 ] "float tex1" [ 0 ] "float tex2" [ 1 ]\n'
           'Texture "a_chk2" "float" "checkerboard" "string aamode" "none" "float uscale" [ 2 ] "float vscale" [ 6 ] "float tex1" [ 1 ] "float tex2" [ 0 ]\n'
           'Texture "a_img" "float" "imagemap" "string filename" "img_pal.png" "bool gamma" "false" "float scale" [ 1 ]\n'
           'Texture "a_cut" "float" "scale" "texture tex1" "a_img" "texture tex2" "a_chk2"\n')
    s = s.replace("WorldBegin\n", "WorldBegin\n" + tex, 1)
    # the two boxes are the 24-vertex meshes; give them masks
    boxes = [m.start() for m in re.finditer(r'Shape "trianglemesh"\n  "integer indices" \[ 0 1 2 0 2 3  4 5 6', s)]
    assert len(boxes) == 2
    s = s[:boxes[1]] + 'Shape "trianglemesh" "texture alpha" "a_cut" "texture shadowalpha" "a_chk"\n  "integer indices" [ 0 1 2 0 2 3  4 5 6' + s[boxes[1] + len('Shape "trianglemesh"\n  "integer indices" [ 0 1 2 0 2 3  4 5 6'):]
    s = s[:boxes[0]] + 'Shape "trianglemesh" "texture alpha" "a_chk"\n  "integer indices" [ 0 1 2 0 2 3  4 5 6' + s[boxes[0] + len('Shape "trianglemesh"\n  "integer indices" [ 0 1 2 0 2 3  4 5 6'):]
    # a masked emitter, an invisible quad and a shadow-masked quad
    s = s.replace("# short box", 'AttributeBegin\n  AreaLightSource "diffuse" "rgb L" [ 6 5 3 ]\n  Shape "trianglemesh" "texture alpha" "a_chk2" "integer indices" [ 0 1 2 0 2 3 ] '
                  '"point P" [ 100 300 200  200 300 200  200 400 250  100 400 250 ] "float uv" [ 0 0 1 0 1 1 0 1 ]\nAttributeEnd\n'
                  'Shape "trianglemesh" "float alpha" [ 0 ] "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 0 100 100  556 100 100  556 400 100  0 400 100 ]\n'
                  'Shape "trianglemesh" "texture shadowalpha" "a_chk" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 350 250 150  500 250 150  500 250 350  350 250 350 ] "float uv" [ 0 0 1 0 1 1 0 1 ]\n# short box', 1)
    return s


def with_uv_boxes(s):
    return with_normals(s, uv=True).replace(' "normal N" [', ' "normal Nunused" [')


def with_textures(s):
    s = with_normals(s, uv=True)
    s = s.replace("WorldBegin\n", "WorldBegin\n" + TEXTURES, 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "matte" "texture Kd" "chk"', 1)        # floor / ceiling / back wall
    s = s.replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "matte" "texture Kd" "scaled"')
    s = s.replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "matte" "texture Kd" "mixed" "texture sigma" "fsc"')
    s = s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "matte" "texture Kd" "chkpt"')
    s = s.replace("# tall box", 'Material "matte" "texture Kd" "chk3"\n# tall box')
    return s


def with_textured_materials(s):
    s = with_normals(s, uv=True)
    s = s.replace("WorldBegin\n", "WorldBegin\n" + TEXTURES +
                  'MakeNamedMaterial "ta" "string type" "plastic" "texture Kd" "chk" "texture roughness" "fmix"\n'
                  'MakeNamedMaterial "tb" "string type" "metal" "texture k" "bil" "float roughness" [ 0.1 ]\n'
                  'MakeNamedMaterial "tmix" "string type" "mix" "string namedmaterial1" "ta" "string namedmaterial2" "tb" "texture amount" "uvt"\n', 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "uber" "texture Kd" "chk" "texture opacity" "bil" "texture Kr" "kconst" "rgb Kt" [ 0.2 0.2 0.2 ] "texture uroughness" "fbil"', 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "substrate" "texture Kd" "uvt" "texture Ks" "kconst" "texture vroughness" "fbil"')
    s = s.replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "translucent" "texture Kd" "chk" "texture transmit" "bil" "texture roughness" "fmix"')
    s = s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "glass" "texture Kt" "chkpt" "texture Kr" "kconst" "texture index" "fmixidx"')
    s = s.replace('Texture "kconst"', 'Texture "fmixidx" "float" "mix" "float tex1" [ 1.2 ] "float tex2" [ 1.7 ] "texture amount" "famt"\nTexture "kconst"')
    s = s.replace("# tall box", 'NamedMaterial "tmix"\n# tall box')
    return s


def with_mappings(s):
    s = with_normals(s, uv=True)
    tex = ('Texture "pl" "spectrum" "checkerboard" "string aamode" "none" "string mapping" "planar" "vector v1" [ 0.02 0 0.01 ] "vector v2" [ 0 0.03 0 ] "float udelta" [ 0.25 ] "rgb tex1" [ 0.9 0.9 0.9 ] "rgb tex2" [ 0.2 0.2 0.7 ]\n'
           'TransformBegin\n  Translate 278 273 280\n  Rotate 30 1 0 0\n'
           '  Texture "sph" "spectrum" "checkerboard" "string mapping" "spherical" "float uscale" [ 1 ] "rgb tex1" [ 0.9 0.5 0.1 ] "rgb tex2" [ 0.1 0.5 0.9 ]\n'
           '  Texture "cyl" "spectrum" "checkerboard" "string mapping" "cylindrical" "float uscale" [ 1 ] "rgb tex1" [ 0.8 0.8 0.2 ] "rgb tex2" [ 0.2 0.7 0.3 ]\n'
           '  Texture "sphuv" "spectrum" "uv" "string mapping" "spherical"\nTransformEnd\n'
           'Texture "sc" "spectrum" "scale" "texture tex1" "sph" "texture tex2" "cyl"\n')
    s = s.replace("WorldBegin\n", "WorldBegin\n" + tex, 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "matte" "texture Kd" "pl"', 1)
    s = s.replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "matte" "texture Kd" "sph"')
    s = s.replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "plastic" "texture Kd" "cyl" "texture Ks" "sphuv"')
    s = s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "matte" "texture Kd" "sc"')
    s = s.replace("# tall box", 'Material "mirror" "texture Kr" "sphuv"\n# tall box')
    s = s.replace("# short box", 'AttributeBegin\n  Translate 420 70 120\n  Rotate 40 0 1 1\n'
                  '  Texture "sphball" "spectrum" "checkerboard" "float uscale" [ 8 ] "float vscale" [ 4 ] "rgb tex1" [ 0.9 0.9 0.9 ] "rgb tex2" [ 0.7 0.1 0.1 ]\n  Material "matte" "texture Kd" "sphball"\n'
                  '  Shape "sphere" "float radius" [ 70 ]\n  Translate -250 0 150\n  Shape "cylinder" "float radius" [ 30 ] "float zmin" [ -60 ] "float zmax" [ 60 ]\n'
                  '  Translate 0 0 61\n  Shape "disk" "float radius" [ 30 ]\nAttributeEnd\n# short box', 1)
    return s


def with_instances(s):
    """Move the two boxes into `ObjectBegin "boxes"`, add a one-sphere object, and instance both several times."""
    i = s.index("# short box")
    j = s.index("WorldEnd")
    boxes = s[i:j].replace("# tall box", 'Material "glass" "float index" [ 1.4 ]\n# tall box')
    inst = ('ObjectBegin "boxes"\n' + boxes + 'ObjectEnd\n'
            'ObjectBegin "ball"\n  Material "mirror"\n  Translate 0 40 0\n  Shape "sphere" "float radius" [ 40 ]\nObjectEnd\n'
            'ObjectBegin "tri"\n  Material "matte" "rgb Kd" [ 0.8 0.6 0.1 ]\n  Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ 0 0 0  90 0 0  0 120 30 ]\nObjectEnd\n'
            'ObjectInstance "boxes"\n'
            'AttributeBegin\n  Translate 50 230 -40\n  Rotate 25 0 1 0.2\n  Scale 0.45 0.45 0.45\n  ObjectInstance "boxes"\nAttributeEnd\n'
            'AttributeBegin\n  Translate 520 330 120\n  Scale -0.4 0.3 0.5\n  ObjectInstance "boxes"\nAttributeEnd\n'
            'AttributeBegin\n  Translate 120 0 420\n  ObjectInstance "ball"\nAttributeEnd\n'
            'AttributeBegin\n  Translate 420 300 80\n  Scale 1.5 0.6 1\n  ObjectInstance "ball"\nAttributeEnd\n'
            'AttributeBegin\n  Translate 40 20 100\n  Rotate -30 0 1 0\n  ObjectInstance "tri"\nAttributeEnd\n'
            'AttributeBegin\n  Translate 330 380 300\n  Scale 1 -1 1\n  ObjectInstance "tri"\nAttributeEnd\n')
    return s[:i] + inst + s[j:]


def with_normals(s, tangents=False, uv=False):
    """Give both Cornell boxes smooth-ish per-vertex normals (outward from the box centre, one of them zero), optionally
    tangents and a uv parameterisation."""
    import re
    def edit(m):
        pts = [float(x) for x in m.group(2).split()]
        n = len(pts) // 3
        c = [sum(pts[k::3]) / n for k in range(3)]
        N, S, UV = [], [], []
        for i in range(n):
            d = [pts[3 * i + k] - c[k] for k in range(3)]
            N += ([0, 0, 0] if i == 5 else [d[0], d[1] * 0.5, d[2]])
            S += [d[2] + 1, 0.25 * d[1], -d[0]]
            UV += [0.01 * pts[3 * i] + 0.002 * pts[3 * i + 1], 0.01 * pts[3 * i + 2]]
        extra = ' "normal N" [ ' + " ".join(f"{x:.6g}" for x in N) + " ]"
        if tangents: extra += ' "vector S" [ ' + " ".join(f"{x:.6g}" for x in S) + " ]"
        if uv: extra += ' "float uv" [ ' + " ".join(f"{x:.6g}" for x in UV) + " ]"
        return m.group(1) + m.group(2) + " ]" + extra
    # the two 24-vertex box meshes
    return re.sub(r'("point P" \[ )((?:[-\d.]+\s+){71}[-\d.]+) \]', edit, s)


TALL = [423, 330, 247, 265, 330, 296, 314, 330, 456, 472, 330, 406, 423, 0, 247, 423, 330, 247, 472, 330, 406, 472, 0, 406,
        472, 0, 406, 472, 330, 406, 314, 330, 456, 314, 0, 456, 314, 0, 456, 314, 330, 456, 265, 330, 296, 265, 0, 296,
        265, 0, 296, 265, 330, 296, 423, 330, 247, 423, 0, 247, 423, 0, 247, 472, 0, 406, 314, 0, 456, 265, 0, 296]
SHORT = [130, 165, 65, 82, 165, 225, 240, 165, 272, 290, 165, 114, 290, 0, 114, 290, 165, 114, 240, 165, 272, 240, 0, 272,
         130, 0, 65, 130, 165, 65, 290, 165, 114, 290, 0, 114, 82, 0, 225, 82, 165, 225, 130, 165, 65, 130, 0, 65,
         240, 0, 272, 240, 165, 272, 82, 165, 225, 82, 0, 225, 130, 0, 65, 290, 0, 114, 240, 0, 272, 82, 0, 225]


def write_plys():
    """The PLY fixtures of cornell_ply (committed next to the scene)."""
    import struct
    # tall box: ASCII, 6 quads, a comment, an extra per-vertex property and an extra element that must be skipped
    with open(os.path.join(GOLD, "tall_quads.ply"), "w") as f:
        f.write("ply\nformat ascii 1.0\ncomment tall Cornell box as quads\nelement vertex 24\nproperty float x\nproperty float y\n"
                "property float z\nproperty uchar red\nelement face 7\nproperty list uchar int vertex_indices\n"
                "element edge 1\nproperty int vertex1\nproperty int vertex2\nend_header\n")
        for i in range(24):
            f.write(f"{TALL[3 * i]} {TALL[3 * i + 1]} {TALL[3 * i + 2]} {i}\n")
        for q in range(6):
            f.write(f"4 {4 * q} {4 * q + 1} {4 * q + 2} {4 * q + 3}\n")
        f.write("5 0 1 2 3 4\n0 1\n")  # a pentagon (ignored with a warning), then the edge element
    # short box: binary, normals + uv; one file per endianness, split in two meshes of 3 quads (as 6 triangles) each
    c = [sum(SHORT[k::3]) / 24 for k in range(3)]
    for name, end, half in (("short_le.ply", "<", 0), ("short_be.ply", ">", 1)):
        with open(os.path.join(GOLD, name), "wb") as f:
            fmt = "binary_little_endian" if end == "<" else "binary_big_endian"
            f.write((f"ply\nformat {fmt} 1.0\nelement vertex 24\nproperty double x\nproperty float y\nproperty float z\n"
                     "property float nx\nproperty float ny\nproperty float nz\nproperty float s\nproperty float t\n"
                     "element face 6\nproperty list uchar ushort vertex_indices\nend_header\n").encode())
            for i in range(24):
                x, y, z = SHORT[3 * i:3 * i + 3]
                f.write(struct.pack(end + "dfffffff", x, y, z, x - c[0], 0.5 * (y - c[1]), z - c[2], 0.01 * x, 0.01 * z))
            for q in range(3 * half, 3 * half + 3):
                for tri in ((0, 1, 2), (0, 2, 3)):
                    f.write(struct.pack(end + "BHHH", 3, *(4 * q + k for k in tri)))


def with_subdiv(s):
    import re
    boxes = list(re.finditer(r'Shape "trianglemesh"\s+"integer indices" \[ 0 1 2 0 2 3  4 5 6[^\]]*\]\s+"point P" \[[^\]]*\]', s))
    assert len(boxes) == 2
    octa = ('AttributeBegin\n Translate 370 200 350\n Scale 120 190 120\n Shape "loopsubdiv" "integer levels" [ 3 ] '
            '"integer indices" [ 0 2 4  2 1 4  1 3 4  3 0 4  2 0 5  1 2 5  3 1 5  0 3 5 ] '
            '"point P" [ 1 0 0  -1 0 0  0 0 1  0 0 -1  0 1 0  0 -1 0 ]\nAttributeEnd\n')
    fan = ('AttributeBegin\n Translate 180 90 170\n Rotate 25 0 1 0\n Scale 95 60 95\n Shape "loopsubdiv" "integer nlevels" [ 2 ] '
           '"integer indices" [ 0 1 2  0 2 3  0 3 4  0 4 5  0 5 6  0 6 1  1 7 2  2 7 8  3 2 8  9 5 4  6 5 10 ] '
           '"point P" [ 0 1 0  1 0 0  0.5 0.2 0.87  -0.5 0 0.87  -1 0.3 0  -0.5 0 -0.87  0.5 0.1 -0.87  1.4 -0.5 0.9  0.2 -0.4 1.7  -1.6 -0.2 -0.7  0.4 -0.3 -1.8 ]\nAttributeEnd\n'
           'AttributeBegin\n Translate 160 40 90\n Scale 40 40 40\n ReverseOrientation\n Shape "loopsubdiv" "integer levels" [ 1 ] '
           '"integer indices" [ 0 1 2  0 3 1  0 2 3  1 3 2 ] "point P" [ 1 1 1  -1 -1 1  -1 1 -1  1 -1 -1 ]\nAttributeEnd\n')
    s = s[:boxes[1].start()] + octa + s[boxes[1].end():]
    s = s[:boxes[0].start()] + fan + s[boxes[0].end():]
    return s


def with_ply(s):
    write_plys()
    import re
    boxes = list(re.finditer(r'Shape "trianglemesh"\s+"integer indices" \[ 0 1 2 0 2 3  4 5 6[^\]]*\]\s+"point P" \[[^\]]*\]', s))
    assert len(boxes) == 2
    s = s[:boxes[1].start()] + 'Shape "plymesh" "string filename" "tall_quads.ply"' + s[boxes[1].end():]
    s = s[:boxes[0].start()] + 'Shape "plymesh" "string filename" "short_le.ply"\nShape "plymesh" "string filename" "short_be.ply"' + s[boxes[0].end():]
    return s


def with_many_lights(s, n=50):
    """The ceiling light as an n x n grid of cells = 2 n^2 emissive triangles (api.cpp:1353-1363: one DiffuseAreaLight each): under
    the default "spatial" strategy every touched voxel holds a distribution over all of them (lightdistrib.cpp:232-300)."""
    xs = [213 + (343 - 213) * i / n for i in range(n + 1)]
    zs = [227 + (332 - 227) * j / n for j in range(n + 1)]
    P = " ".join(f"{x:.6g} 548.7 {z:.6g}" for z in zs for x in xs)
    idx = []
    for j in range(n):
        for i in range(n):
            a, b, c, d = j * (n + 1) + i, j * (n + 1) + i + 1, (j + 1) * (n + 1) + i + 1, (j + 1) * (n + 1) + i
            idx += [a, c, d, a, b, c]  # normals point down (-y), like the original quad
    mesh = 'Shape "trianglemesh" "integer indices" [ ' + " ".join(map(str, idx)) + ' ]\n    "point P" [ ' + P + " ]"
    old = 'Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ]\n    "point P" [ 343 548.7 227   343 548.7 332   213 548.7 332   213 548.7 227 ]'
    assert old in s
    return s.replace(old, mesh)


def moving(directives, end_motion, start_motion=""):
    """`directives` (shapes, instances, materials) under an animated transformation: `start_motion` acts on the START transform only, `end_motion`
    on the END transform only (ActiveTransform StartTime / EndTime, api.cpp:395-403), so the CTM's two ends differ and every Shape becomes a
    TransformedPrimitive over an AnimatedTransform (api.cpp:1386-1419), every ObjectInstance too (:1576-1586)."""
    pre = ("ActiveTransform StartTime\n" + start_motion + "\n") if start_motion else ""
    return "AttributeBegin\n" + pre + "ActiveTransform EndTime\n" + end_motion + "\nActiveTransform All\n" + directives + "AttributeEnd\n"


def with_moving_boxes(s, short_motion="Translate 70 0 -50", tall_motion="Translate -40 60 0\nScale 1 0.8 1", times=""):
    """The two Cornell boxes move during the exposure: each mesh becomes a BVHAccel of its own under one TransformedPrimitive."""
    i = s.index("# short box")
    j = s.index("# tall box")
    k = s.index("WorldEnd")
    out = s[:i] + moving(s[i:j], short_motion) + moving('Material "plastic" "rgb Kd" [ 0.2 0.5 0.3 ] "float roughness" [ 0.2 ]\n' + s[j:k], tall_motion) + s[k:]
    return out.replace("WorldBegin", times + "WorldBegin", 1) if times else out


def with_moving_instances(s, spin=("", "", "", "")):
    """with_instances' objects, some uses of them under a motion: a BVHAccel object, a lone sphere (a quadric inside a moving instance), a lone
    triangle, and a moving SHAPE (a sphere, created at the identity) beside them.  `spin`: directives added to the four END transforms."""
    s = with_instances(s)
    s = s.replace('AttributeBegin\n  Translate 50 230 -40\n  Rotate 25 0 1 0.2\n  Scale 0.45 0.45 0.45\n  ObjectInstance "boxes"\nAttributeEnd\n',
                  'AttributeBegin\n  Translate 50 230 -40\n  Rotate 25 0 1 0.2\n  Scale 0.45 0.45 0.45\n  ActiveTransform EndTime\n  Translate 120 -80 60\n' + spin[0] + '  ActiveTransform All\n  ObjectInstance "boxes"\nAttributeEnd\n')
    s = s.replace('AttributeBegin\n  Translate 120 0 420\n  ObjectInstance "ball"\nAttributeEnd\n',
                  'AttributeBegin\n  Translate 120 0 420\n  ActiveTransform EndTime\n  Translate 60 30 -90\n  Scale 1.3 0.8 1\n' + spin[1] + '  ActiveTransform All\n  ObjectInstance "ball"\nAttributeEnd\n')
    s = s.replace('AttributeBegin\n  Translate 330 380 300\n  Scale 1 -1 1\n  ObjectInstance "tri"\nAttributeEnd\n',
                  'AttributeBegin\n  Translate 330 380 300\n  Scale 1 -1 1\n  ObjectInstance "tri"\nAttributeEnd\n'
                  'AttributeBegin\n  Translate 300 300 250\n  Scale 1 1.2 1\n  ActiveTransform StartTime\n  Translate -50 0 0\n  ActiveTransform EndTime\n  Translate 40 -30 20\n' + spin[2] + '  ActiveTransform All\n  ObjectInstance "tri"\nAttributeEnd\n'
                  + moving('  Translate 300 90 60\n  Material "glass" "float index" [ 1.5 ]\n  Shape "sphere" "float radius" [ 45 ]\n', "Translate 0 120 0\n" + spin[3]))
    assert s.count("ActiveTransform EndTime") == 4
    return s


def with_nested_motion(s, tall="  Translate 30 20 -40\n  Rotate 20 0 1 0\n", ball="  Translate 70 10 0\n", tri="  Translate 10 30 0\n"):
    """Moving SHAPES inside the object definitions of an instanced scene (with_instances / with_moving_instances): pbrtShape under an animated
    transformation between ObjectBegin and ObjectEnd adds its TransformedPrimitive to the definition (api.cpp:1386-1419), so every ObjectInstance
    of it is a TransformedPrimitive around a TransformedPrimitive.  "boxes": the tall box (a BVHAccel of its own) moves beside the still short box;
    "ball": a second, moving sphere beside the still one; "tri": the definition's ONLY primitive moves (no accelerator at either level)."""
    n0 = s.count("ActiveTransform EndTime")
    s = s.replace('# tall box\nShape', '# tall box: moves inside the object definition\nActiveTransform EndTime\n' + tall + 'ActiveTransform All\nShape', 1)
    s = s.replace('  Shape "sphere" "float radius" [ 40 ]\nObjectEnd', '  Shape "sphere" "float radius" [ 40 ]\n  ActiveTransform StartTime\n' + ball +
                  '  ActiveTransform All\n  Material "matte" "rgb Kd" [ 0.2 0.3 0.8 ]\n  Shape "sphere" "float radius" [ 25 ]\nObjectEnd', 1)
    s = s.replace('  Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ 0 0 0  90 0 0  0 120 30 ]\nObjectEnd',
                  '  ActiveTransform EndTime\n' + tri + '  ActiveTransform All\n  Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ 0 0 0  90 0 0  0 120 30 ]\nObjectEnd', 1)
    assert s.count("ActiveTransform EndTime") == n0 + 2 and s.count("ActiveTransform StartTime") >= 1
    return s



def cam_anim(s, end_motion, times=""):
    """Give the camera an end-of-motion transform: the directives `end_motion` act on the END transform only (ActiveTransform
    EndTime, api.cpp:395-403) on top of the LookAt both share, so CameraToWorld[0] != CameraToWorld[1]."""
    assert "Camera " in s
    return s.replace("Camera ", times + "ActiveTransform EndTime\n" + end_motion + "\nActiveTransform All\nCamera ", 1)


def with_sampler(s, spec):
    import re as _re
    out, n = _re.subn(r'Sampler "halton" "integer pixelsamples" \[ \d+ \]', 'Sampler ' + spec, s)
    assert n == 1
    return out


SCENES = {
    # the samplers that draw from one RNG stream per tile (integrator.cpp:247-248): RandomSampler, and the PixelSamplers with their
    # per-pixel sample arrays and the RNG fallback beyond "dimensions" (sampler.cpp:100-134); clipped edge tiles, crop windows,
    # pixel bounds (StartPixel runs for skipped pixels too), a lens, wide filters, rounded sample counts, volpath
    "sampler_random": with_sampler(cornell(24, 24, 4), '"random" "integer pixelsamples" [ 5 ]'),
    "sampler_stratified": with_sampler(cornell(40, 24, 4), '"stratified" "integer xsamples" [ 3 ] "integer ysamples" [ 2 ]'),
    "sampler_stratified_dims": with_sampler(cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 3 ] "integer pixelbounds" [ 5 20 3 17 ]'),
                                            '"stratified" "integer xsamples" [ 2 ] "integer ysamples" [ 2 ] "bool jitter" "false" "integer dimensions" [ 14 ]'),
    "filter_02sequence_lens": with_sampler(cornell(36, 20, 4, extra_film='"float cropwindow" [ 0.1 0.9 0.2 1 ]'), '"02sequence" "integer pixelsamples" [ 6 ]')
                          .replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "perspective" "float fov" [ 39.3 ] "float lensradius" [ 8 ] "float focaldistance" [ 900 ]')
                          .replace('PixelFilter "box"', 'PixelFilter "gaussian"'),
    "sampler_maxmindist": with_sampler(cornell(24, 24, 4), '"maxmindist" "integer pixelsamples" [ 8 ] "integer dimensions" [ 2 ]'),
    # "dimensions" covering every draw a path of this depth can make (1 + 2 maxdepth one-dimensional, 2 + 3 maxdepth two-dimensional): no path
    # touches the tile's stream after StartPixel -- the device generates a tile's arrays ahead and traces all its pixels as one wavefront
    "filter_02sequence_dims": with_sampler(cornell(36, 20, 4, extra_film='"float cropwindow" [ 0.1 0.9 0.2 1 ]', integrator='Integrator "path" "integer maxdepth" [ 3 ]'),
                                           '"02sequence" "integer pixelsamples" [ 4 ] "integer dimensions" [ 11 ]')
                          .replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "perspective" "float fov" [ 39.3 ] "float lensradius" [ 8 ] "float focaldistance" [ 900 ]')
                          .replace('PixelFilter "box"', 'PixelFilter "gaussian"'),
    "sampler_maxmindist_dims": with_sampler(cornell(24, 24, 4), '"maxmindist" "integer pixelsamples" [ 8 ] "integer dimensions" [ 17 ]'),  # maxdepth 5: the roulette draws too
    "sampler_stratified_dims_tex": with_sampler(cornell(32, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 4 ]', world_edit=lambda s: with_image_textures(s)),
                                                '"stratified" "integer xsamples" [ 2 ] "integer ysamples" [ 2 ] "integer dimensions" [ 14 ]')
                          .replace('"float fov" [ 39.3 ]', '"float fov" [ 39.3 ] "float lensradius" [ 10 ] "float focaldistance" [ 700 ]'),
    "sampler_lowdisc_vol": with_sampler(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 4 ]', world_edit=lambda s: with_smoke(s)),
                                        '"lowdiscrepancy" "integer pixelsamples" [ 4 ] "integer dimensions" [ 3 ]'),
    # 5 000 area lights under the default "spatial" strategy: the device fills its voxel tables on first touch (sparse), as the
    # reference's hash table does
    "many_lights": cornell(20, 20, 2, integrator='Integrator "path" "integer maxdepth" [ 2 ]', world_edit=lambda s: with_many_lights(s, 50)),
    # plain Cornell, tile-aligned and not
    "cornell_32": cornell(32, 32, 8),
    "cornell_40x24": cornell(40, 24, 4),
    # crop window + non-default light strategies + depth variants (RR kicks in after 4 bounces)
    "cornell_crop": cornell(48, 48, 4, extra_film='"float cropwindow" [ 0.25 0.8 0.1 0.6 ]'),
    "cornell_uniform": cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 3 ] "string lightsamplestrategy" "uniform"'),
    "cornell_power": cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 8 ] "string lightsamplestrategy" "power"'),
    "cornell_depth1": cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 1 ]'),
    "cornell_rr": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 12 ] "float rrthreshold" [ 0.5 ]'),
    # two-sided light, reversed orientation, transforms, black material, scaled light, maxsampleluminance, film scale
    "cornell_twosided": cornell(24, 24, 4, world_edit=lambda s: s.replace('"rgb L" [ 17 12 4 ]', '"rgb L" [ 17 12 4 ] "bool twosided" "true" "rgb scale" [ 0.5 0.5 2 ]')),
    "cornell_reverse": cornell(24, 24, 4, world_edit=lambda s: s.replace("# short box", "ReverseOrientation\n# short box")),
    "cornell_xform": cornell(24, 24, 4, world_edit=lambda s: s.replace("# tall box", "Translate 30 0 -20\nRotate 15 0 1 0\nScale 1 0.8 -1\n# tall box")),
    "cornell_black": cornell(24, 24, 4, world_edit=lambda s: s.replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "matte" "rgb Kd" [ 0 0 0 ]')),
    "cornell_filmopts": cornell(24, 24, 8, extra_film='"float scale" [ 1.5 ] "float maxsampleluminance" [ 2.0 ]'),
    "cornell_center": cornell(20, 20, 4).replace('Sampler "halton"', 'Sampler "halton" "bool samplepixelcenter" "true"'),
    # plastic = Lambertian + TrowbridgeReitz microfacet lobe (plastic.cpp:45-70): defaults; specular-only without roughness
    # remapping; and a camera looking straight down at a plastic floor (normal-incidence branch of TrowbridgeReitzSample11)
    "cornell_plastic": cornell(32, 32, 8, world_edit=lambda s: s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "plastic"')
                               .replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "plastic" "rgb Kd" [ 0.65 0.05 0.05 ] "rgb Ks" [ 0.3 0.3 0.3 ] "float roughness" [ 0.05 ]')),
    "cornell_plastic_spec": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 7 ]',
                                    world_edit=lambda s: s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]',
                                        '# short box\nMaterial "plastic" "rgb Kd" [ 0 0 0 ] "rgb Ks" [ 0.6 0.7 0.8 ] "float roughness" [ 0.3 ] "bool remaproughness" "false"')),
    "plastic_topdown": cornell(24, 24, 8, world_edit=lambda s: s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "plastic" "rgb Kd" [ 0.2 0.3 0.4 ] "rgb Ks" [ 0.5 0.5 0.5 ] "float roughness" [ 0.02 ]', 1))
        .replace("LookAt 278 273 -800  278 273 0  0 1 0", "LookAt 278 500 279  278 0 279  0 0 1"),
    # per-vertex shading normals N, tangents S and uv on the boxes (triangle.cpp:350-419), incl. reversed orientation,
    # an emitter with normals (Triangle::Sample's Faceforward, triangle.cpp:593-597) and a zero normal
    "cornell_normals": cornell(32, 32, 8, world_edit=lambda s: with_normals(s)),
    "cornell_tangents": cornell(24, 24, 8, world_edit=lambda s: with_normals(s, tangents=True, uv=True).replace("# tall box", "ReverseOrientation\n# tall box")),
    "cornell_lightnormals": cornell(24, 24, 8, world_edit=lambda s: s.replace(
        '"point P" [ 343 548.7 227   343 548.7 332   213 548.7 332   213 548.7 227 ]',
        '"point P" [ 343 548.7 227   343 548.7 332   213 548.7 332   213 548.7 227 ] "normal N" [ 0.2 -1 0  0 -1 0.3  0 1 0  -0.2 -1 -0.1 ]')),
    # Shape "plymesh" (plymesh.cpp): the tall box as an ASCII PLY of quads, the short box as binary PLYs (little and big
    # endian, doubles and uchar/ushort index types) with normals and texture coordinates
    "cornell_ply": cornell(32, 32, 8, world_edit=lambda s: with_ply(s)),
    # delta lights (point.cpp, spot.cpp, distant.cpp) alone and mixed with the area light under every light-sampling strategy
    "cornell_point": cornell(24, 24, 8, world_edit=lambda s: s.replace("# light\nAttributeBegin", DELTA_POINT + "# light\nAttributeBegin")),
    "cornell_spot_power": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 4 ] "string lightsamplestrategy" "power"',
                                  world_edit=lambda s: s.replace("# light\nAttributeBegin", DELTA_SPOT + DELTA_DISTANT + "# light\nAttributeBegin")),
    "cornell_delta_only": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 3 ] "string lightsamplestrategy" "uniform"',
                                  world_edit=lambda s: s.replace("  AreaLightSource", "#  AreaLightSource").replace("# light\nAttributeBegin", DELTA_SPOT + DELTA_POINT + DELTA_DISTANT + "# light\nAttributeBegin")),
    # specular BSDFs: mirror (SpecularReflection) on the tall box, glass (FresnelSpecular: reflection + refraction, etaScale in
    # the Russian roulette, emission seen through specular bounces) on the short box; deep paths
    "cornell_mirror_glass": cornell(32, 32, 16, integrator='Integrator "path" "integer maxdepth" [ 9 ] "float rrthreshold" [ 0.8 ]',
                                    world_edit=lambda s: s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "glass" "float index" [ 1.45 ] "rgb Kt" [ 0.9 1 0.9 ]')
                                    .replace("# tall box", 'Material "mirror" "rgb Kr" [ 0.8 0.8 0.9 ]\n# tall box')),
    "cornell_glass_eta": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]',
                                 world_edit=lambda s: s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "glass" "float eta" [ 2.2 ] "rgb Kr" [ 0 0 0 ]')
                                 .replace("# tall box", 'Material "mirror" "rgb Kr" [ 0 0 0 ]\n# tall box')),
    # pixel filters other than the default box (filters/*.cpp, FilmTile::AddSample's table path, overlapping tile merges)
    "filter_gaussian": cornell(40, 24, 4).replace('PixelFilter "box"', 'PixelFilter "gaussian"'),
    "filter_mitchell_crop": cornell(48, 48, 4, extra_film='"float cropwindow" [ 0.2 0.75 0.3 0.9 ]').replace('PixelFilter "box"', 'PixelFilter "mitchell" "float xwidth" [ 1.5 ] "float ywidth" [ 2.5 ] "float B" [ 0.2 ]'),
    "filter_sinc": cornell(24, 24, 4).replace('PixelFilter "box"', 'PixelFilter "sinc" "float xwidth" [ 3 ] "float ywidth" [ 3 ]'),
    "filter_triangle_box": cornell(24, 24, 4).replace('PixelFilter "box"', 'PixelFilter "triangle" "float xwidth" [ 0.5 ] "float ywidth" [ 1 ]'),
    "filter_widebox": cornell(24, 24, 4).replace('PixelFilter "box"', 'PixelFilter "box" "float xwidth" [ 1.25 ] "float ywidth" [ 0.75 ]'),
    # A box-filter frame in which film positions round UP onto the next pixel: 1920 pixels wide, pixels from x = 1024 on (float
    # spacing 2^-13), Halton sample indices beyond 2 097 024 (sample 607 of a pixel under a stride of 128 x 27): `1050 + u0` with
    # u0 = 1 - 2^-14 is 1051.0, and FilmTile::AddSample adds the sample to pixel 1051 BEFORE that pixel's own samples
    # (film.h:121-161).  Such frames take the gathering film path (pg_box_filter_needs_gather, include/pbrt_gpu.h); found at the
    # full size of BASELINE config 4 (1920x1080 @ 256 spp: 76 of its 36 864 window pixels)
    "filter_box_round_up": (
        'LookAt 0 -17 9  0 -1.5 0.3  0 0 1\nCamera "perspective" "float fov" [ 42 ]\n'
        'Film "image" "integer xresolution" [ 1920 ] "integer yresolution" [ 1080 ] "float cropwindow" [ 0.53307292 0.59973958 0.43287037 0.44768519 ] '
        '"string filename" "filter_box_round_up.pfm"\nSampler "halton" "integer pixelsamples" [ 640 ]\nPixelFilter "box"\n'
        'Integrator "path" "integer maxdepth" [ 1 ]\nWorldBegin\nLightSource "infinite" "rgb L" [ 1 1 1 ]\nMaterial "matte" "rgb Kd" [ 0.5 0.5 0.5 ]\n'
        'Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ -50 -50 0  50 -50 0  50 50 0  -50 50 0 ]\nWorldEnd\n'),
    # Shape "loopsubdiv" (loopsubdiv.cpp): a closed octahedron (valence-4 extraordinary vertices), an open fan with boundary
    # vertices of valence 2, 3, 4 and 6, and a tetrahedron (valence 3), at several levels, replacing the Cornell boxes
    "cornell_loopsubdiv": cornell(40, 40, 8, world_edit=lambda s: with_subdiv(s)),
    # matte with sigma != 0: the OrenNayar BRDF (reflection.cpp:197-219), incl. sigma clamped to 90
    "cornell_orennayar": cornell(24, 24, 8, world_edit=lambda s: s.replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ] "float sigma" [ 35 ]')
                                 .replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ] "float sigma" [ 120 ]')),
    # OrthographicCamera (orthographic.cpp), plain and with a thin lens and a screen window
    "cornell_ortho": cornell(24, 24, 8).replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "orthographic" "float screenwindow" [ -300 300 -290 310 ]'),
    "cornell_ortho_lens": cornell(20, 28, 8).replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "orthographic" "float screenwindow" [ -280 280 -280 280 ] "float lensradius" [ 12 ] "float focaldistance" [ 1100 ]'),
    # InfiniteAreaLight with constant radiance (infinite.cpp): alone (camera rays that escape see it, MIS against BSDF samples
    # that escape), rotated, and mixed with area + spot lights under the power strategy with mirror/glass (specular escapes)
    "env_only": cornell(32, 32, 8, world_edit=lambda s: s.replace("  AreaLightSource", "#  AreaLightSource")
                        .replace("# light\nAttributeBegin", 'AttributeBegin\n  Rotate 30 1 0.3 0\n  LightSource "infinite" "rgb L" [ 0.6 0.8 1.2 ] "rgb scale" [ 1.5 1.5 1.5 ]\nAttributeEnd\n# light\nAttributeBegin')),
    "env_mixed_power": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ] "string lightsamplestrategy" "power"',
                               world_edit=lambda s: s.replace("# light\nAttributeBegin", 'LightSource "infinite" "rgb L" [ 0.3 0.3 0.35 ]\n' + DELTA_SPOT + "# light\nAttributeBegin")
                               .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "glass"').replace("# tall box", 'Material "mirror"\n# tall box')),
    "env_uniform_open": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 4 ] "string lightsamplestrategy" "uniform"',
                                world_edit=lambda s: s.replace("# light\nAttributeBegin", 'LightSource "infinite"\n# light\nAttributeBegin')
                                .replace('Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ]\n  "point P" [ 556 548.8 0   556 548.8 559.2   0 548.8 559.2   0 548.8 0 ]', "")),
    # Shape "sphere" (sphere.cpp): as the area light (cone sampling of the subtended solid angle, Sphere::Pdf in the MIS) with
    # matte / mirror / glass spheres in the room; partial spheres (zmin/zmax/phimax) under a non-uniform transform with
    # reversed orientation, a tiny far-away emitter (the small-angle Taylor branch) and the quad under the spatial strategy;
    # and an emitting sphere that encloses the whole room (reference points inside it: area sampling + Shape::Pdf)
    "sphere_light": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: s.replace(
        'Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ]\n  "point P" [ 343 548.7 227   343 548.7 332   213 548.7 332   213 548.7 227 ]',
        'Translate 278 480 280\n  Shape "sphere" "float radius" [ 35 ]')
        .replace("# short box", SPHERES + "# short box")),
    "sphere_partial": cornell(32, 32, 8, world_edit=lambda s: s.replace("# light\nAttributeBegin", SPHERES_PARTIAL + "# light\nAttributeBegin")),
    "sphere_enclosing": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 3 ] "string lightsamplestrategy" "power"',
                                world_edit=lambda s: s.replace("# light\nAttributeBegin", SPHERE_ENCLOSING + "# light\nAttributeBegin")
                                .replace('Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ]\n  "point P" [ 556 548.8 0   556 548.8 559.2   0 548.8 559.2   0 548.8 0 ]', "")),
    # the materials that are BxDF lists only (PG_MAT_LOBES): uber (opacity, specular reflection + transmission lobes,
    # anisotropic roughness), metal (FresnelConductor; the default copper spectra), substrate (FresnelBlend), translucent
    # (Lambertian + microfacet transmission), rough glass (MicrofacetTransmission) and mix (ScaledBxDF, nested, with specular lobes)
    "mat_uber": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: s
                        .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "uber" "rgb Kd" [ 0.3 0.5 0.2 ] "rgb Kr" [ 0.2 0.2 0.2 ] "rgb Kt" [ 0.4 0.4 0.5 ] "rgb opacity" [ 0.6 0.7 1 ] "float uroughness" [ 0.05 ] "float vroughness" [ 0.3 ] "float index" [ 1.3 ]')
                        .replace("# tall box", 'Material "uber"\n# tall box')
                        .replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "uber" "rgb Kd" [ 0.65 0.05 0.05 ] "rgb Ks" [ 0.4 0.4 0.4 ] "float roughness" [ 0.2 ] "bool remaproughness" "false" "float eta" [ 2 ]')),
    "mat_metal": cornell(32, 32, 8, world_edit=lambda s: s
                         .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "metal" "rgb eta" [ 0.2 0.9 1.1 ] "rgb k" [ 3.9 2.4 2.1 ] "float uroughness" [ 0.02 ] "float vroughness" [ 0.2 ]')
                         .replace("# tall box", 'Material "metal"\n# tall box')
                         .replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "metal" "float roughness" [ 0.3 ] "bool remaproughness" "false" "rgb eta" [ 1.5 0.4 0.3 ] "rgb k" [ 1.8 2.5 3 ]')),
    "mat_substrate": cornell(32, 32, 8, world_edit=lambda s: s
                             .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "substrate" "rgb Kd" [ 0.1 0.3 0.6 ] "rgb Ks" [ 0.2 0.2 0.2 ] "float uroughness" [ 0.02 ] "float vroughness" [ 0.4 ]')
                             .replace("# tall box", 'Material "substrate"\n# tall box')
                             .replace('Material "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', 'Material "substrate" "rgb Kd" [ 0.7 0.7 0.7 ] "rgb Ks" [ 0 0 0 ] "bool remaproughness" "false"', 1)),
    "mat_translucent": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 7 ]', world_edit=lambda s: s
                               .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "translucent" "rgb Kd" [ 0.5 0.6 0.3 ] "rgb Ks" [ 0.3 0.3 0.3 ] "rgb reflect" [ 0.3 0.3 0.3 ] "rgb transmit" [ 0.8 0.8 0.8 ] "float roughness" [ 0.15 ]')
                               .replace("# tall box", 'Material "translucent" "rgb reflect" [ 0 0 0 ]\n# tall box')),
    "mat_roughglass": cornell(32, 32, 16, integrator='Integrator "path" "integer maxdepth" [ 8 ]', world_edit=lambda s: s
                              .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "glass" "float uroughness" [ 0.1 ] "float vroughness" [ 0.35 ] "float index" [ 1.4 ] "rgb Kt" [ 0.9 1 0.9 ]')
                              .replace("# tall box", 'Material "glass" "float uroughness" [ 0.05 ] "float vroughness" [ 0.05 ] "bool remaproughness" "false" "rgb Kr" [ 0 0 0 ]\n# tall box')),
    "mat_mix": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 7 ]', world_edit=lambda s: s
                       .replace("# light\nAttributeBegin", MIX_MATERIALS + "# light\nAttributeBegin")
                       .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nNamedMaterial "mixA"')
                       .replace("# tall box", 'NamedMaterial "mixNested"\n# tall box')
                       .replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'NamedMaterial "mixSpec"')),
    # object instancing (api.cpp:1509-1588, TransformedPrimitive primitive.cpp:76-103): the two boxes become an object
    # definition with its own BVH, instanced under translations, a rotation, a non-uniform and a mirroring scale and the
    # identity; a one-primitive object (no accelerator) holding a sphere; materials bound inside the definition
    "instance_boxes": cornell(40, 40, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_instances(s)),
    "instance_accel": cornell(32, 32, 8, world_edit=lambda s: with_instances(s)).replace('WorldBegin', 'Accelerator "bvh" "integer maxnodeprims" [ 3 ] "string splitmethod" "middle"\nWorldBegin'),
    # Shape "cylinder" and "disk" (cylinder.cpp, disk.cpp): as geometry (partial sweeps, annulus, non-uniform transforms,
    # reversed orientation) and as area-light shapes (Shape::Sample / Shape::Pdf over their Sample(u))
    "quadrics": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: s.replace(
        'Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ]\n  "point P" [ 343 548.7 227   343 548.7 332   213 548.7 332   213 548.7 227 ]',
        'Translate 278 548 280\n  Rotate 90 1 0 0\n  Shape "disk" "float radius" [ 70 ] "float innerradius" [ 20 ] "float phimax" [ 300 ]')
        .replace("# short box", QUADRICS + "# short box")),
    "quadric_lights": cornell(24, 24, 8, integrator='Integrator "path" "integer maxdepth" [ 4 ] "string lightsamplestrategy" "power"',
                              world_edit=lambda s: s.replace("# short box", QUADRIC_LIGHTS + "# short box")),
    # Accelerator "bvh" "string splitmethod" "hlbvh" (bvh.cpp:404-638): Morton-sorted LBVH treelets under an SAH top tree
    "hlbvh_cornell": cornell(24, 24, 8, world_edit=lambda s: with_instances(s)).replace('WorldBegin', 'Accelerator "bvh" "string splitmethod" "hlbvh" "integer maxnodeprims" [ 2 ]\nWorldBegin'),
    # EnvironmentCamera (environment.cpp): the whole sphere of directions from inside the box
    "cornell_envcam": cornell(48, 24, 8).replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "environment"').replace("LookAt 278 273 -800  278 273 0  0 1 0", "LookAt 200 300 150  278 273 400  0 1 0"),
    # textures (core/texture.cpp, textures/): checkerboards (2D closed-form / none, 3D), scale, mix, uv, bilerp under every 2D
    # mapping, as parameters of every material kind; the camera rays' differentials (perspective with and without a lens,
    # orthographic, environment) drive the closed-form filter at the first hit
    "tex_checker": cornell(40, 40, 8, world_edit=lambda s: with_textures(s)),
    "tex_materials": cornell(40, 40, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_textured_materials(s)),
    "tex_mappings_lens": cornell(36, 36, 8, world_edit=lambda s: with_mappings(s)).replace('"float fov" [ 39.3 ]', '"float fov" [ 39.3 ] "float lensradius" [ 6 ] "float focaldistance" [ 900 ]'),
    "tex_ortho": cornell(32, 32, 4, world_edit=lambda s: with_textures(s)).replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "orthographic" "float screenwindow" [ -300 300 -290 310 ] "float lensradius" [ 4 ] "float focaldistance" [ 1000 ]'),
    "tex_envcam": cornell(48, 24, 4, world_edit=lambda s: with_mappings(s)).replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "environment"').replace("LookAt 278 273 -800  278 273 0  0 1 0", "LookAt 200 300 150  278 273 400  0 1 0"),
    # ImageTexture / MIPMap (imagemap.cpp, mipmap.h): PNG / TGA / PFM inputs, Lanczos resampling to powers of two, EWA and
    # trilinear lookups, the three wrap modes, gamma / scale conversion, float and spectrum textures, a missing file
    "tex_image": cornell(40, 40, 8, world_edit=lambda s: with_image_textures(s)),
    "tex_image_lens": cornell(32, 32, 4, world_edit=lambda s: with_image_textures(s)).replace('"float fov" [ 39.3 ]', '"float fov" [ 39.3 ] "float lensradius" [ 10 ] "float focaldistance" [ 700 ]'),
    # InfiniteAreaLight with an environment map (infinite.cpp:46-135): the map's MIPMap times L, the Distribution2D over
    # 2w x 2h trilinear lookups, importance sampling, escaped-ray Le; a non-power-of-two PFM and a PNG, rotated lights
    "env_map": cornell(40, 40, 8, integrator='Integrator "path" "integer maxdepth" [ 4 ]',
                       world_edit=lambda s: s.replace("  AreaLightSource", "#  AreaLightSource")
                       .replace("# light\nAttributeBegin", 'AttributeBegin\n  Rotate 40 1 0.2 0\n  LightSource "infinite" "string mapname" "img_color.pfm" "rgb L" [ 2 2.5 3 ]\nAttributeEnd\n# light\nAttributeBegin')
                       .replace('Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ]\n  "point P" [ 556 548.8 0   556 548.8 559.2   0 548.8 559.2   0 548.8 0 ]', "")
                       .replace("# tall box", 'Material "mirror"\n# tall box')),
    "env_map_mixed": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 5 ] "string lightsamplestrategy" "power"',
                             world_edit=lambda s: s.replace("# light\nAttributeBegin", 'AttributeBegin\n  Rotate -70 0 0 1\n  LightSource "infinite" "string mapname" "img_rgb.png" "rgb scale" [ 0.5 0.5 0.5 ]\nAttributeEnd\n# light\nAttributeBegin')
                             .replace('Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ]\n  "point P" [ 556 548.8 0   556 548.8 559.2   0 548.8 559.2   0 548.8 0 ]', "")
                             .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "glass"')),
    # alpha / shadowalpha textures on triangle meshes (triangle.cpp:333-338, :531-569): checkerboard and image masks, a
    # constant 0 (an invisible mesh), a shadow-only mask, and a masked emitter (sampled and MIS-weighted without the mask)
    "alpha_masks": cornell(40, 40, 8, integrator='Integrator "path" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_alpha(s)),
    # "bumpmap" displacement textures (Material::Bump, material.cpp:46-85) on flat and smooth-shaded triangles, quadrics,
    # instanced geometry; procedural and image displacements, a constant one, bump on the first material of a mix
    "bump_maps": cornell(40, 40, 8, integrator='Integrator "path" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_bump(s)),
    # VolPathIntegrator + HomogeneousMedium (integrators/volpath.cpp, media/homogeneous.cpp): fog everywhere with the camera
    # inside it; media bounded by "none" surfaces; delta / infinite lights through media; the plain path integrator in a
    # scene that has medium boundaries (path.cpp:107-113)
    "vol_fog": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_fog(s)),
    "vol_fog_halfspace": cornell(24, 24, 8, integrator='Integrator "volpath" "integer maxdepth" [ 6 ] "string lightsamplestrategy" "power"',
                                 world_edit=lambda s: with_fog(s, world_interface="")),
    "vol_smoke": cornell(40, 40, 8, integrator='Integrator "volpath" "integer maxdepth" [ 8 ] "float rrthreshold" [ 0.7 ]',
                         world_edit=lambda s: with_smoke(s).replace("# tall box", 'Material "glass"\n# tall box')),
    "vol_smoke_delta": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 5 ] "string lightsamplestrategy" "uniform"',
                               world_edit=lambda s: with_smoke(s).replace("# light\nAttributeBegin", DELTA_POINT + DELTA_SPOT + 'LightSource "infinite" "rgb L" [ 0.3 0.4 0.6 ]\n# light\nAttributeBegin')),
    "vol_path_none": cornell(24, 24, 8, world_edit=lambda s: with_smoke(s)),
    "vol_path_none_glass": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 8 ]',
                                   world_edit=lambda s: with_smoke(s).replace("# tall box", 'Material "glass"\n# tall box').replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "mirror"')),
    "vol_presets": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_smoke(s))
                   .replace('"rgb sigma_a" [ 0.2 0.5 1.0 ] "rgb sigma_s" [ 3 3 3 ] "float scale" [ 0.01 ]', '"string preset" "Skin1" "float scale" [ 0.05 ]')
                   .replace('"rgb sigma_a" [ 0.001 0.001 0.001 ] "rgb sigma_s" [ 0.004 0.002 0.001 ]', '"string preset" "Regular Milk" "rgb sigma_a" [ 0.0005 0.001 0.002 ] "float scale" [ 0.002 ]'),
    "vol_instances": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_fog(with_instances(s), camera_in_fog=False)),
    # SobolSampler (samplers/sobol.cpp): power-of-two and rounded-up sample counts, non-square frames, crop windows and wide
    # filters (sample bounds that start below / above 0), under both integrators
    "sobol_cornell": cornell(32, 32, 8).replace('Sampler "halton"', 'Sampler "sobol"'),
    "sobol_round_crop": cornell(40, 24, 6, extra_film='"float cropwindow" [ 0.3 0.9 0.2 0.7 ]').replace('Sampler "halton"', 'Sampler "sobol"'),
    "filter_sobol_gaussian": cornell(40, 24, 4).replace('PixelFilter "box"', 'PixelFilter "gaussian"').replace('Sampler "halton"', 'Sampler "sobol"'),
    "sobol_vol_smoke": cornell(32, 32, 4, integrator='Integrator "volpath" "integer maxdepth" [ 6 ]',
                               world_edit=lambda s: with_fog(with_smoke(s)).replace("# tall box", 'Material "glass"\n# tall box')).replace('Sampler "halton"', 'Sampler "sobol"'),
    "sobol_tex_lens": cornell(32, 32, 4, world_edit=lambda s: with_image_textures(s)).replace('"float fov" [ 39.3 ]', '"float fov" [ 39.3 ] "float lensradius" [ 10 ] "float focaldistance" [ 700 ]').replace('Sampler "halton"', 'Sampler "sobol"'),
    # "blackbody", "xyz" and "spectrum" (inline pairs, unsorted pairs, .spd files) parameters -> RGB (paramset.cpp:122-208)
    "spectrum_params": cornell(32, 32, 8, world_edit=lambda s: s.replace('"rgb L" [ 17 12 4 ]', '"blackbody L" [ 4500 14 ]')
                               .replace('Material "matte" "rgb Kd" [ 0.12 0.45 0.15 ]', 'Material "matte" "xyz Kd" [ 0.2 0.35 0.1 ]')
                               .replace('Material "matte" "rgb Kd" [ 0.65 0.05 0.05 ]', 'Material "matte" "spectrum Kd" [ 400 0.05 700 0.7 550 0.1 620 0.65 ]')
                               .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "plastic" "spectrum Kd" "test_kd.spd" "spectrum Ks" [ 300 0.3 900 0.5 ]')
                               .replace("# tall box", 'Material "metal" "spectrum eta" "test_eta.spd" "spectrum k" "test_k.spd" "float roughness" [ 0.05 ]\n# tall box')
                               .replace("# light\nAttributeBegin", 'LightSource "point" "point from" [ 100 400 100 ] "blackbody I" [ 2800 30000 ] "xyz scale" [ 1 1 1.2 ]\n# light\nAttributeBegin')),
    "mesh_shapes": cornell(40, 40, 8, world_edit=lambda s: with_mesh_shapes(s)),
    # cone / paraboloid / hyperboloid: full and partial, transformed, reversed, specular (error bounds), inside an instance
    "quadrics3": cornell(40, 40, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: s.replace("# short box", QUADRICS3 + "# short box", 1)),
    # Perlin-noise textures: fbm, wrinkled, windy, marble, dots (float and spectrum; as Kd, sigma, roughness, opacity, bump maps)
    "tex_noise": cornell(40, 40, 8, world_edit=lambda s: with_noise_textures(s)),
    "tex_noise_lens": cornell(32, 32, 4, world_edit=lambda s: with_noise_textures(s)).replace('"float fov" [ 39.3 ]', '"float fov" [ 39.3 ] "float lensradius" [ 10 ] "float focaldistance" [ 700 ]'),
    # ProjectionLight and GonioPhotometricLight: with and without a map, rotated, under the power strategy, through a medium
    "light_projection": cornell(32, 32, 8, world_edit=lambda s: s.replace("# light\nAttributeBegin", IMAGE_LIGHTS + "# light\nAttributeBegin")),
    "light_gonio_power": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 4 ] "string lightsamplestrategy" "power"',
                                 world_edit=lambda s: with_fog(s).replace("# light\nAttributeBegin", IMAGE_LIGHTS + "# light\nAttributeBegin").replace("  AreaLightSource", "#  AreaLightSource")),
    "cornell_lens": cornell(24, 24, 8).replace('"float fov" [ 39.3 ]', '"float fov" [ 39.3 ] "float lensradius" [ 8 ] "float focaldistance" [ 1000 ]'),
    # A moving camera: CameraToWorld is an AnimatedTransform interpolated at every camera ray's time (transform.cpp:1144-1181,
    # perspective.cpp:89-91, :139).  Translation only; a rotation large enough for Slerp's acos / sin / cos branch; one so small that
    # Slerp normalises the blend (cosTheta > .9995); translation + rotation + scale with TransformTimes inside the shutter interval
    # (rays before the start time and after the end time take the end transforms themselves); then the rays' differentials through
    # the same interpolated transform (filtered image textures) under every sampler family, a lens, the other two cameras, and volpath.
    # moving shapes and instances (TransformedPrimitive over an AnimatedTransform, primitive.cpp:76-103; api.cpp:1386-1419, :1576-1586): translation,
    # scale, a rotation small enough to count as none (Dot(R[0], R[1]) >= 0.9995: the bounds stay the union of the ends, Interpolate still blends the
    # quaternions), TransformTimes inside the shutter (the end transforms outside the motion), textures on a moving mesh, every sampler family (the
    # ray's time comes from the camera sample), volpath, a camera that moves as well
    "motion_boxes": with_moving_boxes(cornell(32, 32, 8)),
    "motion_boxes_times": with_moving_boxes(cornell(32, 24, 8), times="TransformTimes 0.25 0.6\n"),
    "motion_small_rotation": with_moving_boxes(cornell(24, 24, 8), short_motion="Rotate 2.5 0 1 0\nTranslate 30 0 0", tall_motion="Translate 0 40 0\nRotate -1.5 0.2 1 0"),
    "motion_instances": with_moving_instances(cornell(40, 32, 8)),
    "motion_instances_shutter": with_moving_instances(cornell(32, 32, 4)).replace('Camera "perspective"', 'Camera "perspective" "float shutteropen" [ 0.3 ] "float shutterclose" [ 0.8 ]'),
    "motion_tex": with_moving_boxes(cornell(32, 32, 4, world_edit=lambda s: with_image_textures(s))),
    "motion_sobol": with_moving_boxes(cornell(24, 24, 4)).replace('Sampler "halton"', 'Sampler "sobol"'),
    "motion_random": with_sampler(with_moving_boxes(cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 3 ]')), '"random" "integer pixelsamples" [ 4 ]'),
    "motion_stratified": with_sampler(with_moving_boxes(cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 4 ]')),
                                      '"stratified" "integer xsamples" [ 2 ] "integer ysamples" [ 2 ] "integer dimensions" [ 14 ]'),
    "motion_vol": with_moving_instances(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 4 ]', world_edit=lambda s: with_fog(s))),
    # a moving shape under a set AreaLightSource: the reference ignores the light (api.cpp:1389-1391) -- the box moves and does not emit; a moving
    # shape placed through a named coordinate system whose two ends differ (CoordinateSystem / CoordSysTransform keep both transforms, api.cpp:478-500)
    "motion_arealight_ignored": with_moving_boxes(cornell(24, 24, 4)).replace('# short box\n', '# short box\nAreaLightSource "diffuse" "rgb L" [ 30 5 5 ]\n', 1),
    "motion_coordsys": cornell(24, 24, 4).replace("WorldBegin\n", 'WorldBegin\nAttributeBegin\nTranslate 100 250 150\nActiveTransform EndTime\nTranslate 80 -60 40\nScale 1.3 1 1\nActiveTransform All\n'
                                                  'CoordinateSystem "mover"\nAttributeEnd\nAttributeBegin\nCoordSysTransform "mover"\nMaterial "metal" "float roughness" [ 0.1 ]\n'
                                                  'Shape "sphere" "float radius" [ 50 ]\nTranslate 120 0 0\nShape "cylinder" "float radius" [ 25 ] "float zmin" [ -40 ] "float zmax" [ 40 ]\nAttributeEnd\n', 1),
    # motion that ROTATES (AnimatedTransform::hasRotation, transform.cpp:411): the TransformedPrimitive's box is the corners' paths bounded at the zeros
    # of their derivatives (MotionBounds / BoundPointMotion / IntervalFindZeros, transform.cpp:1215-1247, :354-394) -- the top-level BVH and, through
    # the scene's world bound, the distant and infinite lights and the spatial light grid depend on it; rays slerp the rotation at their own time
    "motion_rotate_boxes": with_moving_boxes(cornell(32, 32, 8), short_motion="Rotate 35 0 1 0\nTranslate 30 0 0", tall_motion="Translate 0 40 0\nRotate -50 0.2 1 0.3\nScale 1 0.8 1.1"),
    "motion_rotate_big_times": with_moving_boxes(cornell(32, 24, 8), short_motion="Translate 60 80 0\nRotate 170 0.1 1 0", tall_motion="Rotate 95 1 0.2 0.1\nTranslate 0 -60 30", times="TransformTimes 0.2 0.9\n"),
    "motion_rotate_instances": with_moving_instances(cornell(40, 32, 8), spin=("  Rotate 40 0 1 0.2\n", "  Rotate -75 1 0 0\n", "  Rotate 120 0 0 1\n", "  Rotate 60 1 1 0\n")),
    "motion_rotate_distant_spatial": with_moving_boxes(cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 5 ] "string lightsamplestrategy" "spatial"'),
                                                       short_motion="Translate 0 250 -700\nRotate 80 1 0 0.3", tall_motion="Rotate 30 0 1 0")
                                     .replace("# light\nAttributeBegin", 'LightSource "distant" "point from" [ 278 500 -600 ] "point to" [ 278 200 200 ] "rgb L" [ 2 2 1.5 ]\n# light\nAttributeBegin', 1),
    "motion_rotate_vol": with_moving_instances(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 4 ]', world_edit=lambda s: with_fog(s)),
                                               spin=("  Rotate -30 0.3 1 0\n", "  Rotate 45 0 1 1\n", "", "  Rotate 90 0 0 1\n")),
    "motion_rotate_camera_too": cam_anim(with_moving_boxes(cornell(32, 24, 8), short_motion="Rotate -60 0 1 0", tall_motion="Rotate 25 0 0 1\nTranslate 20 0 0"), "Translate 30 0 -40\nRotate 12 0.1 1 0.2"),
    # moving shapes INSIDE object definitions (round 6, ABI 29): a TransformedPrimitive under the TransformedPrimitive of every ObjectInstance -- still and
    # moving instances around them, a rotating inner motion, a shutter inside the motion, volpath through fog, the samplers over one RNG stream per tile
    "nest_motion": with_nested_motion(with_instances(cornell(40, 32, 8))),
    "nest_motion_moving_instances": with_nested_motion(with_moving_instances(cornell(32, 32, 4))).replace('Camera "perspective"', 'Camera "perspective" "float shutteropen" [ 0.3 ] "float shutterclose" [ 0.8 ]'),
    "nest_motion_rotate": with_nested_motion(with_moving_instances(cornell(32, 32, 8), spin=("  Rotate 40 0 1 0.2\n", "  Rotate -75 1 0 0\n", "  Rotate 120 0 0 1\n", "  Rotate 60 1 1 0\n")),
                                             tall="  Rotate 65 0.1 1 0\n  Translate 20 0 -30\n", ball="  Rotate 100 0 0 1\n  Translate 60 0 0\n", tri="  Rotate -45 1 0 0.3\n"),
    "nest_motion_vol": with_nested_motion(with_moving_instances(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 4 ]', world_edit=lambda s: with_fog(s)))),
    "nest_motion_sobol": with_nested_motion(with_instances(cornell(24, 24, 4))).replace('Sampler "halton"', 'Sampler "sobol"'),
    "nest_motion_random": with_sampler(with_nested_motion(with_moving_instances(cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 3 ]'))), '"random" "integer pixelsamples" [ 4 ]'),
    "motion_camera_too": cam_anim(with_moving_boxes(cornell(32, 24, 8)), "Translate 30 0 -40\nRotate 12 0.1 1 0.2"),
    "camanim_translate": cam_anim(cornell(32, 32, 8), "Translate 40 -20 60"),
    "camanim_rotate": cam_anim(cornell(32, 32, 8), "Translate 30 0 -40\nRotate 25 0.1 1 0.2"),
    "camanim_small_rotate": cam_anim(cornell(24, 24, 8), "Rotate 1 0 1 0"),
    "camanim_times_scale": cam_anim(cornell(32, 24, 8), "Translate -35 10 20\nRotate -18 1 0.3 0\nScale 1.1 0.9 1", times="TransformTimes 0.2 0.7\n")
                           .replace('"float fov" [ 39.3 ]', '"float fov" [ 39.3 ] "float shutteropen" [ 0 ] "float shutterclose" [ 1 ]'),
    "camanim_lens_tex": cam_anim(cornell(32, 32, 4, world_edit=lambda s: with_image_textures(s)), "Translate 25 15 0\nRotate 12 0 1 0.1")
                        .replace('"float fov" [ 39.3 ]', '"float fov" [ 39.3 ] "float lensradius" [ 10 ] "float focaldistance" [ 700 ] "float shutteropen" [ 0.1 ] "float shutterclose" [ 0.9 ]'),
    "camanim_sobol_tex": cam_anim(cornell(32, 32, 4, world_edit=lambda s: with_image_textures(s)), "Rotate 15 0.2 1 0").replace('Sampler "halton"', 'Sampler "sobol"'),
    "camanim_strat_dims_tex": with_sampler(cam_anim(cornell(32, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 4 ]', world_edit=lambda s: with_image_textures(s)), "Translate 20 0 30\nRotate 10 0 1 0"),
                                           '"stratified" "integer xsamples" [ 2 ] "integer ysamples" [ 2 ] "integer dimensions" [ 14 ]'),
    "camanim_random_tex": with_sampler(cam_anim(cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 3 ]', world_edit=lambda s: with_image_textures(s)), "Rotate 20 0 1 0"),
                                       '"random" "integer pixelsamples" [ 3 ]'),
    "camanim_ortho": cam_anim(cornell(24, 24, 8).replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "orthographic" "float screenwindow" [ -300 300 -290 310 ]'), "Translate 0 40 0\nRotate 14 0 0 1"),
    "camanim_env": cam_anim(cornell(32, 16, 8, world_edit=lambda s: with_image_textures(s)).replace("LookAt 278 273 -800  278 273 0  0 1 0", "LookAt 278 273 100  278 273 400  0 1 0")
                            .replace('Camera "perspective" "float fov" [ 39.3 ]', 'Camera "environment"'), "Translate 60 0 0\nRotate 30 0 1 0"),
    "camanim_vol": cam_anim(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 4 ]', world_edit=lambda s: with_fog(s)), "Translate 0 0 50\nRotate 16 0.3 1 0"),
}


# ---- GridDensityMedium ("heterogeneous", media/grid.cpp).  ABI 23 carries its tables and the CPU oracle renders it; the device
# (round 2: refused by the device library, goldens apart in tests/golden_grid/; round 3: two-phase kernels, goldens in tests/golden/) ----
GOLD_GRID = GOLD  # (apart in tests/golden_grid/ until the device had the two-phase kernels: round 3)


def density_values(nx, ny, nz, peak, seed):
    """A puff with some structure: a smooth blob times a lattice pattern, one empty slab, deterministic."""
    import math
    out = []
    for z in range(nz):
        for y in range(ny):
            for x in range(nx):
                u, v, w = (x + .5) / nx - .5, (y + .5) / ny - .5, (z + .5) / nz - .5
                blob = math.exp(-6 * (u * u + v * v + w * w))
                lat = 0.55 + 0.45 * math.sin(3.1 * x + seed) * math.cos(2.3 * y - seed) * math.sin(1.7 * z + 0.5 * seed)
                out.append(0.0 if (y == 1 and ny > 3) else peak * blob * lat)
    return " ".join(f"{v:.6g}" for v in out)


def grid_medium(name, nx, ny, nz, p0, p1, sigma_a, sigma_s, g, peak, seed, extra=""):
    return (f'MakeNamedMedium "{name}" "string type" "heterogeneous" "rgb sigma_a" [ {sigma_a} ] "rgb sigma_s" [ {sigma_s} ] "float g" [ {g} ] {extra}\n'
            f'  "integer nx" [ {nx} ] "integer ny" [ {ny} ] "integer nz" [ {nz} ] "point p0" [ {p0} ] "point p1" [ {p1} ]\n'
            f'  "float density" [ {density_values(nx, ny, nz, peak, seed)} ]\n')


BOX_MESH = ('Shape "trianglemesh" "integer indices" [ 0 2 1 0 3 2  4 5 6 4 6 7  0 1 5 0 5 4  2 3 7 2 7 6  1 2 6 1 6 5  3 0 4 3 4 7 ]\n'
            '    "point P" [ %(x0)g %(y0)g %(z0)g  %(x1)g %(y0)g %(z0)g  %(x1)g %(y1)g %(z0)g  %(x0)g %(y1)g %(z0)g  %(x0)g %(y0)g %(z1)g  %(x1)g %(y0)g %(z1)g  %(x1)g %(y1)g %(z1)g  %(x0)g %(y1)g %(z1)g ]\n')


def with_grid_puff(s, dense=False):
    """A GridDensityMedium inside a "none"-material box in front of the tall box (medium defined in the world block: identity CTM)."""
    if dense:  # optical depth ~ 20 through the middle: VisibilityTester::Tr's roulette (grid.cpp:108-116) decides most shadow rays
        med = grid_medium("puff", 5, 4, 6, "150 20 100", "400 350 400", "0.03125 0.0625 0.015625", "0.0625 0.03125 0.078125", 0.6, 3.0, 1.0)
    else:
        med = grid_medium("puff", 6, 5, 4, "150 20 100", "400 350 400", "0.0078125 0.015625 0.00390625", "0.015625 0.0078125 0.01953125", -0.3, 1.5, 2.0)
    box = 'AttributeBegin\n  MediumInterface "puff" ""\n  Material "none"\n  ' + BOX_MESH % dict(x0=150, y0=20, z0=100, x1=400, y1=350, z1=400) + 'AttributeEnd\n'
    return s.replace("WorldBegin\n", "WorldBegin\n" + med, 1).replace("# short box", box + "# short box", 1)


def with_grid_fog(s):
    """The whole box filled by a coarse GridDensityMedium, the camera inside it; the medium is declared before LookAt (identity CTM)."""
    med = grid_medium("gfog", 4, 4, 8, "-50 -50 -900", "600 600 600", "0.00048828125 0.00048828125 0.00048828125", "0.00146484375 0.00146484375 0.00146484375", 0.2, 1.0, 0.7)
    s = med + s
    s = s.replace("Camera ", 'MediumInterface "" "gfog"\nCamera ', 1)
    return s.replace("WorldBegin\n", 'WorldBegin\nMediumInterface "gfog" "gfog"\n', 1)


def with_grid_transformed(s):
    """The medium declared under a rotated, non-uniformly scaled CTM with "scale" and a preset's coefficients replaced by uniform
    ones; a HomogeneousMedium next to it in the table (media_grid = -1 for that one); the grid's box is a "none" sphere that is
    larger than the grid in places, so rays enter the medium but miss the unit cube."""
    med = ('AttributeBegin\n  Translate 278 200 250\n  Rotate 35 0.3 1 0.2\n  Scale 160 120 140\n' +
           grid_medium("cloud", 3, 7, 5, "-1 -1 -1", "1 1 1", "0.125 0.125 0.125", "1.875 1.875 1.875", 0.4, 1.2, 3.0, extra='"float scale" [ 0.01 ]') +
           'AttributeEnd\n'
           'MakeNamedMedium "thin" "string type" "homogeneous" "rgb sigma_a" [ 0.001 0.002 0.001 ] "rgb sigma_s" [ 0.003 0.002 0.004 ]\n')
    ball = ('AttributeBegin\n  Translate 278 200 250\n  MediumInterface "cloud" "thin"\n  Material "none"\n  Shape "sphere" "float radius" [ 170 ]\nAttributeEnd\n')
    s = s.replace("WorldBegin\n", 'WorldBegin\n' + med + 'MediumInterface "thin" "thin"\n', 1)
    s = s.replace("Camera ", 'MakeNamedMedium "thin0" "string type" "homogeneous" "rgb sigma_a" [ 0.001 0.002 0.001 ] "rgb sigma_s" [ 0.003 0.002 0.004 ]\nMediumInterface "" "thin0"\nCamera ', 1)
    return s.replace("# short box", ball + "# short box", 1)


GRID_SCENES = {
    "grid_puff": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=with_grid_puff),
    "grid_puff_dense": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 6 ] "string lightsamplestrategy" "uniform"',
                               world_edit=lambda s: with_grid_puff(s, dense=True).replace("# light\nAttributeBegin", DELTA_POINT + DELTA_SPOT + "# light\nAttributeBegin")),
    "grid_fog_camera": cornell(32, 24, 8, integrator='Integrator "volpath" "integer maxdepth" [ 4 ]',
                               world_edit=lambda s: with_grid_fog(s).replace("# light\nAttributeBegin", DELTA_POINT + "# light\nAttributeBegin")),
    "grid_transformed": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 5 ] "float rrthreshold" [ 0.5 ]', world_edit=with_grid_transformed),
    # a GridDensityMedium beside MOVING boxes: the transmittance rays of ratio tracking and both shading phases carry the rays' times
    "grid_puff_motion": with_moving_boxes(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=with_grid_puff)),
    "grid_puff_motion_random": with_moving_boxes(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 4 ]', world_edit=lambda s: with_grid_puff(s, dense=True)),
                                                 times="TransformTimes 0.2 0.9\n").replace('Sampler "halton"', 'Sampler "random"'),
    "grid_puff_sobol": cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=with_grid_puff).replace('Sampler "halton"', 'Sampler "sobol"'),
    "grid_puff_random": cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=with_grid_puff).replace('Sampler "halton"', 'Sampler "random"'),
    "grid_puff_stratified": cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_grid_puff(s, dense=True))
                            .replace('Sampler "halton" "integer pixelsamples" [ 4 ]', 'Sampler "stratified" "integer xsamples" [ 2 ] "integer ysamples" [ 2 ] "integer dimensions" [ 3 ]'),
    # a mirror and a glass box INSIDE the puff: specular paths cross the medium's material-less boundary on their way to the light, and
    # `specularBounce` must survive that surface (path.cpp:107-113) for the light's emission to count
    "grid_puff_specular": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]',
                                  world_edit=lambda s: with_grid_puff(s).replace("# tall box", 'Material "mirror"\n# tall box')
                                  .replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\nMaterial "glass"')),
    # PathIntegrator ignores media (path.cpp): the "none" box is simply passed through
    "grid_path_ignores": cornell(24, 24, 4, world_edit=with_grid_puff),
}
# A GridDensityMedium AND subsurface materials in one scene (VolPathIntegrator::Li handles both in one loop, volpath.cpp:76-176; the device refused the
# pair until round 6): the entry vertex's ratio-tracking draws come before its BSDF / Sample_S draws, the exit vertex's between its light sample and
# its next direction.  (Defined below GRID_SCENES / SSS_SCENES' helpers; merged into GRID_SCENES at the end of the SSS block.)
GRID_SSS_SCENES = lambda: {
    "grid_sss_puff": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_sss(with_grid_puff(s), SSS_PLAIN)),
    "grid_sss_dense_delta": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 6 ] "string lightsamplestrategy" "uniform"',
                                    world_edit=lambda s: with_sss(with_grid_puff(s, dense=True), SSS_PLAIN, tall='Material "matte" "rgb Kd" [ 0.6 0.6 0.3 ]')
                                    .replace("# light\nAttributeBegin", DELTA_POINT + DELTA_SPOT + "# light\nAttributeBegin")),
    "grid_sss_fog_camera": cornell(32, 24, 8, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_sss(with_grid_fog(s), SSS_PLAIN)),
    "grid_sss_random": cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_sss(with_grid_puff(s, dense=True), SSS_PLAIN))
                       .replace('Sampler "halton"', 'Sampler "random"'),
    "grid_sss_sobol": cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(with_grid_puff(s), SSS_PLAIN))
                      .replace('Sampler "halton"', 'Sampler "sobol"'),
    "grid_sss_motion": with_moving_boxes(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_sss(with_grid_puff(s), SSS_PLAIN))),
}


# ---- Subsurface scattering: SubsurfaceMaterial / KdSubsurfaceMaterial + the BSSRDF branch of PathIntegrator::Li / VolPathIntegrator::Li
# (path.cpp:152-174, core/bssrdf.cpp).  ABI 24 carries the tables and the CPU oracle renders it; the device library refuses it
# (round 2: PG_ERR_UNSUPPORTED, goldens apart in tests/golden_sss/; round 3: kernels, goldens in tests/golden/ like all others) ----
GOLD_SSS = GOLD  # (kept apart in tests/golden_sss/ until the device had kernels for the BSSRDF branch: round 3)
SHORT_BOX_MATTE = '# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]'
SSS_PLAIN = 'Material "subsurface" "rgb sigma_a" [ 0.002 0.004 0.02 ] "rgb sigma_s" [ 0.05 0.06 0.08 ] "float eta" [ 1.33 ]'


def with_sss(s, mat, tall=None):
    """The short box (and, unless `tall` names another material, the tall box after it -- the same Material object) made of `mat`."""
    assert SHORT_BOX_MATTE in s
    s = s.replace(SHORT_BOX_MATTE, "# short box\n" + mat, 1)
    if tall is not None:
        s = s.replace("# tall box", tall + "\n# tall box", 1)
    return s


def with_moving_sss_boxes(s):
    """Both Cornell boxes of ONE subsurface Material object (a named material), each moving on its own: a probe segment started on one box can
    leave through the other (bssrdf.cpp:301), and both are TransformedPrimitives interpolated at the path's time."""
    s = with_moving_boxes(s, tall_motion="Translate 30 0 -20")
    assert SHORT_BOX_MATTE in s
    s = s.replace(SHORT_BOX_MATTE, '# short box\nNamedMaterial "skin"', 1)
    s = s.replace('Material "plastic" "rgb Kd" [ 0.2 0.5 0.3 ] "float roughness" [ 0.2 ]\n', 'NamedMaterial "skin"\n', 1)
    return s.replace("WorldBegin\n", 'WorldBegin\nMakeNamedMaterial "skin" "string type" "subsurface" "rgb sigma_a" [ 0.002 0.004 0.008 ] "rgb sigma_s" [ 0.1 0.08 0.06 ] "float scale" [ 1 ] "float eta" [ 1.33 ]\n', 1)


def with_nested_sss(s):
    """with_nested_motion's scene with ONE subsurface Material object on both boxes of "boxes" (the still short one and the tall one that moves inside the
    definition) and on the moving sphere inside "ball": probe chains run through hits under two transforms, started on one box they can leave through the other."""
    skin = 'NamedMaterial "skin"'
    n = s.count(skin)
    s = s.replace('# short box\nMaterial "matte" "rgb Kd" [ 0.73 0.73 0.73 ]', '# short box\n' + skin, 1)
    s = s.replace('Material "glass" "float index" [ 1.4 ]\n# tall box', skin + '\n# tall box', 1)
    s = s.replace('  Material "matte" "rgb Kd" [ 0.2 0.3 0.8 ]\n  Shape "sphere" "float radius" [ 25 ]', '  ' + skin + '\n  Shape "sphere" "float radius" [ 25 ]', 1)
    assert s.count(skin) == n + 3
    return s.replace("WorldBegin\n", 'WorldBegin\nMakeNamedMaterial "skin" "string type" "subsurface" "rgb sigma_a" [ 0.002 0.004 0.008 ] "rgb sigma_s" [ 0.1 0.08 0.06 ] "float scale" [ 1 ] "float eta" [ 1.33 ]\n', 1)


SSS_SCENES = {
    # both boxes share one SubsurfaceMaterial: probe rays started on one box may leave through the other (bssrdf.cpp:301)
    "sss_subsurface": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(s, SSS_PLAIN)),
    # the same parameters declared twice: two Material objects, the probe rays of one never accept the other
    "sss_two_materials": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(s, SSS_PLAIN, tall=SSS_PLAIN)),
    "sss_preset_volpath": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 6 ]',
                                  world_edit=lambda s: with_fog(with_sss(s, 'Material "subsurface" "string name" "Skin1" "float scale" [ 0.05 ] "float eta" [ 1.4 ]', tall='Material "matte" "rgb Kd" [ 0.6 0.6 0.6 ]'))),
    # MOVING boxes of a subsurface material: the probe rays of Sample_Sp enter the moving instance at the path's time, the exit vertex comes back
    # through the transform interpolated for the chosen probe ray
    "sss_motion": with_moving_sss_boxes(cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 5 ]')),
    "sss_motion_volpath": with_moving_boxes(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]',
                                                    world_edit=lambda s: with_fog(with_sss(s, 'Material "subsurface" "string name" "Skin1" "float scale" [ 0.05 ] "float eta" [ 1.4 ]'))),
                                            times="TransformTimes 0.1 0.8\n").replace('Sampler "halton"', 'Sampler "sobol"'),
    # moving shapes of a subsurface material INSIDE object definitions (ABI 29): the probe chains' hits lie under two transforms
    "sss_nest_motion": with_nested_sss(with_nested_motion(with_moving_instances(cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 5 ]')))),
    "sss_nest_motion_volpath": with_nested_sss(with_nested_motion(with_moving_instances(cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_fog(s)),
                                                                                     spin=("  Rotate 40 0 1 0.2\n", "  Rotate -75 1 0 0\n", "  Rotate 120 0 0 1\n", "  Rotate 60 1 1 0\n")))).replace('Sampler "halton"', 'Sampler "sobol"'),
    "sss_kd_rough": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]',
                            world_edit=lambda s: with_sss(s, 'Material "kdsubsurface" "rgb Kd" [ 0.6 0.4 0.3 ] "rgb mfp" [ 8 12 20 ] "float uroughness" [ 0.1 ] "float vroughness" [ 0.2 ] "float g" [ 0.3 ]',
                                                          tall='Material "kdsubsurface" "rgb Kd" [ 0.2 0.5 0.7 ] "rgb mfp" [ 30 30 30 ] "float scale" [ 0.5 ] "float eta" [ 1.2 ] "bool remaproughness" "false" "float uroughness" [ 0.05 ] "float vroughness" [ 0.05 ]')),
    "sss_g_power_delta": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 8 ] "string lightsamplestrategy" "power"',
                                 world_edit=lambda s: with_sss(s, 'Material "subsurface" "rgb sigma_a" [ 0.01 0.004 0.002 ] "rgb sigma_s" [ 0.2 0.3 0.1 ] "float g" [ 0.5 ] "float eta" [ 1.5 ] "float scale" [ 0.5 ] "rgb Kr" [ 0.5 0.5 0.5 ]')
                                 .replace("# light\nAttributeBegin", DELTA_POINT + DELTA_SPOT + "# light\nAttributeBegin")),
    # Kr = Kt = 0: ComputeScatteringFunctions returns before it sets the BSSRDF (subsurface.cpp:55): a black, BxDF-less surface
    "sss_black_no_bssrdf": cornell(24, 24, 4, world_edit=lambda s: with_sss(s, 'Material "subsurface" "rgb Kr" [ 0 0 0 ] "rgb Kt" [ 0 0 0 ]', tall='Material "matte" "rgb Kd" [ 0.6 0.6 0.6 ]')),
    "sss_named_spheres": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]',
                                 world_edit=lambda s: s.replace("# short box", 'MakeNamedMaterial "wax" "string type" "subsurface" "rgb sigma_a" [ 0.004 0.01 0.03 ] "rgb sigma_s" [ 0.08 0.08 0.06 ] "float eta" [ 1.45 ]\n'
                                                                'AttributeBegin\n  NamedMaterial "wax"\n  Translate 420 70 120\n  Shape "sphere" "float radius" [ 70 ]\nAttributeEnd\n'
                                                                'AttributeBegin\n  NamedMaterial "wax"\n  Translate 150 390 330\n  Rotate 30 1 0 0\n  Shape "sphere" "float radius" [ 60 ] "float zmax" [ 40 ]\nAttributeEnd\n# short box', 1)),
    # a texture / a bump map among the parameters: the BSDF and the BSSRDF's coefficients are evaluated per hit
    "sss_textured_kd_bump": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(s,
        'Texture "chk" "spectrum" "checkerboard" "float uscale" [ 3 ] "float vscale" [ 3 ] "rgb tex1" [ 0.7 0.3 0.2 ] "rgb tex2" [ 0.2 0.5 0.8 ]\n'
        'Texture "bmp" "float" "fbm" "integer octaves" [ 3 ]\nTexture "bmps" "float" "scale" "texture tex1" "bmp" "float tex2" [ 4 ]\n'
        'Material "kdsubsurface" "texture Kd" "chk" "rgb mfp" [ 10 14 20 ] "texture bumpmap" "bmps" "float eta" [ 1.3 ]')),
    # textured sigma_a; Kt is black on half of the squares: there the material returns before it sets the BSSRDF (subsurface.cpp:55)
    "sss_textured_sigma_kt": cornell(32, 32, 8, integrator='Integrator "volpath" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(s,
        'Texture "sa" "spectrum" "checkerboard" "integer dimension" [ 3 ] "rgb tex1" [ 0.002 0.004 0.02 ] "rgb tex2" [ 0.02 0.004 0.002 ]\n'
        'Texture "kt" "spectrum" "checkerboard" "float uscale" [ 2 ] "float vscale" [ 2 ] "rgb tex1" [ 1 1 1 ] "rgb tex2" [ 0 0 0 ]\n'
        'Material "subsurface" "texture sigma_a" "sa" "rgb sigma_s" [ 0.05 0.06 0.08 ] "float scale" [ 2 ] "texture Kt" "kt" "rgb Kr" [ 0 0 0 ] "float uroughness" [ 0.2 ] "float vroughness" [ 0.1 ]')),
    # a MixMaterial whose first component is a subsurface material: the mix's BSSRDF is the component's, and its probe rays accept
    # hits on primitives that carry the component ITSELF (the tall box) -- never the mixed box they start from (mixmat.cpp:52-53, bssrdf.cpp:301)
    "sss_mix_component": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(s,
        'MakeNamedMaterial "wax" "string type" "subsurface" "rgb sigma_a" [ 0.002 0.004 0.02 ] "rgb sigma_s" [ 0.02 0.03 0.04 ] "float eta" [ 1.33 ]\n'
        'MakeNamedMaterial "paint" "string type" "matte" "rgb Kd" [ 0.6 0.2 0.2 ]\n'
        'MakeNamedMaterial "waxpaint" "string type" "mix" "string namedmaterial1" "wax" "string namedmaterial2" "paint" "rgb amount" [ 0.7 0.6 0.5 ]\n'
        'NamedMaterial "waxpaint"', tall='NamedMaterial "wax"')),
    "sss_instances": cornell(32, 32, 8, integrator='Integrator "path" "integer maxdepth" [ 5 ]', world_edit=lambda s: with_instances(with_sss(s, SSS_PLAIN))),
    "sss_sobol": cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(s, SSS_PLAIN)).replace('Sampler "halton"', 'Sampler "sobol"'),
    "sss_random": cornell(24, 24, 4, integrator='Integrator "volpath" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(s, SSS_PLAIN)).replace('Sampler "halton"', 'Sampler "random"'),
    "sss_stratified": cornell(24, 24, 4, integrator='Integrator "path" "integer maxdepth" [ 6 ]', world_edit=lambda s: with_sss(s, SSS_PLAIN))
                      .replace('Sampler "halton" "integer pixelsamples" [ 4 ]', 'Sampler "stratified" "integer xsamples" [ 2 ] "integer ysamples" [ 2 ] "integer dimensions" [ 6 ]'),
}


GRID_SCENES.update(GRID_SSS_SCENES())


def run(name, scene_path, outdir=GOLD):
    ref = os.path.join(HERE, "_ref", "pbrt_oracle")
    out = os.path.join(outdir, name + ".pfm")
    # one thread for the wide-filter scenes: overlapping FilmTiles are then merged in tile order (film.cpp:117-130)
    # (also maxmindist: its first film sample of a pixel lies ON the pixel's edge, i / spp with i = 0, and reaches the neighbouring pixel -- across
    # a tile boundary that is a sum whose order depends on which thread merges its tile first)
    nthreads = "1" if name.startswith("filter_") or "maxmindist" in name else "4"
    txt = subprocess.run([ref, "--nthreads", nthreads, "--outfile", out, scene_path], capture_output=True, text=True, check=True).stdout
    stats = parse_stats(txt)
    json.dump(stats, open(os.path.join(outdir, name + ".json"), "w"))
    print(name, stats)


def parse_stats(txt):
    """The reference's printed statistics a golden keeps (stats.cpp:108-186; a counter that stayed 0 is not printed)."""
    g = lambda pat: int(re.search(pat, txt).group(1)) if re.search(pat, txt) else 0
    stats = {"camera_rays": g(r"Camera rays traced\s+(\d+)"),
             "closest_rays": g(r"Regular ray intersection tests\s+(\d+)"),
             "shadow_rays": g(r"Shadow ray intersection tests\s+(\d+)"),
             "tri_tests": g(r"Ray-triangle intersection tests\s+\d+ /\s+(\d+)"),
             # the integrators' own (path.cpp:45-46, volpath.cpp:45-47): PgCounters since ABI 28
             "paths_zero_radiance": g(r"Zero-radiance paths\s+(\d+) /"), "paths_total": g(r"Zero-radiance paths\s+\d+ /\s+(\d+)"),
             "volume_interactions": g(r"Volume interactions\s+(\d+)"), "surface_interactions": g(r"Surface interactions\s+(\d+)")}
    m = re.search(r"Path length\s+([0-9.]+) avg \[range (\d+) - (\d+)\]", txt)
    if m: stats.update(path_length_avg=m.group(1), path_length_min=int(m.group(2)), path_length_max=int(m.group(3)))  # avg as printed: "%.3f" of sum / count
    return stats


def refresh_stats(paths):
    """`make_golden.py --stats-only [scene.pbrt ...]`: re-run the reference on committed scenes and rewrite only their statistics (the image goes to a
    scratch directory and must equal the committed one)."""
    import tempfile
    ref = os.path.join(HERE, "_ref", "pbrt_oracle")
    for scene in paths:
        name = os.path.basename(scene)[:-5]
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, name + ".pfm")
            nthreads = "1" if name.startswith("filter_") or "maxmindist" in name else "4"
            txt = subprocess.run([ref, "--nthreads", nthreads, "--outfile", out, scene], capture_output=True, text=True, check=True, timeout=1800).stdout
            committed = scene[:-5] + ".pfm"
            if os.path.exists(committed) and open(out, "rb").read() != open(committed, "rb").read():
                sys.exit(f"{name}: the reference's image differs from the committed golden")
        stats = parse_stats(txt)
        old = json.load(open(scene[:-5] + ".json"))
        for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests"):
            if old[k] != stats[k]: sys.exit(f"{name}: {k} {old[k]} -> {stats[k]}")
        json.dump(stats, open(scene[:-5] + ".json", "w"))
        print(name, {k: v for k, v in stats.items() if k.startswith(("path", "volume", "surface"))}, flush=True)


def main():
    os.makedirs(GOLD, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == "--stats-only":
        import glob
        paths = sys.argv[2:] or sorted(glob.glob(os.path.join(GOLD, "*.pbrt")) + glob.glob(os.path.join(ROOT, "tests", "golden_large", "*.pbrt")) +
                                       glob.glob(os.path.join(ROOT, "tests", "golden_large", "config0", "*.pbrt")))
        return refresh_stats([q for q in paths if os.path.exists(q[:-5] + ".json")])  # (included geometry files have no statistics of their own)
    only = sys.argv[1:]
    write_test_images(GOLD)
    write_test_spds(GOLD)
    for name, text in SCENES.items():
        if only and name not in only: continue
        p = os.path.join(GOLD, name + ".pbrt")
        open(p, "w").write(text)
        run(name, p)
    os.makedirs(GOLD_SSS, exist_ok=True)
    for name, text in SSS_SCENES.items():
        if only and name not in only: continue
        p = os.path.join(GOLD_SSS, name + ".pbrt")
        open(p, "w").write(text)
        run(name, p, outdir=GOLD_SSS)
    os.makedirs(GOLD_GRID, exist_ok=True)
    for name, text in GRID_SCENES.items():
        if only and name not in only: continue
        p = os.path.join(GOLD_GRID, name + ".pbrt")
        open(p, "w").write(text)
        run(name, p, outdir=GOLD_GRID)
    # small synthetic heightfield (3 042 + 12 triangles): SAH BVH with real depth
    if not only or "hlbvh_synthetic" in only:  # 3 042 triangles: several treelets, deep LBVH bit splits
        p = os.path.join(GOLD, "hlbvh_synthetic.pbrt")
        gen_synthetic.write_scene(p, n=40, xres=40, yres=24, spp=4, filename="hlbvh_synthetic.pfm")
        txt = open(p).read().replace("WorldBegin", 'Accelerator "bvh" "string splitmethod" "hlbvh"\nWorldBegin', 1).replace("synthetic_n40_mesh", "hlbvh_synthetic_mesh")
        open(p, "w").write(txt)
        run("hlbvh_synthetic", p)
    if not only or "synthetic_n40" in only:
        p = os.path.join(GOLD, "synthetic_n40.pbrt")
        gen_synthetic.write_scene(p, n=40, xres=48, yres=27, spp=4, filename="synthetic_n40.pfm")
        run("synthetic_n40", p)
    # the stand-ins of BASELINE.json configs 4 / 5 (scenes/gen_divergent.py) in miniature: PLY meshes with normals and uv under 100+ object
    # instances, image / bump textures, an alpha-masked card mesh (a plain image map: k_trace's inline mask path), eight materials,
    # environment + area light; "_vol": inside a HomogeneousMedium under volpath.  Assets: tests/golden/div_*.ply / .png / .pfm
    for name, vol in (("divergent_small", False), ("divergent_small_vol", True)):
        if only and name not in only: continue
        p = os.path.join(GOLD, name + ".pbrt")
        gen_divergent.write_scene(p, tris=6000, xres=48, yres=27, spp=4, volumetric=vol, filename=name + ".pfm", n_defs=6, tex_res=64, prefix="div_")
        run(name, p)
    # full-size geometry of BASELINE.json config 3 (999 710 triangles) at a small resolution: the 40 MB scene file is
    # regenerated by scenes/gen_synthetic.py at test time, only the reference's image and statistics are committed
    if not only or "synthetic_1m" in only:
        import tempfile
        large = os.path.join(ROOT, "tests", "golden_large")
        os.makedirs(large, exist_ok=True)
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "synthetic_1m.pbrt")
            gen_synthetic.write_scene(p, n=708, xres=96, yres=54, spp=4, filename="synthetic_1m.pfm")
            run("synthetic_1m", p, outdir=large)
    # BASELINE.json config 0: scenes/killeroo-simple.pbrt as the reference ships it (two Loop-subdivided killeroos, plastic, a SPHERE
    # area light, Halton 8 spp) at the 400x400 BASELINE names, written to a PFM.  The two scene files are input DATA of the reference
    # (a fixture, like a golden vector): copied verbatim but for the film's size and file name; tests/golden_large/config0/
    ref_scenes = "/root/reference/scenes"
    if (not only or "config0" in only) and os.path.exists(os.path.join(ref_scenes, "killeroo-simple.pbrt")):
        import shutil
        d = os.path.join(ROOT, "tests", "golden_large", "config0")
        os.makedirs(os.path.join(d, "geometry"), exist_ok=True)
        shutil.copy(os.path.join(ref_scenes, "geometry", "killeroo.pbrt"), os.path.join(d, "geometry", "killeroo.pbrt"))
        os.chmod(os.path.join(d, "geometry", "killeroo.pbrt"), 0o644)
        txt = open(os.path.join(ref_scenes, "killeroo-simple.pbrt")).read()
        a, b = '"integer xresolution" [700] "integer yresolution" [700]', '"string filename" "killeroo-simple.exr"'
        assert a in txt and b in txt
        txt = txt.replace(a, '"integer xresolution" [400] "integer yresolution" [400]').replace(b, '"string filename" "config0.pfm"')
        open(os.path.join(d, "config0.pbrt"), "w").write(txt)
        run("config0", os.path.join(d, "config0.pbrt"), outdir=d)
    # BASELINE.json config 2 (Cornell box) at a quarter of its resolution and spp
    if not only or "cornell_128" in only:
        large = os.path.join(ROOT, "tests", "golden_large")
        os.makedirs(large, exist_ok=True)
        p = os.path.join(large, "cornell_128.pbrt")
        open(p, "w").write(cornell(128, 128, 64))
        run("cornell_128", p, outdir=large)


if __name__ == "__main__":
    main()
