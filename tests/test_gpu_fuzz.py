"""Differential fuzzing of the HIP path against the CPU oracle: seeded random scenes (triangle soups with degenerate, duplicate
and extreme-scale triangles; every material, light type, filter and light strategy of the closed set; random cameras) and
random rays.  Traversal results, film buffers, stray samples and counters are compared bit for bit, in the reference's shadow-ray order and in the product's default (free) order: 500 scenes."""
import os
import subprocess

import re
import numpy as np
import pytest

from conftest import GOLD, INTEGRATOR_STATS, ROOT

pytestmark = pytest.mark.gpu


def random_scene(seed, res=16, spp=4):
    rng = np.random.default_rng(seed)
    f = lambda a: " ".join(f"{x:.9g}" for x in np.asarray(a, np.float32).ravel())
    eye = rng.normal(size=3) * 3 + np.array([0, 0, -6])
    filt = ['PixelFilter "box"', 'PixelFilter "gaussian"', 'PixelFilter "mitchell" "float xwidth" [ 1.2 ]', 'PixelFilter "triangle"',
            'PixelFilter "box" "float xwidth" [ 0.9 ] "float ywidth" [ 0.3 ]'][seed % 5]
    strat = ["spatial", "power", "uniform"][seed % 3]
    lens = '"float lensradius" [ 0.05 ] "float focaldistance" [ 6 ]' if seed % 4 == 0 else ""
    out = [f"LookAt {f(eye)}  0 0 0  0 1 0", f'Camera "perspective" "float fov" [ {30 + 40 * rng.random():.4g} ] {lens}',
           f'Film "image" "integer xresolution" [ {res} ] "integer yresolution" [ {res + seed % 7} ] "string filename" "fuzz.pfm"',
           f'Sampler "halton" "integer pixelsamples" [ {spp} ]', filt,
           f'Integrator "path" "integer maxdepth" [ {1 + seed % 6} ] "string lightsamplestrategy" "{strat}"', "WorldBegin"]
    mats = ['Material "matte" "rgb Kd" [ %s ]' % f(rng.random(3)), 'Material "plastic" "rgb Kd" [ %s ] "rgb Ks" [ %s ] "float roughness" [ %.4g ]'
            % (f(rng.random(3) * 0.6), f(rng.random(3) * 0.4), 0.02 + 0.5 * rng.random()), 'Material "mirror"',
            'Material "glass" "float index" [ %.4g ]' % (1.1 + rng.random()), 'Material "matte" "rgb Kd" [ 0 0 0 ]']
    if seed % 3 == 1:
        out.append(f'LightSource "point" "point from" [ {f(rng.normal(size=3) * 2 + [0, 4, 0])} ] "rgb I" [ {f(20 + 40 * rng.random(3))} ]')
    if seed % 3 == 2:
        out.append(f'LightSource "spot" "point from" [ {f(rng.normal(size=3) + [0, 5, -2])} ] "point to" [ 0 0 0 ] "rgb I" [ {f(80 + 80 * rng.random(3))} ] "float coneangle" [ 40 ]')
        out.append(f'LightSource "distant" "point from" [ {f(rng.normal(size=3) + [0, 3, 0])} ] "rgb L" [ {f(rng.random(3))} ]')
    # an emissive quad above (every third scene relies on delta lights only)
    if seed % 3 != 1 or seed % 2 == 0:
        two = '"bool twosided" "true"' if seed % 5 == 2 else ""
        out.append(f'AttributeBegin\n AreaLightSource "diffuse" "rgb L" [ {f(5 + 10 * rng.random(3))} ] {two}\n Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] '
                   f'"point P" [ -1 3 -1  1 3 -1  1 3 1  -1 3 1 ]\nAttributeEnd')
    # floor + random soup in a few meshes, with degenerate / duplicate / tiny / huge triangles mixed in
    out.append(mats[0])
    out.append('Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ -6 -2 -6  6 -2 -6  6 -2 6  -6 -2 6 ]')
    for m in range(1 + seed % 4):
        out.append(mats[int(rng.integers(len(mats)))])
        nt = int(rng.integers(1, 40))
        P = rng.normal(size=(nt, 3, 3)) * 0.7 + rng.normal(size=(nt, 1, 3)) * 1.5
        P[::7, 2] = P[::7, 1]                      # zero-area triangles
        if nt > 3: P[3] = P[2]                     # exact duplicate (ties)
        P[::11] *= 1e-3                            # tiny
        if seed % 6 == 0: P[::13] *= 1e3           # huge
        extra = ""
        if m % 2 == 1:
            N = rng.normal(size=(nt * 3, 3)); N[::5] = 0
            extra += f' "normal N" [ {f(N)} ]'
        if m % 3 == 2:
            extra += f' "vector S" [ {f(rng.normal(size=(nt * 3, 3)))} ] "float uv" [ {f(rng.random((nt * 3, 2)))} ]'
        if m == 1 and seed % 2: out.append("ReverseOrientation")
        out.append(f'AttributeBegin\n Rotate {90 * rng.random():.4g} 0 1 0\n Scale 1 {1 - 2 * (seed % 2)} 1\n Shape "trianglemesh" "integer indices" [ {" ".join(map(str, range(3 * nt)))} ] '
                   f'"point P" [ {f(P)} ]{extra}\nAttributeEnd')
    out.append("WorldEnd")
    return "\n".join(out) + "\n"


def random_scene_ext(seed, res=16, spp=4):
    """The wider closed set on top of random_scene: the BxDF-list materials, spheres (geometry, partial, as emitters), the
    infinite light and object instances (of soups, of single spheres, mirrored / non-uniformly scaled)."""
    rng = np.random.default_rng(7000 + seed)
    f = lambda a: " ".join(f"{x:.9g}" for x in np.asarray(a, np.float32).ravel())
    base = random_scene(seed, res, spp)
    head, world = base.split("WorldBegin\n")
    world = world.replace("WorldEnd\n", "")
    mats = ['Material "uber" "rgb Kd" [ %s ] "rgb Kr" [ %s ] "rgb Kt" [ %s ] "rgb opacity" [ %s ] "float uroughness" [ %.4g ] "float vroughness" [ %.4g ]'
            % (f(rng.random(3)), f(rng.random(3) * 0.4), f(rng.random(3) * 0.5), f(0.5 + 0.5 * rng.random(3)), 0.01 + 0.3 * rng.random(), 0.01 + 0.5 * rng.random()),
            'Material "metal" "float uroughness" [ %.4g ] "float vroughness" [ %.4g ]' % (0.005 + 0.2 * rng.random(), 0.005 + 0.4 * rng.random()),
            'Material "metal" "rgb eta" [ %s ] "rgb k" [ %s ] "float roughness" [ %.4g ] "bool remaproughness" "false"' % (f(0.2 + 2 * rng.random(3)), f(1 + 3 * rng.random(3)), 0.05 + 0.3 * rng.random()),
            'Material "substrate" "rgb Kd" [ %s ] "rgb Ks" [ %s ] "float uroughness" [ %.4g ]' % (f(rng.random(3)), f(rng.random(3) * 0.5), 0.01 + 0.4 * rng.random()),
            'Material "translucent" "rgb Kd" [ %s ] "rgb reflect" [ %s ] "rgb transmit" [ %s ]' % (f(rng.random(3)), f(rng.random(3)), f(rng.random(3))),
            'Material "glass" "float uroughness" [ %.4g ] "float vroughness" [ %.4g ] "float index" [ %.4g ]' % (0.02 + 0.3 * rng.random(), 0.02 + 0.3 * rng.random(), 1.2 + 0.6 * rng.random()),
            'NamedMaterial "fzmix"', 'NamedMaterial "fzmix2"']
    img = lambda n: os.path.join(GOLD, n)
    textures = ['Texture "fz_chk" "spectrum" "checkerboard" "float uscale" [ %.4g ] "float vscale" [ %.4g ] "rgb tex1" [ %s ] "rgb tex2" [ %s ]'
                % (1 + 6 * rng.random(), 1 + 6 * rng.random(), f(rng.random(3)), f(rng.random(3))),
                'Texture "fz_img" "spectrum" "imagemap" "string filename" "%s" "float uscale" [ %.4g ] "float vscale" [ %.4g ] "bool trilinear" "%s" "string wrap" "%s"'
                % (img(["img_rgb.png", "img_rle.tga", "img_color.pfm", "img_pal.png"][seed % 4]), 0.5 + 3 * rng.random(), 0.5 + 3 * rng.random(),
                   "true" if seed % 2 else "false", ["repeat", "black", "clamp"][seed % 3]),
                'Texture "fz_f" "float" "checkerboard" "string aamode" "none" "float uscale" [ 3 ] "float vscale" [ 2 ] "float tex1" [ 0 ] "float tex2" [ 1 ]',
                'Texture "fz_fimg" "float" "imagemap" "string filename" "%s" "float scale" [ %.4g ]' % (img("img_gray16.png"), 0.2 + rng.random()),
                'Texture "fz_mix" "spectrum" "mix" "texture tex1" "fz_chk" "texture tex2" "fz_img" "texture amount" "fz_f"',
                'Texture "fz_sph" "spectrum" "checkerboard" "string mapping" "%s" "float uscale" [ 1 ] "rgb tex1" [ 0.9 0.6 0.2 ] "rgb tex2" [ 0.1 0.3 0.8 ]'
                % ["spherical", "cylindrical", "planar"][seed % 3],
                'Texture "fz_scale" "spectrum" "scale" "texture tex1" "fz_mix" "rgb tex2" [ 0.9 0.8 0.7 ]']
    mats += ['Material "matte" "texture Kd" "fz_mix"', 'Material "plastic" "texture Kd" "fz_img" "texture roughness" "fz_fimg" "texture bumpmap" "fz_fimg"',
             'Material "uber" "texture Kd" "fz_chk" "texture opacity" "fz_sph"', 'Material "metal" "texture k" "fz_scale" "float roughness" [ 0.1 ] "texture bumpmap" "fz_f"',
             'Material "matte" "texture Kd" "fz_sph" "texture sigma" "fz_fimg"', 'Material "glass" "texture Kt" "fz_chk"']
    out = [head, "WorldBegin"] + textures + [
           'MakeNamedMaterial "fza" "string type" "plastic" "rgb Kd" [ %s ]' % f(rng.random(3)),
           'MakeNamedMaterial "fzb" "string type" "metal"', 'MakeNamedMaterial "fzc" "string type" "glass"',
           'MakeNamedMaterial "fzmix" "string type" "mix" "string namedmaterial1" "fza" "string namedmaterial2" "fzb" "rgb amount" [ %s ]' % f(rng.random(3)),
           'MakeNamedMaterial "fzmix2" "string type" "mix" "string namedmaterial1" "fzmix" "string namedmaterial2" "fzc"',
           'MakeNamedMaterial "fzt" "string type" "matte" "texture Kd" "fz_img" "texture bumpmap" "fz_f"',
           'MakeNamedMaterial "fzmix3" "string type" "mix" "string namedmaterial1" "fzt" "string namedmaterial2" "fzb" "texture amount" "fz_chk"']
    mats += ['NamedMaterial "fzmix3"']
    if seed % 4 == 2:
        out.append('AttributeBegin\n Rotate %.4g 0 1 0.5\n LightSource "infinite" "string mapname" "%s" "rgb L" [ %s ]\nAttributeEnd'
                   % (360 * rng.random(), img(["img_color.pfm", "img_rgb.png"][seed % 8 // 4]), f(0.3 + rng.random(3))))
    if seed % 4 == 0:
        out.append('AttributeBegin\n Rotate %.4g 1 0.3 0\n LightSource "infinite" "rgb L" [ %s ]\nAttributeEnd' % (360 * rng.random(), f(0.2 + 0.5 * rng.random(3))))
    out.append(world)
    # spheres: plain, partial under a non-uniform transform, and an emitter
    for k in range(1 + seed % 3):
        out.append(mats[int(rng.integers(len(mats)))])
        part = ' "float zmin" [ -0.3 ] "float zmax" [ 0.5 ] "float phimax" [ 240 ]' if (seed + k) % 3 == 0 else ""
        out.append('AttributeBegin\n Translate %s\n Rotate %.4g 1 1 0\n Scale 1 %.4g %.4g\n Shape "sphere" "float radius" [ %.4g ]%s\nAttributeEnd'
                   % (f(rng.normal(size=3) * 1.5), 90 * rng.random(), 0.6 + rng.random(), 0.6 + rng.random(), 0.3 + 0.7 * rng.random(), part))
    # cylinders and disks: geometry (partial, annulus) and, every third scene, an emitting disk / cylinder
    out.append(mats[int(rng.integers(len(mats)))])
    out.append('AttributeBegin\n Translate %s\n Rotate %.4g 1 0 1\n Scale %.4g 1 1\n Shape "cylinder" "float radius" [ %.4g ] "float zmin" [ -0.5 ] "float zmax" [ 0.7 ] "float phimax" [ %.4g ]\nAttributeEnd'
               % (f(rng.normal(size=3) * 1.5), 180 * rng.random(), 0.5 + rng.random(), 0.2 + 0.5 * rng.random(), 120 + 240 * rng.random()))
    out.append('AttributeBegin\n Translate %s\n Rotate %.4g 0 1 1\n Shape "disk" "float radius" [ %.4g ] "float innerradius" [ %.4g ] "float height" [ 0.2 ] "float phimax" [ %.4g ]\nAttributeEnd'
               % (f(rng.normal(size=3) * 1.5), 180 * rng.random(), 0.5 + 0.5 * rng.random(), 0.3 * rng.random(), 150 + 210 * rng.random()))
    if seed % 3 == 0:
        shape = 'Shape "disk" "float radius" [ 0.5 ]' if seed % 2 else 'Shape "cylinder" "float radius" [ 0.1 ] "float zmin" [ -0.6 ] "float zmax" [ 0.6 ]'
        out.append('AttributeBegin\n Translate %s\n Rotate %.4g 1 0 0\n AreaLightSource "diffuse" "rgb L" [ %s ] "bool twosided" "%s"\n %s\nAttributeEnd'
                   % (f(rng.normal(size=3) + [0, 2.2, 0]), 60 + 60 * rng.random(), f(4 + 8 * rng.random(3)), "true" if seed % 2 else "false", shape))
    if seed % 2 == 0:
        out.append('AttributeBegin\n Translate %s\n AreaLightSource "diffuse" "rgb L" [ %s ]\n Shape "sphere" "float radius" [ %.4g ]\nAttributeEnd'
                   % (f(rng.normal(size=3) + [0, 2.5, 0]), f(4 + 8 * rng.random(3)), 0.05 + 0.4 * rng.random()))
    # object definitions: a soup with its own BVH, and a lone sphere; instanced a few times
    nt = int(rng.integers(2, 30))
    P = rng.normal(size=(nt, 3, 3)) * 0.4
    out.append('%s\nShape "trianglemesh" "texture alpha" "fz_f" "texture shadowalpha" "fz_fimg" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ %s ] "float uv" [ 0 0 2 0 2 2 0 2 ]'
               % (mats[int(rng.integers(len(mats)))], f(rng.normal(size=(4, 3)) * 1.5)))
    out.append('ObjectBegin "soup"\n %s\n Shape "trianglemesh" "texture alpha" "fz_f" "integer indices" [ %s ] "point P" [ %s ]\n %s\n Translate 0.5 0 0\n Shape "sphere" "float radius" [ 0.3 ]\nObjectEnd'
               % (mats[int(rng.integers(len(mats)))], " ".join(map(str, range(3 * nt))), f(P), mats[int(rng.integers(len(mats)))]))
    out.append('ObjectBegin "ball"\n %s\n Shape "sphere" "float radius" [ 0.4 ]\nObjectEnd' % mats[int(rng.integers(len(mats)))])
    for k in range(2 + seed % 3):
        sx = -1 if (seed + k) % 2 else 1
        out.append('AttributeBegin\n Translate %s\n Rotate %.4g 0 1 1\n Scale %.4g %.4g %.4g\n ObjectInstance "%s"\nAttributeEnd'
                   % (f(rng.normal(size=3) * 2), 360 * rng.random(), sx * (0.5 + rng.random()), 0.5 + rng.random(), 0.5 + rng.random(), "soup" if k % 2 == 0 else "ball"))
    if seed % 5 == 0: out.append('ObjectInstance "soup"')  # identity transform
    out.append("WorldEnd")
    return "\n".join(out) + "\n"


def random_scene_vol(seed, res=16, spp=4):
    """random_scene_ext under the VolPathIntegrator: homogeneous media (isotropic, forward / backward scattering, one absorbing
    only), the camera inside a medium in every other scene, media bounded by "none" surfaces (a sphere, loose triangles, a
    sphere inside an instanced object), medium interfaces on ordinary surfaces, and the cone / paraboloid / hyperboloid shapes."""
    rng = np.random.default_rng(9000 + seed)
    f = lambda a: " ".join(f"{x:.9g}" for x in np.asarray(a, np.float32).ravel())
    text = random_scene_ext(seed, res, spp)
    media = ['MakeNamedMedium "m0" "string type" "homogeneous" "rgb sigma_a" [ %s ] "rgb sigma_s" [ %s ] "float g" [ %.4g ]' % (f(0.02 * rng.random(3)), f(0.15 * rng.random(3)), 1.6 * rng.random() - 0.8),
             'MakeNamedMedium "m1" "string type" "homogeneous" "rgb sigma_a" [ %s ] "rgb sigma_s" [ %s ] "float g" [ 0.0002 ] "float scale" [ %.4g ]' % (f(rng.random(3)), f(2 * rng.random(3)), 0.2 + rng.random()),
             'MakeNamedMedium "m2" "string type" "homogeneous" "rgb sigma_a" [ %s ] "rgb sigma_s" [ 0 0 0 ]' % f(0.5 * rng.random(3)),
             'MakeNamedMedium "m3" "string type" "homogeneous" "float g" [ %.4g ] "float scale" [ 0.1 ]' % (-0.9 * rng.random())]
    cam = 'MediumInterface "" "m0"\n' if seed % 2 == 0 else ""
    text = text.replace("Camera ", "\n".join(media) + "\n" + cam + "Camera ", 1)
    if seed % 4 == 1: text = text.replace('Sampler "halton"', 'Sampler "sobol"', 1)
    text = text.replace('Integrator "path"', 'Integrator "volpath"' + (' "float rrthreshold" [ 0.6 ]' if seed % 3 == 0 else ""), 1)
    extra = ['AttributeBegin\n MediumInterface "m1" "%s"\n Material "none"\n Translate %s\n Shape "sphere" "float radius" [ %.4g ]\nAttributeEnd'
             % ("m0" if seed % 2 == 0 else "", f(rng.normal(size=3)), 0.5 + rng.random()),
             'AttributeBegin\n MediumInterface "m2" "m3"\n Material ""\n Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ %s ]\nAttributeEnd' % f(rng.normal(size=(4, 3)) * 2),
             'AttributeBegin\n MediumInterface "m3" ""\n Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ %s ]\nAttributeEnd' % f(rng.normal(size=(3, 3)) * 2),
             'ObjectBegin "cloud"\n MediumInterface "m1" "m2"\n Material "none"\n Shape "sphere" "float radius" [ 0.5 ]\n Material "matte"\n Translate 0 0.6 0\n Shape "disk" "float radius" [ 0.4 ]\nObjectEnd',
             'AttributeBegin\n Translate %s\n Scale %.4g 1 -1\n ObjectInstance "cloud"\nAttributeEnd' % (f(rng.normal(size=3) * 1.5), 0.5 + rng.random())]
    # the intersect-only quadrics: cone, paraboloid, hyperboloid (full / partial, non-uniform transforms, every material class)
    extra += ['AttributeBegin\n Translate %s\n Rotate %.4g 1 0.2 0\n Scale 1 %.4g 1\n Material "%s"\n Shape "cone" "float radius" [ %.4g ] "float height" [ %.4g ] "float phimax" [ %.4g ]\nAttributeEnd'
              % (f(rng.normal(size=3) * 1.5), 360 * rng.random(), 0.5 + rng.random(), ["matte", "glass", "mirror", "plastic"][seed % 4], 0.3 + 0.5 * rng.random(), 0.5 + rng.random(), 150 + 210 * rng.random()),
              'AttributeBegin\n Translate %s\n Rotate %.4g 0 1 0.3\n Material "%s"\n Shape "paraboloid" "float radius" [ %.4g ] "float zmin" [ %.4g ] "float zmax" [ %.4g ]\nAttributeEnd'
              % (f(rng.normal(size=3) * 1.5), 360 * rng.random(), ["glass", "mirror", "matte", "uber"][seed % 4], 0.3 + 0.5 * rng.random(), 0.2 * rng.random(), 0.5 + 0.5 * rng.random()),
              'AttributeBegin\n Translate %s\n Rotate %.4g 1 1 0\n ReverseOrientation\n Material "%s"\n Shape "hyperboloid" "point p1" [ %s ] "point p2" [ %s ] "float phimax" [ %.4g ]\nAttributeEnd'
              % (f(rng.normal(size=3) * 1.5), 360 * rng.random(), ["mirror", "matte", "glass", "substrate"][seed % 4], f(rng.normal(size=3) * 0.4 + [0.5, 0, -0.4]), f(rng.normal(size=3) * 0.4 + [0.3, 0.2, 0.5]),
                 200 + 160 * rng.random())]
    # Perlin-noise textures (fbm, wrinkled, windy, marble, dots) as reflectances, float parameters and bump maps
    extra += ['TransformBegin\n Scale %.4g %.4g %.4g\n Texture "vz_fbm" "float" "fbm" "integer octaves" [ %d ] "float roughness" [ %.4g ]\n Texture "vz_wr" "spectrum" "wrinkled"\n'
              ' Texture "vz_windy" "float" "windy"\n Texture "vz_marble" "spectrum" "marble" "float scale" [ %.4g ] "float variation" [ %.4g ]\nTransformEnd'
              % (1 + 3 * rng.random(), 1 + 3 * rng.random(), 1 + 3 * rng.random(), 2 + seed % 6, 0.3 + 0.4 * rng.random(), 0.5 + 2 * rng.random(), 0.1 + 0.4 * rng.random()),
              'Texture "vz_dots" "spectrum" "dots" "float uscale" [ %.4g ] "float vscale" [ %.4g ] "texture inside" "vz_marble" "rgb outside" [ %s ]'
              % (2 + 6 * rng.random(), 2 + 6 * rng.random(), f(rng.random(3))),
              'AttributeBegin\n Translate %s\n Material "plastic" "texture Kd" "vz_dots" "texture roughness" "vz_windy" "texture bumpmap" "vz_fbm"\n Shape "sphere" "float radius" [ %.4g ]\nAttributeEnd'
              % (f(rng.normal(size=3) * 1.5), 0.4 + 0.5 * rng.random()),
              'Material "matte" "texture Kd" "vz_wr" "texture sigma" "vz_windy"\nShape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ %s ] "float uv" [ 0 0 3 0 3 3 0 3 ]'
              % f(rng.normal(size=(4, 3)) * 1.5),
              'Material "uber" "texture Kd" "vz_marble" "texture opacity" "vz_wr" "texture bumpmap" "vz_windy"\nShape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ %s ]'
              % f(rng.normal(size=(3, 3)) * 1.5)]
    return text.replace("WorldEnd\n", "\n".join(extra) + "\nWorldEnd\n")


def random_scene_sss_grid(seed, features="both"):
    """The volumetric / extended random scenes with two more features mixed in (features = "both", "sss" or "grid"; the device takes one of them per scene): one of the media becomes a GridDensityMedium
    (random dimensions, densities with empty voxels, box and CTM), and the glass / mirror / plastic / matte materials become
    subsurface / kdsubsurface materials with random coefficients (smooth and rough, presets, textured Kd)."""
    rng = np.random.default_rng(77000 + seed)
    f = lambda a: " ".join(f"{x:.9g}" for x in np.asarray(a, np.float32).ravel())
    text = random_scene_vol(seed) if seed % 3 else random_scene_ext(seed)
    if features != "sss" and 'MakeNamedMedium "m1"' in text:
        line = [l for l in text.splitlines() if l.startswith('MakeNamedMedium "m1"')][0]
        nx, ny, nz = (int(v) for v in rng.integers(1, 6, size=3))
        den = rng.random(nx * ny * nz) * (rng.random(nx * ny * nz) > 0.25) * (0.5 + 3 * rng.random())
        if den.max() == 0: den[0] = 1
        sa, ss = 0.3 * rng.random(), 0.5 + 3 * rng.random()
        p0 = rng.normal(size=3) - 1.0
        grid = ('MakeNamedMedium "m1" "string type" "heterogeneous" "rgb sigma_a" [ %s ] "rgb sigma_s" [ %s ] "float g" [ %.4g ] "integer nx" [ %d ] "integer ny" [ %d ] "integer nz" [ %d ] '
                '"point p0" [ %s ] "point p1" [ %s ] "float density" [ %s ]' % (f([sa] * 3), f([ss] * 3), 1.4 * rng.random() - 0.7, nx, ny, nz, f(p0), f(p0 + 1 + 2 * rng.random(3)), f(den)))
        text = text.replace(line, grid, 1)
    def sss():
        k = rng.integers(0, 4)
        rough = ' "float uroughness" [ %.4g ] "float vroughness" [ %.4g ]' % (0.3 * rng.random(), 0.3 * rng.random()) if rng.random() < 0.4 else ""
        if k == 0: return 'Material "subsurface" "rgb sigma_a" [ %s ] "rgb sigma_s" [ %s ] "float eta" [ %.4g ] "float g" [ %.4g ]%s' % (f(rng.random(3)), f(0.2 + 5 * rng.random(3)), 1.1 + 0.5 * rng.random(), 1.2 * rng.random() - 0.5, rough)
        if k == 1: return 'Material "kdsubsurface" "rgb Kd" [ %s ] "rgb mfp" [ %s ] "float eta" [ %.4g ]%s' % (f(rng.random(3)), f(0.05 + rng.random(3)), 1.2 + 0.4 * rng.random(), rough)
        if k == 2: return 'Material "subsurface" "string name" "%s" "float scale" [ %.4g ]%s' % (["Skin1", "Marble", "Ketchup", "Wholemilk"][seed % 4], 0.5 + 20 * rng.random(), rough)
        return 'Material "kdsubsurface" "rgb Kd" [ %s ] "rgb mfp" [ %s ] "rgb Kr" [ %s ] "float scale" [ %.4g ]' % (f(rng.random(3)), f(0.2 + rng.random(3)), f(rng.random(3)), 0.5 + rng.random())
    out = []
    for l in text.splitlines():
        st = l.strip()
        if features != "grid" and (st.startswith('Material "glass"') or st.startswith('Material "mirror"') or st.startswith('Material "plastic"') or st.startswith('Material "matte"')) and "texture" not in st and rng.random() < 0.6:
            l = l[:len(l) - len(l.lstrip())] + sss()
        out.append(l)
    return "\n".join(out) + "\n"


def random_scene_pixel_sampler(seed):
    """The plain / extended random scenes under a PixelSampler whose "dimensions" cover every draw of a path (2 + 3 maxdepth): the form in
    which the device generates all pixels' sample arrays ahead and traces one wavefront (k_ts_start_tile, DESIGN.md section 4 "Samplers")."""
    import re
    text = random_scene_ext(seed) if seed % 2 else random_scene(seed)
    depth = int(re.search(r'"integer maxdepth" \[ (\d+) \]', text).group(1))
    dims = 2 + 3 * depth
    spec = ('"stratified" "integer xsamples" [ 2 ] "integer ysamples" [ 2 ] "bool jitter" "%s" "integer dimensions" [ %d ]' % ("true" if seed % 4 < 2 else "false", dims),
            '"02sequence" "integer pixelsamples" [ 4 ] "integer dimensions" [ %d ]' % dims,
            '"maxmindist" "integer pixelsamples" [ 4 ] "integer dimensions" [ %d ]' % dims)[seed % 3]
    out, n = re.subn(r'Sampler "halton" "integer pixelsamples" \[ 4 \]', "Sampler " + spec, text)
    assert n == 1
    return out


def random_scene_moving_camera(seed):
    """One of the random scenes above under a camera that moves (AnimatedTransform CameraToWorld, interpolated at every camera ray's
    time): a random end-of-motion transform -- translation, a rotation that is sometimes tiny (Slerp's normalised-lerp branch), sometimes a
    scale --, TransformTimes that lie inside, across or outside the shutter interval, and every sampler family the front end has."""
    rng = np.random.default_rng(1000 + seed)
    gen = (random_scene, random_scene_ext, random_scene_vol, random_scene_pixel_sampler)[seed % 4]
    text = gen(seed)
    motion = ["Translate %.6g %.6g %.6g" % tuple(rng.normal(size=3) * 0.6)]
    if seed % 3 != 2:
        angle = rng.choice([0.7, 6.0, 25.0, 70.0]) * (1 if rng.random() < 0.5 else -1)
        motion.append("Rotate %.6g %.6g %.6g %.6g" % (angle, *(rng.normal(size=3) + np.array([0, 1e-3, 0]))))
    if seed % 5 == 0:
        motion.append("Scale %.6g %.6g %.6g" % tuple(1 + 0.2 * rng.random(3)))
    t0, t1 = sorted(rng.random(2) * 1.4 - 0.2)
    times = "TransformTimes %.6g %.6g\n" % (t0, max(t1, t0 + 1e-3)) if seed % 2 else ""
    shutter = ' "float shutteropen" [ %.6g ] "float shutterclose" [ %.6g ]' % (0.1 * (seed % 3), 1 - 0.15 * (seed % 4)) if seed % 3 else ""
    assert "Camera " in text
    head, rest = text.split("Camera ", 1)
    line, tail = rest.split("\n", 1)
    return head + times + "ActiveTransform EndTime\n" + "\n".join(motion) + "\nActiveTransform All\nCamera " + line + shutter + "\n" + tail


def add_motion(text, rng, spin=False):
    """Put attribute blocks of `text` that hold shapes or object instances (no lights, no mirroring; `spin`: rotations of any angle -- without it
    rotations below the hasRotation threshold only) under an end-of-motion transform."""
    chunks = text.split("AttributeBegin\n")
    moved = 0
    for k in range(1, len(chunks)):
        body = chunks[k].split("AttributeEnd", 1)[0]
        if "LightSource" in body or re.search(r"Scale[^\n]*-", body) or not ('Shape "' in body or "ObjectInstance" in body): continue
        if moved and rng.random() < 0.35: continue
        motion = [" ActiveTransform EndTime", " Translate %.6g %.6g %.6g" % tuple(rng.normal(size=3) * 0.5)]
        r = rng.random()
        if r < 0.4: motion.append(" Scale %.6g %.6g %.6g" % tuple(0.8 + 0.5 * rng.random(3)))
        elif r < 0.6: motion.append(" Rotate %.6g %.6g %.6g %.6g" % (rng.uniform(-1.5, 1.5), *(rng.normal(size=3) + np.array([0, 1e-3, 0]))))
        if spin and rng.random() < 0.8:  # (before or after the translation / scale: the end transform's decomposition sees a shear in the second case)
            turn = " Rotate %.6g %.6g %.6g %.6g" % (rng.uniform(-175, 175) if rng.random() < 0.7 else rng.uniform(-6, 6), *(rng.normal(size=3) + np.array([0, 1e-3, 0])))
            motion.insert(1 if rng.random() < 0.5 else len(motion), turn)
        motion.append(" ActiveTransform All")
        lines = chunks[k].split("\n")
        at = next(i for i, l in enumerate(lines) if l.lstrip().startswith(("Shape ", "ObjectInstance ")))
        chunks[k] = "\n".join(lines[:at] + motion + lines[at:])
        moved += 1
    text = "AttributeBegin\n".join(chunks)
    if moved == 0:  # (random_scene_pixel_sampler's scenes have no attribute blocks: move a sphere of our own)
        text = text.replace("WorldEnd", 'AttributeBegin\n Translate 0.3 0.2 0\n ActiveTransform EndTime\n Translate 0.5 -0.3 0.4\n ActiveTransform All\n Material "plastic"\n'
                            ' Shape "sphere" "float radius" [ 0.5 ]\nAttributeEnd\nWorldEnd')
    return text


def random_scene_motion(seed):
    """One of the random scenes above with MOVING shapes and object instances (TransformedPrimitive over an AnimatedTransform, interpolated at
    every ray's time: primitive.cpp:76-103): an end-of-motion translation, sometimes a scale, sometimes a rotation small enough to count as none
    (Dot(R[0], R[1]) >= 0.9995), on quadrics, on instances of a BVH object and of a lone sphere; TransformTimes inside, across or outside the
    shutter, every sampler family, volpath; every fourth scene under a moving camera as well.  (Mirrored instances are left still: the reference
    takes the quaternion of an improper rotation -- not a unit one -- and slerps it; the front end refuses such motions.)"""
    rng = np.random.default_rng(2000 + seed)
    gen = (random_scene_ext, random_scene_vol, random_scene_ext, random_scene_pixel_sampler)[seed % 4]
    text = random_scene_moving_camera(seed) if seed % 4 == 2 else gen(seed)
    text = add_motion(text, rng)
    if seed % 4 != 2 and seed % 3 == 1:
        t0, t1 = sorted(rng.random(2) * 1.4 - 0.2)
        text = text.replace("WorldBegin", "TransformTimes %.6g %.6g\nWorldBegin" % (t0, max(t1, t0 + 1e-3)), 1)
    return text


@pytest.mark.parametrize("seed", range(48))
def test_random_scene_with_moving_shapes_and_instances(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_motion(seed), seed)


def random_scene_motion_sss_grid(seed):
    """Moving shapes and instances beside subsurface materials (the probe chains of Sample_Sp run at the path's time, the exit vertex comes back
    through the moving instance's interpolated transform) or a GridDensityMedium (ratio tracking's transmittance rays, both shading phases)."""
    rng = np.random.default_rng(3000 + seed)
    return add_motion(random_scene_sss_grid(seed, ("sss", "grid", "both")[seed % 3] if seed % 3 != 2 else "sss"), rng)


def random_scene_rotating_motion(seed):
    """random_scene_motion's scenes with motions that ROTATE (AnimatedTransform::hasRotation): every ray slerps the two rotations at its own time
    inside TransformedPrimitive::Intersect[P]; the primitives' boxes in the top-level BVH are MotionBounds' (host/motion_bounds.cpp, pinned
    against the reference in tests/test_motion_bounds.py).  Mirrored blocks stay still: the reference slerps the non-unit quaternion of an improper
    rotation, and its own renders of such motions abort or do not terminate; the front end refuses them.  Every third scene with subsurface
    materials or a grid medium."""
    rng = np.random.default_rng(4000 + seed)
    if seed % 3 == 2:
        return add_motion(random_scene_sss_grid(seed, ("sss", "grid")[seed % 2]), rng, spin=True)
    gen = (random_scene_ext, random_scene_vol, random_scene_ext, random_scene_pixel_sampler)[seed % 4]
    text = random_scene_moving_camera(seed) if seed % 4 == 2 else gen(seed)
    text = add_motion(text, rng, spin=True)
    if seed % 4 != 2 and seed % 5 == 1:
        t0, t1 = sorted(rng.random(2) * 1.4 - 0.2)
        text = text.replace("WorldBegin", "TransformTimes %.6g %.6g\nWorldBegin" % (t0, max(t1, t0 + 1e-3)), 1)
    return text


@pytest.mark.parametrize("seed", range(36))
def test_random_scene_with_rotating_shapes_and_instances(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_rotating_motion(seed), seed)


def add_nested_motion(text, rng, spin=False):
    """Put shapes INSIDE the object definitions of `text` under an end-of-motion transform: pbrtShape then adds a TransformedPrimitive to the definition
    (api.cpp:1386-1419) and every ObjectInstance of it is a TransformedPrimitive around a TransformedPrimitive (PG_PRIM_INSTANCE inside an object's run,
    ABI 29).  Before a definition's first shape every shape of it moves (the CTM keeps its two ends until ObjectEnd), before a later one the earlier stay still."""
    parts = text.split('ObjectBegin "')
    for k in range(1, len(parts)):
        body, rest = parts[k].split("ObjectEnd", 1)
        lines = body.split("\n")
        shapes = [i for i, l in enumerate(lines) if l.lstrip().startswith("Shape ")]
        if not shapes or (k > 1 and rng.random() < 0.25): continue
        at = shapes[int(rng.integers(len(shapes)))]
        motion = [" ActiveTransform EndTime", " Translate %.6g %.6g %.6g" % tuple(rng.normal(size=3) * 0.4)]
        if rng.random() < 0.4: motion.append(" Scale %.6g %.6g %.6g" % tuple(0.8 + 0.5 * rng.random(3)))
        if spin and rng.random() < 0.8: motion.insert(1 if rng.random() < 0.5 else len(motion), " Rotate %.6g %.6g %.6g %.6g" % (rng.uniform(-175, 175), *(rng.normal(size=3) + np.array([0, 1e-3, 0]))))
        motion.append(" ActiveTransform All")
        parts[k] = "\n".join(lines[:at] + motion + lines[at:]) + "ObjectEnd" + rest
    return 'ObjectBegin "'.join(parts)


def random_scene_nested_motion(seed):
    """The random scenes above -- extended, volumetric, with moving / rotating instances and shapes, under a moving camera, every sampler family -- with
    moving shapes inside their object definitions as well: a soup with alpha masks and a sphere, a lone sphere, a medium boundary."""
    rng = np.random.default_rng(6000 + seed)
    base = (random_scene_ext, random_scene_vol, random_scene_motion, random_scene_rotating_motion)[seed % 4]
    text = base(seed if seed % 4 < 2 or seed % 3 != 2 else seed + 1)  # (random_scene_rotating_motion's every third scene has BSSRDF materials: refused beside nested motion)
    assert "subsurface" not in text
    return add_nested_motion(text, rng, spin=seed % 2 == 1)


@pytest.mark.parametrize("seed", range(40))
def test_random_scene_with_moving_shapes_inside_object_definitions(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_nested_motion(seed), seed)


def random_scene_nested_motion_sss_grid(seed):
    """Moving shapes inside object definitions beside subsurface materials (the probe chains' hits lie under two transforms: k_sss_probe<., ., NEST>, k_sss_exit), a
    GridDensityMedium (both shading phases around the transmittance rays), or both; every other scene with moving / rotating instances around them."""
    rng = np.random.default_rng(7000 + seed)
    text = random_scene_sss_grid(seed, ("sss", "grid", "both", "sss")[seed % 4])
    if seed % 2: text = add_motion(text, rng, spin=seed % 4 == 3)
    return add_nested_motion(text, rng, spin=seed % 3 == 0)


@pytest.mark.parametrize("seed", range(24))
def test_random_sss_or_grid_scene_with_moving_shapes_inside_object_definitions(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_nested_motion_sss_grid(seed), seed)


@pytest.mark.parametrize("seed", range(24))
def test_random_sss_or_grid_scene_with_moving_shapes(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_motion_sss_grid(seed), seed)


@pytest.mark.parametrize("seed", range(48))
def test_random_scene_with_a_moving_camera(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_moving_camera(seed), seed)


@pytest.mark.parametrize("seed", range(60))
def test_random_scene_under_a_batched_pixel_sampler(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_pixel_sampler(seed), seed)


@pytest.mark.parametrize("seed", range(110))
def test_random_scene_film_matches_oracle(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene(seed), seed)


@pytest.mark.parametrize("seed", range(110))
def test_random_extended_scene_film_matches_oracle(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_ext(seed), seed)


@pytest.mark.parametrize("seed", range(100))
def test_random_volumetric_scene_film_matches_oracle(gpu, oracle, seed):
    check_scene(gpu, oracle, random_scene_vol(seed), seed)


@pytest.mark.parametrize("seed", range(120))
def test_random_sss_or_grid_scene_film_matches_oracle(gpu, oracle, seed):
    """Subsurface materials (odd seeds) or a GridDensityMedium (even seeds) mixed into the random scenes: BSSRDF probe chains through
    random soups, ratio tracking through random grids -- against the oracle, bit for bit."""
    check_scene(gpu, oracle, random_scene_sss_grid(seed, "sss" if seed % 2 else "grid"), seed)


@pytest.mark.parametrize("seed", range(48))
def test_random_sss_and_grid_scene_film_matches_oracle(gpu, oracle, seed):
    """BOTH in one scene (round 6; the device refused the pair before): VolPathIntegrator::Li's loop with medium sampling, ratio tracking and the BSSRDF branch
    (volpath.cpp:76-176) -- the entry vertex in two shading phases, then the probe chains, then the exit vertex in two phases of its own (k_sss_exit)."""
    check_scene(gpu, oracle, random_scene_sss_grid(seed, "both"), seed)


@pytest.mark.parametrize("seed", [59, 68])
def test_grid_scene_keeps_the_specular_flag_across_a_material_less_surface(gpu, oracle, seed):
    """Two scenes of a 300-scene sweep on the GPU (tools/fuzz_emulated_device.py): a path leaves a specular surface, crosses the
    material-less boundary of a medium and hits an area light -- `specularBounce` survives the boundary (path.cpp:107-113: `continue`),
    so the light's emission counts.  The second shading phase of grid-medium scenes had written the state of vertices that the first
    phase finished back without their flags."""
    check_scene(gpu, oracle, random_scene_sss_grid(seed, "grid"), seed)


def check_scene(gpu, oracle, text, seed):
    scene = gpu.HostScene(text=text)
    gs = gpu.GpuScene(scene.desc)  # (tests/conftest.py: shadow rays in the reference's visiting order)
    rd = scene.render_desc()
    film, strays = gs.render(rd)
    cn = gs.counters()
    # the CPU restatement (pinned against the reference binary): film, stray samples and every counter bit for bit
    ofilm, ostrays, ocn = oracle.render(scene.desc, rd)
    assert np.array_equal(film["weight"], ofilm["weight"])
    assert np.array_equal(film["rgb"], ofilm["rgb"]), f"{(film['rgb'] != ofilm['rgb']).any(axis=1).sum()} pixels differ from the oracle"
    key = lambda s: np.lexsort((s["src_px"], s["src_py"], s["px"], s["py"]))
    a, b = strays[key(strays)], ostrays[key(ostrays)]
    assert len(a) == len(b)
    for f in ("px", "py", "src_px", "src_py", "weight", "rgb"):
        assert np.array_equal(a[f], b[f]), f
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests", "node_visits") + INTEGRATOR_STATS:
        assert cn[k] == ocn[k], (k, cn[k], ocn[k])
    # the PRODUCT'S DEFAULT: shadow rays in free order (k_trace<2, .>: the child the ray enters first).  Same occlusion answers, so the
    # same film, strays and ray counts; only the two statistics that count what a traversal read (triangle tests, node visits) may differ.
    saved = os.environ.pop("PG_ANYHIT_ORDER", None)
    try:
        gf = gpu.GpuScene(scene.desc)
    finally:
        if saved is not None: os.environ["PG_ANYHIT_ORDER"] = saved
    ffilm, fstrays = gf.render(rd)
    fcn = gf.counters()
    gf.close()
    assert np.array_equal(ffilm["rgb"], ofilm["rgb"]) and np.array_equal(ffilm["weight"], ofilm["weight"]), "free-order shadow rays changed the film"
    c = fstrays[key(fstrays)]
    assert len(c) == len(b)
    for f in ("px", "py", "src_px", "src_py", "weight", "rgb"):
        assert np.array_equal(c[f], b[f]), f
    for k in ("camera_rays", "closest_rays", "shadow_rays") + INTEGRATOR_STATS:
        assert fcn[k] == ocn[k], (k, fcn[k], ocn[k])
    # rays through the same soup: bit-exact, including the reference's counters
    rng = np.random.default_rng(1000 + seed)
    n = 4096
    nodes = scene.nodes()
    lo, hi = nodes["bmin"][0], nodes["bmax"][0]
    o = (lo + (hi - lo) * (rng.random((n, 3)) * 1.4 - 0.2)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:64, seed % 3] = 0  # axis-parallel directions: infinite inverse components
    tmax = np.where(rng.random(n) < 0.3, rng.random(n) * 3, np.inf).astype(np.float32)
    gs.counters_reset()
    prim, t, bary = gs.intersect(o, d, tmax)
    op, ot, ob, on = oracle.intersect(scene.desc, o, d, tmax)
    assert np.array_equal(prim, op) and np.array_equal(t, ot) and np.array_equal(bary, ob)
    c2 = gs.counters()
    assert c2["closest_node_visits"] == on["node_visits"] and c2["closest_tri_tests"] == on["tri_tests"]
    occ = gs.intersect_p(o, d, tmax)
    oocc, on2 = oracle.intersect_p(scene.desc, o, d, tmax)
    assert np.array_equal(occ, oocc)
    c3 = gs.counters()
    assert c3["shadow_node_visits"] == on2["node_visits"] and c3["shadow_tri_tests"] == on2["tri_tests"]
    gs.close()


def test_cli_end_to_end(gpu, tmp_path):
    """pbrt_amd scene.pbrt --outfile x.pfm, the drop-in for `pbrt scene.pbrt --outfile x.pfm`."""
    exe = os.path.join(ROOT, "pbrt-v3_amd", "pbrt_amd")
    out = tmp_path / "cli.pfm"
    for name in ("cornell_32", "filter_gaussian", "cornell_ply"):
        r = subprocess.run([exe, "--quiet", "--outfile", str(out), os.path.join(GOLD, name + ".pbrt")], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr
        img, ref = gpu.read_pfm(str(out)), gpu.read_pfm(os.path.join(GOLD, name + ".pfm"))
        assert img.shape == ref.shape
        assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), name  # the reference binary's file, bit for bit
