"""Film positions that round UP onto the next pixel (found at the full size of BASELINE config 4: 76 of its 36 864 window pixels
differed from the reference binary's in their last bits).

GetCameraSample forms `pFilm = (Float)pixel + u` (sampler.cpp:46-52).  1920 pixels wide, floats from 1024 on are 2^-13 apart: once
the Halton sampler's u0 = RadicalInverse(0, index >> 7) reaches 1 - 2^-14 -- sample index 2 097 024, the 68th sample of a pixel --
`1050 + u0` IS 1051.0, the box filter's footprint covers pixels 1050 and 1051, and FilmTile::AddSample (film.h:121-161) adds the
sample to pixel 1051 BEFORE that pixel's own samples.  The fast film path (filter_general = 0) appends such samples after the
pixel's own sum: the same numbers in another order, a few ulps apart.  Frames in which it can happen therefore take the gathering
path (pgh_box_filter_needs_gather, include/pbrt_gpu.h), whose summation order is the reference's for any footprint;
tests/golden/filter_box_round_up.* (reference binary) is such a frame in miniature."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLD

FRAME = ('LookAt 0 0 5  0 0 0  0 1 0\nCamera "perspective" "float fov" [ 40 ]\n'
         'Film "image" "integer xresolution" [ {xres} ] "integer yresolution" [ {yres} ] "string filename" "x.pfm"\n'
         'Sampler "{sampler}" "integer pixelsamples" [ {spp} ]\nPixelFilter "box"\nIntegrator "path"\nWorldBegin\n'
         'LightSource "distant" "rgb L" [ 1 1 1 ]\nMaterial "matte"\nShape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ 0 0 0 1 0 0 0 1 0 ]\nWorldEnd\n')


def rd_of(pkg, xres, yres, spp, sampler="halton"):
    return pkg.HostScene(text=FRAME.format(xres=xres, yres=yres, spp=spp, sampler=sampler)).render_desc()


def test_which_frames_take_the_gathering_path(pkg):
    # BASELINE's sizes under Halton: config 0, config 2, config 3 stay on the fast path; 1920 wide, the 68th sample is the first that rounds up
    for xres, yres, spp, want in ((400, 400, 8, 0), (512, 512, 256, 0), (1920, 1080, 64, 0), (1920, 1080, 67, 0), (1920, 1080, 68, 1),
                                  (1920, 1080, 128, 1), (1920, 1080, 256, 1), (64, 64, 1024, 0)):
        rd = rd_of(pkg, xres, yres, spp)
        assert rd.filter_general == want, (xres, yres, spp)
        assert rd.tile_pixels == (18 * 18 if want else 256) and list(rd.tile_halo) == ([1, 1, 1, 1] if want else [0, 0, 0, 0])
    # numbers up to OneMinusEpsilon (Sobol', the PixelSamplers' RNG) round up from pixel 1 on
    for sampler in ("sobol", "random", "02sequence"):
        assert rd_of(pkg, 64, 64, 4, sampler).filter_general == 1, sampler
    assert rd_of(pkg, 1, 1, 4, "random").filter_general == 0
    # a wider filter gathers anyway
    assert pkg.HostScene(text=FRAME.format(xres=32, yres=32, spp=4, sampler="halton").replace('PixelFilter "box"', 'PixelFilter "gaussian"')).render_desc().filter_general == 1


def test_fast_film_path_on_such_a_frame_differs_from_the_reference(pkg, oracle):
    """The golden frame through the gathering path is the reference's image bit for bit (tests/test_oracle_vs_reference.py and the
    GPU golden tests hold that); through the fast path, forced, it is not -- which is what the rule is for, and what shows that the
    golden exercises it."""
    scene = pkg.HostScene(os.path.join(GOLD, "filter_box_round_up.pbrt"))
    ref = pkg.read_pfm(os.path.join(GOLD, "filter_box_round_up.pfm"))
    rd = scene.render_desc()
    assert rd.filter_general == 1 and list(rd.cropped_pixel_bounds) == [1024, 468, 1152, 484]
    img, _ = oracle.render_image(scene)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    fast = scene.render_desc()
    fast.filter_general = 0
    fast.tile_pixels = 256
    for i in range(4): fast.tile_halo[i] = 0
    film, strays, _ = oracle.render(scene.desc, fast)
    assert len(strays) > 0
    scene.film_clear(); scene.film_merge(fast, film, strays)
    img_fast = scene.film_image()
    differing = int((img_fast.view(np.uint32) != ref.view(np.uint32)).any(axis=2).sum())
    assert 0 < differing < 200
    assert np.abs(img_fast - ref).max() < 1e-6  # the same samples in another order


@pytest.mark.gpu
def test_device_refuses_the_fast_path_where_it_would_be_inexact(gpu):
    scene = gpu.HostScene(os.path.join(GOLD, "filter_box_round_up.pbrt"))
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    assert gpu.gpu_lib().pg_box_filter_needs_gather(C.byref(rd)) == 1
    fast = scene.render_desc()
    fast.filter_general = 0
    fast.tile_pixels = 256
    for i in range(4): fast.tile_halo[i] = 0
    with pytest.raises(gpu.PbrtGpuError, match="round up onto the next pixel"):
        gs.render(fast)
    film, strays = gs.render(rd)  # the frame itself: identical to the reference binary's image
    assert len(strays) == 0
    scene.film_clear(); scene.film_merge(rd, film, strays)
    ref = gpu.read_pfm(os.path.join(GOLD, "filter_box_round_up.pfm"))
    assert np.array_equal(scene.film_image().view(np.uint32), ref.view(np.uint32))
    gs.close()
