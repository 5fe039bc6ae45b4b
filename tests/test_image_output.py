"""Film::WriteImage's 8-bit formats (core/imageio.cpp:81-122): gamma-encoded PNG and TGA next to PFM."""
import os

import numpy as np
import pytest

from conftest import GOLD
from kat_util import read_png_rgb8, read_tga_rgb8


def to_byte(rgb):  # TO_BYTE, imageio.cpp:97
    v = np.asarray(rgb, np.float32)
    g = np.where(v <= np.float32(0.0031308), np.float32(12.92) * v,
                 np.float32(1.055) * np.power(np.maximum(v, 0), np.float32(1 / 2.4), dtype=np.float32) - np.float32(0.055))
    return np.clip(np.float32(255) * g.astype(np.float32) + np.float32(0.5), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("ext,reader", [("png", read_png_rgb8), ("tga", read_tga_rgb8)])
def test_8bit_round_trip(pkg, tmp_path, ext, reader):
    rng = np.random.default_rng(3)
    img = (rng.random((37, 53, 3)) ** 3 * 1.4 - 0.05).astype(np.float32)  # below 0, the linear toe, above 1
    path = str(tmp_path / f"out.{ext}")
    assert pkg.host_lib().pbrt_host_write_image(path.encode(), img.ctypes.data, 53, 37) == 0
    got = reader(path)
    want = to_byte(img)
    assert got.shape == want.shape
    diff = np.abs(got.astype(int) - want.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 2e-3  # numpy's powf vs glibc's at a rounding boundary


def test_8bit_output_equals_reference_when_present(pkg, oracle, tmp_path):
    """The unmodified reference writes the same scene as .png and .tga; this repository's writers produce the same pixels."""
    if not os.path.exists(oracle.REF_BINARY):
        pytest.skip("reference binary not available here")
    scene_file = os.path.join(GOLD, "cornell_40x24.pbrt")
    scene = pkg.HostScene(scene_file)
    img, _ = oracle.render_image(scene)
    h, w, _ = img.shape
    buf = np.ascontiguousarray(img, np.float32)
    for ext, reader in (("png", read_png_rgb8), ("tga", read_tga_rgb8)):
        ref, mine = str(tmp_path / f"ref.{ext}"), str(tmp_path / f"mine.{ext}")
        oracle.run_reference(scene_file, ref, nthreads=2)
        assert pkg.host_lib().pbrt_host_write_image(mine.encode(), buf.ctypes.data, w, h) == 0
        assert np.array_equal(reader(ref), reader(mine)), ext


def test_unknown_suffix_is_an_error(pkg, tmp_path):
    img = np.zeros((2, 2, 3), np.float32)
    n0 = pkg.host_lib().pbrt_host_error_count()
    assert pkg.host_lib().pbrt_host_write_image(str(tmp_path / "x.bmp").encode(), img.ctypes.data, 2, 2) != 0
    assert pkg.host_lib().pbrt_host_error_count() == n0 + 1
