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


def read_exr_scanline_half(path):
    """A reader for exactly what WriteImageEXR (host/film.cpp) says it writes: single-part scanline OpenEXR 2, no compression, HALF
    channels B, G, R.  Returns (rgb float16 [h, w, 3], data window, display window, attribute dict)."""
    import struct
    b = open(path, "rb").read()
    assert b[:4] == bytes([0x76, 0x2f, 0x31, 0x01]) and struct.unpack_from("<I", b, 4)[0] == 2
    pos, attrs = 8, {}
    def cstr():
        nonlocal pos
        end = b.index(b"\0", pos)
        s = b[pos:end].decode()
        pos = end + 1
        return s
    while b[pos] != 0:
        name, typ = cstr(), cstr()
        size = struct.unpack_from("<i", b, pos)[0]
        attrs[name] = (typ, b[pos + 4:pos + 4 + size])
        pos += 4 + size
    pos += 1
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    ch, names = attrs["channels"][1], []
    q = 0
    while ch[q] != 0:
        end = ch.index(b"\0", q)
        names.append(ch[q:end].decode())
        ptype, _, xs, ys = struct.unpack_from("<iIii", ch, end + 1)
        assert (ptype, xs, ys) == (1, 1, 1)
        q = end + 1 + 16
    assert names == ["B", "G", "R"]
    dw, disp = struct.unpack("<4i", attrs["dataWindow"][1]), struct.unpack("<4i", attrs["displayWindow"][1])
    w, h = dw[2] - dw[0] + 1, dw[3] - dw[1] + 1
    offsets = struct.unpack_from(f"<{h}Q", b, pos)
    img = np.zeros((h, w, 3), np.float16)
    for y in range(h):
        yy, n = struct.unpack_from("<ii", b, offsets[y])
        assert yy == dw[1] + y and n == 6 * w
        row = np.frombuffer(b, np.float16, 3 * w, offsets[y] + 8).reshape(3, w)
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = row[0], row[1], row[2]
    assert offsets[-1] + 8 + 6 * w == len(b)
    return img, dw, disp, attrs


def test_exr_output_is_half_float_rgb(pkg, tmp_path):
    """Film "image" files default to OpenEXR (film.cpp:213-252); the host writes the half-float RGB scanline file the reference asks the
    OpenEXR library for (imageio.cpp:186-211), uncompressed.  float -> half: round to nearest even, overflow to infinity."""
    rng = np.random.default_rng(5)
    img = (rng.random((19, 31, 3)).astype(np.float32) ** 4 * 90000 - 2).astype(np.float32)  # negatives, denormal halfs, beyond 65504
    img[0, 0] = [65519.996, 65520.0, 5.9604645e-08]      # just below / at the overflow tie; the smallest denormal half
    img[0, 1] = [2.9802322e-08, 2.9802326e-08, 6.1e-05]  # the tie below the smallest denormal (to zero), just above it, the denormal / normal border
    img[0, 2] = [np.inf, -np.inf, 1.0009766]             # 1 + 2^-10: exactly representable
    img[0, 3] = [1.00048828125, 1.00146484375, -0.0]     # ties: to even (down), to even (up)
    path = str(tmp_path / "out.exr")
    assert pkg.host_lib().pbrt_host_write_image(path.encode(), np.ascontiguousarray(img).ctypes.data, 31, 19) == 0
    got, dw, disp, attrs = read_exr_scanline_half(path)
    with np.errstate(over="ignore"):
        want = img.astype(np.float16)  # numpy converts with round-to-nearest-even, like half::half(float)
    assert np.array_equal(got.view(np.uint16), want.view(np.uint16))
    assert dw == (0, 0, 30, 18) and disp == (0, 0, 30, 18)


def test_exr_data_window_is_the_crop_window(pkg, tmp_path):
    """A cropped film: displayWindow = the full resolution, dataWindow = the crop (WriteImageEXR's totalXRes / xOffset arguments,
    imageio.cpp:186-196)."""
    img = np.linspace(0, 2, 7 * 5 * 3, dtype=np.float32).reshape(5, 7, 3)
    path = str(tmp_path / "crop.exr")
    assert pkg.host_lib().pbrt_host_write_image_window(path.encode(), img.ctypes.data, 7, 5, 12, 4, 48, 32) == 0
    got, dw, disp, _ = read_exr_scanline_half(path)
    assert dw == (12, 4, 18, 8) and disp == (0, 0, 47, 31) and np.array_equal(got, img.astype(np.float16))
