// Film for the MI355X path (reference: src/core/film.{h,cpp}).  The device
// accumulates FilmTilePixel{contribSum, filterWeightSum} per 16x16 tile in
// sample order; this file performs MergeFilmTile (RGB->XYZ add, film.cpp:117-130)
// and WriteImage's normalisation (film.cpp:169-211) with the same arithmetic,
// and writes PFM (core/imageio.cpp:437-482).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <map>
#include "api.h"
#include "error.h"
#include <zlib.h>
#include <cstring>
#include "scene.h"

namespace pbrt {
static inline void RGBToXYZ(const Float rgb[3], Float xyz[3]) {  // spectrum.h:62-66
    xyz[0] = 0.412453f * rgb[0] + 0.357580f * rgb[1] + 0.180423f * rgb[2];
    xyz[1] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];
    xyz[2] = 0.019334f * rgb[0] + 0.119193f * rgb[1] + 0.950227f * rgb[2];
}
static inline void XYZToRGB(const Float xyz[3], Float rgb[3]) {  // spectrum.h:56-60
    rgb[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    rgb[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    rgb[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
}

Film::Film(const int resolution[2], const Float crop[4], Float frx, Float fry, const std::string &fn, Float scale,
           Float maxSampleLuminance)
    : filename(fn), scale(scale), maxSampleLuminance(maxSampleLuminance) {
    fullResolution[0] = resolution[0]; fullResolution[1] = resolution[1];
    filterRadius[0] = frx; filterRadius[1] = fry;
    // film.cpp:54-59: crop = {xmin, xmax, ymin, ymax}
    croppedPixelBounds[0] = (int)std::ceil(fullResolution[0] * crop[0]);
    croppedPixelBounds[1] = (int)std::ceil(fullResolution[1] * crop[2]);
    croppedPixelBounds[2] = (int)std::ceil(fullResolution[0] * crop[1]);
    croppedPixelBounds[3] = (int)std::ceil(fullResolution[1] * crop[3]);
    int w = croppedPixelBounds[2] - croppedPixelBounds[0], h = croppedPixelBounds[3] - croppedPixelBounds[1];
    pixels.resize((size_t)std::max(0, w) * std::max(0, h));
}
void Film::GetSampleBounds(int out[4]) const {  // film.cpp:80-86
    out[0] = (int)std::floor((Float)croppedPixelBounds[0] + 0.5f - filterRadius[0]);
    out[1] = (int)std::floor((Float)croppedPixelBounds[1] + 0.5f - filterRadius[1]);
    out[2] = (int)std::ceil((Float)croppedPixelBounds[2] - 0.5f + filterRadius[0]);
    out[3] = (int)std::ceil((Float)croppedPixelBounds[3] - 0.5f + filterRadius[1]);
}
void Film::Clear() { for (auto &p : pixels) p = Pixel(); }

// ---- pixel filters: filters/{box,gaussian,mitchell,sinc,triangle}.{h,cpp}, filter.h:50-66 ------------------------
void FilterRadiusFor(const std::string &name, const ParamSet &ps, Float *xw, Float *yw) {
    Float d = name == "box" ? 0.5f : (name == "sinc" ? 4.f : 2.f);  // each Create*Filter's default "xwidth"/"ywidth"
    *xw = ps.FindOneFloat("xwidth", d);
    *yw = ps.FindOneFloat("ywidth", d);
}
bool SetFilmFilter(Film *film, const std::string &name, const ParamSet &ps) {
    const Float rx = film->filterRadius[0], ry = film->filterRadius[1];
    std::function<Float(Float, Float)> eval;
    if (name == "box") eval = [](Float, Float) { return (Float)1.; };
    else if (name == "gaussian") {  // gaussian.h:50-68
        Float alpha = ps.FindOneFloat("alpha", 2.f);
        Float expX = std::exp(-alpha * rx * rx), expY = std::exp(-alpha * ry * ry);
        auto G = [alpha](Float d, Float expv) { return std::max((Float)0, Float(std::exp(-alpha * d * d) - expv)); };
        eval = [=](Float x, Float y) { return G(x, expX) * G(y, expY); };
    } else if (name == "mitchell") {  // mitchell.h:51-67
        Float B = ps.FindOneFloat("B", 1.f / 3.f), C = ps.FindOneFloat("C", 1.f / 3.f);
        Float invX = 1 / rx, invY = 1 / ry;
        auto M = [B, C](Float x) {
            x = std::abs(2 * x);
            if (x > 1) return ((-B - 6 * C) * x * x * x + (6 * B + 30 * C) * x * x + (-12 * B - 48 * C) * x + (8 * B + 24 * C)) * (1.f / 6.f);
            else return ((12 - 9 * B - 6 * C) * x * x * x + (-18 + 12 * B + 6 * C) * x * x + (6 - 2 * B)) * (1.f / 6.f);
        };
        eval = [=](Float x, Float y) { return M(x * invX) * M(y * invY); };
    } else if (name == "sinc") {  // sinc.h:50-66
        Float tau = ps.FindOneFloat("tau", 3.f);
        auto Sinc = [](Float x) { x = std::abs(x); if (x < 1e-5) return (Float)1; return std::sin(Pi * x) / (Pi * x); };
        auto W = [=](Float x, Float radius) { x = std::abs(x); if (x > radius) return (Float)0; Float lanczos = Sinc(x / tau); return Sinc(x) * lanczos; };
        eval = [=](Float x, Float y) { return W(x, rx) * W(y, ry); };
    } else if (name == "triangle")  // triangle.cpp:41-45
        eval = [=](Float x, Float y) { return std::max((Float)0, rx - std::abs(x)) * std::max((Float)0, ry - std::abs(y)); };
    else return false;
    const int filterTableWidth = 16;
    int offset = 0;
    for (int y = 0; y < filterTableWidth; ++y)  // film.cpp:68-77
        for (int x = 0; x < filterTableWidth; ++x, ++offset) {
            Float px = (x + 0.5f) * rx / filterTableWidth, py = (y + 0.5f) * ry / filterTableWidth;
            film->filterTable[offset] = eval(px, py);
        }
    film->filterGeneral = !(name == "box" && rx <= 0.5f && ry <= 0.5f && rx > 0 && ry > 0);
    return true;
}
void Film::TileHalo(int h[4]) const {  // GetFilmTile, film.cpp:95-106, for a tile [x0, x0+16) x [y0, y0+16)
    // low: x0 - ceil(x0 - 0.5 - r); high: floor(x1 - 0.5 + r) + 1 - x1 (integers x0, x1 drop out)
    h[0] = -(int)std::ceil(-0.5f - filterRadius[0]); h[1] = -(int)std::ceil(-0.5f - filterRadius[1]);
    h[2] = (int)std::floor(-0.5f + filterRadius[0]) + 1; h[3] = (int)std::floor(-0.5f + filterRadius[1]) + 1;
}
int Film::TilePixels() const {
    if (!filterGeneral) return 256;
    int h[4];
    TileHalo(h);
    return (16 + h[0] + h[2]) * (16 + h[1] + h[3]);
}

void Film::MergeShard(const PgRenderDesc &rd, const PgFilmPixel *film, const PgStraySample *strays, int nStrays) {
    // one shard = the tiles tile_first, tile_first + tile_step, ... of the frame, in that order
    MergeTiles(rd, rd.tile_first, rd.tile_step, [&](int t) { return film + (size_t)((t - rd.tile_first) / rd.tile_step) * rd.tile_pixels; }, &strays, &nStrays, 1);
}
void Film::MergeShards(const PgRenderDesc &full, int n, const PgFilmPixel *const *film, const PgStraySample *const *strays, const int *nStrays) {
    // n shards of one frame (rank r rendered the tiles t = r (mod n)): merged in the frame's own tile order 0, 1, 2, ... -- tile t's
    // block is the (t / n)-th of shard t % n -- so that overlapping tile blocks of a filter wider than half a pixel (float sums)
    // add up in the order a one-device render and the single-threaded reference use, whatever n is
    MergeTiles(full, 0, 1, [&](int t) { return film[t % n] + (size_t)(t / n) * full.tile_pixels; }, strays, nStrays, n);
}
void Film::MergeTiles(const PgRenderDesc &rd, int tileFirst, int tileStep, const std::function<const PgFilmPixel *(int)> &blockOf, const PgStraySample *const *strayLists,
                      const int *nStrayLists, int nLists) {
    const int tileSize = 16;
    if (rd.filter_general) {
        // MergeFilmTile (film.cpp:117-130) for every tile in tile order: each FilmTile pixel is converted to
        // XYZ and added to the film, clipped to the cropped pixel bounds as GetFilmTile's Intersect does.
        const int sx0 = rd.sample_bounds[0], sy0 = rd.sample_bounds[1];
        const int nTilesX = (rd.sample_bounds[2] - sx0 + tileSize - 1) / tileSize;
        const int nTilesY = (rd.sample_bounds[3] - sy0 + tileSize - 1) / tileSize;
        const int width = croppedPixelBounds[2] - croppedPixelBounds[0];
        const int tw = tileSize + rd.tile_halo[0] + rd.tile_halo[2];
        for (int t = tileFirst; t < nTilesX * nTilesY; t += tileStep) {
            const PgFilmPixel *block = blockOf(t);
            int tx = t % nTilesX, ty = t / nTilesX;
            int x0 = sx0 + tx * tileSize, y0 = sy0 + ty * tileSize;
            int x1 = std::min(x0 + tileSize, rd.sample_bounds[2]), y1 = std::min(y0 + tileSize, rd.sample_bounds[3]);
            int px0 = std::max(x0 - rd.tile_halo[0], croppedPixelBounds[0]), py0 = std::max(y0 - rd.tile_halo[1], croppedPixelBounds[1]);
            int px1 = std::min(x1 + rd.tile_halo[2], croppedPixelBounds[2]), py1 = std::min(y1 + rd.tile_halo[3], croppedPixelBounds[3]);
            for (int y = py0; y < py1; ++y)
                for (int x = px0; x < px1; ++x) {
                    const PgFilmPixel &fp = block[(size_t)(y - (y0 - rd.tile_halo[1])) * tw + (x - (x0 - rd.tile_halo[0]))];
                    Float xyz[3];
                    RGBToXYZ(fp.rgb, xyz);
                    Pixel &mp = pixels[(size_t)(y - croppedPixelBounds[1]) * width + (x - croppedPixelBounds[0])];
                    for (int c = 0; c < 3; ++c) mp.xyz[c] += xyz[c];
                    mp.filterWeightSum += fp.weight;
                }
        }
        return;
    }
    const int sx0 = rd.sample_bounds[0], sy0 = rd.sample_bounds[1];
    const int nTilesX = (rd.sample_bounds[2] - sx0 + tileSize - 1) / tileSize;
    const int nTilesY = (rd.sample_bounds[3] - sy0 + tileSize - 1) / tileSize;
    const int width = croppedPixelBounds[2] - croppedPixelBounds[0];
    auto tileOf = [&](int x, int y) { return ((y - sy0) / tileSize) * nTilesX + (x - sx0) / tileSize; };
    auto inCrop = [&](int x, int y) {
        return x >= croppedPixelBounds[0] && x < croppedPixelBounds[2] && y >= croppedPixelBounds[1] && y < croppedPixelBounds[3];
    };
    // Stray samples (a sample whose offset is exactly 0 also lands in the
    // previous pixel, film.h:127-132): the contributions one FilmTile makes to
    // one pixel are summed in RGB in sample order, then merged.
    struct Key { int tile, x, y; bool operator<(const Key &o) const { return tile != o.tile ? tile < o.tile : (y != o.y ? y < o.y : x < o.x); } };
    std::map<Key, std::vector<const PgStraySample *>> groups;
    for (int l = 0; l < nLists; ++l)  // (a tile's strays all come from the one shard that rendered it: their order inside a group is that shard's)
        for (int i = 0; i < nStrayLists[l]; ++i) {
            const PgStraySample &s = strayLists[l][i];
            if (!inCrop(s.px, s.py)) continue;
            groups[Key{tileOf(s.src_px, s.src_py), s.px, s.py}].push_back(&s);
        }
    for (auto &kv : groups)  // source order inside a tile = row-major pixel order (integrator.cpp:263)
        std::sort(kv.second.begin(), kv.second.end(), [](const PgStraySample *a, const PgStraySample *b) {
            return a->src_py != b->src_py ? a->src_py < b->src_py : a->src_px < b->src_px;
        });
    for (int t = tileFirst; t < nTilesX * nTilesY; t += tileStep) {
        const PgFilmPixel *block = blockOf(t);
        int tx = t % nTilesX, ty = t / nTilesX;
        int x0 = sx0 + tx * tileSize, y0 = sy0 + ty * tileSize;
        int x1 = std::min(x0 + tileSize, rd.sample_bounds[2]), y1 = std::min(y0 + tileSize, rd.sample_bounds[3]);
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                if (!inCrop(x, y)) continue;
                const PgFilmPixel &fp = block[(y - y0) * tileSize + (x - x0)];
                Float rgb[3] = {fp.rgb[0], fp.rgb[1], fp.rgb[2]};
                Float wsum = fp.weight;
                auto g = groups.find(Key{t, x, y});
                if (g != groups.end()) {
                    for (const PgStraySample *s : g->second) {
                        for (int c = 0; c < 3; ++c) rgb[c] += s->rgb[c];
                        wsum += s->weight;
                    }
                    groups.erase(g);
                }
                Float xyz[3];
                RGBToXYZ(rgb, xyz);
                Pixel &mp = pixels[(size_t)(y - croppedPixelBounds[1]) * width + (x - croppedPixelBounds[0])];
                for (int c = 0; c < 3; ++c) mp.xyz[c] += xyz[c];
                mp.filterWeightSum += wsum;
            }
    }
    // contributions a tile makes to pixels owned by a neighbouring tile
    for (auto &kv : groups) {
        Float rgb[3] = {0, 0, 0}, wsum = 0;
        for (const PgStraySample *s : kv.second) {
            for (int c = 0; c < 3; ++c) rgb[c] += s->rgb[c];
            wsum += s->weight;
        }
        Float xyz[3];
        RGBToXYZ(rgb, xyz);
        Pixel &mp = pixels[(size_t)(kv.first.y - croppedPixelBounds[1]) * width + (kv.first.x - croppedPixelBounds[0])];
        for (int c = 0; c < 3; ++c) mp.xyz[c] += xyz[c];
        mp.filterWeightSum += wsum;
    }
}

void Film::ComputeImage(std::vector<Float> *out) const {  // film.cpp:169-206 (no splats)
    std::vector<Float> &rgb = *out;
    rgb.resize(3 * pixels.size());
    size_t offset = 0;
    for (const Pixel &pixel : pixels) {
        XYZToRGB(pixel.xyz, &rgb[3 * offset]);
        Float filterWeightSum = pixel.filterWeightSum;
        if (filterWeightSum != 0) {
            Float invWt = (Float)1 / filterWeightSum;
            rgb[3 * offset] = std::max((Float)0, rgb[3 * offset] * invWt);
            rgb[3 * offset + 1] = std::max((Float)0, rgb[3 * offset + 1] * invWt);
            rgb[3 * offset + 2] = std::max((Float)0, rgb[3 * offset + 2] * invWt);
        }
        Float splatRGB[3], splatXYZ[3] = {0, 0, 0};
        XYZToRGB(splatXYZ, splatRGB);
        const Float splatScale = 1;
        rgb[3 * offset] += splatScale * splatRGB[0];
        rgb[3 * offset + 1] += splatScale * splatRGB[1];
        rgb[3 * offset + 2] += splatScale * splatRGB[2];
        rgb[3 * offset] *= scale;
        rgb[3 * offset + 1] *= scale;
        rgb[3 * offset + 2] *= scale;
        ++offset;
    }
}

bool WriteImagePFM(const std::string &filename, const Float *rgb, int width, int height) {  // imageio.cpp:437-482
    FILE *fp = fopen(filename.c_str(), "wb");
    if (!fp) { Error("Unable to open output PFM file \"%s\"", filename.c_str()); return false; }
    bool ok = fprintf(fp, "PF\n") >= 0 && fprintf(fp, "%d %d\n", width, height) >= 0 && fprintf(fp, "%f\n", -1.f) >= 0;
    for (int y = height - 1; ok && y >= 0; y--)  // bottom-to-top scanlines, little endian
        ok = fwrite(&rgb[(size_t)y * width * 3], sizeof(float), (size_t)width * 3, fp) == (size_t)width * 3;
    fclose(fp);
    if (!ok) Error("Error writing PFM file \"%s\"", filename.c_str());
    return ok;
}

// ---- 8-bit output formats, imageio.cpp:81-120: gamma-encoded RGB8 as PNG (lodepng_encode24_file in the reference) or TGA
// (tga_write_bgr, uncompressed 24 bit).  Readers see the same pixels as in the reference's files; the PNG's deflate stream
// itself comes from zlib here.
static Float GammaCorrect(Float value) {  // pbrt.h:293-296
    if (value <= 0.0031308f) return 12.92f * value;
    return 1.055f * std::pow(value, (Float)(1.f / 2.4f)) - 0.055f;
}
static void ToRGB8(const Float *rgb, int width, int height, std::vector<uint8_t> *out) {
    out->resize((size_t)3 * width * height);
    for (size_t i = 0; i < out->size(); ++i) {
        Float v = 255.f * GammaCorrect(rgb[i]) + 0.5f;
        (*out)[i] = (uint8_t)(v < 0.f ? 0.f : (v > 255.f ? 255.f : v));  // (uint8_t)Clamp(..., 0.f, 255.f)
    }
}
static void put32be(std::vector<uint8_t> &v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
static void pngChunk(std::vector<uint8_t> &file, const char type[4], const std::vector<uint8_t> &data) {
    put32be(file, (uint32_t)data.size());
    std::vector<uint8_t> body(type, type + 4);
    body.insert(body.end(), data.begin(), data.end());
    file.insert(file.end(), body.begin(), body.end());
    put32be(file, (uint32_t)crc32(0L, body.data(), (uInt)body.size()));
}
bool WriteImagePNG(const std::string &filename, const uint8_t *rgb8, int width, int height) {
    std::vector<uint8_t> raw;  // filter type 0 in front of every scanline
    raw.reserve((size_t)height * (3 * (size_t)width + 1));
    for (int y = 0; y < height; ++y) { raw.push_back(0); raw.insert(raw.end(), rgb8 + (size_t)3 * width * y, rgb8 + (size_t)3 * width * (y + 1)); }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { Error("Error writing PNG \"%s\": deflate failed", filename.c_str()); return false; }
    z.resize(zlen);
    std::vector<uint8_t> file = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'}, ihdr;
    put32be(ihdr, (uint32_t)width); put32be(ihdr, (uint32_t)height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);  // 8 bits, RGB, deflate, adaptive, no interlace
    pngChunk(file, "IHDR", ihdr);
    pngChunk(file, "IDAT", z);
    pngChunk(file, "IEND", {});
    FILE *fp = fopen(filename.c_str(), "wb");
    if (!fp) { Error("Error writing PNG \"%s\": cannot open the file", filename.c_str()); return false; }
    bool ok = fwrite(file.data(), 1, file.size(), fp) == file.size();
    fclose(fp);
    if (!ok) Error("Error writing PNG \"%s\"", filename.c_str());
    return ok;
}
bool WriteImageTGA(const std::string &filename, const uint8_t *rgb8, int width, int height) {
    uint8_t hdr[18] = {0};
    hdr[2] = 2;  // uncompressed true colour
    hdr[12] = width & 255; hdr[13] = width >> 8; hdr[14] = height & 255; hdr[15] = height >> 8;
    hdr[16] = 24; hdr[17] = 0x20;  // top-to-bottom rows
    std::vector<uint8_t> bgr((size_t)3 * width * height);
    for (size_t i = 0; i < (size_t)width * height; ++i) { bgr[3 * i] = rgb8[3 * i + 2]; bgr[3 * i + 1] = rgb8[3 * i + 1]; bgr[3 * i + 2] = rgb8[3 * i]; }
    FILE *fp = fopen(filename.c_str(), "wb");
    if (!fp) { Error("Unable to write output file \"%s\"", filename.c_str()); return false; }
    bool ok = fwrite(hdr, 1, 18, fp) == 18 && fwrite(bgr.data(), 1, bgr.size(), fp) == bgr.size();
    fclose(fp);
    if (!ok) Error("Unable to write output file \"%s\"", filename.c_str());
    return ok;
}
// OpenEXR output: the file the reference's WriteImageEXR (imageio.cpp:186-211) asks the OpenEXR library for -- half-float R, G, B
// (Rgba with WRITE_RGB), display window = the full resolution, data window = the crop window -- written directly as a scanline
// file WITHOUT compression (the library's default would be PIZ; any reader accepts either).  float -> half rounds to nearest
// even, overflows to infinity and keeps NaN, like half::half(float).  Unpinned against the reference: the reference build used
// as the oracle carries no OpenEXR (oracle/ref_stubs); tests/test_image_output.py checks the file against this description.
static uint16_t FloatToHalf(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | (x > 0x7f800000u ? (0x200u | ((x >> 13) & 0x3ffu)) : 0u));  // inf / NaN
    if (x >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  // rounds to a value beyond 65504: infinity
    if (x < 0x38800000u) {  // below the smallest normal half: denormal (or zero)
        if (x < 0x33000000u) return (uint16_t)sign;  // less than half of the smallest denormal
        const int shift = 126 - (int)(x >> 23);      // 14 .. 24
        const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
        uint32_t h = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1u), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (h & 1u))) ++h;
        return (uint16_t)(sign | h);
    }
    uint32_t h = ((x - 0x38000000u) >> 13);
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;  // a carry out of the mantissa moves to the next exponent, as it must
    return (uint16_t)(sign | h);
}
bool WriteImageEXR(const std::string &filename, const Float *rgb, int width, int height, int xOffset, int yOffset, int totalX, int totalY) {
    std::vector<uint8_t> out;
    auto bytes = [&](const void *p, size_t n) { const uint8_t *b = (const uint8_t *)p; out.insert(out.end(), b, b + n); };
    auto i32 = [&](int32_t v) { bytes(&v, 4); };
    auto str = [&](const char *t) { bytes(t, strlen(t) + 1); };
    auto attr = [&](const char *name, const char *type, int32_t size) { str(name); str(type); i32(size); };
    const uint8_t magic[8] = {0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0};  // magic number, version 2, no flags: single-part scanline
    bytes(magic, 8);
    attr("channels", "chlist", 3 * 18 + 1);
    for (const char *c : {"B", "G", "R"}) {  // alphabetical; HALF = 1, pLinear 0, sampling 1 x 1
        str(c); i32(1);
        const uint8_t lin[4] = {0, 0, 0, 0};
        bytes(lin, 4); i32(1); i32(1);
    }
    out.push_back(0);
    attr("compression", "compression", 1); out.push_back(0);  // NO_COMPRESSION
    attr("dataWindow", "box2i", 16); i32(xOffset); i32(yOffset); i32(xOffset + width - 1); i32(yOffset + height - 1);
    attr("displayWindow", "box2i", 16); i32(0); i32(0); i32(totalX - 1); i32(totalY - 1);
    attr("lineOrder", "lineOrder", 1); out.push_back(0);  // INCREASING_Y
    const float one = 1.f, zero = 0.f;
    attr("pixelAspectRatio", "float", 4); bytes(&one, 4);
    attr("screenWindowCenter", "v2f", 8); bytes(&zero, 4); bytes(&zero, 4);
    attr("screenWindowWidth", "float", 4); bytes(&one, 4);
    out.push_back(0);  // end of header
    const size_t rowBytes = 8 + (size_t)3 * 2 * width;
    uint64_t offset = out.size() + (uint64_t)8 * height;
    for (int y = 0; y < height; ++y) { bytes(&offset, 8); offset += rowBytes; }
    std::vector<uint16_t> row((size_t)3 * width);
    for (int y = 0; y < height; ++y) {
        i32(yOffset + y); i32((int32_t)(rowBytes - 8));
        for (int x = 0; x < width; ++x) {
            const Float *px = rgb + 3 * ((size_t)y * width + x);
            row[x] = FloatToHalf(px[2]); row[width + x] = FloatToHalf(px[1]); row[2 * (size_t)width + x] = FloatToHalf(px[0]);
        }
        bytes(row.data(), row.size() * 2);
    }
    FILE *fp = fopen(filename.c_str(), "wb");
    if (!fp) { Error("Unable to write output file \"%s\"", filename.c_str()); return false; }
    const bool ok = fwrite(out.data(), 1, out.size(), fp) == out.size();
    fclose(fp);
    if (!ok) Error("Unable to write output file \"%s\"", filename.c_str());
    return ok;
}
// WriteImage, imageio.cpp:81-122 (outputBounds = [offset, offset + size), totalResolution)
bool WriteImage(const std::string &filename, const Float *rgb, int width, int height) { return WriteImage(filename, rgb, width, height, 0, 0, width, height); }
bool WriteImage(const std::string &filename, const Float *rgb, int width, int height, int xOffset, int yOffset, int totalX, int totalY) {
    auto hasExt = [&](const char *e) { size_t n = filename.size(), m = strlen(e); return n >= m && filename.compare(n - m, m, e) == 0; };
    if (hasExt(".pfm")) return WriteImagePFM(filename, rgb, width, height);
    if (hasExt(".png") || hasExt(".tga")) {
        std::vector<uint8_t> rgb8;
        ToRGB8(rgb, width, height, &rgb8);
        return hasExt(".png") ? WriteImagePNG(filename, rgb8.data(), width, height) : WriteImageTGA(filename, rgb8.data(), width, height);
    }
    if (hasExt(".exr")) return WriteImageEXR(filename, rgb, width, height, xOffset, yOffset, totalX, totalY);
    Error("Can't determine image file type from suffix of filename \"%s\"", filename.c_str());
    return false;
}

void Film::WriteImage() const {
    std::vector<Float> rgb;
    ComputeImage(&rgb);
    int w = croppedPixelBounds[2] - croppedPixelBounds[0], h = croppedPixelBounds[3] - croppedPixelBounds[1];
    pbrt::WriteImage(filename, rgb.data(), w, h, croppedPixelBounds[0], croppedPixelBounds[1], fullResolution[0], fullResolution[1]);
}

Film *CreateFilm(const ParamSet &params, Float frx, Float fry) {  // film.cpp:213-252
    std::string filename;
    if (PbrtOptions.imageFile != "") {
        filename = PbrtOptions.imageFile;
        std::string paramsFilename = params.FindOneString("filename", "");
        if (paramsFilename != "")
            Warning("Output filename supplied on command line, \"%s\" is overriding filename provided in scene description file, \"%s\".",
                    PbrtOptions.imageFile.c_str(), paramsFilename.c_str());
    } else
        filename = params.FindOneString("filename", "pbrt.exr");
    int xres = params.FindOneInt("xresolution", 1280);
    int yres = params.FindOneInt("yresolution", 720);
    if (PbrtOptions.quickRender) xres = std::max(1, xres / 4);
    if (PbrtOptions.quickRender) yres = std::max(1, yres / 4);
    Float crop[4];
    auto clamp01 = [](Float v) { return v < 0.f ? 0.f : (v > 1.f ? 1.f : v); };
    const std::vector<Float> *cr = params.FindFloat("cropwindow");
    if (cr && cr->size() == 4) {
        crop[0] = clamp01(std::min((*cr)[0], (*cr)[1])); crop[1] = clamp01(std::max((*cr)[0], (*cr)[1]));
        crop[2] = clamp01(std::min((*cr)[2], (*cr)[3])); crop[3] = clamp01(std::max((*cr)[2], (*cr)[3]));
    } else {
        if (cr) Error("%d values supplied for \"cropwindow\". Expected 4.", (int)cr->size());
        crop[0] = clamp01(PbrtOptions.cropWindow[0][0]); crop[1] = clamp01(PbrtOptions.cropWindow[0][1]);
        crop[2] = clamp01(PbrtOptions.cropWindow[1][0]); crop[3] = clamp01(PbrtOptions.cropWindow[1][1]);
    }
    Float scale = params.FindOneFloat("scale", 1.);
    params.FindOneFloat("diagonal", 35.);
    Float maxSampleLuminance = params.FindOneFloat("maxsampleluminance", Infinity);
    int res[2] = {xres, yres};
    return new Film(res, crop, frx, fry, filename, scale, maxSampleLuminance);
}
}  // namespace pbrt
