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
    const int tileSize = 16;
    if (rd.filter_general) {
        // MergeFilmTile (film.cpp:117-130) for every tile of the shard in tile order: each FilmTile pixel is converted to
        // XYZ and added to the film, clipped to the cropped pixel bounds as GetFilmTile's Intersect does.
        const int sx0 = rd.sample_bounds[0], sy0 = rd.sample_bounds[1];
        const int nTilesX = (rd.sample_bounds[2] - sx0 + tileSize - 1) / tileSize;
        const int nTilesY = (rd.sample_bounds[3] - sy0 + tileSize - 1) / tileSize;
        const int width = croppedPixelBounds[2] - croppedPixelBounds[0];
        const int tw = tileSize + rd.tile_halo[0] + rd.tile_halo[2];
        int local = 0;
        for (int t = rd.tile_first; t < nTilesX * nTilesY; t += rd.tile_step, ++local) {
            int tx = t % nTilesX, ty = t / nTilesX;
            int x0 = sx0 + tx * tileSize, y0 = sy0 + ty * tileSize;
            int x1 = std::min(x0 + tileSize, rd.sample_bounds[2]), y1 = std::min(y0 + tileSize, rd.sample_bounds[3]);
            int px0 = std::max(x0 - rd.tile_halo[0], croppedPixelBounds[0]), py0 = std::max(y0 - rd.tile_halo[1], croppedPixelBounds[1]);
            int px1 = std::min(x1 + rd.tile_halo[2], croppedPixelBounds[2]), py1 = std::min(y1 + rd.tile_halo[3], croppedPixelBounds[3]);
            for (int y = py0; y < py1; ++y)
                for (int x = px0; x < px1; ++x) {
                    const PgFilmPixel &fp = film[(size_t)local * rd.tile_pixels + (size_t)(y - (y0 - rd.tile_halo[1])) * tw + (x - (x0 - rd.tile_halo[0]))];
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
    for (int i = 0; i < nStrays; ++i) {
        const PgStraySample &s = strays[i];
        if (!inCrop(s.px, s.py)) continue;
        groups[Key{tileOf(s.src_px, s.src_py), s.px, s.py}].push_back(&s);
    }
    for (auto &kv : groups)  // source order inside a tile = row-major pixel order (integrator.cpp:263)
        std::sort(kv.second.begin(), kv.second.end(), [](const PgStraySample *a, const PgStraySample *b) {
            return a->src_py != b->src_py ? a->src_py < b->src_py : a->src_px < b->src_px;
        });
    int local = 0;
    for (int t = rd.tile_first; t < nTilesX * nTilesY; t += rd.tile_step, ++local) {
        int tx = t % nTilesX, ty = t / nTilesX;
        int x0 = sx0 + tx * tileSize, y0 = sy0 + ty * tileSize;
        int x1 = std::min(x0 + tileSize, rd.sample_bounds[2]), y1 = std::min(y0 + tileSize, rd.sample_bounds[3]);
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                if (!inCrop(x, y)) continue;
                const PgFilmPixel &fp = film[(size_t)local * 256 + (y - y0) * tileSize + (x - x0)];
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

void Film::WriteImage() const {
    std::vector<Float> rgb;
    ComputeImage(&rgb);
    int w = croppedPixelBounds[2] - croppedPixelBounds[0], h = croppedPixelBounds[3] - croppedPixelBounds[1];
    size_t n = filename.size();
    if (n >= 4 && filename.substr(n - 4) == ".pfm") WriteImagePFM(filename, rgb.data(), w, h);
    else {
        std::string alt = filename + ".pfm";
        Warning("Image format of \"%s\" is not supported by this build (PFM only); writing \"%s\".", filename.c_str(), alt.c_str());
        WriteImagePFM(alt, rgb.data(), w, h);
    }
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
