// Spectrum-valued parameters other than "rgb": "xyz", "blackbody" and "spectrum" (inline (lambda, value) pairs or .spd files)
// are converted to the RGB coefficients the reference's RGBSpectrum build stores for them (core/paramset.cpp:122-208,
// core/spectrum.h:466-487, core/spectrum.cpp:41-57,179-188,939-964).  The CIE colour matching functions are data
// (data/cie_tables.bin, tools/extract_reference_tables.py) embedded into the library.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "api.h"
#include "error.h"
#include "paramset.h"

#ifndef PG_CIE_BIN
#error "PG_CIE_BIN (path of data/cie_tables.bin) must be defined by the build"
#endif
__asm__(".section .rodata\n"
        ".balign 4\n"
        ".global pg_cie_blob\n"
        "pg_cie_blob:\n"
        ".incbin \"" PG_CIE_BIN "\"\n"
        ".global pg_cie_blob_end\n"
        "pg_cie_blob_end:\n"
        ".previous\n");
extern "C" const float pg_cie_blob[];
extern "C" const unsigned char pg_cie_blob_end[];

namespace pbrt {
namespace {
const int nCIESamples = 471;                 // spectrum.h:76
const Float CIE_Y_integral = 106.856895;     // spectrum.h:81
const Float *CIE_X() { return pg_cie_blob; }
const Float *CIE_Y() { return pg_cie_blob + nCIESamples; }
const Float *CIE_Z() { return pg_cie_blob + 2 * nCIESamples; }
const Float *CIE_lambda() { return pg_cie_blob + 3 * nCIESamples; }

RGB FromXYZ(const Float xyz[3]) {  // XYZToRGB, spectrum.h:56-60
    RGB r;
    r.c[0] = 3.240479f * xyz[0] - 1.537150f * xyz[1] - 0.498535f * xyz[2];
    r.c[1] = -0.969256f * xyz[0] + 1.875991f * xyz[1] + 0.041556f * xyz[2];
    r.c[2] = 0.055648f * xyz[0] - 0.204043f * xyz[1] + 1.057311f * xyz[2];
    return r;
}
Float InterpolateSpectrumSamples(const Float *lambda, const Float *vals, int n, Float l) {  // spectrum.cpp:179-188
    if (l <= lambda[0]) return vals[0];
    if (l >= lambda[n - 1]) return vals[n - 1];
    // FindInterval(n, lambda[index] <= l), pbrt.h:403-415
    int first = 0, len = n;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (lambda[middle] <= l) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = std::min(std::max(first - 1, 0), n - 2);
    Float t = (l - lambda[offset]) / (lambda[offset + 1] - lambda[offset]);
    return (1 - t) * vals[offset] + t * vals[offset + 1];
}
RGB FromSampled(const Float *lambda, const Float *v, int n) {  // RGBSpectrum::FromSampled, spectrum.h:466-487
    bool sorted = true;
    for (int i = 0; i < n - 1; ++i) if (lambda[i] > lambda[i + 1]) sorted = false;  // SpectrumSamplesSorted, spectrum.cpp:41-45
    if (!sorted) {  // SortSpectrumSamples, spectrum.cpp:47-57
        std::vector<std::pair<Float, Float>> sortVec;
        for (int i = 0; i < n; ++i) sortVec.push_back(std::make_pair(lambda[i], v[i]));
        std::sort(sortVec.begin(), sortVec.end());
        std::vector<Float> sl(n), sv(n);
        for (int i = 0; i < n; ++i) { sl[i] = sortVec[i].first; sv[i] = sortVec[i].second; }
        return FromSampled(sl.data(), sv.data(), n);
    }
    for (int i = 0; i < n - 1; ++i)
        if (!(lambda[i + 1] > lambda[i])) { Error("Spectrum samples with equal wavelengths (%g nm): the reference aborts on this input.", lambda[i]); Fatal(); }
    Float xyz[3] = {0, 0, 0};
    for (int i = 0; i < nCIESamples; ++i) {
        Float val = InterpolateSpectrumSamples(lambda, v, n, CIE_lambda()[i]);
        xyz[0] += val * CIE_X()[i];
        xyz[1] += val * CIE_Y()[i];
        xyz[2] += val * CIE_Z()[i];
    }
    Float scale = Float(CIE_lambda()[nCIESamples - 1] - CIE_lambda()[0]) / Float(CIE_Y_integral * nCIESamples);
    xyz[0] *= scale; xyz[1] *= scale; xyz[2] *= scale;
    return FromXYZ(xyz);
}
// Planck's law, spectral radiance of a black body at T kelvin for a wavelength in nm (spectrum.cpp:939-955).  Computed in
// float with the reference's association -- [order] l = lambda * 1e-9 (double product, rounded), l^5 = (l*l)*(l*l)*l,
// ((2h)c)c / (l^5 * (exp((hc) / ((l kB) T)) - 1)) -- because "blackbody" parameters must give the same RGB coefficients.
static Float PlanckRadiance(Float lambdaNm, Float T) {
    const Float lightSpeed = 299792458, planck = 6.62606957e-34, boltzmann = 1.3806488e-23;
    const Float l = lambdaNm * 1e-9;
    const Float l2 = l * l, l5 = l2 * l2 * l;
    const Float exponent = (planck * lightSpeed) / (l * boltzmann * T);
    return (2 * planck * lightSpeed * lightSpeed) / (l5 * (std::exp(exponent) - 1));
}
// the curve divided by its peak, which Wien's displacement law puts at 2.8977721e-3 / T metres (spectrum.cpp:957-964)
void BlackbodyNormalized(const Float *lambda, int n, Float T, Float *Le) {
    if (T <= 0) { std::fill(Le, Le + n, std::numeric_limits<Float>::quiet_NaN()); return; }  // the reference divides 0 by 0 here
    const Float peakNm = 2.8977721e-3 / T * 1e9;  // [order] double arithmetic, rounded once
    const Float peak = PlanckRadiance(peakNm, T);
    for (int i = 0; i < n; ++i) Le[i] = PlanckRadiance(lambda[i], T) / peak;
}
// Whitespace-separated numbers with '#' comments (core/floatfile.cpp:40-82).  Kept quirks, because .spd files in the wild
// rely on them: a number is [digit . - +] followed by [digit . e - +]*; the character that ends a number is swallowed
// (so "1#x" does not start a comment); a number that runs into the end of the file is dropped; other text is warned about.
bool ReadFloatFile(const char *filename, std::vector<Float> *values) {
    FILE *f = fopen(filename, "r");
    if (!f) { Error("Unable to open file \"%s\"", filename); return false; }
    std::string text;
    char chunk[1 << 14];
    for (size_t got; (got = fread(chunk, 1, sizeof(chunk), f)) > 0;) text.append(chunk, got);
    fclose(f);
    auto startsNumber = [](unsigned char c) { return isdigit(c) || c == '.' || c == '-' || c == '+'; };
    int line = 1;
    for (size_t i = 0; i < text.size();) {
        const unsigned char c = text[i];
        if (startsNumber(c)) {
            size_t j = i + 1;
            while (j < text.size() && (startsNumber(text[j]) || text[j] == 'e')) ++j;
            if (j == text.size()) break;  // ran into the end of the file
            if (j - i >= 32) { Error("Overflowed buffer for parsing number in file: %s, at line %d", filename, line); Fatal(); }
            values->push_back(atof(text.substr(i, j - i).c_str()));
            if (text[j] == '\n') ++line;
            i = j + 1;  // the terminating character goes with the number
            continue;
        }
        if (c == '#') {
            while (i < text.size() && text[i] != '\n') ++i;
            ++line; ++i;
            continue;
        }
        if (c == '\n') ++line;
        else if (!isspace(c)) Warning("Unexpected text found at line %d of float file \"%s\"", line, filename);
        ++i;
    }
    return true;
}
std::map<std::string, RGB> cachedSpectra;  // ParamSet::cachedSpectra, paramset.cpp:210
}  // namespace

// ParamSet::AddXYZSpectrum / AddBlackbodySpectrum / AddSampledSpectrum / AddSampledSpectrumFiles, paramset.cpp:122-208
void XYZToRGBValues(const std::vector<Float> &xyz, std::vector<Float> *rgb) {
    rgb->clear();
    for (size_t i = 0; i + 2 < xyz.size(); i += 3) { RGB r = FromXYZ(&xyz[i]); rgb->insert(rgb->end(), r.c, r.c + 3); }
}
void BlackbodyToRGBValues(const std::vector<Float> &values, std::vector<Float> *rgb) {
    rgb->clear();
    std::vector<Float> v(nCIESamples);
    for (size_t i = 0; i + 1 < values.size(); i += 2) {
        BlackbodyNormalized(CIE_lambda(), nCIESamples, values[i], v.data());
        RGB s = FromSampled(CIE_lambda(), v.data(), nCIESamples);
        for (int k = 0; k < 3; ++k) rgb->push_back(values[i + 1] * s.c[k]);
    }
}
void SampledToRGBValues(const std::vector<Float> &values, std::vector<Float> *rgb) {
    std::vector<Float> wl, v;
    for (size_t i = 0; i + 1 < values.size(); i += 2) { wl.push_back(values[i]); v.push_back(values[i + 1]); }
    rgb->clear();
    if (wl.empty()) return;
    RGB s = FromSampled(wl.data(), v.data(), (int)wl.size());
    rgb->assign(s.c, s.c + 3);
}
void SpectrumFilesToRGBValues(const std::vector<std::string> &names, std::vector<Float> *rgb) {
    rgb->clear();
    for (const std::string &name : names) {
        std::string fn = AbsolutePath(ResolveFilename(name));
        RGB s{{0, 0, 0}};
        auto it = cachedSpectra.find(fn);
        if (it != cachedSpectra.end()) s = it->second;
        else {
            std::vector<Float> vals;
            if (!ReadFloatFile(fn.c_str(), &vals)) Warning("Unable to read SPD file \"%s\".  Using black distribution.", fn.c_str());
            else {
                if (vals.size() % 2) Warning("Extra value found in spectrum file \"%s\". Ignoring it.", fn.c_str());
                std::vector<Float> wls, v;
                for (size_t j = 0; j < vals.size() / 2; ++j) { wls.push_back(vals[2 * j]); v.push_back(vals[2 * j + 1]); }
                if (!wls.empty()) s = FromSampled(wls.data(), v.data(), (int)wls.size());
            }
            cachedSpectra[fn] = s;
        }
        rgb->insert(rgb->end(), s.c, s.c + 3);
    }
}
}  // namespace pbrt
