// ParamSet: typed name->values store filled by the scene-file parser and read
// by the Create*() factories, with pbrt's "report unused" behaviour
// (src/core/paramset.h:95-117, paramset.cpp).  Only the parameter types the
// supported directives use are kept.
#ifndef PBRT_AMD_HOST_PARAMSET_H
#define PBRT_AMD_HOST_PARAMSET_H
#include <map>
#include <string>
#include <vector>
#include "error.h"
#include "geometry.h"

namespace pbrt {
struct RGB { Float c[3]; };
// "xyz" / "blackbody" / "spectrum" parameter values as the RGB triples the reference's RGBSpectrum holds (host/spectrum.cpp)
void XYZToRGBValues(const std::vector<Float> &xyz, std::vector<Float> *rgb);
void BlackbodyToRGBValues(const std::vector<Float> &temperatureScalePairs, std::vector<Float> *rgb);
void SampledToRGBValues(const std::vector<Float> &lambdaValuePairs, std::vector<Float> *rgb);
void SpectrumFilesToRGBValues(const std::vector<std::string> &filenames, std::vector<Float> *rgb);

class ParamSet {
  public:
    template <typename T> struct Item { std::vector<T> v; mutable bool lookedUp = false; };
    std::map<std::string, Item<int>> ints;
    std::map<std::string, Item<Float>> floats;      // "float", and point/vector/normal/rgb as flat floats
    std::map<std::string, Item<Float>> point2s, point3s, vector3s, normals, spectra;
    std::map<std::string, Item<std::string>> strings, textures;
    std::map<std::string, Item<bool>> bools;

    int FindOneInt(const std::string &n, int d) const { auto p = look(ints, n); return p && !p->v.empty() ? p->v[0] : d; }
    Point3f FindOnePoint3f(const std::string &n, Point3f d) const { auto p = look(point3s, n); return p && p->v.size() >= 3 ? Point3f(p->v[0], p->v[1], p->v[2]) : d; }
    Float FindOneFloat(const std::string &n, Float d) const { auto p = look(floats, n); return p && !p->v.empty() ? p->v[0] : d; }
    bool FindOneBool(const std::string &n, bool d) const { auto p = look(bools, n); return p && !p->v.empty() ? p->v[0] : d; }
    std::string FindOneString(const std::string &n, const std::string &d) const {
        auto p = look(strings, n); return p && !p->v.empty() ? p->v[0] : d;
    }
    std::string FindTexture(const std::string &n) const { auto p = look(textures, n); return p && !p->v.empty() ? p->v[0] : ""; }
    const std::vector<int> *FindInt(const std::string &n) const { auto p = look(ints, n); return p ? &p->v : nullptr; }
    const std::vector<Float> *FindFloat(const std::string &n) const { auto p = look(floats, n); return p ? &p->v : nullptr; }
    const std::vector<Float> *FindPoint2f(const std::string &n) const { auto p = look(point2s, n); return p ? &p->v : nullptr; }
    const std::vector<Float> *FindPoint3f(const std::string &n) const { auto p = look(point3s, n); return p ? &p->v : nullptr; }
    const std::vector<Float> *FindVector3f(const std::string &n) const { auto p = look(vector3s, n); return p ? &p->v : nullptr; }
    const std::vector<Float> *FindNormal3f(const std::string &n) const { auto p = look(normals, n); return p ? &p->v : nullptr; }
    // FindOneSpectrum (paramset.cpp): RGB spectra only (RGBSpectrum build, pbrt.h:158-160).
    bool FindSpectrum(const std::string &n, RGB *out) const {
        auto p = look(spectra, n);
        if (!p || p->v.size() < 3) return false;
        out->c[0] = p->v[0]; out->c[1] = p->v[1]; out->c[2] = p->v[2];
        return true;
    }
    RGB FindOneSpectrum(const std::string &n, RGB d) const { RGB r; return FindSpectrum(n, &r) ? r : d; }
    void ReportUnused() const {  // paramset.cpp ReportUnused
        report(ints); report(floats); report(point2s); report(point3s); report(vector3s);
        report(normals); report(spectra); report(strings); report(textures); report(bools);
    }
    // shapeMaySetMaterialParameters heuristics need raw access (api.cpp:1427-1470).
  private:
    template <typename M> static const typename M::mapped_type *look(const M &m, const std::string &n) {
        auto it = m.find(n);
        if (it == m.end()) return nullptr;
        it->second.lookedUp = true;
        return &it->second;
    }
    template <typename M> static void report(const M &m) {
        for (auto &kv : m)
            if (!kv.second.lookedUp) Warning("Parameter \"%s\" not used", kv.first.c_str());
    }
};
}  // namespace pbrt
#endif
