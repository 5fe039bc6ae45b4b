// Graphics-state machine and Create*() factories of the scene front end
// (reference: src/core/api.cpp).  Same directive names, parameter names and
// defaults, so a .pbrt file means the same thing; plugins outside the
// hot-path closed set are reported with Error() and
// skipped, which is pbrt's own behaviour for unknown plugin names.
#include "api.h"
#include <libgen.h>
#include <climits>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <map>
#include "error.h"
#include "scene.h"

namespace pbrt {
Options PbrtOptions;
std::unique_ptr<LoadedScene> lastLoadedScene;

// ---- fileutil.cpp -----------------------------------------------------------
static std::string searchDirectory;
std::string DirectoryContaining(const std::string &filename) {
    char *t = strdup(filename.c_str());
    std::string result = dirname(t);
    free(t);
    return result;
}
void SetSearchDirectory(const std::string &dirname) { searchDirectory = dirname; }
static bool IsAbsolutePath(const std::string &filename) { return !filename.empty() && filename[0] == '/'; }
std::string ResolveFilename(const std::string &filename) {
    if (searchDirectory.empty() || filename.empty()) return filename;
    if (IsAbsolutePath(filename)) return filename;
    if (searchDirectory.back() == '/') return searchDirectory + filename;
    return searchDirectory + "/" + filename;
}
std::string AbsolutePath(const std::string &filename) {
    char full[PATH_MAX];
    if (realpath(filename.c_str(), full)) return std::string(full);
    return filename;
}

// ---- API state (api.cpp:120-330) -------------------------------------------------
namespace {
constexpr int MaxTransforms = 2;
constexpr int StartTransformBits = 1, EndTransformBits = 2, AllTransformsBits = 3;
struct TransformSet {
    Transform t[MaxTransforms];
    Transform &operator[](int i) { return t[i]; }
    const Transform &operator[](int i) const { return t[i]; }
    bool IsAnimated() const {  // t[i] != t[i + 1]: Transform::operator!= compares m AND mInv (transform.h:139-141, api.cpp:101-105)
        return !(t[0].GetMatrix() == t[1].GetMatrix()) || !(t[0].GetInverseMatrix() == t[1].GetInverseMatrix());
    }
};
TransformSet Inverse(const TransformSet &ts) {
    TransformSet r;
    for (int i = 0; i < MaxTransforms; ++i) r.t[i] = Inverse(ts.t[i]);
    return r;
}
struct MaterialInstance { std::string name; int material = -1; ParamSet params; };
struct GraphicsState {
    // named textures: a constant (tex == -1) or a node of renderOptions->textures
    std::map<std::string, PgTexRef> spectrumTextures;
    std::map<std::string, PgTexRef> floatTextures;
    std::map<std::string, MaterialInstance> namedMaterials;
    MaterialInstance currentMaterial;
    ParamSet areaLightParams;
    std::string areaLight;
    bool reverseOrientation = false;
    std::string currentInsideMedium, currentOutsideMedium;  // MediumInterface, api.cpp:1107-1116
};
struct RenderOptions {
    Float transformStartTime = 0, transformEndTime = 1;
    bool refused = false;  // the scene asks for something whose absence would change the image (animated shapes / instances / camera): no frame is rendered
    std::string FilterName = "box"; ParamSet FilterParams;
    std::string FilmName = "image"; ParamSet FilmParams;
    std::string SamplerName = "halton"; ParamSet SamplerParams;
    std::string AcceleratorName = "bvh"; ParamSet AcceleratorParams;
    std::string IntegratorName = "path"; ParamSet IntegratorParams;
    std::string CameraName = "perspective"; ParamSet CameraParams;
    TransformSet CameraToWorld;
    std::vector<GeometricPrimitive> primitives;
    std::map<std::string, std::shared_ptr<ObjectDefinition>> instances;  // api.cpp:129
    std::shared_ptr<ObjectDefinition> currentInstance;
    std::vector<PgLight> lights;           // prim index filled at flatten time
    std::vector<size_t> lightPrimSerial;   // serial number of the emitting primitive
    std::vector<PgMaterial> materials;
    std::vector<PgBxDF> bxdfs;  // the materials' BxDF lists, concatenated
    std::vector<PgTexture> textures;  // texture nodes (Texture "name" ... with a non-constant class)
    std::vector<PgTexturedMaterial> textured;
    std::vector<PgImage> images;      // MIPMaps of the image textures, shared through imageCache (imagemap.cpp:55-59)
    std::vector<float> texels;
    std::vector<float> envTables;     // the infinite lights' Distribution2D tables
    std::vector<PgAlphaMask> alphas;  // alpha / shadow-alpha textures of the meshes that have them
    std::vector<PgMedium> media;      // MakeNamedMedium "homogeneous" / "heterogeneous"
    std::vector<int32_t> mediaGrid;   // per medium: index into grids, -1 = homogeneous
    std::vector<PgDensityGrid> grids; // GridDensityMedium (media/grid.h)
    std::vector<float> gridDensity;
    std::vector<PgBSSRDF> bssrdfs;          // SubsurfaceMaterial / KdSubsurfaceMaterial: one per material of those types
    std::vector<int32_t> materialBssrdf;    // per material (grown on demand): index into bssrdfs, -1 = none
    std::vector<float> bssrdfTables;        // BSSRDFTable(100, 64) per distinct (g, eta)
    std::map<std::pair<Float, Float>, int64_t> bssrdfTableOf;
    std::map<std::string, int> namedMedia;
    std::map<std::string, int> imageCache;
    bool haveScatteringMedia = false;
    bool usesNoise = false;  // a Perlin-noise texture was declared: the scene carries NoisePerm
};
enum class APIState { Uninitialized, OptionsBlock, WorldBlock };
APIState currentApiState = APIState::Uninitialized;
TransformSet curTransform;
uint32_t activeTransformBits = AllTransformsBits;
std::map<std::string, TransformSet> namedCoordinateSystems;
std::unique_ptr<RenderOptions> renderOptions;
GraphicsState graphicsState;
std::vector<GraphicsState> pushedGraphicsStates;
std::vector<TransformSet> pushedTransforms;
std::vector<uint32_t> pushedActiveTransformBits;
}  // namespace

#define VERIFY_INITIALIZED(func)                                                        \
    if (currentApiState == APIState::Uninitialized) {                                   \
        Error("pbrtInit() must be before calling \"%s()\". Ignoring.", func);           \
        return;                                                                         \
    } else /* swallow trailing semicolon */
#define VERIFY_OPTIONS(func)                                                            \
    VERIFY_INITIALIZED(func);                                                           \
    if (currentApiState == APIState::WorldBlock) {                                      \
        Error("Options cannot be set inside world block; \"%s\" not allowed.  Ignoring.", func); \
        return;                                                                         \
    } else /* swallow trailing semicolon */
#define VERIFY_WORLD(func)                                                              \
    VERIFY_INITIALIZED(func);                                                           \
    if (currentApiState == APIState::OptionsBlock) {                                    \
        Error("Scene description must be inside world block; \"%s\" not allowed. Ignoring.", func); \
        return;                                                                         \
    } else /* swallow trailing semicolon */

// ---- materials (api.cpp:537-620; matte.cpp:64-72; plastic.cpp:72-84) ---------------
// Textures are restricted to constants: a "texture" parameter must name a
// constant texture declared with Texture "name" "spectrum|float" "constant".
static PgTexRef constRef(RGB v) { PgTexRef r; r.tex = -1; r.v[0] = v.c[0]; r.v[1] = v.c[1]; r.v[2] = v.c[2]; return r; }
static PgTexRef constRef(Float v) { PgTexRef r; r.tex = -1; r.v[0] = v; r.v[1] = r.v[2] = 0; return r; }
// TextureParams::GetSpectrumTexture / GetFloatTexture (paramset.cpp:720-800): shape parameters first, a named texture
// before a literal value
static PgTexRef spectrumRef(const ParamSet &geom, const ParamSet &mat, const std::string &n, RGB def, const GraphicsState &gs) {
    for (const ParamSet *ps : {&geom, &mat}) {
        std::string tex = ps->FindTexture(n);
        if (!tex.empty()) {
            auto it = gs.spectrumTextures.find(tex);
            if (it != gs.spectrumTextures.end()) return it->second;
            Error("Couldn't find spectrum texture named \"%s\" for parameter \"%s\"", tex.c_str(), n.c_str());
            return constRef(def);  // paramset.cpp:760-764: no further lookup once a texture name was given
        }
        RGB sp;
        if (ps->FindSpectrum(n, &sp)) return constRef(sp);
    }
    return constRef(def);
}
static PgTexRef floatRef(const ParamSet &geom, const ParamSet &mat, const std::string &n, Float def, const GraphicsState &gs) {
    for (const ParamSet *ps : {&geom, &mat}) {
        std::string tex = ps->FindTexture(n);
        if (!tex.empty()) {
            auto it = gs.floatTextures.find(tex);
            if (it != gs.floatTextures.end()) return it->second;
            Error("Couldn't find float texture named \"%s\" for parameter \"%s\"", tex.c_str(), n.c_str());
            return constRef(def);
        }
        const std::vector<Float> *f = ps->FindFloat(n);
        if (f && !f->empty()) return constRef((*f)[0]);
    }
    return constRef(def);
}
static RGB spectrumParam(const ParamSet &geom, const ParamSet &mat, const std::string &n, RGB def, const GraphicsState &gs) {
    PgTexRef r = spectrumRef(geom, mat, n, def, gs);
    return RGB{{r.v[0], r.v[1], r.v[2]}};
}
static Float floatParam(const ParamSet &geom, const ParamSet &mat, const std::string &n, Float def, const GraphicsState &gs) {
    return floatRef(geom, mat, n, def, gs).v[0];
}
// A material is interned with its BxDF list: equal parameters and equal lists share one table entry.
static int internMaterial(PgMaterial m, const std::vector<PgBxDF> &lobes) {
    auto &tab = renderOptions->materials;
    auto &bx = renderOptions->bxdfs;
    m.n_bxdfs = (int)lobes.size();
    for (size_t i = 0; i < tab.size(); ++i) {
        PgMaterial probe = m;
        probe.first_bxdf = tab[i].first_bxdf;
        if (memcmp(&tab[i], &probe, sizeof(m)) != 0) continue;
        if (lobes.empty() || memcmp(&bx[tab[i].first_bxdf], lobes.data(), lobes.size() * sizeof(PgBxDF)) == 0) return (int)i;
    }
    m.first_bxdf = (int)bx.size();
    bx.insert(bx.end(), lobes.begin(), lobes.end());
    tab.push_back(m);
    return (int)tab.size() - 1;
}
// ---- BxDF lists: what each material's ComputeScatteringFunctions adds, with its constant textures evaluated ----------
static RGB rgbClamp(RGB v) { for (int i = 0; i < 3; ++i) v.c[i] = v.c[i] < 0 ? 0 : v.c[i]; return v; }  // Spectrum::Clamp(), spectrum.h:180-186
static RGB rgbMul(RGB a, RGB b) { RGB r; for (int i = 0; i < 3; ++i) r.c[i] = a.c[i] * b.c[i]; return r; }
static bool rgbBlack(RGB v) { return v.c[0] == 0 && v.c[1] == 0 && v.c[2] == 0; }
static PgBxDF lobe(int type) { PgBxDF b; memset(&b, 0, sizeof(b)); b.type = type; b.eta_a = b.eta_b = 1; b.alpha_x = b.alpha_y = 1; b.on_a = 1; return b; }
static void setR(PgBxDF &b, RGB v) { for (int i = 0; i < 3; ++i) b.R[i] = v.c[i]; }
static void setT(PgBxDF &b, RGB v) { for (int i = 0; i < 3; ++i) b.T[i] = v.c[i]; }
static Float RoughnessToAlpha(Float roughness) {  // TrowbridgeReitzDistribution::RoughnessToAlpha, microfacet.h:127-132
    roughness = std::max(roughness, (Float)1e-3);
    Float x = std::log(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
static void setTR(PgBxDF &b, Float ax, Float ay) {  // TrowbridgeReitzDistribution ctor, microfacet.h:109-113
    b.alpha_x = std::max(Float(0.001), ax);
    b.alpha_y = std::max(Float(0.001), ay);
}
static void setDielectric(PgBxDF &b, Float etaA, Float etaB) { b.fresnel = PG_FRESNEL_DIELECTRIC; b.eta_a = etaA; b.eta_b = etaB; }
static PgBxDF lambertOrOrenNayar(RGB r, Float sigmaDeg) {  // matte.cpp:55-61, OrenNayar ctor reflection.h:425-431
    Float sig = Clamp(sigmaDeg, 0, 90);
    PgBxDF b = lobe(sig == 0 ? PG_BXDF_LAMBERT_R : PG_BXDF_OREN_NAYAR);
    setR(b, r);
    if (sig != 0) {
        Float sigma = Radians(sig), sigma2 = sigma * sigma;
        b.on_a = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
        b.on_b = 0.45f * sigma2 / (sigma2 + 0.09f);
    }
    return b;
}
// MetalMaterial's default spectra as the reference's RGB build derives them (Spectrum::FromSampled of metal.cpp's copper
// tables over the CIE curves); printed from the unmodified reference by oracle/ref_probe.cpp
static const RGB kCopperN = {{0.19999069f, 0.92208463f, 1.09987593f}};
static const RGB kCopperK = {{3.90463543f, 2.44763327f, 2.13765264f}};

static bool hasParam(const ParamSet &geom, const ParamSet &mat, const std::string &n) {  // GetFloatTextureOrNull finds something
    return geom.FindFloat(n) || mat.FindFloat(n) || !geom.FindTexture(n).empty() || !mat.FindTexture(n).empty();
}
static int MakeMaterial(const std::string &name, const ParamSet &geom, const ParamSet &mat) {
    PgMaterial m;
    memset(&m, 0, sizeof(m));
    std::vector<PgBxDF> lobes;
    m.bsdf_eta = 1;  // BSDF(si, eta = 1), reflection.h:167
    if (name == "" || name == "none") { m.type = PG_MAT_NONE; return internMaterial(m, lobes); }
    const GraphicsState &gs = graphicsState;
    // Parameter slots of PgTexturedMaterial (include/pbrt_gpu.h), per material kind.  Every parameter read below is also
    // recorded in `tm`; when one of them is a non-constant texture the material is handed over as PG_MAT_TEXTURED and its
    // BxDFs are evaluated per hit, otherwise as the BxDF list built here.
    struct KindSlots { const char *name; int kind; const char *s[5]; const char *f[4]; };
    static const KindSlots kSlots[] = {
        {"matte", PG_KIND_MATTE, {"Kd"}, {"sigma"}}, {"plastic", PG_KIND_PLASTIC, {"Kd", "Ks"}, {"roughness"}},
        {"mirror", PG_KIND_MIRROR, {"Kr"}, {}}, {"glass", PG_KIND_GLASS, {"Kr", "Kt"}, {"uroughness", "vroughness", "eta"}},
        {"uber", PG_KIND_UBER, {"Kd", "Ks", "Kr", "Kt", "opacity"}, {"roughness", "uroughness", "vroughness", "eta"}},
        {"metal", PG_KIND_METAL, {"eta", "k"}, {"roughness", "uroughness", "vroughness"}},
        {"substrate", PG_KIND_SUBSTRATE, {"Kd", "Ks"}, {"uroughness", "vroughness"}},
        {"translucent", PG_KIND_TRANSLUCENT, {"Kd", "Ks", "reflect", "transmit"}, {"roughness"}}, {"mix", PG_KIND_MIX, {"amount"}, {}}};
    const KindSlots *slots = &kSlots[0];  // unknown names fall back to matte (api.cpp:588-591)
    for (const KindSlots &k : kSlots) if (name == k.name) slots = &k;
    PgTexturedMaterial tm;
    memset(&tm, 0, sizeof(tm));
    tm.kind = slots->kind;
    for (int i = 0; i < 5; ++i) tm.s[i].tex = -1;
    for (int i = 0; i < 4; ++i) tm.f[i].tex = -1;
    tm.sub[0] = tm.sub[1] = -1;
    bool anyTexture = false;
    auto specRGB = [&](const char *n, RGB d) {
        PgTexRef r = spectrumRef(geom, mat, n, d, gs);
        for (int i = 0; i < 5; ++i) if (slots->s[i] && !strcmp(slots->s[i], n)) tm.s[i] = r;
        if (r.tex >= 0) anyTexture = true;
        return RGB{{r.v[0], r.v[1], r.v[2]}};
    };
    auto spec = [&](const char *n, Float d) { return specRGB(n, RGB{{d, d, d}}); };
    auto flt = [&](const char *n, Float d) {
        PgTexRef r = floatRef(geom, mat, n, d, gs);
        const char *slotName = !strcmp(n, "index") ? "eta" : n;  // "eta" and its alias "index" share a slot
        for (int i = 0; i < 4; ++i) if (slots->f[i] && !strcmp(slots->f[i], slotName)) tm.f[i] = r;
        if (r.tex >= 0) anyTexture = true;
        return r.v[0];
    };
    auto remapParam = [&]() { return geom.FindOneBool("remaproughness", mat.FindOneBool("remaproughness", true)); };
    // GetFloatTextureOrNull("bumpmap") of every Create*Material: a displacement texture (or a literal float)
    const bool hasBump = hasParam(geom, mat, "bumpmap") && name != "mix";
    PgTexRef bumpRef = constRef(0.f);
    if (hasBump) bumpRef = floatRef(geom, mat, "bumpmap", 0.f, gs);
    if (name == "matte" || (name != "plastic" && name != "mirror" && name != "glass" && name != "uber" && name != "metal" &&
                            name != "substrate" && name != "translucent" && name != "mix" && name != "subsurface" && name != "kdsubsurface")) {
        if (name != "matte") {
            if (name == "hair" || name == "disney" || name == "fourier")
            {   // a material the REFERENCE renders and this build does not: no frame with a stand-in (the scene is refused at WorldEnd)
                Error("Material \"%s\" is outside this build's closed set (matte, plastic, mirror, glass, uber, metal, substrate, translucent, mix, subsurface, kdsubsurface); the scene will not be rendered.", name.c_str());
                renderOptions->refused = true;
            }
            else Warning("Material \"%s\" unknown. Using \"matte\".", name.c_str());  // api.cpp:588-591
        }
        m.type = PG_MAT_MATTE;  // matte.cpp:45-72
        RGB kd = spec("Kd", 0.5f);
        for (int i = 0; i < 3; ++i) m.kd[i] = kd.c[i];
        m.sigma = flt("sigma", 0.f);
        RGB r = rgbClamp(kd);
        if (!rgbBlack(r)) lobes.push_back(lambertOrOrenNayar(r, m.sigma));
    } else if (name == "plastic") {  // plastic.cpp:45-84
        m.type = PG_MAT_PLASTIC;
        RGB kd = spec("Kd", 0.25f), ks = spec("Ks", 0.25f);
        for (int i = 0; i < 3; ++i) { m.kd[i] = kd.c[i]; m.ks[i] = ks.c[i]; }
        m.roughness = flt("roughness", .1f);
        m.remap_roughness = remapParam() ? 1 : 0;
        if (!rgbBlack(rgbClamp(kd))) { PgBxDF b = lobe(PG_BXDF_LAMBERT_R); setR(b, rgbClamp(kd)); lobes.push_back(b); }
        if (!rgbBlack(rgbClamp(ks))) {
            PgBxDF b = lobe(PG_BXDF_MICROFACET_R);
            setR(b, rgbClamp(ks));
            setDielectric(b, 1.5f, 1.f);
            Float rough = m.roughness;
            if (m.remap_roughness) rough = RoughnessToAlpha(rough);
            setTR(b, rough, rough);
            lobes.push_back(b);
        }
    } else if (name == "mirror") {  // mirror.cpp:44-64
        m.type = PG_MAT_MIRROR;
        RGB kr = spec("Kr", 0.9f);
        for (int i = 0; i < 3; ++i) m.kr[i] = kr.c[i];
        if (!rgbBlack(rgbClamp(kr))) { PgBxDF b = lobe(PG_BXDF_SPECULAR_R); setR(b, rgbClamp(kr)); b.fresnel = PG_FRESNEL_NOOP; lobes.push_back(b); }
    } else if (name == "glass") {  // glass.cpp:45-115
        RGB kr = spec("Kr", 1.f), kt = spec("Kt", 1.f);
        for (int i = 0; i < 3; ++i) { m.kr[i] = kr.c[i]; m.kt[i] = kt.c[i]; }
        m.eta = hasParam(geom, mat, "eta") ? flt("eta", 1.5f) : flt("index", 1.5f);
        Float ur = flt("uroughness", 0.f), vr = flt("vroughness", 0.f);
        const bool remap = remapParam();
        const RGB R = rgbClamp(kr), T = rgbClamp(kt);
        m.bsdf_eta = m.eta;
        const bool isSpecular = ur == 0 && vr == 0;
        m.type = isSpecular ? PG_MAT_GLASS : PG_MAT_LOBES;
        if (!(rgbBlack(R) && rgbBlack(T))) {
            if (isSpecular) {  // allowMultipleLobes is true on the path integrator's call (path.cpp:106)
                PgBxDF b = lobe(PG_BXDF_FRESNEL_SPECULAR); setR(b, R); setT(b, T); b.eta_a = 1.f; b.eta_b = m.eta; lobes.push_back(b);
            } else {
                if (remap) { ur = RoughnessToAlpha(ur); vr = RoughnessToAlpha(vr); }
                if (!rgbBlack(R)) { PgBxDF b = lobe(PG_BXDF_MICROFACET_R); setR(b, R); setDielectric(b, 1.f, m.eta); setTR(b, ur, vr); lobes.push_back(b); }
                if (!rgbBlack(T)) { PgBxDF b = lobe(PG_BXDF_MICROFACET_T); setT(b, T); b.eta_a = 1.f; b.eta_b = m.eta; setTR(b, ur, vr); lobes.push_back(b); }
            }
        }
    } else if (name == "subsurface" || name == "kdsubsurface") {
        // CreateSubsurfaceMaterial (subsurface.cpp:93-136) / CreateKdSubsurfaceMaterial (kdsubsurface.cpp:95-123).  The surface BSDF
        // has glass's shape with "eta" as the index (subsurface.cpp:57-86); with constant parameters the BSSRDF's coefficients are
        // evaluated here once (:87-90 / kdsubsurface.cpp:88-92), with a texture (or a bump map) among them per hit; the table is the
        // constructor's ComputeBeamDiffusionBSSRDF(g, eta) (subsurface.h:73-75).
        const bool kdForm = name == "kdsubsurface";
        bool textured = false;
        auto specC = [&](const char *n, RGB d) { PgTexRef r = spectrumRef(geom, mat, n, d, gs); if (r.tex >= 0) textured = true; return r; };
        auto fltC = [&](const char *n, Float d) { PgTexRef r = floatRef(geom, mat, n, d, gs); if (r.tex >= 0) textured = true; return r; };
        auto rgbOf = [](const PgTexRef &r) { return RGB{{r.v[0], r.v[1], r.v[2]}}; };
        RGB sig_a{{.0011f, .0024f, .014f}}, sig_s{{2.55f, 3.21f, 3.77f}};
        Float g = geom.FindOneFloat("g", mat.FindOneFloat("g", 0.0f));
        if (!kdForm) {
            const std::string preset = geom.FindOneString("name", mat.FindOneString("name", ""));
            const bool found = GetMediumScatteringProperties(preset, sig_a.c, sig_s.c);
            if (preset != "") {
                if (!found) Warning("Named material \"%s\" not found.  Using defaults.", preset.c_str());
                else g = 0;  // the database specifies reduced scattering coefficients
            }
        }
        const Float scale = geom.FindOneFloat("scale", mat.FindOneFloat("scale", 1.f));
        const Float eta = geom.FindOneFloat("eta", mat.FindOneFloat("eta", 1.33f));
        // the two coefficient parameters: (sigma_a, sigma_s) or (Kd, mfp)
        const PgTexRef ca = kdForm ? specC("Kd", RGB{{.5f, .5f, .5f}}) : specC("sigma_a", sig_a);
        const PgTexRef cb = kdForm ? specC("mfp", RGB{{1.f, 1.f, 1.f}}) : specC("sigma_s", sig_s);
        const PgTexRef krRef = specC("Kr", RGB{{1.f, 1.f, 1.f}}), ktRef = specC("Kt", RGB{{1.f, 1.f, 1.f}});
        const PgTexRef urRef = fltC("uroughness", 0.f), vrRef = fltC("vroughness", 0.f);
        const bool remap = remapParam();
        mat.ReportUnused();
        const bool perHit = textured || hasBump;
        bool hasBssrdf = true;
        if (!perHit) {
            const RGB kr = rgbOf(krRef), kt = rgbOf(ktRef);
            Float ur = urRef.v[0], vr = vrRef.v[0];
            for (int i = 0; i < 3; ++i) { m.kr[i] = kr.c[i]; m.kt[i] = kt.c[i]; }
            m.eta = eta;
            m.bsdf_eta = eta;  // BSDF(*si, eta)
            m.type = PG_MAT_LOBES;
            const RGB R = rgbClamp(kr), T = rgbClamp(kt);
            const bool isSpecular = ur == 0 && vr == 0;
            hasBssrdf = !(rgbBlack(R) && rgbBlack(T));  // the early return of ComputeScatteringFunctions (subsurface.cpp:55)
            if (hasBssrdf) {
                if (isSpecular) { PgBxDF b = lobe(PG_BXDF_FRESNEL_SPECULAR); setR(b, R); setT(b, T); b.eta_a = 1.f; b.eta_b = eta; lobes.push_back(b); }
                else {
                    if (remap) { ur = RoughnessToAlpha(ur); vr = RoughnessToAlpha(vr); }
                    if (!rgbBlack(R)) { PgBxDF b = lobe(PG_BXDF_MICROFACET_R); setR(b, R); setDielectric(b, 1.f, eta); setTR(b, ur, vr); lobes.push_back(b); }
                    if (!rgbBlack(T)) { PgBxDF b = lobe(PG_BXDF_MICROFACET_T); setT(b, T); b.eta_a = 1.f; b.eta_b = eta; setTR(b, ur, vr); lobes.push_back(b); }
                }
            }
        } else {  // evaluated per hit: glass's parameter slots (Kr, Kt | uroughness, vroughness, eta), bump map
            tm.kind = PG_KIND_GLASS;
            tm.s[0] = krRef; tm.s[1] = ktRef;
            tm.f[0] = urRef; tm.f[1] = vrRef; tm.f[2] = constRef(eta);
            tm.remap_roughness = remap ? 1 : 0;
            tm.has_bump = hasBump ? 1 : 0; tm.bump = bumpRef;
            auto &tab = renderOptions->textured;
            int found = -1;
            for (size_t i = 0; i < tab.size(); ++i) if (memcmp(&tab[i], &tm, sizeof(tm)) == 0) found = (int)i;
            if (found < 0) { found = (int)tab.size(); tab.push_back(tm); }
            m.type = PG_MAT_TEXTURED;
            m.bsdf_eta = 1;
            m.textured_index = found;
        }
        // never interned: the probe rays of Sample_Sp compare Material objects (bssrdf.cpp:301)
        m.n_bxdfs = (int)lobes.size();
        m.first_bxdf = (int)renderOptions->bxdfs.size();
        renderOptions->bxdfs.insert(renderOptions->bxdfs.end(), lobes.begin(), lobes.end());
        renderOptions->materials.push_back(m);
        const int index = (int)renderOptions->materials.size() - 1;
        if (hasBssrdf) {
            PgBSSRDF b;
            memset(&b, 0, sizeof(b));
            b.eta = eta; b.n_rho = 100; b.n_radius = 64;
            b.scale = scale; b.a = ca; b.b = cb;
            b.textured = !perHit ? 0 : (kdForm ? 2 : 1);
            auto key = std::make_pair(g, eta);
            auto it = renderOptions->bssrdfTableOf.find(key);
            if (it == renderOptions->bssrdfTableOf.end()) {
                std::vector<float> table;
                ComputeBeamDiffusionTable(g, eta, b.n_rho, b.n_radius, &table);
                it = renderOptions->bssrdfTableOf.emplace(key, (int64_t)renderOptions->bssrdfTables.size()).first;
                renderOptions->bssrdfTables.insert(renderOptions->bssrdfTables.end(), table.begin(), table.end());
            }
            b.table = it->second;
            // the constant parts' coefficients (final when nothing is textured)
            Float sa[3], ss[3];
            const RGB ac = rgbClamp(rgbOf(ca)), bc = rgbClamp(rgbOf(cb));
            if (kdForm) {
                Float mf[3];
                for (int i = 0; i < 3; ++i) mf[i] = scale * bc.c[i];
                SubsurfaceFromDiffuse(renderOptions->bssrdfTables.data() + b.table, b.n_rho, b.n_radius, ac.c, mf, sa, ss);
            } else for (int i = 0; i < 3; ++i) { sa[i] = scale * ac.c[i]; ss[i] = scale * bc.c[i]; }
            for (int i = 0; i < 3; ++i) {  // the TabulatedBSSRDF constructor, bssrdf.h:146-150
                b.sigma_t[i] = sa[i] + ss[i];
                b.rho[i] = b.sigma_t[i] != 0 ? (ss[i] / b.sigma_t[i]) : 0;
            }
            b.match_material = index;
            renderOptions->materialBssrdf.resize(index + 1, -1);
            renderOptions->materialBssrdf[index] = (int)renderOptions->bssrdfs.size();
            renderOptions->bssrdfs.push_back(b);
        }
        return index;
    } else if (name == "uber") {  // uber.cpp:45-128
        m.type = PG_MAT_LOBES;
        RGB Kd = spec("Kd", 0.25f), Ks = spec("Ks", 0.25f), Kr = spec("Kr", 0.f), Kt = spec("Kt", 0.f);
        Float roughness = flt("roughness", .1f);
        const bool hasU = hasParam(geom, mat, "uroughness"), hasV = hasParam(geom, mat, "vroughness");
        Float uro = hasU ? flt("uroughness", 0.f) : 0.f, vro = hasV ? flt("vroughness", 0.f) : 0.f;
        Float e = hasParam(geom, mat, "eta") ? flt("eta", 1.5f) : flt("index", 1.5f);
        RGB opacity = spec("opacity", 1.f);
        const bool remap = remapParam();
        RGB op = rgbClamp(opacity), t;
        for (int i = 0; i < 3; ++i) t.c[i] = -op.c[i] + 1.f;
        t = rgbClamp(t);
        if (!rgbBlack(t)) { PgBxDF b = lobe(PG_BXDF_SPECULAR_T); setT(b, t); b.eta_a = 1.f; b.eta_b = 1.f; lobes.push_back(b); m.bsdf_eta = 1.f; }
        else m.bsdf_eta = e;
        RGB kd = rgbMul(op, rgbClamp(Kd));
        if (!rgbBlack(kd)) { PgBxDF b = lobe(PG_BXDF_LAMBERT_R); setR(b, kd); lobes.push_back(b); }
        RGB ks = rgbMul(op, rgbClamp(Ks));
        if (!rgbBlack(ks)) {
            Float roughu = hasU ? uro : roughness, roughv = hasV ? vro : roughu;
            if (remap) { roughu = RoughnessToAlpha(roughu); roughv = RoughnessToAlpha(roughv); }
            PgBxDF b = lobe(PG_BXDF_MICROFACET_R); setR(b, ks); setDielectric(b, 1.f, e); setTR(b, roughu, roughv); lobes.push_back(b);
        }
        RGB kr = rgbMul(op, rgbClamp(Kr));
        if (!rgbBlack(kr)) { PgBxDF b = lobe(PG_BXDF_SPECULAR_R); setR(b, kr); setDielectric(b, 1.f, e); lobes.push_back(b); }
        RGB kt = rgbMul(op, rgbClamp(Kt));
        if (!rgbBlack(kt)) { PgBxDF b = lobe(PG_BXDF_SPECULAR_T); setT(b, kt); b.eta_a = 1.f; b.eta_b = e; lobes.push_back(b); }
    } else if (name == "metal") {  // metal.cpp:61-136
        m.type = PG_MAT_LOBES;
        RGB eta = specRGB("eta", kCopperN), k = specRGB("k", kCopperK);
        Float roughness = flt("roughness", .01f);
        const bool hasU = hasParam(geom, mat, "uroughness"), hasV = hasParam(geom, mat, "vroughness");
        Float uRough = hasU ? flt("uroughness", 0.f) : roughness, vRough = hasV ? flt("vroughness", 0.f) : roughness;
        if (remapParam()) { uRough = RoughnessToAlpha(uRough); vRough = RoughnessToAlpha(vRough); }
        PgBxDF b = lobe(PG_BXDF_MICROFACET_R);
        setR(b, RGB{{1.f, 1.f, 1.f}});
        b.fresnel = PG_FRESNEL_CONDUCTOR;
        for (int i = 0; i < 3; ++i) { b.cond_eta[i] = eta.c[i]; b.cond_k[i] = k.c[i]; }
        setTR(b, uRough, vRough);
        lobes.push_back(b);
    } else if (name == "substrate") {  // substrate.cpp:45-82
        m.type = PG_MAT_LOBES;
        RGB d = rgbClamp(spec("Kd", .5f)), sp = rgbClamp(spec("Ks", .5f));
        Float roughu = flt("uroughness", .1f), roughv = flt("vroughness", .1f);
        const bool remap = remapParam();
        if (!rgbBlack(d) || !rgbBlack(sp)) {
            if (remap) { roughu = RoughnessToAlpha(roughu); roughv = RoughnessToAlpha(roughv); }
            PgBxDF b = lobe(PG_BXDF_FRESNEL_BLEND); setR(b, d); setT(b, sp); setTR(b, roughu, roughv); lobes.push_back(b);
        }
    } else if (name == "translucent") {  // translucent.cpp:45-99
        m.type = PG_MAT_LOBES;
        const Float eta = 1.5f;
        m.bsdf_eta = eta;
        RGB Kd = spec("Kd", 0.25f), Ks = spec("Ks", 0.25f), refl = spec("reflect", 0.5f), trans = spec("transmit", 0.5f);
        Float rough = flt("roughness", .1f);
        const bool remap = remapParam();
        RGB r = rgbClamp(refl), t = rgbClamp(trans);
        if (!(rgbBlack(r) && rgbBlack(t))) {
            RGB kd = rgbClamp(Kd);
            if (!rgbBlack(kd)) {
                if (!rgbBlack(r)) { PgBxDF b = lobe(PG_BXDF_LAMBERT_R); setR(b, rgbMul(r, kd)); lobes.push_back(b); }
                if (!rgbBlack(t)) { PgBxDF b = lobe(PG_BXDF_LAMBERT_T); setT(b, rgbMul(t, kd)); lobes.push_back(b); }
            }
            RGB ks = rgbClamp(Ks);
            if (!rgbBlack(ks) && (!rgbBlack(r) || !rgbBlack(t))) {
                if (remap) rough = RoughnessToAlpha(rough);
                if (!rgbBlack(r)) { PgBxDF b = lobe(PG_BXDF_MICROFACET_R); setR(b, rgbMul(r, ks)); setDielectric(b, 1.f, eta); setTR(b, rough, rough); lobes.push_back(b); }
                if (!rgbBlack(t)) { PgBxDF b = lobe(PG_BXDF_MICROFACET_T); setT(b, rgbMul(t, ks)); b.eta_a = 1.f; b.eta_b = eta; setTR(b, rough, rough); lobes.push_back(b); }
            }
        }
    } else if (name == "mix") {  // api.cpp:556-577, mixmat.cpp:45-73
        m.type = PG_MAT_LOBES;
        int sub[2];
        const char *pn[2] = {"namedmaterial1", "namedmaterial2"};
        for (int j = 0; j < 2; ++j) {
            std::string mn = geom.FindOneString(pn[j], mat.FindOneString(pn[j], ""));
            auto it = gs.namedMaterials.find(mn);
            if (it == gs.namedMaterials.end()) {
                Error("Named material \"%s\" undefined.  Using \"matte\"", mn.c_str());
                sub[j] = MakeMaterial("matte", geom, mat);
            } else sub[j] = it->second.material;
        }
        RGB s1 = rgbClamp(spec("amount", 0.5f)), s2;
        for (int i = 0; i < 3; ++i) s2.c[i] = 1.f - s1.c[i];
        s2 = rgbClamp(s2);
        bool bad = false;
        for (int j = 0; j < 2; ++j) {
            const PgMaterial sm = renderOptions->materials[sub[j]];  // copy: the table may grow
            if (sm.type == PG_MAT_NONE) { Error("mix: a \"none\" material cannot be mixed; ignoring it."); continue; }

            if (j == 0) m.bsdf_eta = sm.bsdf_eta;  // si->bsdf stays m1's BSDF
            const RGB &sc = j == 0 ? s1 : s2;
            for (int i = 0; i < sm.n_bxdfs; ++i) {
                PgBxDF b = renderOptions->bxdfs[sm.first_bxdf + i];
                if (b.n_scales >= PG_MAX_BXDF_SCALES) { bad = true; continue; }
                for (int c = 0; c < 3; ++c) b.scale[b.n_scales][c] = sc.c[c];
                ++b.n_scales;
                if ((int)lobes.size() < PG_MAX_BXDFS) lobes.push_back(b); else bad = true;
            }
        }
        if (bad) {  // (the reference's BSDF::Add stops the process at more than MaxBxDFs, reflection.h:176-179; deeper nesting it renders: no frame with lobes left out)
            Error("mix: more than %d BxDFs or more than %d nested mixes; the scene will not be rendered.", PG_MAX_BXDFS, PG_MAX_BXDF_SCALES);
            renderOptions->refused = true;
        }
        tm.sub[0] = sub[0]; tm.sub[1] = sub[1];
        for (int j = 0; j < 2; ++j) if (renderOptions->materials[sub[j]].type == PG_MAT_TEXTURED) anyTexture = true;
        // si->bssrdf stays what the FIRST component's ComputeScatteringFunctions set (mixmat.cpp:52-53; the second one works on a copy
        // of si): the mix then carries that component's BSSRDF, whose probe rays look for the component itself.  The mix is its own
        // table entry (never merged), like every material with a BSSRDF.
        auto bssrdfOf = [&](int mi) { return mi < (int)renderOptions->materialBssrdf.size() ? renderOptions->materialBssrdf[mi] : -1; };
        if (bssrdfOf(sub[0]) >= 0) {
            const PgBSSRDF comp = renderOptions->bssrdfs[bssrdfOf(sub[0])];
            if (comp.textured || anyTexture) {
                Error("mix: a textured subsurface component (or a textured mix around one) is outside this build's closed set; the scene will not be rendered.");
                renderOptions->refused = true;  // (no frame with the mix's BSSRDF left out)
            } else {
                mat.ReportUnused();
                m.n_bxdfs = (int)lobes.size();
                m.first_bxdf = (int)renderOptions->bxdfs.size();
                renderOptions->bxdfs.insert(renderOptions->bxdfs.end(), lobes.begin(), lobes.end());
                renderOptions->materials.push_back(m);
                const int index = (int)renderOptions->materials.size() - 1;
                renderOptions->materialBssrdf.resize(index + 1, -1);
                renderOptions->materialBssrdf[index] = (int)renderOptions->bssrdfs.size();
                renderOptions->bssrdfs.push_back(comp);  // (match_material stays the component)
                return index;
            }
        }
    }
    mat.ReportUnused();
    if (hasBump) anyTexture = true;
    tm.has_bump = hasBump ? 1 : 0; tm.bump = bumpRef;
    if (anyTexture) {  // a texture among the parameters: ComputeScatteringFunctions runs per hit on the device
        tm.has_u = hasParam(geom, mat, "uroughness") ? 1 : 0;
        tm.has_v = hasParam(geom, mat, "vroughness") ? 1 : 0;
        tm.remap_roughness = geom.FindOneBool("remaproughness", mat.FindOneBool("remaproughness", true)) ? 1 : 0;
        if (slots->kind == PG_KIND_GLASS && !hasParam(geom, mat, "eta") && !hasParam(geom, mat, "index")) tm.f[2] = constRef(1.5f);
        if (slots->kind == PG_KIND_UBER && !hasParam(geom, mat, "eta") && !hasParam(geom, mat, "index")) tm.f[3] = constRef(1.5f);
        PgMaterial t;
        memset(&t, 0, sizeof(t));
        t.type = PG_MAT_TEXTURED;
        t.bsdf_eta = 1;
        auto &tab = renderOptions->textured;
        int found = -1;
        for (size_t i = 0; i < tab.size(); ++i) if (memcmp(&tab[i], &tm, sizeof(tm)) == 0) found = (int)i;
        if (found < 0) { found = (int)tab.size(); tab.push_back(tm); }
        t.textured_index = found;
        return internMaterial(t, std::vector<PgBxDF>());
    }
    return internMaterial(m, lobes);
}

// api.cpp:1427-1470
static bool shapeMaySetMaterialParameters(const ParamSet &ps) {
    for (auto &kv : ps.textures)
        if (kv.first != "alpha" && kv.first != "shadowalpha") return true;
    for (auto &kv : ps.floats)
        if (kv.second.v.size() == 1 && kv.first != "radius") return true;
    for (auto &kv : ps.strings)
        if (kv.second.v.size() == 1 && kv.first != "filename" && kv.first != "type" && kv.first != "scheme") return true;
    for (auto &kv : ps.bools) if (kv.second.v.size() == 1) return true;
    for (auto &kv : ps.ints) if (kv.second.v.size() == 1) return true;
    for (auto &kv : ps.point2s) if (kv.second.v.size() == 2) return true;
    for (auto &kv : ps.point3s) if (kv.second.v.size() == 3) return true;
    for (auto &kv : ps.vector3s) if (kv.second.v.size() == 3) return true;
    for (auto &kv : ps.normals) if (kv.second.v.size() == 3) return true;
    for (auto &kv : ps.spectra) if (kv.second.v.size() == 3) return true;
    return false;
}
static int GetMaterialForShape(const ParamSet &shapeParams) {  // api.cpp:1472-1487
    if (shapeMaySetMaterialParameters(shapeParams))
        return MakeMaterial(graphicsState.currentMaterial.name, shapeParams, graphicsState.currentMaterial.params);
    return graphicsState.currentMaterial.material;
}

// ---- API functions ----------------------------------------------------------
void pbrtInit(const Options &opt) {  // api.cpp:871-886
    PbrtOptions = opt;
    quietWarnings = opt.quiet;
    if (currentApiState != APIState::Uninitialized) Error("pbrtInit() has already been called.");
    currentApiState = APIState::OptionsBlock;
    renderOptions.reset(new RenderOptions);
    graphicsState = GraphicsState();
    curTransform = TransformSet();
    activeTransformBits = AllTransformsBits;
    namedCoordinateSystems.clear();
    pushedGraphicsStates.clear(); pushedTransforms.clear(); pushedActiveTransformBits.clear();
}
void pbrtCleanup() {  // api.cpp:888-897
    if (currentApiState == APIState::Uninitialized) Error("pbrtCleanup() called without pbrtInit().");
    else if (currentApiState == APIState::WorldBlock) Error("pbrtCleanup() called while inside world block.");
    currentApiState = APIState::Uninitialized;
    renderOptions.reset();
}
// The CTM directives (api.cpp:899-963): each acts on the transforms ActiveTransform selected -- the start one, the end one or both.
// `replace`: Identity / Transform set them; everything else post-multiplies.
static void ActOnActiveTransforms(const char *directive, const Transform &t, bool replace = false) {
    VERIFY_INITIALIZED(directive);
    for (int which = 0; which < MaxTransforms; ++which) {
        if (!(activeTransformBits & (1 << which))) continue;
        curTransform[which] = replace ? t : curTransform[which] * t;
    }
}
static Transform FromColumnMajor(const Float tr[16]) {  // the file gives a matrix column by column (api.cpp:918-921)
    Matrix4x4 m;
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m.m[r][c] = tr[4 * c + r];
    return Transform(m);
}
void pbrtIdentity() { ActOnActiveTransforms("Identity", Transform(), true); }
void pbrtTransform(Float tr[16]) { ActOnActiveTransforms("Transform", FromColumnMajor(tr), true); }
void pbrtConcatTransform(Float tr[16]) { ActOnActiveTransforms("ConcatTransform", FromColumnMajor(tr)); }
void pbrtTranslate(Float dx, Float dy, Float dz) { ActOnActiveTransforms("Translate", Translate(Vector3f(dx, dy, dz))); }
void pbrtRotate(Float angle, Float dx, Float dy, Float dz) { ActOnActiveTransforms("Rotate", Rotate(angle, Vector3f(dx, dy, dz))); }
void pbrtScale(Float sx, Float sy, Float sz) { ActOnActiveTransforms("Scale", Scale(sx, sy, sz)); }
void pbrtLookAt(Float ex, Float ey, Float ez, Float lx, Float ly, Float lz, Float ux, Float uy, Float uz) {
    ActOnActiveTransforms("LookAt", LookAt(Point3f(ex, ey, ez), Point3f(lx, ly, lz), Vector3f(ux, uy, uz)));
}
void pbrtCoordinateSystem(const std::string &name) { VERIFY_INITIALIZED("CoordinateSystem"); namedCoordinateSystems[name] = curTransform; }
void pbrtCoordSysTransform(const std::string &name) {
    VERIFY_INITIALIZED("CoordSysTransform");
    if (namedCoordinateSystems.find(name) != namedCoordinateSystems.end()) curTransform = namedCoordinateSystems[name];
    else Warning("Couldn't find named coordinate system \"%s\"", name.c_str());
}
void pbrtActiveTransformAll() { activeTransformBits = AllTransformsBits; }
void pbrtActiveTransformEndTime() { activeTransformBits = EndTransformBits; }
void pbrtActiveTransformStartTime() { activeTransformBits = StartTransformBits; }
void pbrtTransformTimes(Float start, Float end) {
    VERIFY_OPTIONS("TransformTimes");
    renderOptions->transformStartTime = start;
    renderOptions->transformEndTime = end;
}
void pbrtPixelFilter(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("PixelFilter"); renderOptions->FilterName = name; renderOptions->FilterParams = params;
}
void pbrtFilm(const std::string &type, const ParamSet &params) {
    VERIFY_OPTIONS("Film"); renderOptions->FilmParams = params; renderOptions->FilmName = type;
}
void pbrtSampler(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("Sampler"); renderOptions->SamplerName = name; renderOptions->SamplerParams = params;
}
void pbrtAccelerator(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("Accelerator"); renderOptions->AcceleratorName = name; renderOptions->AcceleratorParams = params;
}
void pbrtIntegrator(const std::string &name, const ParamSet &params) {
    VERIFY_OPTIONS("Integrator"); renderOptions->IntegratorName = name; renderOptions->IntegratorParams = params;
}
void pbrtCamera(const std::string &name, const ParamSet &params) {  // api.cpp:1076-1087
    VERIFY_OPTIONS("Camera");
    renderOptions->CameraName = name;
    renderOptions->CameraParams = params;
    renderOptions->CameraToWorld = Inverse(curTransform);
    namedCoordinateSystems["camera"] = renderOptions->CameraToWorld;
}
void pbrtMakeNamedMedium(const std::string &name, const ParamSet &params) {  // api.cpp:1087-1105, MakeMedium :681-727
    VERIFY_INITIALIZED("MakeNamedMedium");
    std::string type = params.FindOneString("type", "");
    if (type == "") { Error("No parameter string \"type\" found in MakeNamedMedium"); return; }
    RGB sig_a{{.0011f, .0024f, .014f}}, sig_s{{2.55f, 3.21f, 3.77f}};
    std::string preset = params.FindOneString("preset", "");
    if (preset != "" && !GetMediumScatteringProperties(preset, sig_a.c, sig_s.c)) Warning("Material preset \"%s\" not found.  Using defaults.", preset.c_str());
    Float scale = params.FindOneFloat("scale", 1.f);
    Float g = params.FindOneFloat("g", 0.0f);
    sig_a = params.FindOneSpectrum("sigma_a", sig_a);
    sig_s = params.FindOneSpectrum("sigma_s", sig_s);
    if (type != "homogeneous" && type != "heterogeneous") {
        Warning("Medium \"%s\" unknown.", type.c_str());
        params.ReportUnused();
        return;
    }
    PgMedium m;
    for (int i = 0; i < 3; ++i) {
        m.sigma_a[i] = sig_a.c[i] * scale; m.sigma_s[i] = sig_s.c[i] * scale;
        m.sigma_t[i] = m.sigma_s[i] + m.sigma_a[i];  // HomogeneousMedium ctor, homogeneous.h:52-56; GridDensityMedium: grid.h:64
    }
    m.g = g;
    int grid = -1;
    if (type == "heterogeneous") {  // MakeMedium, api.cpp:700-722 + the GridDensityMedium constructor, grid.h:49-73
        const std::vector<Float> *data = params.FindFloat("density");
        if (!data) { Error("No \"density\" values provided for heterogeneous medium?"); return; }
        const int nx = params.FindOneInt("nx", 1), ny = params.FindOneInt("ny", 1), nz = params.FindOneInt("nz", 1);
        const Point3f p0 = params.FindOnePoint3f("p0", Point3f(0.f, 0.f, 0.f)), p1 = params.FindOnePoint3f("p1", Point3f(1.f, 1.f, 1.f));
        // (the reference multiplies the three ints unchecked; counts that do not fit are an error here, never an overflow)
        const long long nVox = nx > 0 && ny > 0 && nz > 0 ? (long long)nx * ny * nz : -1;
        if (nVox < 0 || nVox > 0x7fffffffLL || (long long)data->size() != nVox) {
            Error("GridDensityMedium has %d density values; expected nx*ny*nz = %lld", (int)data->size(), nVox < 0 ? 0LL : nVox);
            return;
        }
        const Transform data2Medium = Translate(Vector3f(p0.x, p0.y, p0.z)) * Scale(p1.x - p0.x, p1.y - p0.y, p1.z - p0.z);
        const Transform worldToMedium = Inverse(curTransform[0] * data2Medium);
        PgDensityGrid gd;
        gd.nx = nx; gd.ny = ny; gd.nz = nz; gd.reserved = 0;
        gd.density_offset = (int64_t)renderOptions->gridDensity.size();
        gd.sigma_t = m.sigma_a[0] + m.sigma_s[0];
        if (m.sigma_t[1] != m.sigma_t[0] || m.sigma_t[2] != m.sigma_t[0])
            Error("GridDensityMedium requires a spectrally uniform attenuation coefficient!");  // (as the reference: reported, then used)
        Float maxDensity = 0;
        for (Float v : *data) maxDensity = std::max(maxDensity, v);
        gd.inv_max_density = 1 / maxDensity;
        const Matrix4x4 &w2m = worldToMedium.GetMatrix();
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) gd.world_to_medium[4 * r + c] = w2m.m[r][c];
        renderOptions->gridDensity.insert(renderOptions->gridDensity.end(), data->begin(), data->end());
        grid = (int)renderOptions->grids.size();
        renderOptions->grids.push_back(gd);
    }
    params.ReportUnused();
    renderOptions->namedMedia[name] = (int)renderOptions->media.size();
    renderOptions->media.push_back(m);
    renderOptions->mediaGrid.push_back(grid);
}
void pbrtMediumInterface(const std::string &insideName, const std::string &outsideName) {
    VERIFY_INITIALIZED("MediumInterface");
    graphicsState.currentInsideMedium = insideName;
    graphicsState.currentOutsideMedium = outsideName;
    renderOptions->haveScatteringMedia = true;
}
// GraphicsState::CreateMediumInterface, api.cpp:1492-1511: (inside, outside) as indices into media, -1 = none
static void CreateMediumInterface(int *inside, int *outside) {
    *inside = *outside = -1;
    const std::string *names[2] = {&graphicsState.currentInsideMedium, &graphicsState.currentOutsideMedium};
    int *dst[2] = {inside, outside};
    for (int k = 0; k < 2; ++k)
        if (*names[k] != "") {
            auto it = renderOptions->namedMedia.find(*names[k]);
            if (it != renderOptions->namedMedia.end()) *dst[k] = it->second;
            else Error("Named medium \"%s\" undefined.", names[k]->c_str());
        }
}
void pbrtWorldBegin() {  // api.cpp:1118-1126
    VERIFY_OPTIONS("WorldBegin");
    currentApiState = APIState::WorldBlock;
    for (int i = 0; i < MaxTransforms; ++i) curTransform[i] = Transform();
    activeTransformBits = AllTransformsBits;
    namedCoordinateSystems["world"] = curTransform;
    // default material: matte (GraphicsState ctor, api.cpp:1596-1601 / :332-340)
    ParamSet empty;
    graphicsState.currentMaterial.name = "matte";
    graphicsState.currentMaterial.params = empty;
    graphicsState.currentMaterial.material = MakeMaterial("matte", empty, empty);
}
void pbrtAttributeBegin() {
    VERIFY_WORLD("AttributeBegin");
    pushedGraphicsStates.push_back(graphicsState);
    pushedTransforms.push_back(curTransform);
    pushedActiveTransformBits.push_back(activeTransformBits);
}
void pbrtAttributeEnd() {
    VERIFY_WORLD("AttributeEnd");
    if (!pushedGraphicsStates.size()) { Error("Unmatched pbrtAttributeEnd() encountered. Ignoring it."); return; }
    graphicsState = std::move(pushedGraphicsStates.back()); pushedGraphicsStates.pop_back();
    // a TransformEnd inside the block may already have taken the transform this block pushed (the reference reads the
    // empty stack's back() there, api.cpp:1158-1163): report it, keep the current transform
    if (pushedTransforms.empty() || pushedActiveTransformBits.empty()) {
        Error("AttributeEnd: the transform saved by its AttributeBegin was popped by an unmatched TransformEnd. Keeping the current transform.");
        return;
    }
    curTransform = pushedTransforms.back(); pushedTransforms.pop_back();
    activeTransformBits = pushedActiveTransformBits.back(); pushedActiveTransformBits.pop_back();
}
void pbrtTransformBegin() {
    VERIFY_WORLD("TransformBegin");
    pushedTransforms.push_back(curTransform);
    pushedActiveTransformBits.push_back(activeTransformBits);
}
void pbrtTransformEnd() {
    VERIFY_WORLD("TransformEnd");
    if (!pushedTransforms.size()) { Error("Unmatched pbrtTransformEnd() encountered. Ignoring it."); return; }
    curTransform = pushedTransforms.back(); pushedTransforms.pop_back();
    activeTransformBits = pushedActiveTransformBits.back(); pushedActiveTransformBits.pop_back();
}
// TextureMapping2D from the texture's parameters: the block every Create*Texture with a 2D mapping repeats (e.g.
// checkerboard.cpp:52-75)
static void readMapping2D(const ParamSet &tp, const Transform &tex2world, PgTexture *t) {
    std::string type = tp.FindOneString("mapping", "uv");
    t->mapping = PG_MAP_UV; t->su = t->sv = 1; t->du = t->dv = 0;
    const Matrix4x4 w2t = Inverse(tex2world).GetMatrix();  // a copy: the inverse is a temporary
    for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) t->w2t[4 * r + c] = w2t.m[r][c];
    if (type == "uv") {
        t->su = tp.FindOneFloat("uscale", 1.); t->sv = tp.FindOneFloat("vscale", 1.);
        t->du = tp.FindOneFloat("udelta", 0.); t->dv = tp.FindOneFloat("vdelta", 0.);
    } else if (type == "spherical") t->mapping = PG_MAP_SPHERICAL;
    else if (type == "cylindrical") t->mapping = PG_MAP_CYLINDRICAL;
    else if (type == "planar") {
        t->mapping = PG_MAP_PLANAR;
        const std::vector<Float> *v1 = tp.FindVector3f("v1"), *v2 = tp.FindVector3f("v2");
        const Float d1[3] = {1, 0, 0}, d2[3] = {0, 1, 0};
        for (int i = 0; i < 3; ++i) { t->vs[i] = (v1 && v1->size() >= 3) ? (*v1)[i] : d1[i]; t->vt[i] = (v2 && v2->size() >= 3) ? (*v2)[i] : d2[i]; }
        t->du = tp.FindOneFloat("udelta", 0.f); t->dv = tp.FindOneFloat("vdelta", 0.f);
    } else Error("2D texture mapping \"%s\" unknown", type.c_str());
}
void pbrtTexture(const std::string &name, const std::string &type, const std::string &texname, const ParamSet &params) {
    VERIFY_WORLD("Texture");  // api.cpp:1183-1238; MakeFloatTexture / MakeSpectrumTexture, api.cpp:603-679
    const bool isFloat = type == "float";
    if (!isFloat && type != "color" && type != "spectrum") { Error("Texture type \"%s\" unknown.", type.c_str()); return; }
    auto &table = isFloat ? graphicsState.floatTextures : graphicsState.spectrumTextures;
    if (table.count(name)) Warning("Texture \"%s\" being redefined", name.c_str());
    // api.cpp:1031 WARN_IF_ANIMATED_TRANSFORM("Texture"): the reference itself gives textures the start transform
    if (curTransform.IsAnimated()) Warning("Animated transformations set; ignoring for \"Texture\" and using the start transform only");
    const GraphicsState &gs = graphicsState;
    ParamSet none;
    auto operand = [&](const char *n, Float d) { return isFloat ? floatRef(params, none, n, d, gs) : spectrumRef(params, none, n, RGB{{d, d, d}}, gs); };
    PgTexture t;
    memset(&t, 0, sizeof(t));
    t.is_float = isFloat ? 1 : 0;
    t.tex1.tex = t.tex2.tex = t.amount.tex = -1;
    PgTexRef ref;
    ref.tex = -1; ref.v[0] = ref.v[1] = ref.v[2] = 0;
    if (texname == "constant") {  // constant.cpp:40-50
        if (isFloat) ref = constRef(params.FindOneFloat("value", 1.f));
        else ref = constRef(params.FindOneSpectrum("value", RGB{{1.f, 1.f, 1.f}}));
        table[name] = ref;
        params.ReportUnused();
        return;
    } else if (texname == "scale") {
        t.type = PG_TEX_SCALE; t.tex1 = operand("tex1", 1.f); t.tex2 = operand("tex2", 1.f);
    } else if (texname == "mix") {
        t.type = PG_TEX_MIX; t.tex1 = operand("tex1", 0.f); t.tex2 = operand("tex2", 1.f);
        t.amount = floatRef(params, none, "amount", 0.5f, gs);
    } else if (texname == "checkerboard") {
        int dim = params.FindOneInt("dimension", 2);
        if (dim != 2 && dim != 3) { Error("%d dimensional checkerboard texture not supported", dim); return; }
        t.tex1 = operand("tex1", 1.f); t.tex2 = operand("tex2", 0.f);
        readMapping2D(dim == 2 ? params : none, curTransform[0], &t);
        if (dim == 2) {
            t.type = PG_TEX_CHECKERBOARD_2D;
            std::string aa = params.FindOneString("aamode", "closedform");
            if (aa == "none") t.aa_none = 1;
            else if (aa != "closedform") Warning("Antialiasing mode \"%s\" not understood by Checkerboard2DTexture; using \"closedform\"", aa.c_str());
        } else {  // checkerboard.cpp:92: IdentityMapping3D(tex2world) -- the mapping's "WorldToTexture" IS tex2world there
            t.type = PG_TEX_CHECKERBOARD_3D;
            const Matrix4x4 &m = curTransform[0].GetMatrix();
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) t.w2t[4 * r + c] = m.m[r][c];
        }
    } else if (texname == "uv" && !isFloat) {
        t.type = PG_TEX_UV;
        readMapping2D(params, curTransform[0], &t);
    } else if (texname == "imagemap") {  // imagemap.cpp:103-188, GetTexture :52-101
        t.type = PG_TEX_IMAGEMAP;
        readMapping2D(params, curTransform[0], &t);
        const Float maxAniso = params.FindOneFloat("maxanisotropy", 8.f);
        const bool trilerp = params.FindOneBool("trilinear", false);
        const std::string wrap = params.FindOneString("wrap", "repeat");
        const int wrapMode = wrap == "black" ? 1 : (wrap == "clamp" ? 2 : 0);
        const Float scale = params.FindOneFloat("scale", 1.f);
        const std::string filename = AbsolutePath(ResolveFilename(params.FindOneString("filename", "")));
        const bool gamma = params.FindOneBool("gamma", ImageGammaDefault(filename));
        char key[64];
        snprintf(key, sizeof(key), "|%d|%d|%a|%d|%a|%d", isFloat ? 1 : 0, trilerp ? 1 : 0, (double)maxAniso, wrapMode, (double)scale, gamma ? 1 : 0);
        const std::string cacheKey = filename + key;
        auto it = renderOptions->imageCache.find(cacheKey);
        if (it != renderOptions->imageCache.end()) t.image = it->second;
        else {
            int xres = 0, yres = 0;
            std::vector<RGB> texels;
            if (!ReadImage(filename, &xres, &yres, &texels)) {
                Warning("Creating a constant grey texture to replace \"%s\".", filename.c_str());
                xres = yres = 1;
                texels.assign(1, RGB{{0.5f, 0.5f, 0.5f}});
            }
            for (int y = 0; y < yres / 2; ++y)  // flip in y: texture space has (0,0) at the lower left corner
                for (int x = 0; x < xres; ++x) std::swap(texels[(size_t)y * xres + x], texels[(size_t)(yres - 1 - y) * xres + x]);
            auto invGamma = [](Float value) {  // InverseGammaCorrect, pbrt.h:298-301
                if (value <= 0.04045f) return value * 1.f / 12.92f;
                return std::pow((value + 0.055f) * 1.f / 1.055f, (Float)2.4f);
            };
            const int nc = isFloat ? 1 : 3;
            std::vector<float> conv((size_t)xres * yres * nc);
            for (size_t i = 0; i < texels.size(); ++i) {  // convertIn, imagemap.h:98-107
                if (isFloat) {
                    const Float yv = 0.212671f * texels[i].c[0] + 0.715160f * texels[i].c[1] + 0.072169f * texels[i].c[2];  // RGBSpectrum::y()
                    conv[i] = scale * (gamma ? invGamma(yv) : yv);
                } else for (int c = 0; c < 3; ++c) conv[3 * i + c] = scale * (gamma ? invGamma(texels[i].c[c]) : texels[i].c[c]);
            }
            PgImage img;
            memset(&img, 0, sizeof(img));
            img.is_float = isFloat ? 1 : 0; img.wrap = wrapMode; img.trilinear = trilerp ? 1 : 0; img.max_anisotropy = maxAniso;
            BuildMIPMap(xres, yres, nc, conv, wrapMode, &img, &renderOptions->texels);
            t.image = (int)renderOptions->images.size();
            renderOptions->images.push_back(img);
            renderOptions->imageCache[cacheKey] = t.image;
        }
    } else if (texname == "bilerp") {
        t.type = PG_TEX_BILERP;
        readMapping2D(params, curTransform[0], &t);
        const char *names[4] = {"v00", "v01", "v10", "v11"};
        float *dst[4] = {t.v00, t.v01, t.v10, t.v11};
        const Float defs[4] = {0.f, 1.f, 0.f, 1.f};
        for (int k = 0; k < 4; ++k) {
            if (isFloat) { dst[k][0] = params.FindOneFloat(names[k], defs[k]); dst[k][1] = dst[k][2] = 0; }
            else { RGB v = params.FindOneSpectrum(names[k], RGB{{defs[k], defs[k], defs[k]}}); for (int c = 0; c < 3; ++c) dst[k][c] = v.c[c]; }
        }
    } else if (texname == "fbm" || texname == "wrinkled" || texname == "windy" || texname == "marble") {
        // fbm.cpp:39-54, wrinkled.cpp:39-55, windy.cpp:39-52, marble.cpp:39-54: all over IdentityMapping3D(tex2world)
        if (texname == "marble" && isFloat) { params.ReportUnused(); return; }  // CreateMarbleFloatTexture returns nullptr: no texture is registered
        t.type = texname == "fbm" ? PG_TEX_FBM : (texname == "wrinkled" ? PG_TEX_WRINKLED : (texname == "windy" ? PG_TEX_WINDY : PG_TEX_MARBLE));
        const Matrix4x4 &m = curTransform[0].GetMatrix();
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) t.w2t[4 * r + c] = m.m[r][c];
        if (texname != "windy") { t.octaves = params.FindOneInt("octaves", 8); t.omega = params.FindOneFloat("roughness", .5f); }
        if (texname == "marble") { t.noise_scale = params.FindOneFloat("scale", 1.f); t.variation = params.FindOneFloat("variation", .2f); }
        renderOptions->usesNoise = true;
    } else if (texname == "dots") {  // dots.cpp:39-100: the constructor's (outsideDot, insideDot) receive "inside" and "outside", in that order
        t.type = PG_TEX_DOTS;
        readMapping2D(params, curTransform[0], &t);
        t.tex1 = operand("inside", 1.f); t.tex2 = operand("outside", 0.f);
        renderOptions->usesNoise = true;
    } else {
        if (texname == "ptex") {  // the one texture class of the reference (api.cpp:603-697) this build has not: no frame without it
            Error("Texture \"%s\": class \"ptex\" is outside this build's closed set (constant, scale, mix, checkerboard, uv, bilerp, imagemap, fbm, wrinkled, windy, marble, dots); the scene will not be rendered.",
                  name.c_str());
            renderOptions->refused = true;
        } else Warning("%s texture \"%s\" unknown.", isFloat ? "Float" : "Spectrum", texname.c_str());  // api.cpp:640, :676 (the reference's own message for a class it does not know)
        return;
    }
    ref.tex = (int)renderOptions->textures.size();
    renderOptions->textures.push_back(t);
    table[name] = ref;
    params.ReportUnused();
}
void pbrtMaterial(const std::string &name, const ParamSet &params) {  // api.cpp:1240-1254
    VERIFY_WORLD("Material");
    ParamSet emptyParams;
    graphicsState.currentMaterial.name = name;
    graphicsState.currentMaterial.params = params;
    graphicsState.currentMaterial.material = MakeMaterial(name, emptyParams, params);
}
void pbrtMakeNamedMaterial(const std::string &name, const ParamSet &params) {  // api.cpp:1256-1284
    VERIFY_WORLD("MakeNamedMaterial");
    ParamSet emptyParams;
    std::string matName = params.FindOneString("type", "");
    if (matName == "") { Error("No parameter string \"type\" found in MakeNamedMaterial"); return; }
    if (graphicsState.namedMaterials.count(name)) Warning("Named material \"%s\" redefined.", name.c_str());
    MaterialInstance mi;
    mi.name = matName; mi.params = params; mi.material = MakeMaterial(matName, emptyParams, params);
    graphicsState.namedMaterials[name] = mi;
}
void pbrtNamedMaterial(const std::string &name) {  // api.cpp:1286-1296
    VERIFY_WORLD("NamedMaterial");
    auto iter = graphicsState.namedMaterials.find(name);
    if (iter == graphicsState.namedMaterials.end()) { Error("NamedMaterial \"%s\" unknown.", name.c_str()); return; }
    graphicsState.currentMaterial = iter->second;
}
// geometry.h:1020-1028
static void CoordinateSystem(const Vector3f &v1, Vector3f *v2, Vector3f *v3) {
    if (std::abs(v1.x) > std::abs(v1.y)) *v2 = Vector3f(-v1.z, 0, v1.x) / std::sqrt(v1.x * v1.x + v1.z * v1.z);
    else *v2 = Vector3f(0, v1.z, -v1.y) / std::sqrt(v1.y * v1.y + v1.z * v1.z);
    *v3 = Cross(v1, *v2);
}
static Point3f point3Param(const ParamSet &ps, const std::string &n, Point3f d) {
    const std::vector<Float> *v = ps.FindPoint3f(n);
    return v && v->size() >= 3 ? Point3f((*v)[0], (*v)[1], (*v)[2]) : d;
}
// MakeLight (api.cpp:690-750): the delta lights PointLight (point.cpp:80-88), SpotLight (spot.cpp:106-127, :42-49) and
// DistantLight (distant.cpp:96-104, :43-48), appended to scene.lights in declaration order (api.cpp:1308-1327).
void pbrtLightSource(const std::string &name, const ParamSet &params) {
    VERIFY_WORLD("LightSource");
    // api.cpp:1296 WARN_IF_ANIMATED_TRANSFORM("LightSource"): the reference itself gives lights the start transform
    if (curTransform.IsAnimated()) Warning("Animated transformations set; ignoring for \"LightSource\" and using the start transform only");
    const Transform &light2world = curTransform[0];
    PgLight l;
    memset(&l, 0, sizeof(l));
    l.prim = -1;
    RGB sc = params.FindOneSpectrum("scale", RGB{{1.f, 1.f, 1.f}});
    if (name == "point") {
        RGB I = params.FindOneSpectrum("I", RGB{{1.f, 1.f, 1.f}});
        Point3f P = point3Param(params, "from", Point3f(0, 0, 0));
        Transform l2w = Translate(Vector3f(P.x, P.y, P.z)) * light2world;
        l.type = PG_LIGHT_POINT;
        for (int i = 0; i < 3; ++i) l.L[i] = I.c[i] * sc.c[i];
        Point3f pLight = l2w.Pt(Point3f(0, 0, 0));  // point.h:51
        l.pos[0] = pLight.x; l.pos[1] = pLight.y; l.pos[2] = pLight.z;
    } else if (name == "spot") {
        RGB I = params.FindOneSpectrum("I", RGB{{1.f, 1.f, 1.f}});
        Float coneangle = params.FindOneFloat("coneangle", 30.);
        Float conedelta = params.FindOneFloat("conedeltaangle", 5.);
        Point3f from = point3Param(params, "from", Point3f(0, 0, 0));
        Point3f to = point3Param(params, "to", Point3f(0, 0, 1));
        Vector3f dir = Normalize(to - from);
        Vector3f du, dv;
        CoordinateSystem(dir, &du, &dv);
        Transform dirToZ = Transform(Matrix4x4(du.x, du.y, du.z, 0., dv.x, dv.y, dv.z, 0., dir.x, dir.y, dir.z, 0., 0, 0, 0, 1.));
        Transform l2w = light2world * Translate(Vector3f(from.x, from.y, from.z)) * Inverse(dirToZ);
        l.type = PG_LIGHT_SPOT;
        for (int i = 0; i < 3; ++i) l.L[i] = I.c[i] * sc.c[i];
        Point3f pLight = l2w.Pt(Point3f(0, 0, 0));
        l.pos[0] = pLight.x; l.pos[1] = pLight.y; l.pos[2] = pLight.z;
        const Matrix4x4 &w2l = l2w.GetInverseMatrix();  // Light::WorldToLight = Inverse(LightToWorld), light.cpp:46-52
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) l.w2l[3 * r + c] = w2l.m[r][c];
        l.cos_total_width = std::cos(Radians(coneangle));
        l.cos_falloff_start = std::cos(Radians(coneangle - conedelta));
    } else if (name == "distant") {
        RGB L = params.FindOneSpectrum("L", RGB{{1.f, 1.f, 1.f}});
        Point3f from = point3Param(params, "from", Point3f(0, 0, 0));
        Point3f to = point3Param(params, "to", Point3f(0, 0, 1));
        Vector3f dir = from - to;
        Vector3f wLight = Normalize(light2world.Vec(dir));
        l.type = PG_LIGHT_DISTANT;
        for (int i = 0; i < 3; ++i) l.L[i] = L.c[i] * sc.c[i];
        l.pos[0] = wLight.x; l.pos[1] = wLight.y; l.pos[2] = wLight.z;
        // world_radius is filled in when the scene is flattened (DistantLight::Preprocess needs the world bound)
    } else if (name == "projection" || name == "goniometric") {
        // CreateProjectionLight (projection.cpp:129-140, ctor :44-69) / CreateGoniometricLight (goniometric.cpp:87-95, ctor goniometric.h:56-66)
        const bool projection = name == "projection";
        RGB I = params.FindOneSpectrum("I", RGB{{1.f, 1.f, 1.f}});
        const Float fov = projection ? params.FindOneFloat("fov", 45.) : 0;
        std::string texname = params.FindOneString("mapname", "");
        if (!texname.empty()) texname = AbsolutePath(ResolveFilename(texname));
        l.type = projection ? PG_LIGHT_PROJECTION : PG_LIGHT_GONIO;
        for (int i = 0; i < 3; ++i) l.L[i] = I.c[i] * sc.c[i];
        Point3f pLight = light2world.Pt(Point3f(0, 0, 0));
        l.pos[0] = pLight.x; l.pos[1] = pLight.y; l.pos[2] = pLight.z;
        const Matrix4x4 &w2l = light2world.GetInverseMatrix();
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) l.w2l[3 * r + c] = w2l.m[r][c];
        // the map as a MIPMap<RGBSpectrum> with its defaults (EWA, max anisotropy 8, repeat); ReadImage reports a missing file
        int resX = 0, resY = 0;
        std::vector<RGB> texels;
        l.env_image = -1;
        l.env_power[0] = l.env_power[1] = l.env_power[2] = 1;
        if (ReadImage(texname, &resX, &resY, &texels)) {
            std::vector<float> conv((size_t)resX * resY * 3);
            for (size_t i = 0; i < texels.size(); ++i) for (int c = 0; c < 3; ++c) conv[3 * i + c] = texels[i].c[c];
            PgImage map;
            memset(&map, 0, sizeof(map));
            map.is_float = 0; map.wrap = 0; map.trilinear = 0; map.max_anisotropy = 8.f;
            BuildMIPMap(resX, resY, 3, conv, 0, &map, &renderOptions->texels);
            l.env_image = (int)renderOptions->images.size();
            renderOptions->images.push_back(map);
            const float st[2] = {.5f, .5f};  // Power(): Lookup(Point2f(.5f, .5f), .5f)
            MIPMapLookup(map, renderOptions->texels, st, .5f, l.env_power);
        }
        if (projection) {
            const Float aspect = l.env_image >= 0 ? (Float(resX) / Float(resY)) : 1;
            if (aspect > 1) { l.screen[0] = -aspect; l.screen[1] = -1; l.screen[2] = aspect; l.screen[3] = 1; }
            else { l.screen[0] = -1; l.screen[1] = -1 / aspect; l.screen[2] = 1; l.screen[3] = 1 / aspect; }
            l.hither = 1e-3f;
            const Float yon = 1e30f;
            Transform lightProjection = Perspective(fov, l.hither, yon);
            const Matrix4x4 &pm = lightProjection.GetMatrix();
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) l.proj[4 * r + c] = pm.m[r][c];
            Transform screenToLight = Inverse(lightProjection);
            Point3f pc = screenToLight.Pt(Point3f(l.screen[2], l.screen[3], 0));
            l.cos_total_width = Normalize(Vector3f(pc.x, pc.y, pc.z)).z;
        }
    } else if (name == "infinite" || name == "exinfinite") {  // CreateInfiniteLight, infinite.cpp:176-188; ctor :44-85
        RGB L = params.FindOneSpectrum("L", RGB{{1.f, 1.f, 1.f}});
        params.FindOneInt("samples", params.FindOneInt("nsamples", 1));
        std::string texmap = params.FindOneString("mapname", "");
        if (!texmap.empty()) texmap = AbsolutePath(ResolveFilename(texmap));
        l.type = PG_LIGHT_INFINITE;
        for (int i = 0; i < 3; ++i) l.L[i] = L.c[i] * sc.c[i];
        const Matrix4x4 &m = light2world.GetMatrix(), &mi = light2world.GetInverseMatrix();
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { l.l2w[3 * r + c] = m.m[r][c]; l.w2l[3 * r + c] = mi.m[r][c]; }
        // Lmap: the map's texels times L (no vertical flip here), or the single texel L (infinite.cpp:50-64)
        int resX = 1, resY = 1;
        std::vector<RGB> texels;
        if (!texmap.empty() && ReadImage(texmap, &resX, &resY, &texels)) {
            for (RGB &t : texels) for (int c = 0; c < 3; ++c) t.c[c] *= l.L[c];
        } else { resX = resY = 1; texels.assign(1, RGB{{l.L[0], l.L[1], l.L[2]}}); }
        std::vector<float> conv((size_t)resX * resY * 3);
        for (size_t i = 0; i < texels.size(); ++i) for (int c = 0; c < 3; ++c) conv[3 * i + c] = texels[i].c[c];
        PgImage lmap;
        memset(&lmap, 0, sizeof(lmap));
        lmap.is_float = 0; lmap.wrap = 0; lmap.trilinear = 0; lmap.max_anisotropy = 8.f;  // MIPMap's defaults (mipmap.h:70-71)
        BuildMIPMap(resX, resY, 3, conv, 0, &lmap, &renderOptions->texels);
        l.env_image = (int)renderOptions->images.size();
        renderOptions->images.push_back(lmap);
        // the scalar image of the sampling distribution (infinite.cpp:66-84)
        const int width = 2 * lmap.width, height = 2 * lmap.height;
        std::vector<Float> img((size_t)width * height);
        float fwidth = 0.5f / std::min(width, height);
        for (int v = 0; v < height; ++v) {
            Float vp = (v + .5f) / (Float)height;
            Float sinTheta = std::sin(Pi * (v + .5f) / height);
            for (int u = 0; u < width; ++u) {
                Float up = (u + .5f) / (Float)width;
                const float st[2] = {up, vp};
                float rgb[3];
                MIPMapLookup(lmap, renderOptions->texels, st, fwidth, rgb);
                img[u + v * width] = 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2];  // .y()
                img[u + v * width] *= sinTheta;
            }
        }
        {   // Power(): Lmap->Lookup(Point2f(.5f, .5f), .5f), infinite.cpp:87-91
            const float st[2] = {.5f, .5f};
            MIPMapLookup(lmap, renderOptions->texels, st, .5f, l.env_power);
        }
        // Distribution2D (sampling.cpp:159-171) of Distribution1Ds (sampling.h:57-70)
        auto dist1d = [](const Float *f, int n, Float *func, Float *cdf, Float *funcInt) {
            for (int i = 0; i < n; ++i) func[i] = f[i];
            cdf[0] = 0;
            for (int i = 1; i < n + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / n;
            *funcInt = cdf[n];
            if (*funcInt == 0) { for (int i = 1; i < n + 1; ++i) cdf[i] = Float(i) / Float(n); }
            else { for (int i = 1; i < n + 1; ++i) cdf[i] /= *funcInt; }
        };
        std::vector<float> &tab = renderOptions->envTables;
        l.env_nu = width; l.env_nv = height; l.env_table = (int64_t)tab.size();
        const size_t rowStride = 2 * (size_t)width + 2, base = tab.size();
        tab.resize(base + rowStride * height + 2 * (size_t)height + 2);
        std::vector<Float> marginalFunc(height);
        for (int v = 0; v < height; ++v) {
            float *row = &tab[base + rowStride * v];
            dist1d(&img[(size_t)v * width], width, row, row + width, row + 2 * width + 1);
            marginalFunc[v] = row[2 * width + 1];
        }
        float *mrow = &tab[base + rowStride * height];
        dist1d(marginalFunc.data(), height, mrow, mrow + height, mrow + 2 * height + 1);
    } else {
        // (every light class of the reference is built above: an unknown name is the file's mistake, reported as the reference reports it)
        Warning("Light \"%s\" unknown.", name.c_str());  // api.cpp:750
        Error("LightSource: light type \"%s\" unknown.", name.c_str());  // api.cpp:1308
        return;
    }
    params.ReportUnused();
    renderOptions->lights.push_back(l);
}
void pbrtAreaLightSource(const std::string &name, const ParamSet &params) {
    VERIFY_WORLD("AreaLightSource");
    graphicsState.areaLight = name;
    graphicsState.areaLightParams = params;
}

// TriangleMesh ctor, shapes/triangle.cpp:60-92: vertices, normals and tangents go to world space once, here
std::shared_ptr<TriangleMesh> BuildTriangleMesh(const Transform &o2w, bool reverseOrientation, int nTriangles, const int *indices,
                                                int nVertices, const Float *P, const Float *S, const Float *N, const Float *UV) {
    auto mesh = std::make_shared<TriangleMesh>();
    mesh->nTriangles = nTriangles;
    mesh->nVertices = nVertices;
    mesh->vertexIndices.assign(indices, indices + 3 * nTriangles);
    mesh->reverseOrientation = reverseOrientation;
    mesh->transformSwapsHandedness = o2w.SwapsHandedness();
    mesh->p.resize(nVertices);
    for (int i = 0; i < nVertices; ++i) mesh->p[i] = o2w.Pt(Point3f(P[3 * i], P[3 * i + 1], P[3 * i + 2]));
    if (UV) mesh->uv.assign(UV, UV + 2 * nVertices);
    if (N) { mesh->n.resize(nVertices); for (int i = 0; i < nVertices; ++i) mesh->n[i] = o2w.Nrm(Normal3f(N[3 * i], N[3 * i + 1], N[3 * i + 2])); }
    if (S) { mesh->s.resize(nVertices); for (int i = 0; i < nVertices; ++i) mesh->s[i] = o2w.Vec(Vector3f(S[3 * i], S[3 * i + 1], S[3 * i + 2])); }
    return mesh;
}
std::shared_ptr<TriangleMesh> CreatePLYMesh(const Transform &o2w, bool reverseOrientation, const ParamSet &params);  // plymesh.cpp
std::shared_ptr<TriangleMesh> CreateLoopSubdiv(const Transform &o2w, bool reverseOrientation, const ParamSet &params);  // loopsubdiv.cpp
std::shared_ptr<TriangleMesh> CreateHeightfield(const Transform &o2w, bool reverseOrientation, const ParamSet &params);  // meshshapes.cpp
std::shared_ptr<TriangleMesh> CreateNURBS(const Transform &o2w, bool reverseOrientation, const ParamSet &params);        // meshshapes.cpp

// The "alpha" / "shadowalpha" parameters of a mesh (triangle.cpp:716-738, plymesh.cpp:259-287): a named float texture, or
// the literal 0 (a constant-zero mask); returns the index of the mesh's PgAlphaMask or -1
int MeshAlphaMask(const ParamSet &params) {
    PgAlphaMask am;
    memset(&am, 0, sizeof(am));
    am.alpha.tex = am.shadow_alpha.tex = -1;
    const char *names[2] = {"alpha", "shadowalpha"};
    for (int k = 0; k < 2; ++k) {
        PgTexRef ref; ref.tex = -1; ref.v[0] = ref.v[1] = ref.v[2] = 0;
        bool has = false;
        const std::string texName = params.FindTexture(names[k]);
        if (!texName.empty()) {
            auto it = graphicsState.floatTextures.find(texName);
            if (it != graphicsState.floatTextures.end()) { ref = it->second; has = true; }
            else Error("Couldn't find float texture \"%s\" for \"%s\" parameter", texName.c_str(), names[k]);
        } else if (params.FindOneFloat(names[k], 1.f) == 0.f) has = true;  // ConstantTexture<Float>(0.f)
        if (k == 0) { am.has_alpha = has; am.alpha = ref; } else { am.has_shadow_alpha = has; am.shadow_alpha = ref; }
    }
    if (!am.has_alpha && !am.has_shadow_alpha) return -1;
    renderOptions->alphas.push_back(am);
    return (int)renderOptions->alphas.size() - 1;
}
// shapes/triangle.cpp:647-743 CreateTriangleMeshShape + :94-110 CreateTriangleMesh
static std::shared_ptr<TriangleMesh> CreateTriangleMeshShape(const Transform &o2w, bool reverseOrientation, const ParamSet &params) {
    const std::vector<int> *vi = params.FindInt("indices");
    const std::vector<Float> *P = params.FindPoint3f("P");
    const std::vector<Float> *uvs = params.FindPoint2f("uv");
    if (!uvs) uvs = params.FindPoint2f("st");
    if (!uvs) { uvs = params.FindFloat("uv"); if (!uvs) uvs = params.FindFloat("st"); }
    if (!vi) { Error("Vertex indices \"indices\" not provided with triangle mesh shape"); return nullptr; }
    if (!P) { Error("Vertex positions \"P\" not provided with triangle mesh shape"); return nullptr; }
    int npi = (int)P->size() / 3, nvi = (int)vi->size();
    if (uvs) {
        int nuvi = (int)uvs->size() / 2;
        if (nuvi < npi) { Error("Not enough of \"uv\"s for triangle mesh.  Expected %d, found %d.  Discarding.", npi, nuvi); uvs = nullptr; }
        else if (nuvi > npi) Warning("More \"uv\"s provided than will be used for triangle mesh.  (%d expcted, %d found)", npi, nuvi);
    }
    const std::vector<Float> *S = params.FindVector3f("S");
    if (S && (int)S->size() / 3 != npi) { Error("Number of \"S\"s for triangle mesh must match \"P\"s"); S = nullptr; }
    const std::vector<Float> *N = params.FindNormal3f("N");
    if (N && (int)N->size() / 3 != npi) { Error("Number of \"N\"s for triangle mesh must match \"P\"s"); N = nullptr; }
    for (int i = 0; i < nvi; ++i)
        if ((*vi)[i] >= npi || (*vi)[i] < 0) {  // (the reference only checks the upper bound and reads out of bounds below zero)
            Error("trianglemesh has out of-bounds vertex index %d (%d \"P\" values were given", (*vi)[i], npi);
            return nullptr;
        }
    const int alphaMask = MeshAlphaMask(params);
    params.FindInt("faceIndices");
    auto mesh = BuildTriangleMesh(o2w, reverseOrientation, nvi / 3, vi->data(), npi, P->data(), S ? S->data() : nullptr, N ? N->data() : nullptr,
                                  uvs ? uvs->data() : nullptr);
    mesh->alphaMask = alphaMask;
    return mesh;
}

// The TransformedPrimitive of a moving shape or instance (api.cpp:1399-1419, :1576-1586), with or without rotation (its bounds: host/motion_bounds.cpp).
// false = outside the closed set, the frame is refused: a motion one of whose ends MIRRORS.  The reference takes the quaternion of the improper
// rotation its polar decomposition finds (quaternion.cpp:61-92) -- not a unit quaternion -- and slerps it: the interpolated "rotation" is no rigid
// motion, and the reference's own renders of such scenes abort (a path that never leaves a surface runs out of sample dimensions) or do not
// terminate (8 of 41 random scenes, tests/test_gpu_fuzz.py::random_scene_rotating_motion before mirrored blocks were left still).
static bool MakeMotion(const char *what, const std::string &name, GeometricPrimitive::InstanceTransforms *xf) {
    xf->InstanceToWorld = curTransform[0];
    xf->WorldToInstance = Inverse(curTransform[0]);
    if (!curTransform.IsAnimated()) return true;
    if (curTransform[0].SwapsHandedness() || curTransform[1].SwapsHandedness()) {
        Error("%s \"%s\" under an animated transformation that MIRRORS (the reference slerps the non-unit quaternion of an improper rotation; its own "
              "renders of such motions abort or do not terminate) is outside this build's closed set; the scene will not be rendered.", what, name.c_str());
        renderOptions->refused = true;
        return false;
    }
    xf->animated = true;
    xf->InstanceToWorldEnd = curTransform[1];
    xf->WorldToInstanceEnd = Inverse(curTransform[1]);
    xf->time[0] = renderOptions->transformStartTime; xf->time[1] = renderOptions->transformEndTime;
    return true;
}
void pbrtShape(const std::string &name, const ParamSet &params) {  // api.cpp:1329-1421
    VERIFY_WORLD("Shape");
    // A shape under an animated transformation is created at the identity and its primitives -- under a BVHAccel of their own when there
    // are several -- become ONE TransformedPrimitive over the AnimatedTransform (api.cpp:1386-1419, primitive.cpp:76-103)
    const bool animated = curTransform.IsAnimated();
    const Transform shapeToWorld = animated ? Transform() : curTransform[0];
    std::shared_ptr<TriangleMesh> mesh;
    std::shared_ptr<Sphere> sphere;
    if (name == "trianglemesh") mesh = CreateTriangleMeshShape(shapeToWorld, graphicsState.reverseOrientation, params);
    else if (name == "plymesh") mesh = CreatePLYMesh(shapeToWorld, graphicsState.reverseOrientation, params);
    else if (name == "loopsubdiv") mesh = CreateLoopSubdiv(shapeToWorld, graphicsState.reverseOrientation, params);
    else if (name == "heightfield") mesh = CreateHeightfield(shapeToWorld, graphicsState.reverseOrientation, params);
    else if (name == "nurbs") mesh = CreateNURBS(shapeToWorld, graphicsState.reverseOrientation, params);
    else if (name == "sphere") {  // CreateSphereShape, sphere.cpp:326-336
        Float radius = params.FindOneFloat("radius", 1.f);
        Float zmin = params.FindOneFloat("zmin", -radius);
        Float zmax = params.FindOneFloat("zmax", radius);
        Float phimax = params.FindOneFloat("phimax", 360.f);
        sphere = std::make_shared<Sphere>(shapeToWorld, Inverse(shapeToWorld), graphicsState.reverseOrientation, radius, zmin, zmax, phimax);
    } else if (name == "cylinder") {  // CreateCylinderShape, cylinder.cpp:225-235
        Float radius = params.FindOneFloat("radius", 1);
        Float zmin = params.FindOneFloat("zmin", -1);
        Float zmax = params.FindOneFloat("zmax", 1);
        Float phimax = params.FindOneFloat("phimax", 360);
        sphere = Sphere::Cylinder(shapeToWorld, Inverse(shapeToWorld), graphicsState.reverseOrientation, radius, zmin, zmax, phimax);
    } else if (name == "disk") {  // CreateDiskShape, disk.cpp:139-149
        Float height = params.FindOneFloat("height", 0.);
        Float radius = params.FindOneFloat("radius", 1);
        Float inner_radius = params.FindOneFloat("innerradius", 0);
        Float phimax = params.FindOneFloat("phimax", 360);
        sphere = Sphere::Disk(shapeToWorld, Inverse(shapeToWorld), graphicsState.reverseOrientation, height, radius, inner_radius, phimax);
    } else if (name == "cone") {  // CreateConeShape, cone.cpp:210-219
        Float radius = params.FindOneFloat("radius", 1);
        Float height = params.FindOneFloat("height", 1);
        Float phimax = params.FindOneFloat("phimax", 360);
        sphere = Sphere::Cone(shapeToWorld, Inverse(shapeToWorld), graphicsState.reverseOrientation, height, radius, phimax);
    } else if (name == "paraboloid") {  // CreateParaboloidShape, paraboloid.cpp:216-226
        Float radius = params.FindOneFloat("radius", 1);
        Float zmin = params.FindOneFloat("zmin", 0);
        Float zmax = params.FindOneFloat("zmax", 1);
        Float phimax = params.FindOneFloat("phimax", 360);
        sphere = Sphere::Paraboloid(shapeToWorld, Inverse(shapeToWorld), graphicsState.reverseOrientation, radius, zmin, zmax, phimax);
    } else if (name == "hyperboloid") {  // CreateHyperboloidShape, hyperboloid.cpp:252-261
        Point3f p1 = params.FindOnePoint3f("p1", Point3f(0, 0, 0));
        Point3f p2 = params.FindOnePoint3f("p2", Point3f(1, 1, 1));
        Float phimax = params.FindOneFloat("phimax", 360);
        sphere = Sphere::Hyperboloid(shapeToWorld, Inverse(shapeToWorld), graphicsState.reverseOrientation, p1, p2, phimax);
    } else if (name == "curve") {  // the one shape of the reference (api.cpp:469-533) this build has not: no frame without it
        Error("Shape \"curve\" is outside this build's closed set (trianglemesh, plymesh, loopsubdiv, heightfield, nurbs, sphere, cylinder, disk, cone, paraboloid, hyperboloid); the scene will not be rendered.");
        renderOptions->refused = true;
    } else Warning("Shape \"%s\" unknown.", name.c_str());  // api.cpp:531
    if (!sphere && (!mesh || mesh->nTriangles == 0)) return;
    int mtl = GetMaterialForShape(params);
    params.ReportUnused();
    int firstLight = -1;
    PgLight lightProto;
    memset(&lightProto, 0, sizeof(lightProto));
    if (animated) {
        if (graphicsState.areaLight != "") Warning("Ignoring currently set area light when creating animated shape");  // api.cpp:1389-1391
    } else if (graphicsState.areaLight != "" && sphere && !sphere->CanEmit())
        Error("Shape \"%s\" cannot be an area light: pbrt-v3 has no Sample() for it and aborts when the light is sampled. The shape is added without emission.", name.c_str());
    else if (graphicsState.areaLight != "") {
        // MakeAreaLight (api.cpp:752-768) + CreateDiffuseAreaLight (diffuse.cpp:135-146)
        if (graphicsState.areaLight == "area" || graphicsState.areaLight == "diffuse") {
            const ParamSet &lp = graphicsState.areaLightParams;
            RGB L = lp.FindOneSpectrum("L", RGB{{1.f, 1.f, 1.f}});
            RGB sc = lp.FindOneSpectrum("scale", RGB{{1.f, 1.f, 1.f}});
            lp.FindOneInt("samples", lp.FindOneInt("nsamples", 1));
            lightProto.two_sided = lp.FindOneBool("twosided", false) ? 1 : 0;
            for (int i = 0; i < 3; ++i) lightProto.L[i] = L.c[i] * sc.c[i];
            lp.ReportUnused();
            firstLight = (int)renderOptions->lights.size();
        } else Warning("Area light \"%s\" unknown.", graphicsState.areaLight.c_str());
    }
    int mediumInside, mediumOutside;
    CreateMediumInterface(&mediumInside, &mediumOutside);  // api.cpp:1380
    const int nShapes = sphere ? 1 : mesh->nTriangles;
    std::shared_ptr<ObjectDefinition> moving = animated ? std::make_shared<ObjectDefinition>() : nullptr;
    for (int i = 0; i < nShapes; ++i) {
        GeometricPrimitive prim;
        if (sphere) prim.sphere = sphere;
        else { prim.shape.mesh = mesh; prim.shape.triIndex = i; }
        prim.material = mtl;
        prim.mediumInside = mediumInside; prim.mediumOutside = mediumOutside;
        if (firstLight >= 0 && !renderOptions->currentInstance) {
            PgLight l = lightProto;
            l.area = sphere ? sphere->Area() : prim.shape.Area();
            l.prim = -1;
            prim.areaLight = (int)renderOptions->lights.size();
            renderOptions->lights.push_back(l);
        }
        // api.cpp:1405-1418: to the scene, or to the instance definition being collected
        if (moving) moving->prims.push_back(prim);
        else if (renderOptions->currentInstance) renderOptions->currentInstance->prims.push_back(prim);
        else renderOptions->primitives.push_back(prim);
    }
    if (moving) {
        if (moving->prims.size() > 1) {  // std::make_shared<BVHAccel>(prims): the defaults (one primitive per leaf, SAH), whatever the file's Accelerator says
            moving->accel = CreateDefaultBVHAccel(std::move(moving->prims));
            moving->prims.clear();
        }
        auto xf = std::make_shared<GeometricPrimitive::InstanceTransforms>();
        if (!MakeMotion("Shape", name, xf.get())) return;
        GeometricPrimitive tp;
        tp.object = moving;
        tp.xf = xf;
        // api.cpp:1405-1418: inside an object definition the moving shape's TransformedPrimitive is one of the instance's primitives -- a
        // TransformedPrimitive under the TransformedPrimitive of every ObjectInstance (PG_PRIM_INSTANCE inside an object's run, ABI 29)
        if (renderOptions->currentInstance) renderOptions->currentInstance->prims.push_back(tp);
        else renderOptions->primitives.push_back(tp);
        return;
    }
    if (firstLight >= 0 && renderOptions->currentInstance)
        Warning("Area lights not supported with object instancing");  // api.cpp:1407-1408 (here the shapes are added without emission)
}
void pbrtReverseOrientation() { VERIFY_WORLD("ReverseOrientation"); graphicsState.reverseOrientation = !graphicsState.reverseOrientation; }
void pbrtObjectBegin(const std::string &name) {  // api.cpp:1509-1519
    VERIFY_WORLD("ObjectBegin");
    pbrtAttributeBegin();
    if (renderOptions->currentInstance) Error("ObjectBegin called inside of instance definition");
    renderOptions->instances[name] = std::make_shared<ObjectDefinition>();
    renderOptions->currentInstance = renderOptions->instances[name];
}
void pbrtObjectEnd() {  // api.cpp:1523-1532
    VERIFY_WORLD("ObjectEnd");
    if (!renderOptions->currentInstance) Error("ObjectEnd called outside of instance definition");
    renderOptions->currentInstance = nullptr;
    pbrtAttributeEnd();
}
void pbrtObjectInstance(const std::string &name) {  // api.cpp:1546-1588
    VERIFY_WORLD("ObjectInstance");
    if (renderOptions->currentInstance) { Error("ObjectInstance can't be called inside instance definition"); return; }
    auto it = renderOptions->instances.find(name);
    if (it == renderOptions->instances.end()) { Error("Unable to find instance named \"%s\"", name.c_str()); return; }
    std::shared_ptr<ObjectDefinition> obj = it->second;
    if (obj->prims.empty() && !obj->accel) return;
    if (obj->prims.size() > 1) {  // the first use builds the instance's own accelerator over its primitives
        RenderOptions &ro = *renderOptions;
        if (ro.AcceleratorName != "bvh")
            Warning("Accelerator \"%s\" is outside this build's closed set; using \"bvh\".", ro.AcceleratorName.c_str());
        obj->accel = CreateBVHAccelerator(std::move(obj->prims), ro.AcceleratorName == "bvh" ? ro.AcceleratorParams : ParamSet());
        obj->prims.clear();
    }
    GeometricPrimitive prim;
    prim.object = obj;
    {   // TransformedPrimitive over the AnimatedTransform of the two current transformations (api.cpp:1576-1586)
        auto xf = std::make_shared<GeometricPrimitive::InstanceTransforms>();
        if (!MakeMotion("Object instance", name, xf.get())) return;
        prim.xf = xf;
    }
    renderOptions->primitives.push_back(prim);
}

// RenderOptions::MakeIntegrator / MakeScene / MakeCamera (api.cpp:1651-1727)
static GpuPathIntegrator *MakeIntegrator() {
    RenderOptions &ro = *renderOptions;
    // MakeFilter, api.cpp:785-803
    std::string filterName = ro.FilterName;
    if (filterName != "box" && filterName != "gaussian" && filterName != "mitchell" && filterName != "sinc" && filterName != "triangle") {
        Error("Filter \"%s\" unknown.", filterName.c_str());
        filterName = "box";
    }
    Float xw, yw;
    FilterRadiusFor(filterName, ro.FilterParams, &xw, &yw);
    if (ro.FilmName != "image") { Error("Film \"%s\" unknown.", ro.FilmName.c_str()); return nullptr; }
    Film *film = CreateFilm(ro.FilmParams, xw, yw);
    SetFilmFilter(film, filterName, ro.FilterParams);
    ro.FilterParams.ReportUnused();
    ro.FilmParams.ReportUnused();
    if (ro.CameraName != "perspective" && ro.CameraName != "orthographic" && ro.CameraName != "environment") {
        Error("Camera \"%s\" is outside this build's closed set (perspective, orthographic, environment).", ro.CameraName.c_str());
        delete film;
        return nullptr;
    }
    std::shared_ptr<PerspectiveCamera> camera(CreatePerspectiveCamera(ro.CameraParams, ro.CameraToWorld[0], film, ro.CameraName == "orthographic"));
    // AnimatedTransform animatedCam2World(CameraToWorld[0], transformStartTime, CameraToWorld[1], transformEndTime), api.cpp:1725-1730: every
    // camera ray is carried to world space by the transform interpolated at its time (cameras/perspective.cpp:89-91, :139)
    camera->CameraToWorldEnd = ro.CameraToWorld[1];
    camera->transformStartTime = ro.transformStartTime; camera->transformEndTime = ro.transformEndTime;
    camera->animated = !(ro.CameraToWorld[0].GetMatrix() == ro.CameraToWorld[1].GetMatrix() &&
                         ro.CameraToWorld[0].GetInverseMatrix() == ro.CameraToWorld[1].GetInverseMatrix());  // Transform::operator!=, transform.h:152-154
    if (ro.CameraName == "environment") { camera->environment = true; camera->lensRadius = 0; }  // environment.cpp:96-97: lens parameters unused
    ro.CameraParams.ReportUnused();
    // the GlobalSamplers (halton, sobol) index every sample by (pixel, sample number) alone; the PixelSamplers (random,
    // stratified, 02sequence, maxmindist) draw from one RNG stream per tile, which serialises a tile's samples
    int sb[4];
    film->GetSampleBounds(sb);
    // MakeSampler, api.cpp:816-836
    std::shared_ptr<HaltonSampler> sampler;
    const std::string &sn = ro.SamplerName;
    if (sn == "halton") sampler.reset(CreateHaltonSampler(ro.SamplerParams, sb));
    else if (sn == "sobol") sampler.reset(CreateSobolSampler(ro.SamplerParams, sb));
    else if (sn == "lowdiscrepancy" || sn == "02sequence") sampler.reset(CreateTileSerialSampler("02sequence", ro.SamplerParams));
    else if (sn == "maxmindist" || sn == "random" || sn == "stratified") sampler.reset(CreateTileSerialSampler(sn, ro.SamplerParams));
    else {
        Warning("Sampler \"%s\" unknown.", sn.c_str());
        Error("Unable to create sampler.");
        return nullptr;
    }
    ro.SamplerParams.ReportUnused();
    if (ro.IntegratorName != "path" && ro.IntegratorName != "volpath") {
        Error("Integrator \"%s\" is outside this build's closed set (path, volpath).", ro.IntegratorName.c_str());
        return nullptr;
    }
    GpuPathIntegrator *integrator = CreatePathIntegrator(ro.IntegratorParams, sampler, camera);  // volpath.cpp:191-214 reads the same parameters
    integrator->volumetric = ro.IntegratorName == "volpath";
    // A PixelSampler (stratified, 02sequence, maxmindist) whose "dimensions" do not cover a whole path (2 + 3 maxdepth two-dimensional
    // draws) falls back to its tile's ONE random stream from the first bounce on (sampler.cpp:108-134): a path's later numbers then depend on how many
    // all earlier paths of the tile drew, and the device can hold one path per tile in flight -- correct, pinned against the reference, and
    // orders of magnitude slower than the same sampler with enough dimensions.  "random" always draws from the stream.  volpath: no bound.
    if (sampler->kind > PG_SAMPLER_RANDOM && !integrator->volumetric && sampler->nSampledDimensions < 2 + 3 * (long long)integrator->maxDepth)
        Warning("Sampler \"%s\" with \"dimensions\" %d samples a path of \"maxdepth\" %d from its tile's random stream after the first bounce: tiles render one "
                "path at a time on the device. Set \"integer dimensions\" [ %lld ] (2 + 3 maxdepth) or more to render the frame as one wavefront.",
                sn.c_str(), sampler->nSampledDimensions, integrator->maxDepth, 2 + 3 * (long long)integrator->maxDepth);
    else if (sampler->kind == PG_SAMPLER_RANDOM || (sampler->kind > PG_SAMPLER_RANDOM && integrator->volumetric))
        Warning("Sampler \"%s\"%s draws from its tile's random stream: tiles render one path at a time on the device (halton / sobol render the frame as one wavefront).",
                sn.c_str(), integrator->volumetric ? " under \"volpath\"" : "");
    {   // MakeCamera (api.cpp:785-790): the camera sits in the graphics state's current outside medium
        int in, out;
        CreateMediumInterface(&in, &out);
        integrator->cameraMedium = out;
    }
    if (ro.haveScatteringMedia && ro.IntegratorName != "volpath")
        Warning("Scene has scattering media but \"path\" integrator doesn't support volume scattering. Consider using \"volpath\".");
    ro.IntegratorParams.ReportUnused();
    if (ro.lights.empty())
        Warning("No light sources defined in scene; rendering a black image.");
    return integrator;
}
static Scene *MakeScene() {
    RenderOptions &ro = *renderOptions;
    Scene *scene = new Scene;
    if (ro.AcceleratorName != "bvh")
        Warning("Accelerator \"%s\" is outside this build's closed set; using \"bvh\".", ro.AcceleratorName.c_str());
    scene->aggregate = CreateBVHAccelerator(std::move(ro.primitives), ro.AcceleratorName == "bvh" ? ro.AcceleratorParams : ParamSet());
    ro.AcceleratorParams.ReportUnused();
    scene->lights = ro.lights;
    scene->materials = ro.materials;
    scene->bxdfs = ro.bxdfs;
    scene->textures = ro.textures;
    scene->textured = ro.textured;
    scene->images = ro.images;
    scene->usesNoise = ro.usesNoise;
    scene->texels = ro.texels;
    scene->envTables = ro.envTables;
    scene->alphas = ro.alphas;
    scene->media = ro.media;
    scene->mediaGrid = ro.mediaGrid; scene->grids = ro.grids; scene->gridDensity = ro.gridDensity;
    scene->bssrdfs = ro.bssrdfs; scene->bssrdfTables = ro.bssrdfTables;
    scene->materialBssrdf = ro.materialBssrdf;
    if (!scene->bssrdfs.empty()) scene->materialBssrdf.resize(scene->materials.size(), -1);
    scene->worldBound = scene->aggregate->WorldBound();
    // resolve each light's emitting triangle to its index in BVH order
    const auto &prims = scene->aggregate->primitives;
    for (size_t i = 0; i < prims.size(); ++i)
        if (prims[i].areaLight >= 0) scene->lights[prims[i].areaLight].prim = (int)i;
    ro.primitives.clear();
    ro.lights.clear();
    return scene;
}

void pbrtWorldEnd() {  // api.cpp:1590-1644
    VERIFY_WORLD("WorldEnd");
    // unmatched Begins: the three stacks are pushed together (AttributeBegin) or as a pair (TransformBegin) and unwind together
    while (pushedGraphicsStates.size()) {
        Warning("Missing end to pbrtAttributeBegin()");
        pushedGraphicsStates.pop_back(); pushedTransforms.pop_back();
        if (!pushedActiveTransformBits.empty()) pushedActiveTransformBits.pop_back();
    }
    while (pushedTransforms.size()) {
        Warning("Missing end to pbrtTransformBegin()");
        pushedTransforms.pop_back();
        if (!pushedActiveTransformBits.empty()) pushedActiveTransformBits.pop_back();
    }
    pushedActiveTransformBits.clear();
    const bool timing = getenv("PBRT_HOST_TIMING") != nullptr;  // stderr: seconds spent building the accelerators
    auto tBuild = std::chrono::steady_clock::now();
    std::unique_ptr<GpuPathIntegrator> integrator(MakeIntegrator());
    const int motionFailuresBefore = MotionBoundsFailures();
    std::unique_ptr<Scene> scene(MakeScene());
    if (MotionBoundsFailures() != motionFailuresBefore) {  // (the reference's CHECK_LE(*nZeros, 8) ends its process here, transform.cpp:385)
        Error("A rotating motion's derivative has more than 8 zeros in the shutter interval (AnimatedTransform::MotionBounds): its box cannot be the reference's");
        renderOptions->refused = true;
    }
    if (timing) fprintf(stderr, "pbrt host: MakeScene (BVH build) %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - tBuild).count());
    if (renderOptions->refused) {
        Error("Scene not rendered: it uses features outside this build's closed set (see the errors above); no approximate image is written.");
        scene.reset(); integrator.reset();
    }
    if (scene && integrator) {
        if (PbrtOptions.loadOnly) {
            lastLoadedScene.reset(new LoadedScene);
            lastLoadedScene->scene = std::move(scene);
            lastLoadedScene->integrator = std::move(integrator);
        } else
            integrator->Render(*scene);
    }
    graphicsState = GraphicsState();
    currentApiState = APIState::OptionsBlock;
    for (int i = 0; i < MaxTransforms; ++i) curTransform[i] = Transform();
    activeTransformBits = AllTransformsBits;
    namedCoordinateSystems.clear();
    // a file may hold several frames: the next WorldBegin starts from fresh options -- camera, film, sampler, integrator,
    // named media, object instances and every material / texture / image table (api.cpp:1630-1640)
    renderOptions.reset(new RenderOptions);
}
}  // namespace pbrt
