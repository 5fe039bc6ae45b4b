// C entry points of libpbrt_host.so (include/pbrt_host.h).
#include "../../include/pbrt_host.h"
#include "api.h"
#include "error.h"
#include "scene.h"

using namespace pbrt;
struct PbrtHostScene {
    std::unique_ptr<LoadedScene> loaded;
    FlatScene flat;
};

static PbrtHostScene *finishLoad() {
    if (!lastLoadedScene) return nullptr;
    PbrtHostScene *s = new PbrtHostScene;
    s->loaded = std::move(lastLoadedScene);
    s->loaded->integrator->Flatten(*s->loaded->scene, &s->flat);
    return s;
}
static bool g_deviceBVH = false;
static Options makeOptions(int quick, const float *crop) {
    Options opt;
    opt.deviceBVH = g_deviceBVH;
    opt.quickRender = quick != 0;
    opt.loadOnly = true;
    if (crop) { opt.cropWindow[0][0] = crop[0]; opt.cropWindow[0][1] = crop[1]; opt.cropWindow[1][0] = crop[2]; opt.cropWindow[1][1] = crop[3]; }
    return opt;
}
// A fatal scene error (FatalError) or an exhausted allocation ends the load with a null scene; nothing is thrown across the C ABI.
template <typename Parse> static PbrtHostScene *load(int quick, const float *crop, Parse parse) {
    lastLoadedScene.reset();
    try {
        pbrtInit(makeOptions(quick, crop));
        parse();
        pbrtCleanup();
        return finishLoad();
    } catch (const FatalError &) {
    } catch (const std::exception &e) { Error("%s", e.what()); }
    try { pbrtCleanup(); } catch (...) {}
    lastLoadedScene.reset();
    return nullptr;
}
extern "C" {
PbrtHostScene *pbrt_host_load_file(const char *filename, int quick, const float *crop) {
    return load(quick, crop, [&]() { pbrtParseFile(filename); });
}
PbrtHostScene *pbrt_host_load_string(const char *text, int quick, const float *crop) {
    return load(quick, crop, [&]() { pbrtParseString(text); });
}
void pbrt_host_free(PbrtHostScene *s) { delete s; }
const PgSceneDesc *pbrt_host_scene_desc(PbrtHostScene *s) { return &s->flat.desc; }
void pbrt_host_render_desc(PbrtHostScene *s, PgRenderDesc *out) { s->loaded->integrator->FillRenderDesc(out); }
void pbrt_host_film_size(PbrtHostScene *s, int *w, int *h) {
    const Film &f = *s->loaded->integrator->camera->film;
    *w = f.croppedPixelBounds[2] - f.croppedPixelBounds[0];
    *h = f.croppedPixelBounds[3] - f.croppedPixelBounds[1];
}
void pbrt_host_film_clear(PbrtHostScene *s) { s->loaded->integrator->camera->film->Clear(); }
void pbrt_host_film_merge(PbrtHostScene *s, const PgRenderDesc *rd, const PgFilmPixel *film, const PgStraySample *strays, int n) {
    s->loaded->integrator->camera->film->MergeShard(*rd, film, strays, n);
}
void pbrt_host_film_merge_shards(PbrtHostScene *s, const PgRenderDesc *full, int n, const PgFilmPixel *const *film, const PgStraySample *const *strays, const int *n_strays) {
    s->loaded->integrator->camera->film->MergeShards(*full, n, film, strays, n_strays);
}
void pbrt_host_film_image(PbrtHostScene *s, float *rgb) {
    std::vector<Float> img;
    s->loaded->integrator->camera->film->ComputeImage(&img);
    std::copy(img.begin(), img.end(), rgb);
}
int pbrt_host_write_image(const char *filename, const float *rgb, int width, int height) {
    return WriteImage(filename, rgb, width, height) ? 0 : -1;
}
int pbrt_host_write_image_window(const char *filename, const float *rgb, int width, int height, int x_offset, int y_offset, int total_x, int total_y) {
    return WriteImage(filename, rgb, width, height, x_offset, y_offset, total_x, total_y) ? 0 : -1;
}
int pbrt_host_write_pfm(const char *filename, const float *rgb, int width, int height) {
    return WriteImagePFM(filename, rgb, width, height) ? 0 : -1;
}
void pbrt_host_hlbvh_build(int n, const float *bounds, int max_prims_in_node, PgBVHNode *nodes, int *n_nodes, int *ordered_prims) {
    std::vector<PgBVHNode> nv;
    std::vector<int> order;
    BVHAccel::HLBVHFromBounds(n, bounds, max_prims_in_node, &nv, &order);
    for (size_t i = 0; i < nv.size(); ++i) nodes[i] = nv[i];
    for (size_t i = 0; i < order.size(); ++i) ordered_prims[i] = order[i];
    *n_nodes = (int)nv.size();
}
int pbrt_host_motion_bounds(const float *start, const float *end, float start_time, float end_time, const float *bounds, float *out) {
    Matrix4x4 a, b;
    for (int i = 0; i < 16; ++i) { a.m[i >> 2][i & 3] = start[i]; b.m[i >> 2][i & 3] = end[i]; }
    const Transform ta(a), tb(b);
    const int failuresBefore = MotionBoundsFailures();
    const Bounds3f r = MotionBounds(ta, start_time, tb, end_time, Bounds3f(Point3f(bounds[0], bounds[1], bounds[2]), Point3f(bounds[3], bounds[4], bounds[5])));
    out[0] = r.pMin.x; out[1] = r.pMin.y; out[2] = r.pMin.z; out[3] = r.pMax.x; out[4] = r.pMax.y; out[5] = r.pMax.z;
    if (MotionBoundsFailures() != failuresBefore) {  // (the reference's CHECK_LE ends its process here)
        Error("MotionBounds: a motion derivative has more than 8 zeros; the box is not the reference's");
        return -1;
    }
    return MotionHasRotation(ta, tb) ? 1 : 0;
}
void pbrt_host_set_device_bvh(int on) { g_deviceBVH = on != 0; }
int pbrt_host_error_count(void) { return ErrorCount(); }
}
