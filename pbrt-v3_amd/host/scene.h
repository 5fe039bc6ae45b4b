// Host-side scene objects of the MI355X path: the closed set of pbrt plugin
// classes the hot path needs (TriangleMesh/Triangle, GeometricPrimitive,
// BVHAccel, Scene, Film, PerspectiveCamera, HaltonSampler, Integrator), with
// the reference's names and parameter semantics, plus FlatScene -- the SoA
// hand-off to the C ABI in include/pbrt_gpu.h.
#ifndef PBRT_AMD_HOST_SCENE_H
#define PBRT_AMD_HOST_SCENE_H
#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <functional>
#include <vector>
#include "../../include/pbrt_gpu.h"
#include "geometry.h"
#include "paramset.h"

namespace pbrt {

// shapes/triangle.h:51-69.  Vertices are stored in world space (triangle.cpp:73-74).
struct TriangleMesh {
    int nTriangles = 0, nVertices = 0;
    std::vector<int> vertexIndices;
    std::vector<Point3f> p;
    std::vector<Normal3f> n;
    std::vector<Vector3f> s;
    std::vector<Float> uv;  // 2 per vertex
    bool reverseOrientation = false, transformSwapsHandedness = false;
    int alphaMask = -1;  // index into the scene's PgAlphaMask table (alphaMask / shadowAlphaMask, triangle.h:66-67), or -1
};
// One Triangle shape per mesh triangle (triangle.cpp:94-110).
struct Triangle {
    std::shared_ptr<TriangleMesh> mesh;
    int triIndex = 0;
    const int *v() const { return &mesh->vertexIndices[3 * triIndex]; }
    Bounds3f WorldBound() const;  // triangle.cpp:180-186
    Float Area() const;           // triangle.cpp:574-580
};
// shapes/sphere.h:46-79: a quadric kept in object space with its two transforms.
struct Sphere {  // also Cylinder (shapes/cylinder.h:46-77) and Disk (shapes/disk.h:46-76): `shape` is a PgQuadricShape
    Sphere(const Transform &o2w, const Transform &w2o, bool reverseOrientation, Float radius, Float zMin, Float zMax, Float phiMax);
    static std::shared_ptr<Sphere> Cylinder(const Transform &o2w, const Transform &w2o, bool ro, Float radius, Float zMin, Float zMax, Float phiMax);
    static std::shared_ptr<Sphere> Disk(const Transform &o2w, const Transform &w2o, bool ro, Float height, Float radius, Float innerRadius, Float phiMax);
    // geometry-only quadrics (no Sample() in the reference): shapes/cone.h:46-73, paraboloid.h:46-74, hyperboloid.h:46-76
    static std::shared_ptr<Sphere> Cone(const Transform &o2w, const Transform &w2o, bool ro, Float height, Float radius, Float phiMax);
    static std::shared_ptr<Sphere> Paraboloid(const Transform &o2w, const Transform &w2o, bool ro, Float radius, Float z0, Float z1, Float phiMax);
    static std::shared_ptr<Sphere> Hyperboloid(const Transform &o2w, const Transform &w2o, bool ro, Point3f point1, Point3f point2, Float phiMax);
    bool CanEmit() const { return shape <= PG_SHAPE_DISK; }
    Transform ObjectToWorld, WorldToObject;
    bool reverseOrientation, transformSwapsHandedness;
    Float radius, zMin, zMax, thetaMin, thetaMax, phiMax;
    int shape = PG_SHAPE_SPHERE;
    Float height = 0, innerRadius = 0;
    Point3f p1, p2;    // hyperboloid
    Float ah = 0, ch = 0;
    Bounds3f WorldBound() const;  // shape.cpp:49 over each shape's ObjectBound()
    Float Area() const;           // sphere.cpp:203, cylinder.cpp:206, disk.cpp:124-126
};
class BVHAccel;
struct GeometricPrimitive;
// What ObjectBegin ... ObjectEnd collects (api.cpp:1509-1544); the first ObjectInstance turns more than one primitive
// into a BVHAccel (api.cpp:1567-1575), which every later instance shares.
struct ObjectDefinition {
    std::vector<GeometricPrimitive> prims;  // before the accelerator is built; afterwards they live in accel->primitives
    std::shared_ptr<BVHAccel> accel;        // null while unbuilt, and for a single primitive
    Bounds3f WorldBound() const;
};
// core/primitive.h:68-89: shape + material + area light (indices into tables).  The shape is a Triangle unless `sphere` is
// set; with `object` set the primitive is a TransformedPrimitive (primitive.h:92-117) of that object definition.
struct GeometricPrimitive {
    Triangle shape;
    std::shared_ptr<Sphere> sphere;
    std::shared_ptr<ObjectDefinition> object;
    // only object instances carry transforms; kept behind a pointer so that the millions of triangle primitives of a large
    // scene stay small (the accelerator build moves every one of them)
    // `animated`: TransformedPrimitive over an AnimatedTransform whose two ends differ (api.cpp:1386-1419 for a moving shape,
    // :1576-1586 for a moving ObjectInstance); the ends' decompositions are taken when the scene is flattened (FlattenScene)
    struct InstanceTransforms {
        Transform InstanceToWorld, WorldToInstance;
        bool animated = false;
        Transform InstanceToWorldEnd, WorldToInstanceEnd;
        Float time[2] = {0, 1};
    };
    std::shared_ptr<const InstanceTransforms> xf;
    int material = -1;
    int areaLight = -1;
    int mediumInside = -1, mediumOutside = -1;  // MediumInterface (core/medium.h:102-116) as indices into Scene::media
    Bounds3f WorldBound() const;
};

// accelerators/bvh.{h,cpp}: SAH / Middle / EqualCounts build + flattenBVHTree.
class BVHAccel {
  public:
    enum class SplitMethod { SAH, HLBVH, Middle, EqualCounts };
    BVHAccel(std::vector<GeometricPrimitive> p, int maxPrimsInNode = 1, SplitMethod splitMethod = SplitMethod::SAH);
    Bounds3f WorldBound() const;
    std::vector<GeometricPrimitive> primitives;  // orderedPrims after construction
    std::vector<PgBVHNode> nodes;
  private:
    struct BuildNode;
    struct PrimInfo;
    // orderedPrims is pre-sized: a leaf over [start, end) writes its primitives to those very positions (they are where the
    // reference's depth-first push_back order puts them), so subtrees can be built on separate threads -- spawnDepth levels
    // of the recursion hand their first child to a new thread -- with a result that does not depend on the thread count
    BuildNode *recursiveBuild(std::vector<PrimInfo> &primitiveInfo, int start, int end, std::atomic<int> *totalNodes,
                              std::vector<GeometricPrimitive> &orderedPrims, int spawnDepth);
    void flattenBVHTree(BuildNode *node, int offset, int spawnDepth);
    struct MortonPrim;
    // the HLBVH build works on primitive numbers: order[k] = number of the k-th primitive of the leaf order
    BuildNode *HLBVHBuild(const std::vector<PrimInfo> &primitiveInfo, int *totalNodes, std::vector<int> &order);
    BuildNode *emitLBVH(const std::vector<PrimInfo> &primitiveInfo, const MortonPrim *mortonPrims, int nPrimitives, int *totalNodes,
                        std::vector<int> &order, int *orderedPrimsOffset, int bitIndex);
    BVHAccel(int maxPrimsInNode, SplitMethod splitMethod);  // no primitives: HLBVHFromBounds
  public:
    // HLBVHBuild + flattenBVHTree over bare bounds (n x {pMin, pMax}): what pg_hlbvh_build computes on the device
    static void HLBVHFromBounds(int n, const float *bounds, int maxPrimsInNode, std::vector<PgBVHNode> *nodes, std::vector<int> *order);
  private:
    BuildNode *buildUpperSAH(std::vector<BuildNode *> &treeletRoots, int start, int end, int *totalNodes);
    const int maxPrimsInNode;
    const SplitMethod splitMethod;
    std::vector<std::unique_ptr<BuildNode[]>> arena;  // chunks of build nodes; threads take whole chunks under arenaMutex
    std::mutex arenaMutex;
    const uint64_t buildId;  // distinguishes this accelerator's arena in the threads' chunk caches (addresses get reused)
    BuildNode *allocNode();
};
std::shared_ptr<BVHAccel> CreateBVHAccelerator(std::vector<GeometricPrimitive> prims, const ParamSet &ps);  // bvh.cpp:740-760
std::shared_ptr<BVHAccel> CreateDefaultBVHAccel(std::vector<GeometricPrimitive> prims);  // std::make_shared<BVHAccel>(prims): one primitive per leaf, SAH (bvh.h:57-58)
// With PbrtOptions.deviceBVH, "hlbvh" accelerators are built by the HIP back end (pg_hlbvh_build, include/pbrt_gpu.h):
// same nodes and primitive order as HLBVHBuild here.  Returns false (after Error) when the back end is missing or fails.
bool DeviceHLBVHBuild(int n, const float *bounds, int maxPrimsInNode, std::vector<PgBVHNode> *nodes, std::vector<int> *order);

struct Scene {  // core/scene.h:50-80
    std::shared_ptr<BVHAccel> aggregate;
    std::vector<PgLight> lights;
    std::vector<PgMaterial> materials;
    std::vector<PgBxDF> bxdfs;
    std::vector<PgTexture> textures;
    std::vector<PgTexturedMaterial> textured;
    std::vector<PgImage> images;
    std::vector<float> texels;
    std::vector<float> envTables;
    std::vector<PgAlphaMask> alphas;
    std::vector<PgMedium> media;
    std::vector<int32_t> mediaGrid;    // per medium: its GridDensityMedium's index in grids, -1 = HomogeneousMedium
    std::vector<PgDensityGrid> grids;
    std::vector<float> gridDensity;
    std::vector<PgBSSRDF> bssrdfs;           // subsurface materials' TabulatedBSSRDF parameters
    std::vector<int32_t> materialBssrdf;     // per material, -1 = none (empty when bssrdfs is)
    std::vector<float> bssrdfTables;
    bool usesNoise = false;
    Bounds3f worldBound;
};

// core/film.{h,cpp} with a box filter (filters/box.cpp).
class Film {
  public:
    Film(const int resolution[2], const Float cropWindow[4], Float filterRadiusX, Float filterRadiusY,
         const std::string &filename, Float scale, Float maxSampleLuminance);
    void GetSampleBounds(int out[4]) const;  // film.cpp:80-86
    // MergeFilmTile (film.cpp:117-130) for one GPU shard's packed tile buffer + stray samples.
    void MergeShard(const PgRenderDesc &rd, const PgFilmPixel *film, const PgStraySample *strays, int nStrays);
    // The n shards of one frame (shard r = the tiles t = r (mod n)), merged in the frame's tile order: what a one-device render merges.
    void MergeShards(const PgRenderDesc &full, int n, const PgFilmPixel *const *film, const PgStraySample *const *strays, const int *nStrays);
    void Clear();
  private:
    void MergeTiles(const PgRenderDesc &rd, int tileFirst, int tileStep, const std::function<const PgFilmPixel *(int)> &blockOf, const PgStraySample *const *strayLists,
                    const int *nStrayLists, int nLists);
  public:
    // WriteImage's arithmetic (film.cpp:169-206): XYZ->RGB, /weight, clamp, *scale.
    void ComputeImage(std::vector<Float> *rgb) const;
    void WriteImage() const;
    int fullResolution[2];
    int croppedPixelBounds[4];
    Float filterRadius[2];
    // filterTable (film.cpp:68-77) and whether the general FilmTile path is needed (anything but a box of radius <= 0.5)
    Float filterTable[256];
    bool filterGeneral = false;
    void TileHalo(int halo[4]) const;  // FilmTile pixel bounds relative to the tile's 16x16 sample block (film.cpp:95-106)
    int TilePixels() const;
    std::string filename;
    Float scale, maxSampleLuminance;
  private:
    struct Pixel { Float xyz[3] = {0, 0, 0}; Float filterWeightSum = 0; };
    std::vector<Pixel> pixels;
};
Film *CreateFilm(const ParamSet &params, Float filterRadiusX, Float filterRadiusY);  // film.cpp:213-252
// MakeFilter (api.cpp:785-803) + the filters' Evaluate(): fills film->filterTable / filterGeneral; returns false for unknown names
bool SetFilmFilter(Film *film, const std::string &name, const ParamSet &params);
void FilterRadiusFor(const std::string &name, const ParamSet &params, Float *xw, Float *yw);
// host/imageio.cpp: image input + MIPMap construction for ImageTexture
bool ReadImage(const std::string &name, int *xres, int *yres, std::vector<RGB> *rgb);
bool ImageGammaDefault(const std::string &filename);
void BuildMIPMap(int resX, int resY, int nc, const std::vector<float> &data, int wrap, PgImage *img, std::vector<float> *pool);
void EWAWeightLut(float lut[128]);
void MIPMapLookup(const PgImage &im, const std::vector<float> &pool, const float st[2], float width, float out[3]);
bool WriteImagePFM(const std::string &filename, const Float *rgb, int width, int height);  // imageio.cpp:437-482
bool WriteImage(const std::string &filename, const Float *rgb, int width, int height);     // imageio.cpp:81-122: .pfm / .png / .tga / .exr by suffix
bool WriteImage(const std::string &filename, const Float *rgb, int width, int height, int xOffset, int yOffset, int totalX, int totalY);

struct PerspectiveCamera {  // ProjectiveCamera (core/camera.h:87-108): cameras/perspective.cpp:45-68 or orthographic.cpp:44-62
    bool orthographic = false;
    bool environment = false;  // EnvironmentCamera (cameras/environment.cpp): no projection, no lens
    Transform CameraToWorld, RasterToCamera;
    // AnimatedTransform CameraToWorld (camera.h:72, transform.h:331-362): the end transform and the two times of a camera that moves
    Transform CameraToWorldEnd;
    Float transformStartTime = 0, transformEndTime = 1;
    bool animated = false;
    Float lensRadius, focalDistance, shutterOpen, shutterClose;
    std::unique_ptr<Film> film;
};
PerspectiveCamera *CreatePerspectiveCamera(const ParamSet &params, const Transform &cam2world, Film *film, bool orthographic = false);

struct HaltonSampler {  // samplers/halton.cpp:65-92
    int samplesPerPixel;
    int baseScales[2], baseExponents[2];
    int sampleStride;
    int multInverse[2];
    bool sampleAtPixelCenter;
    // SobolSampler (samplers/sobol.h:48-71) when sobol is set: the fields above other than samplesPerPixel are unused then
    bool sobol = false;
    int resolution = 0, log2Resolution = 0;
    // the samplers that draw from one RNG stream per tile (PgSamplerKind 2 .. 5): random, stratified, 02sequence, maxmindist
    int kind = 0;                 // PgSamplerKind; 0 / 1 are the two GlobalSamplers above
    int nSampledDimensions = 0;   // PixelSampler::samples1D.size() (sampler.cpp:100-106)
    int xPixelSamples = 1, yPixelSamples = 1;
    bool jitterSamples = true;
};
HaltonSampler *CreateTileSerialSampler(const std::string &name, const ParamSet &params);  // random.cpp:72-75, stratified.cpp:79-87, zerotwosequence.cpp:78-83, maxmin.cpp:82-87
const uint32_t *GetMaxMinDistTable();  // CMaxMinDist [17][32] (data/cmaxmin.bin)
HaltonSampler *CreateSobolSampler(const ParamSet &params, const int sampleBounds[4]);  // sobol.cpp:64-69
// The Sobol' generator matrices embedded into this library (host/sobol.cpp, data/sobol_tables.bin; core/sobolmatrices.h:49-52)
struct SobolTables { const uint32_t *matrices32; const uint64_t *vdc, *vdcInv; int nDims, matrixSize, vdcRows, vdcInvRows; };
const SobolTables &GetSobolTables();
const int32_t *GetNoisePermutation();  // 512 entries
bool GetMediumScatteringProperties(const std::string &name, Float sigma_a[3], Float sigma_prime_s[3]);  // core/medium.cpp:181-191
// bssrdf.cpp: the subsurface materials' tables (core/bssrdf.cpp:43-191)
Float FresnelMoment1(Float eta);
Float FresnelMoment2(Float eta);
void ComputeBeamDiffusionTable(Float g, Float eta, int nRho, int nRadius, std::vector<float> *out);
void SubsurfaceFromDiffuse(const float *table, int nRho, int nRadius, const Float kd[3], const Float mfp[3], Float sigma_a[3], Float sigma_s[3]);
HaltonSampler *CreateHaltonSampler(const ParamSet &params, const int sampleBounds[4]);  // halton.cpp:133-139
// lowdiscrepancy.cpp:2490-2504 with the default-seeded RNG (halton.cpp:69-72).
void ComputeRadicalInversePermutations(int nDims, std::vector<uint16_t> *perms, std::vector<int32_t> *sums);

// Everything pg_scene_create needs, owning the arrays PgSceneDesc points into.
struct FlatScene {
    PgSceneDesc desc;
    std::vector<int32_t> indices, triMaterial, triLight, permSums;
    std::vector<uint32_t> triFlags;
    std::vector<float> P, N, UV, S;
    std::vector<uint16_t> perms;
    std::vector<PgBVHNode> nodes;
    std::vector<PgMaterial> materials;
    std::vector<PgLight> lights;
    std::vector<PgSphere> spheres;
    std::vector<PgBxDF> bxdfs;
    std::vector<PgObject> objects;
    std::vector<PgInstance> instances;
    std::vector<PgTexture> textures;
    std::vector<PgTexturedMaterial> textured;
    std::vector<PgImage> images;
    std::vector<float> texels;
    std::vector<float> envTables;
    std::vector<PgAlphaMask> alphas;
    std::vector<int32_t> triAlpha;
    std::vector<PgMedium> media;
    std::vector<int32_t> mediaGrid;
    std::vector<PgDensityGrid> grids;
    std::vector<float> gridDensity;
    std::vector<PgBSSRDF> bssrdfs;
    std::vector<int32_t> materialBssrdf;
    std::vector<float> bssrdfTables;
    std::vector<int32_t> triMediumInside, triMediumOutside;
    float ewaLut[128];
};

// core/integrator.h:53-58.
class Integrator {
  public:
    virtual ~Integrator() {}
    virtual void Render(const Scene &scene) = 0;
};
// Stands where SamplerIntegrator+PathIntegrator stand in the reference
// (integrator.cpp:228-339, path.cpp:64-188): Render flattens the scene, hands
// it to the HIP back end through the C ABI and merges the film.
class GpuPathIntegrator : public Integrator {
  public:
    GpuPathIntegrator(int maxDepth, std::shared_ptr<PerspectiveCamera> camera, std::shared_ptr<HaltonSampler> sampler,
                      const int pixelBounds[4], Float rrThreshold, const std::string &lightSampleStrategy);
    void Render(const Scene &scene) override;
    void Flatten(const Scene &scene, FlatScene *flat) const;
    void FillRenderDesc(PgRenderDesc *rd) const;
    bool volumetric = false;  // VolPathIntegrator (integrators/volpath.cpp) instead of PathIntegrator
    int cameraMedium = -1;
    std::shared_ptr<PerspectiveCamera> camera;
    std::shared_ptr<HaltonSampler> sampler;
    int maxDepth;
    int pixelBounds[4];
    Float rrThreshold;
    std::string lightSampleStrategy;
};
GpuPathIntegrator *CreatePathIntegrator(const ParamSet &params, std::shared_ptr<HaltonSampler> sampler,
                                        std::shared_ptr<PerspectiveCamera> camera);  // path.cpp:190-213

// Result of a loadOnly parse (Options::loadOnly), consumed by the C API in pbrt_host.h.
struct LoadedScene {
    std::unique_ptr<Scene> scene;
    std::unique_ptr<GpuPathIntegrator> integrator;
};
extern std::unique_ptr<LoadedScene> lastLoadedScene;
}  // namespace pbrt
#endif
