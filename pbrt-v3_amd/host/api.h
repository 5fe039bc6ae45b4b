// pbrt's scene-description API (src/core/api.h): the functions the parser
// calls, with the same names, argument meaning and Error()/Warning()
// behaviour.  WorldEnd builds the Scene (BVHAccel) and the Integrator and
// calls Integrator::Render, which is the single host<->device boundary
// (api.cpp:1590-1623 in the reference).
#ifndef PBRT_AMD_HOST_API_H
#define PBRT_AMD_HOST_API_H
#include <string>
#include "geometry.h"
#include "paramset.h"

namespace pbrt {
struct Options {  // src/core/pbrt.h:171-185
    int nThreads = 0;
    bool quickRender = false;
    bool quiet = false;
    std::string imageFile;
    Float cropWindow[2][2] = {{0, 1}, {0, 1}};
    // additions for the MI355X back end
    bool loadOnly = false;  // WorldEnd keeps the Scene/Integrator instead of rendering
    int device = 0;         // HIP device the integrator renders on
    bool deviceBVH = false; // build "hlbvh" accelerators on the device (pg_hlbvh_build) instead of on the host
};
extern Options PbrtOptions;

// fileutil.cpp
std::string DirectoryContaining(const std::string &filename);
void SetSearchDirectory(const std::string &dirname);
std::string ResolveFilename(const std::string &filename);
std::string AbsolutePath(const std::string &filename);

void pbrtInit(const Options &opt);
void pbrtCleanup();
void pbrtParseFile(const std::string &filename);
void pbrtParseString(const std::string &str);

void pbrtIdentity();
void pbrtTranslate(Float dx, Float dy, Float dz);
void pbrtRotate(Float angle, Float ax, Float ay, Float az);
void pbrtScale(Float sx, Float sy, Float sz);
void pbrtLookAt(Float ex, Float ey, Float ez, Float lx, Float ly, Float lz, Float ux, Float uy, Float uz);
void pbrtConcatTransform(Float transform[16]);
void pbrtTransform(Float transform[16]);
void pbrtCoordinateSystem(const std::string &);
void pbrtCoordSysTransform(const std::string &);
void pbrtActiveTransformAll();
void pbrtActiveTransformEndTime();
void pbrtActiveTransformStartTime();
void pbrtTransformTimes(Float start, Float end);
void pbrtPixelFilter(const std::string &name, const ParamSet &params);
void pbrtFilm(const std::string &type, const ParamSet &params);
void pbrtSampler(const std::string &name, const ParamSet &params);
void pbrtAccelerator(const std::string &name, const ParamSet &params);
void pbrtIntegrator(const std::string &name, const ParamSet &params);
void pbrtCamera(const std::string &, const ParamSet &cameraParams);
void pbrtMakeNamedMedium(const std::string &name, const ParamSet &params);
void pbrtMediumInterface(const std::string &insideName, const std::string &outsideName);
void pbrtWorldBegin();
void pbrtAttributeBegin();
void pbrtAttributeEnd();
void pbrtTransformBegin();
void pbrtTransformEnd();
void pbrtTexture(const std::string &name, const std::string &type, const std::string &texname, const ParamSet &params);
void pbrtMaterial(const std::string &name, const ParamSet &params);
void pbrtMakeNamedMaterial(const std::string &name, const ParamSet &params);
void pbrtNamedMaterial(const std::string &name);
void pbrtLightSource(const std::string &name, const ParamSet &params);
void pbrtAreaLightSource(const std::string &name, const ParamSet &params);
void pbrtShape(const std::string &name, const ParamSet &params);
void pbrtReverseOrientation();
void pbrtObjectBegin(const std::string &name);
void pbrtObjectEnd();
void pbrtObjectInstance(const std::string &name);
void pbrtWorldEnd();
}  // namespace pbrt
#endif
