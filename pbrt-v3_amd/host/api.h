// pbrt's scene-description API (src/core/api.h): the functions the parser
// calls, with the same names, argument meaning and Error()/Warning()
// behaviour.  WorldEnd builds the Scene (BVHAccel) and the Integrator and
// calls Integrator::Render, which is the single host<->device boundary
// (api.cpp:1590-1623 in the reference).
#ifndef PBRT_AMD_HOST_API_H
#define PBRT_AMD_HOST_API_H
#include <string>
#include <vector>
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
    std::vector<int> devices;  // --gpus / --gpu-ids: render the frame sharded over these devices (one host thread each); empty = `device` alone
    bool deviceBVH = false; // build "hlbvh" accelerators on the device (pg_hlbvh_build) instead of on the host
};
extern Options PbrtOptions;

// fileutil.cpp
std::string DirectoryContaining(const std::string &filename);
void SetSearchDirectory(const std::string &dirname);
std::string ResolveFilename(const std::string &filename);
std::string AbsolutePath(const std::string &filename);

// ---- session: pbrtInit .. pbrtCleanup bracket one scene description; the two parse entry points feed the directives below
void pbrtInit(const Options &options);
void pbrtParseFile(const std::string &sceneFile);
void pbrtParseString(const std::string &sceneText);
void pbrtCleanup();

// ---- block structure of a scene description
void pbrtWorldBegin();
void pbrtWorldEnd();
void pbrtAttributeBegin();
void pbrtAttributeEnd();
void pbrtTransformBegin();
void pbrtTransformEnd();
void pbrtObjectBegin(const std::string &objectName);
void pbrtObjectEnd();
void pbrtObjectInstance(const std::string &objectName);

// ---- current transformation matrix (CTM) and named coordinate systems
void pbrtIdentity();
void pbrtTransform(Float rowMajor4x4[16]);
void pbrtConcatTransform(Float rowMajor4x4[16]);
void pbrtTranslate(Float tx, Float ty, Float tz);
void pbrtScale(Float x, Float y, Float z);
void pbrtRotate(Float degrees, Float axisX, Float axisY, Float axisZ);
void pbrtLookAt(Float eyeX, Float eyeY, Float eyeZ, Float atX, Float atY, Float atZ, Float upX, Float upY, Float upZ);
void pbrtCoordinateSystem(const std::string &csName);
void pbrtCoordSysTransform(const std::string &csName);
void pbrtTransformTimes(Float tStart, Float tEnd);
void pbrtActiveTransformStartTime();
void pbrtActiveTransformEndTime();
void pbrtActiveTransformAll();
void pbrtReverseOrientation();

// ---- render options (before WorldBegin)
void pbrtCamera(const std::string &cameraType, const ParamSet &params);
void pbrtFilm(const std::string &filmType, const ParamSet &params);
void pbrtPixelFilter(const std::string &filterType, const ParamSet &params);
void pbrtSampler(const std::string &samplerType, const ParamSet &params);
void pbrtIntegrator(const std::string &integratorType, const ParamSet &params);
void pbrtAccelerator(const std::string &acceleratorType, const ParamSet &params);

// ---- graphics state and scene content
void pbrtMakeNamedMedium(const std::string &mediumName, const ParamSet &params);
void pbrtMediumInterface(const std::string &insideMedium, const std::string &outsideMedium);
void pbrtTexture(const std::string &textureName, const std::string &valueType, const std::string &textureClass, const ParamSet &params);
void pbrtMaterial(const std::string &materialType, const ParamSet &params);
void pbrtMakeNamedMaterial(const std::string &materialName, const ParamSet &params);
void pbrtNamedMaterial(const std::string &materialName);
void pbrtLightSource(const std::string &lightType, const ParamSet &params);
void pbrtAreaLightSource(const std::string &lightType, const ParamSet &params);
void pbrtShape(const std::string &shapeType, const ParamSet &params);
}  // namespace pbrt
#endif
