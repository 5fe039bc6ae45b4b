// Definitions for the glog FLAGS_ globals and the Ptex texture factories that
// the oracle build leaves out (textures/ptex.cpp needs the absent Ptex lib).
#include <glog/logging.h>
#include "pbrt.h"
#include "textures/ptex.h"
int FLAGS_v = 0;
int FLAGS_minloglevel = 0;
int FLAGS_stderrthreshold = 2;
bool FLAGS_logtostderr = false;
std::string FLAGS_log_dir;
namespace pbrt {
PtexTexture<Float> *CreatePtexFloatTexture(const Transform &, const TextureParams &) { return nullptr; }
PtexTexture<Spectrum> *CreatePtexSpectrumTexture(const Transform &, const TextureParams &) { return nullptr; }
}
