// Stand-in for OpenEXR's ImfRgbaFile.h: constructors throw (no EXR support in
// the oracle build). Test infrastructure.
#ifndef PBRT_ORACLE_STUB_IMFRGBAFILE_H
#define PBRT_ORACLE_STUB_IMFRGBAFILE_H
#include "ImfRgba.h"
namespace Imf {
class RgbaInputFile {
  public:
    explicit RgbaInputFile(const char *) { throw std::runtime_error("EXR unsupported in oracle build"); }
    Imath::Box2i dataWindow() const { return Imath::Box2i(); }
    Imath::Box2i displayWindow() const { return Imath::Box2i(); }
    void setFrameBuffer(Rgba *, size_t, size_t) {}
    void readPixels(int, int) {}
};
class RgbaOutputFile {
  public:
    RgbaOutputFile(const char *, const Imath::Box2i &, const Imath::Box2i &, RgbaChannels) {
        throw std::runtime_error("EXR unsupported in oracle build");
    }
    void setFrameBuffer(const Rgba *, size_t, size_t) {}
    void writePixels(int) {}
};
}
#endif
