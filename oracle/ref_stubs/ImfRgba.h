// Stand-in for OpenEXR's ImfRgba.h (submodule absent). Any attempt to read or
// write an EXR throws, which pbrt's imageio catches and reports. The oracle
// always renders to .pfm. Test infrastructure.
#ifndef PBRT_ORACLE_STUB_IMFRGBA_H
#define PBRT_ORACLE_STUB_IMFRGBA_H
#include <stdexcept>
namespace Imath {
struct V2i { int x, y; V2i() : x(0), y(0) {} V2i(int a, int b) : x(a), y(b) {} };
struct Box2i { V2i min, max; Box2i() {} Box2i(V2i a, V2i b) : min(a), max(b) {} };
}
namespace Imf {
struct Rgba {
    float r, g, b, a;
    Rgba() : r(0), g(0), b(0), a(1) {}
    Rgba(float r_, float g_, float b_, float a_ = 1.f) : r(r_), g(g_), b(b_), a(a_) {}
};
enum RgbaChannels { WRITE_RGB = 7 };
}
#endif
