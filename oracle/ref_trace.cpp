// Test-infrastructure debugging aid, built against the reference sources where they lie (like gpupath_binding.cpp: the reference's
// api.cpp compiled once more with the path-integrator factory renamed): the UNMODIFIED PathIntegrator::Li, with the radiance of
// every sample of ONE pixel (REF_TRACE_PIXEL="x y") printed as bit patterns -- to be laid beside ORACLE_TRACE_PIXEL's output of
// oracle/pbrt_oracle.c when an image differs.  Run with --nthreads 1.
//   g++ (Makefile.ref's CXXFLAGS) -fno-access-control -c ref_trace.cpp; link with main/pbrt.o, api_gpubind.o, libpbrt_ref.a
#include "integrators/path.h"
#include "integrators/volpath.h"
#include "sampler.h"
#include "paramset.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
namespace pbrt {
PathIntegrator *CreatePathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera);
VolPathIntegrator *CreateVolPathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera);
namespace {
class TracePathIntegrator : public PathIntegrator {
  public:
    TracePathIntegrator(const PathIntegrator &h, std::shared_ptr<const Camera> camera, std::shared_ptr<Sampler> sampler)
        : PathIntegrator(h.maxDepth, camera, sampler, h.pixelBounds, h.rrThreshold, h.lightSampleStrategy) {
        const char *tp = getenv("REF_TRACE_PIXEL");
        if (tp) sscanf(tp, "%d %d", &px, &py);
    }
    Spectrum Li(const RayDifferential &ray, const Scene &scene, Sampler &sampler, MemoryArena &arena, int depth) const override {
        Spectrum L = PathIntegrator::Li(ray, scene, sampler, arena, depth);
        if (sampler.currentPixel.x == px && sampler.currentPixel.y == py) {
            Float rgb[3]; L.ToRGB(rgb);
            unsigned b[3]; memcpy(b, rgb, 12);
            fprintf(stderr, "ref-trace %d %d %lld %08x %08x %08x\n", px, py, (long long)sampler.CurrentSampleNumber(), b[0], b[1], b[2]);
        }
        return L;
    }
  private:
    int px = -1, py = -1;
};
}
PathIntegrator *GpuBind_CreatePathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera) {
    std::unique_ptr<PathIntegrator> host(CreatePathIntegrator(params, sampler, camera));
    return new TracePathIntegrator(*host, camera, sampler);
}
VolPathIntegrator *GpuBind_CreateVolPathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera) {
    return CreateVolPathIntegrator(params, sampler, camera);
}
}
