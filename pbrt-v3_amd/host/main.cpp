// pbrt_amd: command-line front end with the reference's flags
// (src/main/pbrt.cpp:76-173): --nthreads --outfile --cropwindow --quick --quiet,
// plus --gpu <id> to pick the HIP device.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "api.h"
#include "error.h"

using namespace pbrt;
static void usage(const char *msg = nullptr) {
    if (msg) fprintf(stderr, "pbrt_amd: %s\n\n", msg);
    fprintf(stderr, R"(usage: pbrt_amd [<options>] <filename.pbrt...>
Rendering options:
  --cropwindow <x0,x1,y0,y1> Specify an image crop window.
  --devicebvh                Build "hlbvh" accelerators on the GPU instead of on the host.
  --gpu <id>                 HIP device to render on (default 0).
  --gpus <n>                 Shard the frame's 16x16 tiles over devices 0..n-1 of this node (one host thread per GPU,
                             film shards gathered peer-to-peer on the first device).
  --gpu-ids <a,b,...>        The same over the listed devices (an id may repeat: several shards on one GPU).
  --help                     Print this help text.
  --nthreads <num>           Host threads for the accelerator build (0 = all cores); rendering runs on the GPU.
  --outfile <filename>       Write the final image to the given filename (.pfm).
  --quick                    Automatically reduce a number of quality settings to render more quickly.
  --quiet                    Suppress all text output other than error messages.
)");
    exit(msg ? 1 : 0);
}
int main(int argc, char *argv[]) {
    Options options;
    std::vector<std::string> filenames;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--nthreads") || !strcmp(argv[i], "-nthreads")) { if (i + 1 == argc) usage("missing value after --nthreads argument"); options.nThreads = atoi(argv[++i]); }
        else if (!strncmp(argv[i], "--nthreads=", 11)) options.nThreads = atoi(&argv[i][11]);
        else if (!strcmp(argv[i], "--outfile") || !strcmp(argv[i], "-outfile")) { if (i + 1 == argc) usage("missing value after --outfile argument"); options.imageFile = argv[++i]; }
        else if (!strncmp(argv[i], "--outfile=", 10)) options.imageFile = &argv[i][10];
        else if (!strcmp(argv[i], "--cropwindow") || !strcmp(argv[i], "-cropwindow")) {
            if (i + 4 >= argc) usage("missing value after --cropwindow argument");
            options.cropWindow[0][0] = atof(argv[++i]); options.cropWindow[0][1] = atof(argv[++i]);
            options.cropWindow[1][0] = atof(argv[++i]); options.cropWindow[1][1] = atof(argv[++i]);
        }
        else if (!strcmp(argv[i], "--gpu")) { if (i + 1 == argc) usage("missing value after --gpu argument"); options.device = atoi(argv[++i]); }
        else if (!strcmp(argv[i], "--gpus")) {
            if (i + 1 == argc) usage("missing value after --gpus argument");
            const int n = atoi(argv[++i]);
            if (n < 1) usage("--gpus needs a positive count");
            options.devices.clear();
            for (int d = 0; d < n; ++d) options.devices.push_back(d);
        } else if (!strcmp(argv[i], "--gpu-ids")) {
            if (i + 1 == argc) usage("missing value after --gpu-ids argument");
            options.devices.clear();
            for (const char *p = argv[++i]; *p;) { options.devices.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p == ',') ++p; }
            if (options.devices.empty()) usage("--gpu-ids needs a comma-separated list of device ids");
        }
        else if (!strcmp(argv[i], "--devicebvh")) options.deviceBVH = true;
        else if (!strcmp(argv[i], "--quick") || !strcmp(argv[i], "-quick")) options.quickRender = true;
        else if (!strcmp(argv[i], "--quiet") || !strcmp(argv[i], "-quiet")) options.quiet = true;
        else if (!strcmp(argv[i], "--help") || !strcmp(argv[i], "-help") || !strcmp(argv[i], "-h")) usage();
        else filenames.push_back(argv[i]);
    }
    if (filenames.empty()) usage("no scene file given");
    try {
        pbrtInit(options);
        for (const std::string &f : filenames) pbrtParseFile(f);
        pbrtCleanup();
    } catch (const FatalError &) {  // already reported through Error(): the reference exits with status 1 here
        return 1;
    }
    return ErrorCount() ? 1 : 0;
}
