// Constant tables embedded into libpbrt_host.so: the Sobol' generator matrices (core/sobolmatrices.h:49-52; data/
// sobol_tables.bin, written by tools/extract_reference_tables.py) and the named medium scattering properties (core/medium.cpp:
// 49-176; data/medium_presets.txt, written by tools/extract_medium_presets.py).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "scene.h"

#ifndef PG_SOBOL_BIN
#error "PG_SOBOL_BIN (path of data/sobol_tables.bin) must be defined by the build"
#endif
__asm__(".section .rodata\n"
        ".balign 8\n"
        ".global pg_sobol_blob\n"
        "pg_sobol_blob:\n"
        ".incbin \"" PG_SOBOL_BIN "\"\n"
        ".global pg_sobol_blob_end\n"
        "pg_sobol_blob_end:\n"
        ".previous\n");
__asm__(".section .rodata\n"
        ".global pg_presets_blob\n"
        "pg_presets_blob:\n"
        ".incbin \"" PG_PRESETS_TXT "\"\n"
        ".global pg_presets_blob_end\n"
        "pg_presets_blob_end:\n"
        ".previous\n");
__asm__(".section .rodata\n"
        ".balign 4\n"
        ".global pg_noise_blob\n"
        "pg_noise_blob:\n"
        ".incbin \"" PG_NOISE_BIN "\"\n"
        ".previous\n");
__asm__(".section .rodata\n"
        ".balign 4\n"
        ".global pg_cmaxmin_blob\n"
        "pg_cmaxmin_blob:\n"
        ".incbin \"" PG_CMAXMIN_BIN "\"\n"
        ".previous\n");
extern "C" const uint32_t pg_cmaxmin_blob[];
extern "C" const int32_t pg_noise_blob[];
extern "C" const unsigned char pg_sobol_blob[], pg_sobol_blob_end[];
extern "C" const char pg_presets_blob[], pg_presets_blob_end[];

namespace pbrt {
const SobolTables &GetSobolTables() {
    static SobolTables t = [] {
        SobolTables r;
        int32_t hdr[5];
        memcpy(hdr, pg_sobol_blob, sizeof(hdr));
        const size_t len = (size_t)(pg_sobol_blob_end - pg_sobol_blob);
        if (hdr[0] != 0x4C424F53 || hdr[1] != 1024 || hdr[2] != 52 ||
            len != sizeof(hdr) + (size_t)hdr[1] * hdr[2] * 4 + (size_t)(hdr[3] + hdr[4]) * hdr[2] * 8) {
            fprintf(stderr, "sobol_tables.bin embedded in libpbrt_host.so is corrupt\n");
            abort();
        }
        r.nDims = hdr[1]; r.matrixSize = hdr[2]; r.vdcRows = hdr[3]; r.vdcInvRows = hdr[4];
        // the blob is 8-aligned and the header is 20 bytes: copy the 64-bit tables to aligned storage once
        static std::vector<uint32_t> m32((size_t)r.nDims * r.matrixSize);
        static std::vector<uint64_t> vdc((size_t)r.vdcRows * r.matrixSize), inv((size_t)r.vdcInvRows * r.matrixSize);
        const unsigned char *p = pg_sobol_blob + sizeof(hdr);
        memcpy(m32.data(), p, m32.size() * 4); p += m32.size() * 4;
        memcpy(vdc.data(), p, vdc.size() * 8); p += vdc.size() * 8;
        memcpy(inv.data(), p, inv.size() * 8);
        r.matrices32 = m32.data(); r.vdc = vdc.data(); r.vdcInv = inv.data();
        return r;
    }();
    return t;
}
const uint32_t *GetMaxMinDistTable() { return pg_cmaxmin_blob; }  // CMaxMinDist, core/lowdiscrepancy.cpp:249-... (data/cmaxmin.bin)
const int32_t *GetNoisePermutation() { return pg_noise_blob; }  // NoisePerm, core/texture.cpp:51-78 (data/noise_perm.bin)
// GetMediumScatteringProperties, core/medium.cpp:181-191
bool GetMediumScatteringProperties(const std::string &name, Float sigma_a[3], Float sigma_prime_s[3]) {
    const std::string text(pg_presets_blob, pg_presets_blob_end);
    size_t pos = 0;
    while (pos < text.size()) {
        size_t eol = text.find('\n', pos);
        if (eol == std::string::npos) eol = text.size();
        const std::string line = text.substr(pos, eol - pos);
        pos = eol + 1;
        const size_t bar = line.find('|');
        if (bar == std::string::npos || line.compare(0, bar, name) != 0 || bar != name.size()) continue;
        float a[3], s[3];
        if (sscanf(line.c_str() + bar + 1, "%f %f %f|%f %f %f", &a[0], &a[1], &a[2], &s[0], &s[1], &s[2]) != 6) return false;
        for (int k = 0; k < 3; ++k) { sigma_a[k] = a[k]; sigma_prime_s[k] = s[k]; }
        return true;
    }
    return false;
}
}  // namespace pbrt
