// The Sobol' generator matrices (core/sobolmatrices.h:49-52) as a data blob inside libpbrt_host.so.
// data/sobol_tables.bin is written by tools/extract_sobol_tables.py; its layout is described there.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
extern "C" const unsigned char pg_sobol_blob[], pg_sobol_blob_end[];

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
}  // namespace pbrt
