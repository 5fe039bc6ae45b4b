#!/usr/bin/env python3
"""Writes the constant tables of the host front end that are numeric data of the reference rather than code:

pbrt-v3_amd/data/cie_tables.bin: the CIE 1931 colour matching functions at 1 nm from 360 to 830 nm (core/spectrum.cpp:190-...:
CIE_X, CIE_Y, CIE_Z, CIE_lambda; 471 floats each, in that order), which turn "spectrum" / "blackbody" parameters into RGB.

pbrt-v3_amd/data/noise_perm.bin: the 512-entry permutation table of the Perlin noise functions (core/texture.cpp:51-78; int32).

pbrt-v3_amd/data/cmaxmin.bin: CMaxMinDist, the generator matrices of the MaxMinDistSampler (core/lowdiscrepancy.cpp:249-...; uint32 [17][32]).

pbrt-v3_amd/data/sobol_tables.bin: the Sobol' generator matrices the SobolSampler reads (core/sobolmatrices.h:49-52:
SobolMatrices32, VdCSobolMatrices, VdCSobolMatricesInv -- Gruenschloss' published tables, numeric constants of the sequence
itself like the table of primes).  They are taken from the read-only data section of the reference binary built by
oracle/Makefile.ref (oracle/_ref/pbrt_oracle), so this script runs in the build container only; the .bin is committed and
embedded into libpbrt_host.so (host/sobol.cpp).

Layout (little endian): int32 magic 'SOBL', nDims (1024), matrixSize (52), vdcRows (25), vdcInvRows (26);
uint32 SobolMatrices32[nDims*matrixSize]; uint64 VdCSobolMatrices[vdcRows][matrixSize]; uint64 VdCSobolMatricesInv[vdcInvRows][matrixSize]."""
import os
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_oracle")
OUT = os.path.join(ROOT, "pbrt-v3_amd", "data", "sobol_tables.bin")
OUT_CIE = os.path.join(ROOT, "pbrt-v3_amd", "data", "cie_tables.bin")
OUT_NOISE = os.path.join(ROOT, "pbrt-v3_amd", "data", "noise_perm.bin")
OUT_CMAXMIN = os.path.join(ROOT, "pbrt-v3_amd", "data", "cmaxmin.bin")


def main():
    if not os.path.exists(REF):
        sys.exit("build oracle/_ref first: make -C oracle -f Makefile.ref")
    syms = {}
    for line in subprocess.run(["nm", "-S", "-C", REF], capture_output=True, text=True, check=True).stdout.splitlines():
        parts = line.split(None, 3)
        if len(parts) == 4 and parts[3] in ("pbrt::SobolMatrices32", "pbrt::VdCSobolMatrices", "pbrt::VdCSobolMatricesInv", "pbrt::CIE_X", "pbrt::CIE_Y",
                                              "pbrt::CIE_Z", "pbrt::CIE_lambda", "pbrt::NoisePerm", "pbrt::CMaxMinDist"):
            syms[parts[3].split("::")[1]] = (int(parts[0], 16), int(parts[1], 16))
    assert len(syms) == 9, syms
    # map virtual addresses to file offsets through the section headers
    secs = []
    for line in subprocess.run(["readelf", "-S", "-W", REF], capture_output=True, text=True, check=True).stdout.splitlines():
        line = line.strip()
        if not line.startswith("["): continue
        f = line.split("]", 1)[1].split()
        if len(f) >= 5 and f[1] in ("PROGBITS",):
            secs.append((int(f[2], 16), int(f[3], 16), int(f[4], 16)))  # addr, offset, size
    blob = open(REF, "rb").read()

    def read(name):
        addr, size = syms[name]
        for a, off, sz in secs:
            if a <= addr and addr + size <= a + sz: return blob[off + addr - a: off + addr - a + size]
        raise SystemExit(f"{name}: address not in a PROGBITS section")

    m32, vdc, inv = read("SobolMatrices32"), read("VdCSobolMatrices"), read("VdCSobolMatricesInv")
    size = 52
    assert len(m32) == 1024 * size * 4 and len(vdc) % (size * 8) == 0 and len(inv) % (size * 8) == 0
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "wb") as fo:
        fo.write(struct.pack("<5i", 0x4C424F53, 1024, size, len(vdc) // (size * 8), len(inv) // (size * 8)))
        fo.write(m32); fo.write(vdc); fo.write(inv)
    print(OUT, os.path.getsize(OUT), "bytes; vdc rows", len(vdc) // (size * 8), "inv rows", len(inv) // (size * 8))
    cie = [read(n) for n in ("CIE_X", "CIE_Y", "CIE_Z", "CIE_lambda")]
    assert all(len(c) == 471 * 4 for c in cie)
    with open(OUT_CIE, "wb") as fo:
        for c in cie: fo.write(c)
    print(OUT_CIE, os.path.getsize(OUT_CIE), "bytes")
    perm = read("NoisePerm")  # Perlin's permutation of 0..255, stored twice (core/texture.cpp:51-78)
    assert len(perm) == 512 * 4
    open(OUT_NOISE, "wb").write(perm)
    print(OUT_NOISE, os.path.getsize(OUT_NOISE), "bytes")
    cmm = read("CMaxMinDist")  # the MaxMinDistSampler's generator matrices for 1 .. 2^16 samples (core/lowdiscrepancy.cpp:249-...; uint32 [17][32])
    assert len(cmm) == 17 * 32 * 4
    open(OUT_CMAXMIN, "wb").write(cmm)
    print(OUT_CMAXMIN, os.path.getsize(OUT_CMAXMIN), "bytes")


if __name__ == "__main__":
    main()
