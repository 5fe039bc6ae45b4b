// Test-infrastructure probe, built against the reference sources where they lie (like Makefile.ref): prints, as hex floats,
// the RGB coefficients the UNMODIFIED reference derives for MetalMaterial's default copper spectra (materials/metal.cpp:
// CreateMetalMaterial -> Spectrum::FromSampled over the CIE matching curves).  The six numbers are constants of the host
// front end (pbrt-v3_amd/host/api.cpp, kCopperN / kCopperK); tests/test_oracle_vs_reference.py re-checks them when
// /root/reference is present.  Build: make -C oracle -f Makefile.ref _ref/ref_probe
#include "materials/metal.cpp"
#include <cstdio>
int main() {
    using namespace pbrt;
    Spectrum n = Spectrum::FromSampled(CopperWavelengths, CopperN, CopperSamples);
    Spectrum k = Spectrum::FromSampled(CopperWavelengths, CopperK, CopperSamples);
    Float rgb[3];
    n.ToRGB(rgb); printf("copperN %a %a %a  (%.9g %.9g %.9g)\n", rgb[0], rgb[1], rgb[2], rgb[0], rgb[1], rgb[2]);
    k.ToRGB(rgb); printf("copperK %a %a %a  (%.9g %.9g %.9g)\n", rgb[0], rgb[1], rgb[2], rgb[0], rgb[1], rgb[2]);
    return 0;
}
