#!/usr/bin/env python3
"""Writes pbrt-v3_amd/data/medium_presets.txt: the named scattering properties "string preset" selects in MakeNamedMedium
(GetMediumScatteringProperties, core/medium.cpp:181-191; measurements published by Jensen et al. 2001 and Narasimhan et al.
2006).  The names are read from the reference's table, the values are what the reference build itself returns for them
(oracle/_ref/ref_probe presets ...), so this runs in the build container only; the text file is committed and embedded into
libpbrt_host.so.  One line per preset: NAME|sigma_a r g b|sigma_prime_s r g b  (mm^-1, as floats)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/core/medium.cpp"
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")
OUT = os.path.join(ROOT, "pbrt-v3_amd", "data", "medium_presets.txt")


def main():
    if not (os.path.exists(SRC) and os.path.exists(PROBE)):
        sys.exit("needs /root/reference and oracle/_ref/ref_probe (make -C oracle -f Makefile.ref _ref/ref_probe)")
    text = open(SRC).read()
    table = text[text.index("SubsurfaceParameterTable[]"):text.index("GetMediumScatteringProperties")]
    names = re.findall(r'\{\s*"([^"]+)"\s*,', table)
    out = subprocess.run([PROBE, "presets"] + names, capture_output=True, text=True, check=True).stdout
    assert len(out.splitlines()) == len(names)
    open(OUT, "w").write(out)
    print(OUT, len(names), "presets")


if __name__ == "__main__":
    main()
