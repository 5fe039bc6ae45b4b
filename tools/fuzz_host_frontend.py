#!/usr/bin/env python3
"""Mutation fuzzing of the host front end (parser, API state machine, mesh producers, BVH build, scene flattening) under
AddressSanitizer + UndefinedBehaviorSanitizer.  CPU only.

    python tools/fuzz_host_frontend.py [seed] [count]

Builds pbrt-v3_amd/host/*.cpp (without main.cpp) with -fsanitize=address,undefined together with a small driver of the C API
(pbrt_host_load_string -> pbrt_host_scene_desc / render_desc, every array of the descriptor walked -> pbrt_host_free) into
/tmp/pbrt_host_fuzz/, mutates the golden scenes token-wise (deleted, duplicated, swapped, truncated tokens, hostile numbers,
unbalanced blocks, missing files) and reports every input on which a sanitizer fires, the process dies of a signal or hangs.
Round 2 found seven defects this way (negative vertex indices, non-finite vertices in the SAH bucket index, the error location
of a destroyed tokenizer, non-manifold loopsubdiv control meshes, a file that includes itself, AttributeEnd after a stray
TransformEnd -- an empty-stack read the reference shares --, and an int overflow on a hostile "maxdepth");
tests/test_host_frontend.py keeps one input of each.  The corpus also holds the grid-medium and subsurface scenes
(FUZZ_ONLY=<directory name> restricts it).  Last sweeps: seeds 111, 121, 131 -- 11 000
mutated scenes, 0 problems."""
import glob
import os
import random
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = "/tmp/pbrt_host_fuzz"
DRIVER = r'''
#include "%s/include/pbrt_host.h"
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>
int main(int argc, char **argv) {
    std::ifstream f(argv[1]); std::stringstream ss; ss << f.rdbuf();
    const std::string text = ss.str();
    PbrtHostScene *s = pbrt_host_load_string(text.c_str(), 0, nullptr);
    if (!s) { printf("no scene\n"); return 0; }
    const PgSceneDesc *d = pbrt_host_scene_desc(s);
    PgRenderDesc rd; pbrt_host_render_desc(s, &rd);
    double acc = 0;
    if (d) {
        for (int i = 0; i < d->n_nodes; ++i) acc += d->nodes[i].bmin[0];
        for (int i = 0; i < d->n_tris * 3; ++i) acc += d->indices[i];
        for (int i = 0; i < d->n_tris; ++i) acc += d->tri_material ? d->tri_material[i] : 0;
        for (int i = 0; i < d->n_materials; ++i) acc += d->materials[i].type;
        for (int i = 0; i < d->n_lights; ++i) acc += d->lights[i].type;
        for (int i = 0; i < d->n_materials; ++i) if (d->material_bssrdf && d->material_bssrdf[i] >= 0) {  // subsurface tables, whole
            const PgBSSRDF &b = d->bssrdfs[d->material_bssrdf[i]];
            for (long long k = 0; k < b.n_rho + b.n_radius + 2LL * b.n_rho * b.n_radius + b.n_rho; ++k) acc += d->bssrdf_tables[b.table + k];
        }
        for (int i = 0; i < d->n_media; ++i) if (d->media_grid && d->media_grid[i] >= 0) {  // GridDensityMedium tables: every voxel
            const PgDensityGrid &g = d->grids[d->media_grid[i]];
            for (long long k = 0; k < (long long)g.nx * g.ny * g.nz; ++k) acc += d->grid_density[g.density_offset + k];
        }
    }
    int w, h; pbrt_host_film_size(s, &w, &h);
    printf("ok %%g %%d %%d\n", acc, w, h);
    pbrt_host_free(s);
    return 0;
}
'''
VALUES = ["[", "]", '"', "-1", "1e38", "-1e38", "nan", "inf", "0", "1e-45", '"integer', '"float x"', "WorldEnd", "WorldBegin", "AttributeEnd",
          "AttributeBegin", "ObjectEnd", 'ObjectBegin "a"', 'ObjectInstance "a"', "TransformEnd", "99999999", "2147483647", "-2147483648",
          "4294967296", '"bool x" "true"', '"string filename" "missing.png"', 'Include "nofile.pbrt"', "#", "\n"]


def build():
    os.makedirs(WORK, exist_ok=True)
    drv = os.path.join(WORK, "driver.cpp")
    open(drv, "w").write(DRIVER % ROOT)
    host = os.path.join(ROOT, "pbrt-v3_amd", "host")
    data = os.path.join(ROOT, "pbrt-v3_amd", "data")
    srcs = [s for s in sorted(glob.glob(os.path.join(host, "*.cpp"))) if not s.endswith("main.cpp")]
    defs = [f'-DPG_SOBOL_BIN="{data}/sobol_tables.bin"', f'-DPG_PRESETS_TXT="{data}/medium_presets.txt"', f'-DPG_CIE_BIN="{data}/cie_tables.bin"',
            f'-DPG_NOISE_BIN="{data}/noise_perm.bin"', f'-DPG_CMAXMIN_BIN="{data}/cmaxmin.bin"']
    exe = os.path.join(WORK, "capi_asan")
    subprocess.check_call(["g++", *defs, "-std=c++14", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize=float-divide-by-zero,float-cast-overflow",
                           "-fno-omit-frame-pointer", "-pthread", "-ffp-contract=off", *srcs, drv, "-o", exe, "-ldl", "-lz"])
    return exe


def mutate(rng, text):
    toks = text.split(" ")
    for _ in range(rng.randint(1, 8)):
        k, i = rng.randrange(7), rng.randrange(len(toks))
        if k == 0: toks[i] = ""
        elif k == 1: toks[i] = toks[rng.randrange(len(toks))]
        elif k == 2: toks[i] = rng.choice(VALUES)
        elif k == 3: toks.insert(i, toks[i])
        elif k == 4: del toks[i:i + rng.randint(1, 8)]
        elif k == 5: toks[i] = toks[i][:len(toks[i]) // 2]
        else:
            j = rng.randrange(len(toks)); toks[i], toks[j] = toks[j], toks[i]
        if not toks: toks = [""]
    return " ".join(toks)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    exe = build()
    rng = random.Random(seed)
    gold = os.path.join(ROOT, "tests", "golden")
    dirs = [gold]
    if os.environ.get("FUZZ_ONLY"):  # mutate only that directory's scenes
        dirs = [d for d in dirs if os.path.basename(d) == os.environ["FUZZ_ONLY"]]
    srcs = [s for d in dirs for s in sorted(glob.glob(os.path.join(d, "*.pbrt"))) if os.path.getsize(s) < 20000]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    bad = []
    for n in range(count):
        p = os.path.join(WORK, f"m{seed}_{n}.pbrt")
        open(p, "w").write(mutate(rng, open(rng.choice(srcs)).read()))
        try:
            r = subprocess.run([exe, p], capture_output=True, text=True, cwd=gold, timeout=60, env=env)  # textures / includes resolve next to the goldens
        except subprocess.TimeoutExpired:
            bad.append((p, "TIMEOUT", "")); continue
        if "AddressSanitizer" in r.stderr or "runtime error" in r.stderr or r.returncode < 0:
            keep = [l.strip() for l in r.stderr.splitlines() if "AddressSanitizer" in l or "runtime error" in l or l.strip().startswith(("#0 ", "#1 ", "#2 "))]
            bad.append((p, r.returncode, " | ".join(keep[:5])[:400]))
        else:
            os.remove(p)
    print(f"{count} mutated scenes (seed {seed}): {len(bad)} problem(s)")
    for b in bad: print(b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
