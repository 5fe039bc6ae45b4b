#!/usr/bin/env python3
"""Builds libpbrt_gpu_emulated.so: the product's device translation units (pbrt-v3_amd/csrc/pg_abi.hip, pg_kernels.hip, pg_traverse.hip, pg_hlbvh.hip)
compiled for the HOST under tests/emu/hip_emu.h, exporting the same C ABI as libpbrt_gpu.so.  TEST INFRASTRUCTURE.

    python tests/emu/build_emulated.py OUTDIR      ->  OUTDIR/libpbrt_gpu_emulated.so

The sources are compiled from temporary copies with four mechanical rewrites, all of them about what a host compiler or a fiber
scheduler cannot take literally (the arithmetic is untouched):
  1. `extern __shared__ uint2 ldsStack[]` (dynamic LDS of k_trace) -> a pointer to the emulator's per-block buffer;
  2. two `asm volatile("" : "+v"(...))` register-scheduling barriers of pg_kernels.hip (no effect on values) are dropped;
  3. wave-level calls that sit in DIVERGENT code are marked, since only the lanes inside the branch take part in them:
     k_trace's slab masks, near / far masks and the free-order any-hit masks (every lane uses its own bit only -> emu_ballot_own), and k_shade's
     readfirstlane / ballot pair that decides about the batched Halton draw (-> emu_*_div: served before the lanes that skipped ahead);
  4. one load that the hardware executes in lockstep for the whole wave before lane 0's atomicAdd (k_trace's "is this region drained"
     test) is exchanged explicitly, because fibers reach it at different times.
pg_hlbvh.hip (the device HLBVH build with its own radix sort) compiles unchanged."""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "pbrt-v3_amd", "csrc")
EM = os.path.join(ROOT, "tests", "emu")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def patched_sources(out):
    t = open(os.path.join(CS, "pg_traverse.hip")).read()
    old = "extern __shared__ uint2 ldsStack[];"
    assert t.count(old) == 1
    t = t.replace(old, "uint2 *ldsStack = (uint2 *)emu::dyn_shared;")
    i = t.index("PG_DEV unsigned long long slab_mask(")
    j = t.index("\n}\n", i)
    assert t[i:j].count("__ballot(") == 5
    t = t[:i] + t[i:j].replace("__ballot(", "emu_ballot_own(") + t[j:]
    for name in ("mNeg", "mNear", "mFar"):  # (every form of the interior step that declares them)
        t, n = re.subn(r"const unsigned long long %s = [^\n]*\n" % name, lambda m: m.group(0).replace("__ballot(", "emu_ballot_own("), t)
        assert n >= 1, name
    m = re.search(r"const bool h0 = [^\n]*\n", t)  # free-order any-hit: the two box-test masks, own bit only
    assert m and m.group(0).count("__ballot(") == 2
    t = t[:m.start()] + m.group(0).replace("__ballot(", "emu_ballot_own(") + t[m.end():]
    old = "__hip_atomic_load(cursor, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)"
    assert t.count(old) == 1
    t = t.replace(old, "emu_readfirstlane(" + old + ")")
    t = "\n".join(l for l in t.split("\n") if "asm volatile(\"\" : \"+v\"" not in l)  # register-scheduling barriers (no effect on values)
    open(os.path.join(out, "pg_traverse.hip"), "w").write(t)

    k = open(os.path.join(CS, "pg_kernels.hip")).read()
    lines = [l for l in k.split("\n") if 'asm volatile("" : "+v"' not in l]
    assert len(k.split("\n")) - len(lines) == 2
    k = "\n".join(lines)
    for old, new in (("const int dimU = __builtin_amdgcn_readfirstlane(dim);", "const int dimU = emu_readfirstlane_div(dim);"),
                     ("if (__ballot(!can || dim != dimU) == 0)", "if (emu_ballot_div(!can || dim != dimU) == 0)")):
        assert k.count(old) == 2, old
        k = k.replace(old, new)
    open(os.path.join(out, "pg_kernels.hip"), "w").write(k)
    shutil.copy(os.path.join(CS, "pg_abi.hip"), os.path.join(out, "pg_abi.hip"))
    shutil.copy(os.path.join(CS, "pg_hlbvh.hip"), os.path.join(out, "pg_hlbvh.hip"))  # (its own radix sort since round 4: nothing to stub)


def main():
    out = os.path.abspath(sys.argv[1])
    os.makedirs(out, exist_ok=True)
    patched_sources(out)
    flags = ["--cuda-host-only", "-O1", "-ffp-contract=off", "-fPIC", "-w", "-DPG_TEST_HOOKS"] + os.environ.get("PBRT_EMU_DEFINES", "").split()  # e.g. -DPG_ORDER_WINDOW=1024: several windows on small scenes
    procs = [subprocess.Popen([HIPCC, *flags, "-I" + CS, "-I" + EM, "-include", os.path.join(EM, "hip_emu.h"), "-c", os.path.join(out, f + ".hip"), "-o", os.path.join(out, f + ".o")])
             for f in ("pg_traverse", "pg_kernels", "pg_abi", "pg_hlbvh")]
    procs.append(subprocess.Popen([HIPCC, *flags, "-U_FORTIFY_SOURCE", "-D_FORTIFY_SOURCE=0", "-I" + EM, "-c", os.path.join(EM, "hip_emu.cpp"), "-o", os.path.join(out, "hip_emu.o")]))
    if any(p.wait() for p in procs):
        sys.exit("build_emulated: compilation failed")
    subprocess.check_call([HIPCC, "--cuda-host-only", "-shared", "-fPIC", *[os.path.join(out, f + ".o") for f in ("pg_abi", "pg_kernels", "pg_traverse", "pg_hlbvh", "hip_emu")],
                           "-o", os.path.join(out, "libpbrt_gpu_emulated.so")])
    print(os.path.join(out, "libpbrt_gpu_emulated.so"))


if __name__ == "__main__":
    main()
