#!/usr/bin/env python3
"""Per-kernel resources of the built device library, read off its code objects (no GPU needed): VGPRs, AGPRs, SGPRs, static LDS,
scratch, spills, launch bound, and the waves per SIMD those registers allow on CDNA4 (512 VGPRs per SIMD lane, granule 8:
MI355X_MICROARCH.md's occupancy table).

    python tools/kernel_resources.py [pbrt-v3_amd/libpbrt_gpu.so] > profiles/<round>_kernel_resources.txt

The .hip_fatbin section holds one clang offload bundle per translation unit; each is unbundled for gfx950 and its
NT_AMDGPU_METADATA note parsed."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def waves_per_simd(vgprs, agprs):
    # .vgpr_count of the metadata is the kernel's TOTAL allocation in the unified file (the compiler's TotalNumVgprs = accum_offset +
    # AGPRs, which already contains .agpr_count); 8-register granules.  (Round 2's version added the AGPRs a second time and reported
    # k_shade<2, .> -- 255 registers, 3 of them AGPRs -- at one wave per SIMD; it runs at two.)
    total = ((max(vgprs, 1) + 7) // 8) * 8
    return max(1, min(8, 512 // total))


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "pbrt-v3_amd", "libpbrt_gpu.so")
    rows = []
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__", blob)]
        for i, p in enumerate(starts):
            part, elf = os.path.join(d, f"b{i}.bin"), os.path.join(d, f"co{i}.elf")
            open(part, "wb").write(blob[p:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={elf}"])
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", elf], capture_output=True, text=True, check=True).stdout
            for blk in re.split(r"\n  - \.agpr_count:", "\n" + notes.split("amdhsa.kernels:", 1)[1])[1:]:
                g = lambda k, blk=blk: re.search(r"\n    \." + k + r":\s+(\S+)", blk).group(1)
                rows.append(dict(name=g("name"), agpr=int(blk.split("\n", 1)[0]), vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")),
                                 lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")),
                                 vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")), wg=int(g("max_flat_workgroup_size"))))
    names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True, check=True).stdout.splitlines()
    for r, n in zip(rows, names):
        r["short"] = (re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "")).replace("void ", "") or r["name"]).strip() or r["name"]
    own = sorted((r for r in rows if "rocprim" not in r["short"] and "hipcub" not in r["short"]), key=lambda r: r["short"])
    print(f"# {os.path.relpath(lib, ROOT)}: {len(rows)} kernels in {len(starts)} code objects (gfx950), {len(rows) - len(own)} of them library (hipCUB / rocPRIM)")
    print("# instantiations (none since round 4: the HLBVH build sorts with its own k_radix_*).  waves = waves per SIMD the register file allows (the launch")
    print("# bound or the LDS of a kernel may allow fewer); LDS = static bytes per workgroup; scratch in bytes per lane; spills in registers.")
    print("%-44s %5s %5s %5s %6s %8s %8s %7s %7s %6s" % ("kernel", "vgpr", "agpr", "sgpr", "waves", "LDS B", "scratch", "v-spill", "s-spill", "maxWG"))
    for r in own:
        print("%-44s %5d %5d %5d %6d %8d %8d %7d %7d %6d" % (r["short"][:44], r["vgpr"], r["agpr"], r["sgpr"], waves_per_simd(r["vgpr"], r["agpr"]), r["lds"], r["scratch"],
                                                            r["vspill"], r["sspill"], r["wg"]))


if __name__ == "__main__":
    main()
