#!/usr/bin/env python3
"""Where a kernel's VGPR pressure peaks, by source line (no GPU needed): the translation unit is compiled to device IR with the product's
flags, the kernel cut out, taken through llc to the machine scheduler and LLVM's GCNRegPressurePrinter (`amdgpu-print-rp`) run over it.
The numbers are the live virtual registers before allocation -- what the allocator has to fit under the occupancy target; spills appear
where they exceed it.

    python tools/reg_pressure.py pg_traverse 'k_traceILi0ELi5EE' [-DFLAG ...] [--top 25] [--blocks]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math"]


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("-")]
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    top = 25
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1]); args = [a for a in args if a != str(top)]
    tu, pat = args[0], args[1]
    work = tempfile.mkdtemp(prefix="regp_")
    bc = os.path.join(work, "tu.bc")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *FLAGS, *extra, "--offload-device-only", "-emit-llvm", "-gline-tables-only", "-c",
                           os.path.join(ROOT, "pbrt-v3_amd", "csrc", tu + ".hip"), "-o", bc], stderr=subprocess.DEVNULL)
    names = subprocess.check_output([os.path.join(LLVM, "llvm-dis"), bc, "-o", "-"], text=True)
    funcs = sorted(set(re.findall(r"define [^@]*amdgpu_kernel [^@]*@(\S+?)\(", names)))
    match = [f for f in funcs if re.search(pat, f)]
    if len(match) != 1:
        sys.exit(f"pattern {pat!r} matches {len(match)} kernels: {match[:8]}")
    one = os.path.join(work, "one.bc")
    subprocess.check_call([os.path.join(LLVM, "opt"), "-passes=internalize,globaldce", "-internalize-public-api-list=" + match[0], bc, "-o", one])
    mir = os.path.join(work, "one.mir")
    subprocess.check_call([os.path.join(LLVM, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-O3", "-stop-after=machine-scheduler", one, "-o", mir], stderr=subprocess.DEVNULL)
    rp = subprocess.run([os.path.join(LLVM, "llc"), "-mtriple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-run-pass=amdgpu-print-rp", mir, "-o", "/dev/null"],
                        capture_output=True, text=True).stderr
    by_line, block, blocks, peak = {}, None, [], (0, "")
    for ln in rp.split("\n"):
        m = re.match(r"\s+(bb\.\d+)", ln)
        if m and "Live-" not in ln and re.match(r"\s+bb\.\d+[ .(:]", ln):
            block = m.group(1); blocks.append([block, 0, None]); continue
        m = re.match(r"\s+(\d+)\s+(\d+)\s+(\S.*)$", ln)
        if not m:
            continue
        v = int(m.group(2))
        locs = re.findall(r"(/\S+?):(\d+):\d+", m.group(3).split("debug-location")[-1]) if "debug-location" in m.group(3) else []
        # innermost position, and (when inlined) the line of the kernel's own source it was inlined at
        key = (os.path.basename(locs[0][0]) + ":" + locs[0][1] + (" @ " + os.path.basename(locs[-1][0]) + ":" + locs[-1][1] if len(locs) > 1 else ""), 0) if locs else ("?", 0)
        if v > by_line.get(key, (0, ""))[0]:
            by_line[key] = (v, m.group(3)[:110])
        if blocks:
            blocks[-1][1] = max(blocks[-1][1], v)
        if v > peak[0]:
            peak = (v, key)
    print(f"# {match[0]} {' '.join(extra)}: peak VGPR pressure {peak[0]} at {peak[1][0]}")
    print("# highest pressure by source line (live virtual VGPRs at the instruction):")
    for key, (v, ins) in sorted(by_line.items(), key=lambda kv: -kv[1][0])[:top]:
        print(f"{v:4d}  {key[0]:60s} {ins[:70]}")
    if "--blocks" in sys.argv:
        print("# per basic block: max pressure")
        for b, v, _ in blocks:
            print(f"{b:8s} {v}")


if __name__ == "__main__":
    main()
