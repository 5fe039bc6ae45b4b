#!/bin/bash
# Experiment build of the device library with -DPG_TRACE_STATS: k_trace<false> counts, per wave step, how many lanes expand an
# interior record / test a triangle / are refilled; pg_render prints the frame's totals to stderr.  Not part of the product.
# (the PG_TRACE_STATS blocks live in tools/experiments/removed/trace_stats.diff: patch -p0 < that file first)
# usage (build container): bash tools/experiments/trace_stats.sh   -> gpurun_in_libpbrt_gpu_stats.so; on the GPU box:
#   PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_stats.so python bench.py --steps 1 --warmup 0 --no-cpu-baseline
cd "$(dirname "$0")/../../pbrt-v3_amd" && mkdir -p build_stats
for f in pg_abi pg_hlbvh pg_kernels pg_traverse; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -DPG_TRACE_STATS -c csrc/$f.hip -o build_stats/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC build_stats/*.o -o ../gpurun_in_libpbrt_gpu_stats.so
