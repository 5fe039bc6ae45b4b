#!/bin/bash
# Builds a variant of the device library for an in-call A/B on the GPU box: usage (build container)
#   bash tools/build_variant.sh NAME "FLAGS" [file ...]     -> gpurun_in_libpbrt_gpu_NAME.so (git-ignored; travels with gpurun)
# Only the listed translation units (default: pg_traverse) are recompiled with FLAGS; the others are the product's objects.
# On the GPU box: PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_NAME.so python bench.py ...
NAME=$1; FLAGS=$2; shift 2
FILES=${@:-pg_traverse}
cd "$(dirname "$0")/../pbrt-v3_amd" && make -j8 libpbrt_gpu.so > /dev/null && mkdir -p build_$NAME || exit 1
OBJS=""
for o in build/*.o; do
  f=$(basename $o .o)
  if [[ " $FILES " == *" $f "* ]]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math $FLAGS -c csrc/$f.hip -o build_$NAME/$f.o &
    OBJS="$OBJS build_$NAME/$f.o"
  else OBJS="$OBJS $o"; fi
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o ../gpurun_in_libpbrt_gpu_$NAME.so && echo gpurun_in_libpbrt_gpu_$NAME.so
