#!/bin/bash
# Timing experiments: variant builds of ONE translation unit of libpb_hip.so, linked with the other objects
# of the tree into build/variants/libpb_<name>.so (select with PB_LIBPATH).  usage:
#   tools/build_variants.sh k_detect3.hip noB="-DD3_DBG=2" d8="-DD3_DEPTH=8" ...
# Everything that makes results INVALID lives in such builds only (CH_ABL, FFT_ABL, D2_ABL, and PB_SKIP behind
# -DPB_EXPERIMENTS=1: `tools/build_variants.sh pb_api.hip exp="-DPB_EXPERIMENTS=1"` or `make -C vlite-fast_amd/csrc exp`).
set -e
cd "$(dirname "$0")/../vlite-fast_amd/csrc"
src=$1; shift
make -s
mkdir -p ../../build/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -I../../include -I. -w"
others=$(ls *.o | grep -v "\.fg\.o$" | grep -v "\.exp\.o$" | grep -v "^${src%.hip}.o$")
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  /opt/rocm/bin/hipcc $FLAGS $defs -c -o ../../build/variants/${src%.hip}_$name.o $src
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/variants/libpb_$name.so ../../build/variants/${src%.hip}_$name.o $others -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
  echo built libpb_$name.so
done
