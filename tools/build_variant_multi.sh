#!/bin/bash
# Timing experiments: a variant build of SEVERAL translation units of libpb_hip.so with the same -D flags, linked with
# the other objects of the tree into build/variants/libpb_<name>.so (select with PB_LIBPATH).  usage:
#   tools/build_variant_multi.sh t2lds "-DFFT_T2_LDS=1" k_channelize.hip k_channelize_pfb.hip
# (tools/build_variants.sh: several variants of ONE unit.)  Builds whose flags make results INVALID live here only.
set -e
cd "$(dirname "$0")/../vlite-fast_amd/csrc"
name=$1; defs=$2; shift 2
make -s
mkdir -p ../../build/variants
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -I../../include -I. -w"
objs=""
skip=""
for src in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $defs -c -o ../../build/variants/${src%.hip}_$name.o $src &
  objs="$objs ../../build/variants/${src%.hip}_$name.o"
  skip="$skip ${src%.hip}.o"
done
wait
others=$(ls *.o | grep -v "\.exp\.o$" | grep -v "\.fg\.o$")
for s in $skip; do others=$(echo "$others" | grep -v "^$s$"); done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../build/variants/libpb_$name.so $objs $others -L/opt/rocm/lib -lhipfft -Wl,-rpath,/opt/rocm/lib
echo built libpb_$name.so
