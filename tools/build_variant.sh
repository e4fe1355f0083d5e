#!/bin/bash
# Dev tool: an A/B build of libort_hip.so with extra -D flags in SOME translation units, next to the production objects:
#   tools/build_variant.sh NAME "-DORT_SCAT_WAVES=8" ort_k_scatter ort_hip   ->  build/ab/libort_NAME.so   (use with ORT_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../opticalraytrace_amd/csrc"
name=$1; flags=$2; shift 2
make -j8 libort_hip.so > /dev/null
mkdir -p ../../build/ab/$name
FL="-O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize"
objs=""
for u in ort_hip ort_k_prog64 ort_k_strict ort_k_wide ort_k_prog32 ort_k_fast ort_k_generic ort_k_scatter ort_k_batch ort_k_exp; do
  if [[ " $* " == *" $u "* ]]; then
    /opt/rocm/bin/hipcc $FL $flags -DORT_BUILD_ID="\"variant-$name\"" -c -o ../../build/ab/$name/$u.o $u.hip &
    objs="$objs ../../build/ab/$name/$u.o"
  else
    objs="$objs ../../build/obj/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/ab/libort_$name.so $objs -ldl
echo build/ab/libort_$name.so
