#!/bin/bash
# Builds variants/libpt_hip_<name>.so with extra compiler flags. usage: scripts/build_variant.sh <name> <flags...>
name=$1; shift
mkdir -p variants
common="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-value"
/opt/rocm/bin/hipcc $common -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt "$@" -c -o variants/pt_hip_$name.o path_tracer_amd/csrc/pt_hip.hip || exit 1
/opt/rocm/bin/hipcc $common -Xarch_host -ffp-contract=off -Xarch_device -ffp-contract=fast -fno-hip-fp32-correctly-rounded-divide-sqrt "$@" -c -o variants/pt_shade_fast_$name.o path_tracer_amd/csrc/pt_shade_fast.hip || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/libpt_hip_$name.so variants/pt_hip_$name.o variants/pt_shade_fast_$name.o -ldl && rm -f variants/pt_hip_$name.o variants/pt_shade_fast_$name.o
