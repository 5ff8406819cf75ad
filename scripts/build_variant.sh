#!/bin/bash
# Builds variants/libpt_hip_<name>.so with extra compiler flags. usage: scripts/build_variant.sh <name> <flags...>
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-value "$@" -shared -o variants/libpt_hip_$name.so path_tracer_amd/csrc/pt_hip.hip
