#!/bin/bash
# Build an A/B variant of libccsm into ccsmeth_amd/lib/variants/ (git-ignored, travels to the GPU box): tools/ab_build.sh <name> <hipcc flags...>
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/ccsmeth_amd/lib/variants
name=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" $R/ccsmeth_amd/csrc/ccsm_api.hip -o $R/ccsmeth_amd/lib/variants/libccsm_$name.so
echo built $R/ccsmeth_amd/lib/variants/libccsm_$name.so
