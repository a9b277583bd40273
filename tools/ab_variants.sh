#!/bin/bash
# A/B of library variants on the GPU box (tools/ab_build.sh <name> <flags> built them into ccsmeth_amd/lib/variants/): per variant one
# tools/gpu_power.py run (coalesced groups for SECS seconds: kernel times by HIP events, package power and sclk by rocm-smi), ROUNDS times
# in turn.  usage: tools/ab_variants.sh <out.log> <name> [<name> ...]    (name "product" = ccsmeth_amd/lib/libccsm.so; "name@ENV=1" sets ENV)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
out=$1; shift
: > $out
for round in $(seq 1 ${ROUNDS:-2}); do
  for v in "$@"; do
    name=${v%%@*}; envs=""
    [ "$v" != "$name" ] && envs=${v#*@}
    lib=$R/ccsmeth_amd/lib/variants/libccsm_$name.so
    [ "$name" = product ] && lib=$R/ccsmeth_amd/lib/libccsm.so
    echo "== $v (round $round)" >> $out
    env CCSM_LIB_PATH=$lib $envs MODES=${MODES:-normal} PREC=${PREC:-4} timeout 120 python $R/tools/gpu_power.py ${SECS:-3} 2>&1 | grep -v amdgpu.ids >> $out
  done
done
cat $out
