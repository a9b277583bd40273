#!/bin/bash
# power attribution: every variant library through tools/gpu_power.py (normal data), 3 s each
R=${GRAFT_REPO_ROOT:-$PWD}
for v in "$@"; do
  if [ "$v" = base ]; then lib=$R/ccsmeth_amd/lib/libccsm.so; else lib=$R/ccsmeth_amd/lib/variants/$v; fi
  echo "== $v"
  CCSM_LIB_PATH=$lib MODES=normal python $R/tools/gpu_power.py 3 2>&1 | grep -v amdgpu.ids
done
