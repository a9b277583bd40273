#!/bin/bash
# A/B runs on the GPU box: variant libraries built here (tools/ab_build.sh) are run alternately through bench.py.
# usage: tools/ab.sh <rounds> <variant.so> [<variant.so> ...]     ("base" = the in-tree libccsm.so)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
ROUNDS=$1; shift
for r in $(seq 1 $ROUNDS); do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=$R/ccsmeth_amd/lib/libccsm.so; else lib=$R/ccsmeth_amd/lib/variants/$v; fi
    CCSM_LIB_PATH=$lib python $R/bench.py --cpu-seconds 0 ${AB_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('round $r %-28s value %.4g sites/s  frac %.4f  gru0 %.4f gru1 %.4f gru2 %.4f attn %.4f' % ('$v', d['value'], d['roofline']['frac'], k['gru0'], k['gru1'], k['gru2'], k['attn_fc']))"
  done
done
