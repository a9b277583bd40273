#!/bin/bash
# The LDS counters of the GRU kernels at the timed launch shape (never collected before round 6's end: profiles/r06_r names them as the first
# thing to look at).  usage: tools/pmc_lds_pass.sh <out.md> [precision]   (on the GPU box; one --pmc pass with --kernel-trace only)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$1; P=${2:-4}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lds
PREC=$P REPS=6 COALESCE=6 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d /tmp/pmc_lds/lds -- python $R/tools/gpu_group.py > /dev/null 2> /tmp/pmc_lds.err
PREC=$P REPS=6 COALESCE=6 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d /tmp/pmc_lds/sq -- python $R/tools/gpu_group.py > /dev/null 2>> /tmp/pmc_lds.err
python $R/tools/pmc_summary.py /tmp/pmc_lds $OUT "LDS and wait counters - coalesced launch (6 x 2048 sites = 512 workgroups), precision $P" || tail -n 20 /tmp/pmc_lds.err
