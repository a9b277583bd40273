#!/bin/bash
# Round profile on the GPU box: bench.py plain and under rocprofv3 --kernel-trace --stats, then five --pmc passes (HBM fetch / write, SQ, L2, LDS) over the
# launch shape bench.py times (tools/gpu_group.py with COALESCE=6: 512 workgroups).  usage: tools/profile_round.sh <tag>   (outputs under gpurun_out/<tag>/)
set -u
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# the driver's command; under rocprofv3 without the secondary measurements (their launches have other shapes and would mix into the
# per-kernel averages) and without the CPU leg
python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_plain.json 2> $O/bench_plain.err
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --extras none --cpu-seconds 0 --ceiling-seconds 0 > $O/bench_under_rocprof.json 2> $O/rocprof.err
cp $(find /tmp/prof_stats -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/kernel_trace_by_shape.py $(find /tmp/prof_stats -name "*kernel_trace.csv" | head -1) $O/bench_kernel_by_shape.md 3 > /dev/null
for pass in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "tcc:TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "lds:SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  PREC=${PREC:-4} REPS=6 COALESCE=6 rocprofv3 --kernel-trace --output-format csv --pmc $ctrs -d /tmp/pmc/$name -- python $R/tools/gpu_group.py > /dev/null 2>> $O/pmc.err
done
python $R/tools/pmc_summary.py /tmp/pmc $O/pmc.md "$TAG PMC - coalesced launch (6 x 2048 sites = 512 workgroups: the shape bench.py times), precision ${PREC:-4}" --emit $O/traffic.json --precision ${PREC:-4} --sites 12288 --source "profiles/${TAG}_pmc.md"
tail -1 $O/bench_plain.json
head -8 $O/bench_kernel_stats.csv
cat $O/bench_kernel_by_shape.md
