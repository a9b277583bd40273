set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/l0s
timeout 600 python tools/experiments/l0_stagger/check.py > gpurun_out/l0s/check.log 2>&1; echo "check rc $?" >> gpurun_out/l0s/check.log
for r in 1 2; do
 for v in lock stag; do
  if [ $v = lock ]; then export CCSM_L0_LOCKSTEP=1; else unset CCSM_L0_LOCKSTEP; fi
  timeout 300 python bench.py --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('round $r %-6s value %.4g sites/s trained %.4g frac %.4f  gru0 %.4f gru1 %.4f gru2 %.4f attn %.4f' % ('$v', d['value'], d.get('value_trained_checkpoint',0), d['roofline']['frac'], k['gru0'], k['gru1'], k['gru2'], k['attn_fc']))" >> gpurun_out/l0s/ab.log 2>&1
 done
done
cat gpurun_out/l0s/check.log gpurun_out/l0s/ab.log
