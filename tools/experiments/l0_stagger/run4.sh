cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04_y
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r04_y/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04_y/pytest_gpu.log
tail -4 gpurun_out/r04_y/pytest_gpu.log
for r in 1 2; do for v in lock stag; do
  if [ $v = lock ]; then export CCSM_L0_LOCKSTEP=1; else unset CCSM_L0_LOCKSTEP; fi
  timeout 300 python bench.py --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms']
print('round $r %-6s value %.4g sites/s trained %.4g frac %.4f  gru0 %.4f gru1 %.4f gru2 %.4f attn %.4f' % ('$v', d['value'], d.get('value_trained_checkpoint',0), d['roofline']['frac'], k['gru0'], k['gru1'], k['gru2'], k['attn_fc']))" >> gpurun_out/r04_y/ab.log 2>&1
done; done
unset CCSM_L0_LOCKSTEP
cat gpurun_out/r04_y/ab.log
