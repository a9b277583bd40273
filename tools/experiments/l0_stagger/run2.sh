# the schedule on its own (all-zero weights: no power cap) and under the cap, lock-step against staggered
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/l0s
for v in lock stag lock stag; do
  if [ $v = lock ]; then export CCSM_L0_LOCKSTEP=1; else unset CCSM_L0_LOCKSTEP; fi
  echo "== $v" >> gpurun_out/l0s/power.log
  MODES=zero_weights,normal timeout 300 python tools/gpu_power.py 3 >> gpurun_out/l0s/power.log 2>&1
done
cat gpurun_out/l0s/power.log
