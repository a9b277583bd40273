cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/l0s
export CCSM_LIB_PATH=$GRAFT_REPO_ROOT/ccsmeth_amd/lib/variants/libccsm_stamps.so
for z in 1 0; do for v in lock stag; do
  if [ $v = lock ]; then export CCSM_L0_LOCKSTEP=1; else unset CCSM_L0_LOCKSTEP; fi
  ZERO=$z timeout 200 python tools/experiments/l0_stagger/phases.py >> gpurun_out/l0s/phases.log 2>&1
done; done
cat gpurun_out/l0s/phases.log
