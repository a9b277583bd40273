cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_y_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r04_y_smoke.log
bash tools/profile_round.sh r04_y > gpurun_out/r04_y_profile_round.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 gpurun_out/r04_y_smoke.log; tail -30 gpurun_out/r04_y_profile_round.log
