cd $GRAFT_REPO_ROOT
bash tools/pmc_profile.sh pmc_r02_a --steps 1 --warmup 0 > gpurun_out/pmc_r02_a.log 2>&1
tail -n 8 gpurun_out/pmc_r02_a.log
