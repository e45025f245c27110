cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
PT_WF_OVERLAP=0 PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 1 --counters > gpurun_out/r02a_stage_grid.log 2>&1
PT_WF_OVERLAP=0 PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 1 --opt-flags 4 --counters > gpurun_out/r02a_stage_kd.log 2>&1
