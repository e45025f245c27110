cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grid or long_normals or golden_hash or config2 or translucent_generated or counters_match or sharded or batches or edge_case" > gpurun_out/r02a_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02a_tests.log
tail -n 3 gpurun_out/r02a_tests.log
PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 2 > gpurun_out/r02a_stage_fused.log 2>&1
PT_OG_FUSE_RNG=0 PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 2 > gpurun_out/r02a_stage_nofuse.log 2>&1
PT_WF_OVERLAP=0 PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 2 > gpurun_out/r02a_stage_serial.log 2>&1
