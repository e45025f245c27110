cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grid or golden_hash or config2 or translucent_generated or sharded or batches or textured or edge_case or counters" > gpurun_out/r02h_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02h_tests.log
tail -n 3 gpurun_out/r02h_tests.log
for v in base noagg; do
  lib=build/variants/libptgpu_$v.so; [ $v = base ] && lib=path-tracer_amd/libptgpu.so
  PT_GPU_LIB=$lib PT_WF_OVERLAP=0 PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 2 > gpurun_out/r02h_stage_$v.log 2>&1
  PT_GPU_LIB=$lib timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 3 > gpurun_out/r02h_stage_${v}_ov.log 2>&1
done
