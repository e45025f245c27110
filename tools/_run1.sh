cd $GRAFT_REPO_ROOT
for v in base mbox ldsleaf ldsleaf5; do
  lib=build/variants/libptgpu_$v.so; [ $v = base ] && lib=path-tracer_amd/libptgpu.so
  PT_GPU_LIB=$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_hash or translucent_generated or grid_kd" > gpurun_out/r02j_tests_$v.log 2>&1; echo "$v tests rc=$?"
  PT_GPU_LIB=$lib PT_WF_OVERLAP=0 PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 2 > gpurun_out/r02j_stage_$v.log 2>&1
  PT_GPU_LIB=$lib PT_WF_OVERLAP=0 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 2 --opt-flags 4 > gpurun_out/r02j_stage_${v}_kd.log 2>&1
done
