cd $GRAFT_REPO_ROOT
for v in base fat; do
  lib=build/variants/libptgpu_$v.so; [ $v = base ] && lib=path-tracer_amd/libptgpu.so
  PT_GPU_LIB=$lib timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_hash or translucent_generated or grid_kd or long_normals" > gpurun_out/r02k_tests_$v.log 2>&1; echo "$v tests rc=$?"
  PT_GPU_LIB=$lib PT_WF_OVERLAP=0 PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 3 > gpurun_out/r02k_stage_$v.log 2>&1
  PT_GPU_LIB=$lib timeout -k 10 300 python bench.py --scene-flags 4 --cpu-seconds 0 --no-counters --steps 2 > gpurun_out/r02k_closed_$v.json 2>/dev/null
done
