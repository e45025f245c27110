cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grid or golden_hash or config2 or translucent_generated or sharded or batches or out_of_memory" > gpurun_out/r02g_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02g_tests.log
tail -n 3 gpurun_out/r02g_tests.log
for v in base c256 c128; do
  lib=build/variants/libptgpu_$v.so; [ $v = base ] && lib=path-tracer_amd/libptgpu.so
  PT_GPU_LIB=$lib timeout -k 10 300 python tools/shard_times.py > gpurun_out/r02g_shards_$v.log 2>&1
done
PT_WF_OVERLAP=0 PT_DEBUG_TIMES=1 timeout -k 10 300 python tools/shard_launches.py --shards 8 > gpurun_out/r02g_shard8.log 2>&1
