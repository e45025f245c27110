cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -m gpu -x -q -k "sharded or assemble or batches or rccl or prep or config3 or golden_hash or several_devices" > gpurun_out/r02q_tests.log 2>&1; echo "tests rc=$?"
tail -n 5 gpurun_out/r02q_tests.log
timeout -k 10 400 python tools/_shard2.py 2>/dev/null | tail -n 9
PT_BENCH_DEVICE=0 PT_BENCH_CHECK=1 timeout -k 10 300 python bench.py --gpus 3 --backend gloo --spp 8 --cpu-seconds 0 --steps 1 > gpurun_out/r02q_spawn3.json 2> gpurun_out/r02q_spawn3.err; echo "spawn rc=$?"; grep "assembled" gpurun_out/r02q_spawn3.err
