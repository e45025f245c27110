cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "textured or preview or config4 or config5 or config3" --durations=8 > gpurun_out/r02c_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02c_tests.log
tail -n 16 gpurun_out/r02c_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r02c_bench.json 2> gpurun_out/r02c_bench.err; echo "bench rc=$?"
PT_BENCH_DEVICE=0 PT_BENCH_CHECK=1 timeout -k 10 300 python bench.py --gpus 2 --backend gloo --spp 16 --cpu-seconds 0 --steps 2 > gpurun_out/r02c_bench_spawn2.json 2> gpurun_out/r02c_bench_spawn2.err; echo "spawn rc=$?"
tail -n 3 gpurun_out/r02c_bench_spawn2.err
