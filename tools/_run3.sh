cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r02_a_bench.json 2> gpurun_out/r02_a_bench.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_a -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-counters > gpurun_out/prof_r02_a.log 2>&1; echo "prof rc=$?"
find gpurun_out/prof_r02_a -name "*kernel_stats.csv" | head
timeout -k 10 300 python bench.py --scene-flags 4 --cpu-seconds 0 > gpurun_out/r02_a_bench_closed.json 2> gpurun_out/r02_a_bench_closed.err; echo "closed rc=$?"
timeout -k 10 300 python bench.py --opt-flags 4 --cpu-seconds 0 > gpurun_out/r02_a_bench_kdonly.json 2> gpurun_out/r02_a_bench_kdonly.err; echo "kdonly rc=$?"
