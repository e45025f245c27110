cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py > gpurun_out/r02_b_bench.json 2> gpurun_out/r02_b_bench.err; echo "bench rc=$?"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02_b -- python3 bench.py --steps 2 --warmup 1 --cpu-seconds 0 --no-counters > gpurun_out/prof_r02_b.log 2>&1; echo "prof rc=$?"
timeout -k 10 300 python bench.py --scene-flags 4 --cpu-seconds 0 > gpurun_out/r02_b_bench_closed.json 2> gpurun_out/r02_b_bench_closed.err; echo "closed rc=$?"
timeout -k 10 300 python bench.py --spp 512 --bounces 8 --cpu-seconds 0 --steps 2 > gpurun_out/r02_b_bench_cfg4.json 2> gpurun_out/r02_b_bench_cfg4.err; echo "cfg4 rc=$?"
timeout -k 10 600 python bench.py --tris 4000000 --width 3840 --height 2160 --spp 256 --bounces 8 --tonemap ACES --scene-flags 1 --cpu-seconds 0 --steps 1 --warmup 1 > gpurun_out/r02_b_bench_cfg5.json 2> gpurun_out/r02_b_bench_cfg5.err; echo "cfg5 rc=$?"
timeout -k 10 300 python tools/host_buffer_rate.py > gpurun_out/r02_b_host_buffer.log 2>&1; echo "hostbuf rc=$?"
