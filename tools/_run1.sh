cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "grid or golden_hash or config2 or translucent_generated or textured or sharded or batches or counters or edge_case or out_of_memory" > gpurun_out/r02m_tests.log 2>&1; echo "tests rc=$?"
tail -n 3 gpurun_out/r02m_tests.log
PT_WF_OVERLAP=0 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 3 2>/dev/null | grep launches
for f in 1 0; do
PT_OG_FUSE_RNG=$f timeout -k 10 300 python tools/stage_times.py --tris 1000000 --width 3840 --height 2160 --spp 32 --bounces 8 --flags 1 --reps 2 2>/dev/null | tail -n 2
done
