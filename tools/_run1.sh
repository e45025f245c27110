cd $GRAFT_REPO_ROOT
PT_TILE_ORDER=morton timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden_hash or sharded or grid_kd or batches" > gpurun_out/r02l_tests.log 2>&1; echo "morton tests rc=$?"
for m in rows morton rows morton; do
  PT_TILE_ORDER=$m PT_WF_OVERLAP=0 timeout -k 10 300 python tools/stage_times.py --spp 128 --reps 3 > gpurun_out/r02l_stage_$m.log 2>&1
  grep launches gpurun_out/r02l_stage_$m.log
done
