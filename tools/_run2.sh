cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=6 > gpurun_out/r02d_tests_full.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02d_tests_full.log
tail -n 14 gpurun_out/r02d_tests_full.log
