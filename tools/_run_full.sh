cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r02b_tests_full.log 2>&1; echo "tests rc=$?" >> gpurun_out/r02b_tests_full.log
tail -n 5 gpurun_out/r02b_tests_full.log
