cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config3 and not config4 and not config5 and not out_of_memory" > gpurun_out/r02o_tests.log 2>&1; echo "tests rc=$?"
tail -n 12 gpurun_out/r02o_tests.log
