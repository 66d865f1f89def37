cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 2400 python3 -m pytest tests/ -q -m gpu --durations=25 > $O/gpu_suite.txt 2>&1
tail -40 $O/gpu_suite.txt
