cd $GRAFT_REPO_ROOT
O=gpurun_out/r06b; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_parallel_gpu.py tests/test_kernels_gpu.py -q -m gpu -k "parallel or two_ranks or overlapped or decoder or stack" 2>&1 | tail -5
echo "== previous build (r05 lib)"; DAHITRA_HIP_LIB=build/exp/lib_r05.so timeout 300 python3 tools/dec_stack_bench.py --save /tmp/ref.pt --only-multi
echo "== new"; timeout 300 python3 tools/dec_stack_bench.py --check /tmp/ref.pt
for rep in 1 2; do
for lib in build/exp/lib_r05.so dahitra_amd/lib/libdahitra_hip.so; do
DAHITRA_HIP_LIB=$lib timeout 600 python3 bench.py --net newUNetTrans --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib newUNetTrans', d['value'], d['ms_per_step'])"
done; done
