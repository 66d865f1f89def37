cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "upsample or up4 or upsampled" 2>&1 | tail -8
timeout 300 python3 tools/up4_bench.py 2>&1 | grep -v amdgpu.ids
DAHITRA_UP4_TAP=1 timeout 300 python3 tools/up4_bench.py 2>&1 | grep -v amdgpu.ids
