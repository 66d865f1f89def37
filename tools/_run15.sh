cd $GRAFT_REPO_ROOT
export DAHITRA_HIP_LIB=build/exp/lib_up4_dbg.so
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "upsample or up4 or upsampled or wreg or resident" 2>&1 | tail -3
timeout 300 python3 tools/up4_bench.py 2>&1 | grep -v amdgpu.ids
for rep in 1 2; do
for lib in build/exp/lib_up4_dbg.so dahitra_amd/lib/libdahitra_hip.so; do
DAHITRA_HIP_LIB=$lib DAHITRA_UP4_TAP=1 timeout 600 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --no-roofline --no-secondary --no-ddp-rehearsal 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done; done
