cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
{
echo "== pytest decoder"; timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "decoder" 2>&1 | tail -5
echo "== previous build (r05 lib)"; DAHITRA_HIP_LIB=build/exp/lib_r05.so timeout 300 python3 tools/dec_stack_bench.py --save /tmp/ref.pt --only-multi
echo "== new, default plans (DAHITRA_DEC_BALANCE=0)"; DAHITRA_DEC_BALANCE=0 timeout 300 python3 tools/dec_stack_bench.py --check /tmp/ref.pt
echo "== new, balanced"; DAHITRA_DEC_BALANCE_LOG=1 timeout 300 python3 tools/dec_stack_bench.py --check /tmp/ref.pt --only-multi
for u in 1 2 3 4 6 8 12 16; do echo "== forced upb $u"; DAHITRA_DEC_UPB_FWD=$u DAHITRA_DEC_UPB_BWD=$u timeout 300 python3 tools/dec_stack_bench.py; done
} > $O/dec_stack_new1.txt 2>&1
cat $O/dec_stack_new1.txt | grep -v amdgpu.ids
