cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
{
echo "== pytest decoder + model"; timeout 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "decoder or unet or UNet or stack or dd8 or xbd" 2>&1 | tail -5
for rep in 1 2; do
echo "== rule"; DAHITRA_DEC_BALANCE_LOG=1 timeout 300 python3 tools/dec_stack_bench.py --only-multi
echo "== forced 16"; DAHITRA_DEC_BALANCE_LOG=1 DAHITRA_DEC_UPB_BWD=16 DAHITRA_DEC_UPB_FWD=8 timeout 300 python3 tools/dec_stack_bench.py --only-multi
echo "== forced 8";  DAHITRA_DEC_BALANCE_LOG=1 DAHITRA_DEC_UPB_BWD=8 DAHITRA_DEC_UPB_FWD=4 timeout 300 python3 tools/dec_stack_bench.py --only-multi
echo "== default";  DAHITRA_DEC_BALANCE=0 timeout 300 python3 tools/dec_stack_bench.py --only-multi
done
} > $O/dec_stack_new4.txt 2>&1
cat $O/dec_stack_new4.txt | grep -v amdgpu.ids
