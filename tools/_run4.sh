cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
{
echo "== pytest decoder + model"; timeout 1500 python3 -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -m gpu -k "decoder or unet or UNet or stack or dd8 or xbd" 2>&1 | tail -5
echo "== previous build (r05 lib)"; DAHITRA_HIP_LIB=build/exp/lib_r05.so timeout 300 python3 tools/dec_stack_bench.py --save /tmp/ref.pt --only-multi
echo "== new"; timeout 300 python3 tools/dec_stack_bench.py --check /tmp/ref.pt
echo "== bench newUNetTrans prev"; DAHITRA_HIP_LIB=build/exp/lib_r05.so timeout 600 python3 bench.py --net newUNetTrans --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --no-roofline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== bench newUNetTrans new"; timeout 600 python3 bench.py --net newUNetTrans --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --no-roofline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== bench newUNetTrans prev"; DAHITRA_HIP_LIB=build/exp/lib_r05.so timeout 600 python3 bench.py --net newUNetTrans --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --no-roofline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo "== bench newUNetTrans new"; timeout 600 python3 bench.py --net newUNetTrans --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --no-roofline | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
} > $O/dec_stack_new3.txt 2>&1
cat $O/dec_stack_new3.txt | grep -v amdgpu.ids
