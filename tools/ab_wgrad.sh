echo "== CIG=2 (new)"; python tools/kbench.py --only wgrad --reps 30 2>&1 | grep -v amdgpu
echo "== CIG=1 (old)"; DAHITRA_WGRAD_CIG1=1 python tools/kbench.py --only wgrad --reps 30 2>&1 | grep -v amdgpu
python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "wgrad" 2>&1 | tail -3
