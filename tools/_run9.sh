cd $GRAFT_REPO_ROOT
run() { env "$@" DAHITRA_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-secondary --no-ddp-rehearsal --no-roofline 2>/tmp/err.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['config']['step_form'])" || tail -3 /tmp/err.txt; }
for rep in 1 2; do
run DAHITRA_OVERLAP=1
run DAHITRA_OVERLAP=1 DAHITRA_OVERLAP_PERSIST_BN=1
done
timeout 600 python3 -m pytest tests/test_kernels_gpu.py -q -m gpu -x -k "bench_scale" 2>&1 | tail -2
