cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 2400 python3 -m pytest tests/ -q -m gpu -x -k "gradients_match_oracle or config1 or xbd_step_at_1024 or train_steps_match_reference_golden or benchmarked_size_fp32 or decoder or conv2d" > $O/calib3.txt 2>&1
tail -3 $O/calib3.txt
run() { env "$@" DAHITRA_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-secondary --no-ddp-rehearsal --no-roofline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', d['value'], d['ms_per_step'], d['config']['step_form'])"; }
for rep in 1 2; do
run DAHITRA_OVERLAP=1
run DAHITRA_OVERLAP=1 DAHITRA_OVERLAP_PERSIST_BN=1
run DAHITRA_OVERLAP=0
done
