cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
DAHITRA_TEST_VERBOSE=1 timeout 2400 python3 -m pytest tests/ -q -m gpu -s -k "gradients_match_oracle or (config1 and fp32) or xbd_step_at_1024 or train_steps_match_reference_golden or benchmarked_size_fp32" > $O/calib2.txt 2>&1
tail -3 $O/calib2.txt
timeout 900 python3 bench.py > $O/bench_default3.json 2> $O/bench_default3.err; tail -3 $O/bench_default3.err
python3 -c "
import json; d=json.load(open('$O/bench_default3.json'))
print(d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'])
print('attention', d['attention']['forward'], d['attention']['backward'], d['attention']['algorithmic_frac'])
print('ddp', json.dumps(d['ddp_rehearsal'])[:900])
"
