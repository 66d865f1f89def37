cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
DAHITRA_TEST_VERBOSE=1 timeout 2400 python3 -m pytest tests/ -q -m gpu -s -k "train_steps_match_reference_golden or benchmarked_size_fp32 or config1 or resnet50_trunk_at_1024 or statistics_buffer or gradients_match_oracle or independent_stacks or sixteen_row" > $O/calib.txt 2>&1
tail -5 $O/calib.txt
grep -c VERBOSE $O/calib.txt
timeout 900 python3 bench.py > $O/bench_default2.json 2> $O/bench_default2.err; tail -3 $O/bench_default2.err
python3 -c "
import json; d=json.load(open('$O/bench_default2.json'))
print(d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'])
print('attention', json.dumps(d['attention'])[:1500])
print('sec attention', json.dumps(d['secondary'].get('attention'))[:1500])
print('sec roofline', json.dumps(d['secondary'].get('roofline'))[:600])
print('fwd only', d['forward_only'])
"
