cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 2400 python3 -m pytest tests/ -q -m gpu --durations=8 > $O/r06c_gpu_tests.txt 2>&1
tail -14 $O/r06c_gpu_tests.txt
timeout 900 python3 bench.py > $O/r06c_bench_bf16_b32.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/r06c_bench_bf16_b32.json'))
print('s4', d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'], 'parity', d['parity_mode']['value'], d['parity_mode']['exact_fp32']['value'], 'roof', d['roofline']['frac'])
print('attention', d['attention']['algorithmic_frac'], 'sec', d['secondary']['attention']['algorithmic_frac'], d['secondary']['attention']['forward']['ms_per_step'], d['secondary']['attention']['backward']['ms_per_step'])
print('fwd_only', {k:(v.get('value'), v.get('graph',{}).get('value')) for k,v in d['forward_only'].items()})
print('ddp', d['ddp_rehearsal']['ms_per_step'], d['ddp_rehearsal']['serial_form']['ms_per_step'])
"
