cd $GRAFT_REPO_ROOT
O=gpurun_out/r06a; mkdir -p $O
timeout 1500 python3 -m pytest tests/ -x -q -m gpu > $O/gpu_suite.txt 2>&1
tail -5 $O/gpu_suite.txt
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 -c "
import json; d=json.load(open('$O/bench_default.json'))
print(d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'], 'parity', d['parity_mode']['value'], 'roof', d['roofline']['frac'], 'ddp', d['ddp_rehearsal'].get('ms_per_step'))
"
