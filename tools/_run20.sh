cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 900 python3 bench.py > $O/r06c_bench_bf16_b32.json 2> $O/bench.err
python3 -c "
import json; d=json.load(open('$O/r06c_bench_bf16_b32.json'))
print('s4', d['value'], d['ms_per_step'], 'secondary', d['secondary']['value'], 'parity', d['parity_mode']['value'], d['parity_mode']['exact_fp32']['value'], 'roof', d['roofline']['frac'], d['roofline']['achieved_minus_event_overhead'])
print('attention', d['attention']['algorithmic_frac'], d['attention']['forward']['ms_per_step'], d['attention']['backward']['ms_per_step'], d['attention']['mfma_busy'])
print('sec attention', d['secondary']['attention']['algorithmic_frac'], d['secondary']['attention']['forward'], d['secondary']['attention']['backward'], d['secondary']['attention']['mfma_busy'])
print('fwd_only', {k:v.get('value') for k,v in d['forward_only'].items()})
print('ddp', d['ddp_rehearsal']['ms_per_step'], d['ddp_rehearsal']['serial_form'])
print('hbm', {k:(v['frac'], v['ms_per_step']) for k,v in d['hbm']['classes'].items()})
print('sec roofline', d['secondary']['roofline']['frac'], d['secondary']['roofline'].get('step_traffic'))
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
A="--steps 20 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --no-secondary --no-ddp-rehearsal --no-roofline"
timeout 600 python3 bench.py --net base_transformer_pos_s4_dd8_o5 --img 512 --batch 8 $A > $O/r06c_bench_o5_512.json 2>/dev/null
timeout 600 python3 bench.py --net xbd_unet_transformer --img 1024 --batch 4 $A > $O/r06c_bench_xbd_b4.json 2>/dev/null
DAHITRA_ATTN_FP8=1 timeout 900 python3 bench.py --net base_transformer_pos_s4_resnet50 --img 1024 --batch 8 $A > $O/r06c_bench_r50_1024.json 2>/dev/null
for f in o5_512 xbd_b4 r50_1024; do python3 -c "
import json; d=json.load(open('$O/r06c_bench_$f.json')); print('$f', d['value'], d['ms_per_step'])"; done
