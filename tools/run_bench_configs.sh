mkdir -p gpurun_out/b3; rm -f gpurun_out/b3/*
python bench.py --steps 30 --warmup 5 > gpurun_out/b3/s4.json 2> gpurun_out/b3/s4.err
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --net newUNetTrans > gpurun_out/b3/unet.json 2> gpurun_out/b3/unet.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fwd-only > gpurun_out/b3/s4_fwd.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fwd-only --net newUNetTrans > gpurun_out/b3/unet_fwd.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype fp32 > gpurun_out/b3/s4_fp32.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --net base_transformer_pos_s4_resnet50 > gpurun_out/b3/r50_256.json 2>/dev/null
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --net base_transformer_pos_s4_resnet50 --img 1024 --batch 8 > gpurun_out/b3/r50_1024.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --net xbd_unet_transformer --img 1024 --batch 1 > gpurun_out/b3/xbd_b1.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --net xbd_unet_transformer --img 1024 --batch 4 > gpurun_out/b3/xbd_b4.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --net base_transformer_pos_s4_dd8_o5 --img 512 --batch 8 > gpurun_out/b3/o5_512.json 2>/dev/null
for f in gpurun_out/b3/*.json; do echo $f; python -c "
import json,sys
d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['config'].get('step_tflops'), d['roofline'] and (d['roofline']['kernel'], d['roofline']['achieved'], d['roofline']['frac']), d.get('cpu_baseline',{}).get('value'))"; done
bash tools/prof_bench.sh s4f > /dev/null
bash tools/prof_bench.sh unetf --net newUNetTrans > /dev/null
