# usage: bash tools/ab_bench.sh "ENV1=.. " "ENV2=.."   -- runs bench.py under each environment prefix, interleaved twice
for rep in 1 2; do
for e in "$@"; do
  env $e python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$e', d['value'], d['step_ms']['median'], 'conv', r['all_mfma_conv_ms_per_step'], 'wgrad', r['all_wgrad_ms_per_step'], 'bn', {k:v['ms_per_step'] for k,v in d['hbm']['classes'].items()})"
done; done
