# same-box A/B of two source trees: bash tools/ab_tree.sh build/prev_tree .   (interleaved, 2 rounds)
for rep in 1 2; do
for t in "$@"; do
  (cd $t && python bench.py --no-cpu-baseline --no-parity-mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$t', d['value'], d['step_ms']['median'], 'conv', r['all_mfma_conv_ms_per_step'], 'wgrad', r['all_wgrad_ms_per_step'], 'bn', {k:v['ms_per_step'] for k,v in d['hbm']['classes'].items()})")
done; done
