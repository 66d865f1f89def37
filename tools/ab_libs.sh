# usage (GPU box): bash tools/ab_libs.sh A V1 V2 ...   -- bench.py (newUNetTrans and the default net) under the product library
# (A) and under build/exp/lib_<V>.so variants (DAHITRA_HIP_LIB), interleaved twice: same-box A/B of compile-flag / kernel variants
for rep in 1 2; do for v in "$@"; do
  L=$PWD/dahitra_amd/lib/libdahitra_hip.so; [ $v != A ] && L=$PWD/build/exp/lib_$v.so
  for net in newUNetTrans base_transformer_pos_s4; do
    DAHITRA_HIP_LIB=$L timeout 200 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --net $net 2>/dev/null |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $net', d['value'], d['ms_per_step'])"
  done; done; done
