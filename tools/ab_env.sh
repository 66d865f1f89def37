# usage (GPU box): bash tools/ab_env.sh <net> "ENV1=.." "ENV2=.." ...   -- bench.py --net <net> under each environment prefix
# ("-" = none), interleaved twice: same-box A/B of engine switches
net=$1; shift
for rep in 1 2; do for e in "$@"; do
  pre=""; [ "$e" != "-" ] && pre="env $e"
  $pre timeout 300 python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-mode --no-class-replay --net $net 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e $net', d['value'], d['ms_per_step'])"
done; done
