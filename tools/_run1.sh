cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
for mb in 128 256 512 1024; do for mf in 512 1024 2048; do
  [ $mb != 256 ] && [ $mf != 1024 ] && continue
  echo "== BWD_MINBLK=$mb FWD_MINBLK=$mf"
  DAHITRA_DEC_BWD_MINBLK=$mb DAHITRA_DEC_FWD_MINBLK=$mf timeout 300 python3 tools/dec_stack_bench.py
done; done > gpurun_out/r06a/dec_stack_baseline.txt 2>&1
tail -60 gpurun_out/r06a/dec_stack_baseline.txt
