cd $GRAFT_REPO_ROOT
T=r06c
bash tools/profile_round.sh $T > gpurun_out/${T}_s4.log 2>&1
bash tools/profile_round.sh $T newUNetTrans > gpurun_out/${T}_unet.log 2>&1
cd $GRAFT_REPO_ROOT
DAHITRA_HIP_LIB=build/exp/lib_wreg_timing.so timeout 300 python3 tools/wreg_timeline.py > gpurun_out/$T/${T}_wreg_timeline.txt 2>&1
timeout 300 python3 tools/dec_stack_bench.py > gpurun_out/$T/${T}_dec_stack_bench.txt 2>&1
ls gpurun_out/$T | head -40; tail -3 gpurun_out/${T}_s4.log gpurun_out/${T}_unet.log
