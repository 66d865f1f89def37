cd $GRAFT_REPO_ROOT
bash tools/trace_last_step.sh r06b_unet --net newUNetTrans --no-secondary --no-ddp-rehearsal --no-roofline > /dev/null 2>&1
tail -n 1 gpurun_out/r06b_unet_last_step.txt
