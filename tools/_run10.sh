cd $GRAFT_REPO_ROOT
export DAHITRA_FORCE_DIST=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29613
DAHITRA_OVERLAP=1 bash tools/trace_last_step.sh r06a_ddp_overlap --no-secondary --no-ddp-rehearsal --no-roofline
DAHITRA_OVERLAP=0 bash tools/trace_last_step.sh r06a_ddp_serial --no-secondary --no-ddp-rehearsal --no-roofline
tail -2 gpurun_out/r06a_ddp_overlap_last_step.txt gpurun_out/r06a_ddp_serial_last_step.txt
awk '$3 > 3.0' gpurun_out/r06a_ddp_overlap_last_step.txt | head -30
