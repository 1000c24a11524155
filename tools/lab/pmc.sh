cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lab1 -o lab --output-format csv -- $GRAFT_REPO_ROOT/tools/lab/abl_lab > $GRAFT_REPO_ROOT/gpurun_out/pmc_lab1.log 2>&1
tail -3 $GRAFT_REPO_ROOT/gpurun_out/pmc_lab1.log; ls $GRAFT_REPO_ROOT/gpurun_out/pmc_lab1
