#!/bin/bash
# Lab: where does the ~1-2 us that rocprofv3 brackets around a decode kernel beyond its in-kernel stamps sit -- before the first
# workgroup starts or after the last wave ends?  One profiled decode step (in-kernel wall-clock stamps, TL_PROFILE_DUMP) inside a
# rocprofv3 kernel trace of the same process; tools/lab/launch_overhead_join.py lays the two side by side.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/launch_overhead; mkdir -p $OUT; rm -f $OUT/stamps.txt
cd /tmp && export TMPDIR=/tmp
TL_PROFILE_DUMP=$OUT/stamps.txt rocprofv3 --kernel-trace -d $OUT/trace -o probe --output-format csv -- python $R/tools/decode_ab.py --batch 1 --prompt-len 128 --steps 8 --warmup 4 --profile-steps 3 - > $OUT/run.log 2>&1
tail -2 $OUT/run.log; ls $OUT/trace | head; wc -l $OUT/stamps.txt
python $R/tools/lab/launch_overhead_join.py $OUT/stamps.txt $OUT/trace/*kernel_trace.csv | tee $OUT/joined.txt | tail -30
