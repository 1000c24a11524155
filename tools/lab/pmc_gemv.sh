#!/bin/bash
# PMC pass over the standalone GEMV lab (rocprofv3 --pmc crashes under the full python bench on this image):
# FETCH_SIZE per dispatch for the stream kernel (known byte count -> calibration) and for every qmv3 variant.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_gemv; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o lab --output-format csv -- $R/tools/lab/gemv_lab pmc > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o lab --output-format csv -- $R/tools/lab/gemv_lab pmc > $OUT/write.log 2>&1
tail -2 $OUT/fetch.log; ls $OUT/fetch $OUT/write
