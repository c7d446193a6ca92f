#!/bin/bash
# usage: prof_fetch.sh <tag> ; FETCH_SIZE / WRITE_SIZE calibration passes of tools/ubench_fetch.bin on the GPU box
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/fetch_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench_fetch.bin > $O/run.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $R/tools/ubench_fetch.bin > $O/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $R/tools/ubench_fetch.bin > $O/pmc_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $R/tools/ubench_fetch.bin > $O/pmc_write.log 2>&1
cat $O/run.log
