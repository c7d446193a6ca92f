#!/bin/bash
# usage: prof.sh <tag> ; run from anywhere on the GPU box
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$1/stats -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_$1/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_$1/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --gops-in-flight 1 > $R/gpurun_out/prof_$1/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_$1/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --gops-in-flight 1 > $R/gpurun_out/prof_$1/pmc_write.log 2>&1
ls -R $R/gpurun_out/prof_$1 | head -30
