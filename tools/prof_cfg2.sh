#!/bin/bash
# usage: prof_cfg2.sh <tag> ; rocprofv3 capture of BASELINE.json configs[2] (1080p60 NV12, main profile, CAVLC):
# kernel times + HBM bytes (FETCH_SIZE / WRITE_SIZE in separate passes) -> gpurun_out/prof_<tag>/
set -o pipefail
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--input nv12 --profile main --fps 60 --instances 1 --gops-in-flight 32 --no-cpu-baseline --no-plugin"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 3 --warmup 1 $ARGS > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 1 $ARGS > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 1 --warmup 1 $ARGS > $O/pmc_write.log 2>&1
tail -1 $O/stats.log | cut -c1-300
