#!/bin/bash
# usage: r03_hub_trace.sh <S> ; kernel trace of the native plugin measurement at S streams (short run)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
from media_amd import synth
with open("/tmp/pool.i420", "wb") as f:
    for fr in synth.sequence("s1", 1920, 1080, 100 + 64 + 1):
        f.write(np.ascontiguousarray(fr).tobytes())
PY
export RO_VMI_DEMO_VIDEO_ENCODE_FORMAT=3 RO_SYS_VMI_CLOUDPHONE=video RO_HARDWARE_WIDTH=1920 RO_HARDWARE_HEIGHT=1080 RO_HARDWARE_FPS=30
export PERSIST_VMI_VIDEO_ENCODE_BITRATE=5000000 PERSIST_VMI_VIDEO_ENCODE_GOPSIZE=30 PERSIST_VMI_VIDEO_ENCODE_PROFILE=baseline
export PERSIST_VMI_VIDEO_ENCODE_PARAM_ADJUSTING=0 PERSIST_VMI_VIDEO_ENCODE_KEYFRAME=0 PERSIST_VMI_VIDEO_ENCODE_SCENEDETECT=1 PERSIST_VMI_VIDEO_ENCODE_DEVICE=0 MEDIA_LOG_QUIET=1
export GPU_MAX_HW_QUEUES=32 MI355X_H264_HUB_VERBOSE=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/hubtrace_$1 -- $R/media_amd/lib/plugin_bench /tmp/pool.i420 1920 1080 165 100 $1 > $O/hubtrace_$1.log 2>&1
grep -E "fps_aggregate|hub " $O/hubtrace_$1.log | cut -c1-300
find $O/hubtrace_$1 -name "*.csv" | head; du -sh $O/hubtrace_$1
