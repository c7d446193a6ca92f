#!/bin/bash
# usage: r03_hub_sweep.sh ; the native plugin measurement under different hub settings (breakdown printed by MI355X_H264_HUB_VERBOSE)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
cd $R
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
from media_amd import synth
with open("/tmp/pool.i420", "wb") as f:
    for fr in synth.sequence("s1", 1920, 1080, 200 + 64 + 1):
        f.write(np.ascontiguousarray(fr).tobytes())
PY
export RO_VMI_DEMO_VIDEO_ENCODE_FORMAT=3 RO_SYS_VMI_CLOUDPHONE=video RO_HARDWARE_WIDTH=1920 RO_HARDWARE_HEIGHT=1080 RO_HARDWARE_FPS=30
export PERSIST_VMI_VIDEO_ENCODE_BITRATE=5000000 PERSIST_VMI_VIDEO_ENCODE_GOPSIZE=30 PERSIST_VMI_VIDEO_ENCODE_PROFILE=baseline
export PERSIST_VMI_VIDEO_ENCODE_PARAM_ADJUSTING=0 PERSIST_VMI_VIDEO_ENCODE_KEYFRAME=0 PERSIST_VMI_VIDEO_ENCODE_SCENEDETECT=1 PERSIST_VMI_VIDEO_ENCODE_DEVICE=0 MEDIA_LOG_QUIET=1
export GPU_MAX_HW_QUEUES=32 MI355X_H264_HUB_VERBOSE=1
run() { echo "== $*"; env "$@" timeout -k 10 300 $R/media_amd/lib/plugin_bench /tmp/pool.i420 1920 1080 265 200 ${STREAMS:-16,64} 2>&1 | grep -E "fps_aggregate|hub " | cut -c1-330; }
export STREAMS=2,4,8,16
run X=1
run MI355X_H264_HUB_WINDOW_US=0
run MI355X_H264_HUB_WINDOW_US=0 MI355X_H264_HUB_CTX=3
run MI355X_H264_HUB_WINDOW_US=0 MI355X_H264_HUB_CTX=4
run MI355X_H264_HUB=0
STREAMS=32,64 run MI355X_H264_HUB_WINDOW_US=0
STREAMS=32,64 run X=1
