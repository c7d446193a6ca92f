import os, sys
os.environ["MI355X_H264_DBG_PRED"] = "1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from media_amd import synth, capi
from oracle_lib import OracleEncoder, lib, _ptr
w, h = 64, 48
enc = capi.Encoder(w, h, qp=26); enc.keep_pre(True)
orc = OracleEncoder(w, h, qp=26)
fr = synth.sequence("s1", w, h, 2)
enc.encode(fr[0]); orc.encode(fr[0])
ref = orc.recon(0).copy()
enc.encode(fr[1])
mb = enc.debug_read(capi.DBG_MBINFO)
pre = enc.debug_read(capi.DBG_PRE_Y)
for i in range(1, 3):
    mvx, mvy = int(mb["mvx"][i]), int(mb["mvy"][i])
    x, y = 16 * (i % 4), 16 * (i // 4)
    exp = np.zeros((16, 16), np.uint8)
    lib().h264o_mc_luma(_ptr(ref), 64, 64, 48, x, y, mvx, mvy, 16, 16, _ptr(exp), 16)
    got = pre[y:y+16, x:x+16]
    print("MB", i, "mv", mvx, mvy, "fx,fy", mvx & 3, mvy & 3, "equal", np.array_equal(exp, got))
    for r in (0, 1, 15):
        print("  exp", exp[r]); print("  got", got[r])
