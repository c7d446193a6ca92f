import sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from media_amd import synth, capi
from oracle_lib import OracleEncoder
w, h = 64, 48
enc = capi.Encoder(w, h, qp=51); enc.keep_pre(True)
orc = OracleEncoder(w, h, qp=51)
fr = synth.sequence("s1", w, h, 2)
enc.encode(fr[0]); orc.encode(fr[0]); enc.encode(fr[1]); orc.encode(fr[1])
mb, omb = enc.debug_read(capi.DBG_MBINFO), orc.mbinfo()
lv, olv = enc.debug_read(capi.DBG_LEVELS), orc.levels()
pre, opre = enc.debug_read(capi.DBG_PRE_Y), orc.recon_pre(0)
for i in range(4):
    print("MB", i, "mv", mb["mvx"][i], mb["mvy"][i], "cbp", mb["cbp"][i], omb["cbp"][i], "tc", mb["tc"][i][:8], omb["tc"][i][:8])
    print("  ours luma blk0", lv[i, 16:32]); print("  orc  luma blk0", olv[i, 16:32])
    x = 16 * (i % 4)
    print("  pre ours row0", pre[0, x:x+16]); print("  pre orc  row0", opre[0, x:x+16])
