import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from media_amd import capi, synth
from oracle_lib import OracleEncoder
w, h = 320, 176
for prof in (66,):
    for qp in (34,):
        enc = capi.Encoder(w, h, qp=qp, gop=3, profile_idc=prof); enc.keep_pre(True)
        orc = OracleEncoder(w, h, qp=qp, gop=3, profile_idc=prof)
        for i, f in enumerate(synth.sequence("s3", w, h, 3)):
            a = enc.encode(f)[0]; b = orc.encode(f)[0]
            mb, omb = enc.debug_read(capi.DBG_MBINFO), orc.mbinfo()
            bad = {k: int((mb[k] != omb[k]).sum()) for k in ("mvx", "mvy", "type", "i16_mode", "chroma_mode", "cbp")}
            tcbad = int((mb["tc"] != omb["tc"]).any(axis=1).sum())
            first = int(np.argmax((mb["type"] != omb["type"]) | (mb["cbp"] != omb["cbp"]) | (mb["tc"] != omb["tc"]).any(axis=1))) if (tcbad or bad["type"] or bad["cbp"]) else -1
            aux_bad = int((enc.debug_read(capi.DBG_MBAUX)[omb["type"] == 4] != orc.mbaux()[omb["type"] == 4]).any(axis=1).sum())
            pre = [bool(np.array_equal(enc.debug_read(capi.DBG_PRE_Y + p), orc.recon_pre(p))) for p in range(3)]
            print("prof", prof, "qp", qp, "pic", i, "same" if a == b else "DIFF", len(a), len(b), bad, "tc", tcbad, "first", first, "aux", aux_bad, "pre", pre,
                  "types gpu", np.bincount(mb["type"], minlength=5), "orc", np.bincount(omb["type"], minlength=5), flush=True)
            if first >= 0:
                print("  gpu", mb[first], "\n  orc", omb[first])
            glv, olv = enc.debug_read(capi.DBG_LEVELS), orc.levels()
            if i == 0:
                print("MB1 gpu", glv[1].tolist()); print("MB1 orc", olv[1].tolist())
            both = (mb["type"] == omb["type"]) & (omb["type"] != 3)
            dm = np.where(both & (mb["tc"] != omb["tc"]).any(axis=1))[0]
            for k in dm[:2]:
                j = np.where(mb["tc"][k] != omb["tc"][k])[0][0]
                off = 16 + j * 16 if j < 16 else 280 + (j - 16) * 16
                print("   mb", k, "type", omb["type"][k], "tc idx", j, "gpu lv", glv[k, off:off + 16], "orc lv", olv[k, off:off + 16], "dc gpu", glv[k, 272:280], "orc", olv[k, 272:280])
        enc.close()
