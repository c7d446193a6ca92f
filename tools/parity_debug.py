import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from media_amd import synth, capi
from oracle_lib import OracleEncoder, OracleDecoder

def run(w, h, n, kind="s1", qp=26, gop=30):
    enc = capi.Encoder(w, h, qp=qp, gop=gop); enc.keep_pre(True)
    orc = OracleEncoder(w, h, qp=qp, gop=gop)
    ok_all = True
    for i, f in enumerate(synth.sequence(kind, w, h, n)):
        t0 = time.time(); bs, ft = enc.encode(f); dt = time.time() - t0
        obs, idr = orc.encode(f)
        same = bs == obs
        mb, omb = enc.debug_read(capi.DBG_MBINFO), orc.mbinfo()
        lv, olv = enc.debug_read(capi.DBG_LEVELS), orc.levels()
        mv_ok = np.array_equal(mb["mvx"], omb["mvx"]) and np.array_equal(mb["mvy"], omb["mvy"])
        ty_ok = np.array_equal(mb["type"], omb["type"])
        md_ok = np.array_equal(mb["i16_mode"], omb["i16_mode"]) and np.array_equal(mb["chroma_mode"], omb["chroma_mode"])
        cbp_ok = np.array_equal(mb["cbp"], omb["cbp"]); tc_ok = np.array_equal(mb["tc"], omb["tc"])
        lv_ok = np.array_equal(lv, olv)
        pre_ok = all(np.array_equal(enc.debug_read(capi.DBG_PRE_Y + p), orc.recon_pre(p)) for p in range(3))
        rec_ok = all(np.array_equal(enc.debug_read(capi.DBG_RECON_Y + p), orc.recon(p)) for p in range(3))
        print(f"{w}x{h} {kind} f{i} {'IDR' if idr else 'P'} len {len(bs)}/{len(obs)} bits={same} mv={mv_ok} type={ty_ok} modes={md_ok} cbp={cbp_ok} tc={tc_ok} lv={lv_ok} pre={pre_ok} rec={rec_ok} {dt*1e3:.2f}ms", flush=True)
        if not (same and mv_ok and ty_ok and md_ok and cbp_ok and tc_ok and lv_ok and pre_ok and rec_ok):
            ok_all = False
            if not mv_ok:
                bad = np.nonzero((mb["mvx"] != omb["mvx"]) | (mb["mvy"] != omb["mvy"]))[0]
                print("  mv mismatch at", bad[:8], [(int(mb["mvx"][j]), int(mb["mvy"][j]), int(omb["mvx"][j]), int(omb["mvy"][j])) for j in bad[:8]], "count", len(bad))
            if not md_ok:
                bad = np.nonzero((mb["i16_mode"] != omb["i16_mode"]) | (mb["chroma_mode"] != omb["chroma_mode"]))[0]
                print("  mode mismatch at", bad[:8], [(int(mb["i16_mode"][j]), int(omb["i16_mode"][j]), int(mb["chroma_mode"][j]), int(omb["chroma_mode"][j])) for j in bad[:8]], "count", len(bad))
            if not lv_ok:
                bad = np.argwhere(lv != olv)
                print("  level mismatch first", bad[:5].tolist(), "count", len(bad), [(int(lv[a,b]), int(olv[a,b])) for a,b in bad[:5]])
            if not tc_ok:
                bad = np.argwhere(mb["tc"] != omb["tc"]); print("  tc mismatch", bad[:5].tolist(), len(bad))
            if not pre_ok:
                for p in range(3):
                    a, b = enc.debug_read(capi.DBG_PRE_Y + p), orc.recon_pre(p)
                    d = np.argwhere(a != b)
                    if len(d): print("  pre plane", p, "first", d[0], "count", len(d))
            if not rec_ok:
                for p in range(3):
                    a, b = enc.debug_read(capi.DBG_RECON_Y + p), orc.recon(p)
                    d = np.argwhere(a != b)
                    if len(d): print("  rec plane", p, "first", d[:3].tolist(), "count", len(d))
            if not same:
                k = next((j for j in range(min(len(bs), len(obs))) if bs[j] != obs[j]), None)
                print("  first differing byte", k)
            break
    enc.close()
    return ok_all

if __name__ == "__main__":
    res = []
    res.append(run(64, 48, 4))
    res.append(run(176, 144, 4, "s2"))
    res.append(run(320, 240, 4))
    res.append(run(200, 120, 3, "s3", 30))
    res.append(run(1920, 1080, 3))
    print("ALL OK" if all(res) else "FAILURES", res)
