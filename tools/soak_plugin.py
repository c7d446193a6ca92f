"""Soak test, GPU box only: `python tools/soak_plugin.py <seed> <cases>` - the plugin surface in the reference's own
operating mode (bitrate control + scene-change IDR) on random sizes / bit rates / frame rates / GOP lengths / content with
cuts, forced key frames and parameter changes; the CPU oracle replays the QP the class reports and the scene-change rule,
the Python controller (media_amd/ratecontrol.py) must predict every QP.  Round 1: seed 4, 300 sequences (10 000 pictures), 0 failures."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, random
from media_amd import synth
from media_amd import videocodec as vc
from media_amd.ratecontrol import RateControl, start_qp
from oracle_lib import OracleEncoder, OracleDecoder
rng = random.Random(int(sys.argv[1])); ncase = int(sys.argv[2])
bad = 0; t0 = time.time()
for case in range(ncase):
    w, h = 16 * rng.randint(4, 40), 16 * rng.randint(3, 30)
    fps = rng.choice([30, 60]); bitrate = rng.choice([1000000, 2000000, 5000000, 10000000]); gop = rng.choice([30, 45, 300])
    prof = rng.choice(["baseline", "main", "high"])
    vc.set_video_mode(w, h, fps=fps, bitrate=bitrate, gop=gop, profile=prof, qp=None)
    e = vc.VideoEncoder()
    assert e.rc_create == vc.SUCCESS and e.init() == vc.SUCCESS and e.start() == vc.SUCCESS
    orc = OracleEncoder(w, h, qp=30, gop=gop, fps=fps, profile_idc={"baseline": 66, "main": 77, "high": 100}[prof])
    dec = OracleDecoder()
    mirror = RateControl(bitrate, fps, qp=start_qp(bitrate, fps, w, h), gop=gop)
    nmb = (w // 16) * (h // 16)
    n = rng.randint(20, 50)
    start = 0
    tag = (case, w, h, fps, bitrate, gop, prof)
    r = np.random.default_rng(case)
    try:
        for i in range(n):
            if rng.random() < 0.08: start = rng.randint(100, 5000)                   # a cut
            kind = rng.random()
            f = synth.frame_s1(w, h, start + i) if kind < 0.85 else r.integers(0, 256, w * h * 3 // 2, dtype=np.uint8)
            force = rng.random() < 0.05
            if force: vc.prop_set("persist.vmi.video.encode.keyframe", "1")
            rc, bs = e.encode(f)
            if rc != vc.SUCCESS: raise RuntimeError("encode rc %d" % rc)
            qp = e.last_qp()
            if qp != mirror.qp: raise RuntimeError("picture %d: class QP %d, mirror %d" % (i, qp, mirror.qp))
            orc.set_qp(qp)
            obs, idr = orc.encode(f, force_idr=force)
            if not idr and orc.me_cost() > 3000 * nmb:
                obs, idr = orc.encode(f, force_idr=True)
            if bs != obs: raise RuntimeError("picture %d differs (%d vs %d bytes)" % (i, len(bs), len(obs)))
            if dec.decode(bs) != 1: raise RuntimeError("picture %d does not decode" % i)
            mirror.update(len(bs), idr)
    except Exception as ex:
        bad += 1; print("FAIL", tag, ex, flush=True)
    e.destroy(); e.delete()
    if case % 20 == 0: print("progress", case, round(time.time() - t0, 1), flush=True)
print("soak_plugin done cases", ncase, "bad", bad)
