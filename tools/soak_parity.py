"""Soak test, GPU box only: `python tools/soak_parity.py <seed> <cases>` - random geometry, QP 10..51 (also changed mid-stream),
GOP, profile, loop filter, NV12 / I420, slices, content (pan, scroll, static, noise, local uncovered areas): every access
unit of the HIP path against the CPU oracle.  Round 1: seeds 1 and 2, 17 000 cases, 0 mismatches (about 17 ms a case).
Round 2 adds 1..3 reference pictures, the 'split' (partitions) and 'cut' (intra in P) contents; no exception is tolerated
(the I_PCM fallback makes a payload overflow impossible); every other case also sends the stream through the oracle's
independent decoder (reconstruction equal, no macroblock above 3 200 bits, no level_prefix above 15)."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, random, torch
torch.cuda.init()
from media_amd import capi, synth
from oracle_lib import OracleEncoder, OracleDecoder
seed = int(sys.argv[1]); ncase = int(sys.argv[2])
rng = random.Random(seed)
bad = 0; t0 = time.time()
def to_nv12(f, w, h):
    y = f[:w*h]; u = f[w*h:w*h*5//4]; v = f[w*h*5//4:]
    return np.concatenate([y, np.stack([u, v], 1).reshape(-1)])
for case in range(ncase):
    w, h = 2 * rng.randint(8, 220), 2 * rng.randint(8, 160)
    qp = rng.randint(10, 51)
    mot = (rng.randint(-20, 20), rng.randint(-20, 20))
    noise = rng.choice([0, 0, 1, 3, 8, 30])
    gop = rng.choice([1, 2, 3, 5, 30])
    prof = rng.choice([66, 77, 100]); nodb = rng.random() < 0.2; nv12 = rng.random() < 0.4
    sl = rng.choice([0, 0, 2, 3, 5, 9])
    kind = rng.choice(['s1', 's1', 'scroll', 'rand', 's2', 'mix', 'split', 'cut'])
    refs = rng.choice([0, 0, 2, 3])
    n = 6
    if kind == 's1': frames = [synth.frame_s1(w, h, i, noise=noise, motion=mot) for i in range(n)]
    elif kind == 'scroll': frames = [synth.frame_scroll(w, h, i) for i in range(n)]
    elif kind == 's2': frames = [synth.frame_s1(w, h, 0)] * n
    elif kind in ('split', 'cut'): frames = synth.sequence(kind, w, h, n)
    elif kind == 'rand':
        r = np.random.default_rng(case + seed * 100000); frames = [r.integers(0, 256, w*h*3//2, dtype=np.uint8) for _ in range(n)]
    else:
        r = np.random.default_rng(case + seed * 100000)
        base = [synth.frame_s1(w, h, i, noise=noise, motion=mot) for i in range(n)]
        frames = []
        for i, f in enumerate(base):
            g = f.copy()
            if i >= 2:   # a block of noise and a flat block appear: local uncovered areas
                yy = g[:w*h].reshape(h, w); y0, x0 = r.integers(0, h - 8), r.integers(0, w - 8)
                yy[y0:y0 + r.integers(4, h // 2 + 5), x0:x0 + r.integers(4, w // 2 + 5)] = r.integers(0, 256)
            frames.append(g)
    tag = (case, w, h, qp, mot, noise, gop, prof, nodb, nv12, sl, kind, refs)
    try:
        enc = capi.Encoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), input_format=int(nv12), slices=sl, refs=refs)
        orc = OracleEncoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), slices=sl, refs=refs)
        odec = OracleDecoder() if case % 2 == 0 else None   # every other case: the stream through the independent decoder too
        for i, f in enumerate(frames):
            if rng.random() < 0.15:
                q2 = rng.randint(10, 51); enc.set_qp(q2); orc.set_qp(q2)
            want = orc.encode(f)[0]
            if odec is not None:
                if odec.decode(want) != 1 or any(not np.array_equal(odec.plane(p), orc.recon(p)) for p in range(3)) or odec.max_mb_bits > 3200 or odec.max_level_prefix > 15:
                    bad += 1; print("ROUNDTRIP", tag, "frame", i, flush=True); break
            if nv12:
                d = torch.from_numpy(to_nv12(f, w, h)).cuda(); got = enc.encode_device(d.data_ptr())[0]
            else:
                got = enc.encode(f)[0]
            if got != want:
                bad += 1; print("MISMATCH", tag, "frame", i, flush=True); break
        enc.close()
    except Exception as ex:
        bad += 1; print("EXC", tag, ex, flush=True)
    if case % 200 == 0: print("progress", case, round(time.time() - t0, 1), flush=True)
print("stress2 done seed", seed, "cases", ncase, "bad", bad)
