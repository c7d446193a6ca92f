"""Soak test, GPU box only: `python tools/soak_decoder.py <seed> <cases>` - the decoder peer on random streams of the CPU oracle
encoder (geometry, QP, GOP, profile, loop filter, slices, 1..3 references, content incl. partitions / cuts / noise): every
decoded plane against the encoder's reconstruction; every fifth case then feeds damaged copies of the same access units, which
must be refused or decoded without a crash, a hang or a time-out flag; damaged units that BOTH the GPU decoder and the oracle's
independent decoder accept must give the same picture."""
import sys, time; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, random
from media_amd import synth, h264dec
from oracle_lib import OracleEncoder, OracleDecoder
seed = int(sys.argv[1]); ncase = int(sys.argv[2])
rng = random.Random(seed)
bad = 0; refused = 0; survived = 0; agreed = 0; t0 = time.time()
dec = h264dec.Decoder()
for case in range(ncase):
    w, h = 2 * rng.randint(8, 160), 2 * rng.randint(8, 120)
    qp = rng.randint(10, 51); gop = rng.choice([1, 2, 3, 5, 30]); prof = rng.choice([66, 77, 100])
    nodb = rng.random() < 0.2; sl = rng.choice([0, 0, 2, 3, 5]); refs = rng.choice([0, 0, 2, 3])
    kind = rng.choice(['s1', 's1', 'scroll', 's3', 's2', 'split', 'cut', 'ramp'])
    tag = (case, w, h, qp, gop, prof, nodb, sl, refs, kind)
    try:
        enc = OracleEncoder(w, h, qp=qp, gop=gop, profile_idc=prof, disable_deblock=int(nodb), slices=sl, refs=refs)
        aus = []
        for i, f in enumerate(synth.sequence(kind, w, h, 5)):
            au = enc.encode(f)[0]
            aus.append(au)
            if not dec.decode(au): raise RuntimeError("no picture")
            for p in range(3):
                if not np.array_equal(dec.plane(p), enc.recon(p)):
                    raise RuntimeError("plane %d of picture %d differs" % (p, i))
        if case % 5 == 0:
            for _ in range(12):
                au = bytearray(rng.choice(aus))
                for _ in range(rng.randint(1, 4)): au[rng.randrange(len(au))] ^= 1 << rng.randrange(8)
                if rng.random() < 0.3: au = au[: rng.randrange(1, len(au))]
                try:
                    dec.decode(bytes(au)); survived += 1
                except h264dec.StreamError:
                    refused += 1
            # pixel-level differential on damaged units: a fresh GPU decoder and a fresh oracle decoder see the same intact
            # prefix, then the same damaged access unit; if both accept it, they must reconstruct the same picture
            for _ in range(6):
                k = rng.randrange(len(aus))
                au = bytearray(aus[k])
                for _ in range(rng.randint(1, 2)): au[rng.randrange(5, len(au))] ^= 1 << rng.randrange(8)
                g, o = h264dec.Decoder(), OracleDecoder()
                for a in aus[:k]:
                    g.decode(a); o.decode(a)
                try: ok_o = o.decode(bytes(au)) == 1
                except Exception: ok_o = False
                try: ok_g = g.decode(bytes(au))
                except h264dec.StreamError: ok_g = False
                if ok_o and ok_g:
                    agreed += 1
                    for p in range(3):
                        if not np.array_equal(g.plane(p), o.plane(p)):
                            import os, pickle
                            os.makedirs("gpurun_out/r02", exist_ok=True)
                            with open("gpurun_out/r02/dec_diff_%d_%d.bin" % (seed, case), "wb") as fh:
                                pickle.dump({"prefix": aus[:k], "au": bytes(au), "gpu": g.plane(p), "orc": o.plane(p), "plane": p, "tag": tag}, fh)
                            raise RuntimeError("damaged unit %d: plane %d differs from the oracle decoder's" % (k, p))
                g.close()
    except Exception as ex:
        bad += 1; print("BAD", tag, ex, flush=True)
        dec.close(); dec = h264dec.Decoder()
    if case % 100 == 0: print("progress", case, round(time.time() - t0, 1), flush=True)
print("soak_decoder done seed", seed, "cases", ncase, "bad", bad, "damaged units refused", refused, "decoded", survived, "decoded by both decoders to the same picture", agreed)
