"""Soak test, GPU box only: `python tools/soak_streams.py <seed> <cases>` - the stream hub (include/mi355x_h264.h "streams") on
random configurations: 2..14 streams of one random geometry / profile / slice count / search mode on as many host threads, each
with its own content, GOP length, start QP and QP walk (so steps mix picture types and QPs), 5..9 pictures each; every access
unit against the CPU oracle's for that stream.  No exception is tolerated."""
import sys, time, threading; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, random, torch
torch.cuda.init()
from media_amd import capi, synth
from oracle_lib import OracleEncoder
seed = int(sys.argv[1]); ncase = int(sys.argv[2])
rng = random.Random(seed)
bad = 0; t0 = time.time(); pics = 0; steps = 0
for case in range(ncase):
    w, h = 2 * rng.randint(8, 160), 2 * rng.randint(8, 120)
    prof = rng.choice([66, 77, 100]); sl = rng.choice([0, 0, 0, 2, 4]); nodb = int(rng.random() < 0.15); search = rng.choice([0, 1, 1])
    S = rng.randint(2, 14); n = rng.randint(5, 9)
    kinds = ['s1', 'scroll', 'split', 'cut', 's3', 'ramp', 's2']
    tag = (case, w, h, prof, sl, nodb, search, S, n)
    try:
        cfgs = [(rng.choice(kinds), rng.randint(1, 6), rng.randint(12, 48), rng.randint(1, 9), rng.randint(0, 400)) for _ in range(S)]
        seqs = [synth.sequence(k, w, h, n, start=st) for (k, _, _, _, st) in cfgs]
        want = []
        for (k, gop, qp0, mul, st), fr in zip(cfgs, seqs):
            orc = OracleEncoder(w, h, qp=qp0, gop=gop, profile_idc=prof, slices=sl, disable_deblock=nodb, search=search)
            exp = []
            for i, f in enumerate(fr):
                orc.set_qp(min(51, max(10, qp0 + (i * mul) % 9 - 4)))
                exp.append(orc.encode(f)[0])
            want.append(exp); orc.close()
        streams = [capi.Stream(w, h, qp=qp0, gop=gop, profile_idc=prof, slices=sl, disable_deblock=nodb, search=search) for (_, gop, qp0, _, _) in cfgs]
        got = [[] for _ in range(S)]
        errs = []
        go = threading.Barrier(S)
        def work(k):
            try:
                _, gop, qp0, mul, _ = cfgs[k]
                go.wait()
                for i, f in enumerate(seqs[k]):
                    streams[k].set_qp(min(51, max(10, qp0 + (i * mul) % 9 - 4)))
                    got[k].append(streams[k].encode(f)[0])
            except Exception as ex:
                errs.append((k, ex))
        ths = [threading.Thread(target=work, args=(k,)) for k in range(S)]
        for t in ths: t.start()
        for t in ths: t.join()
        st = streams[0].hub_stats(); pics += st["pictures"]; steps += st["steps"]
        for s in streams: s.close()
        if errs:
            bad += 1; print("EXC", tag, errs[0], flush=True); continue
        for k in range(S):
            for i in range(n):
                if got[k][i] != want[k][i]:
                    bad += 1; print("MISMATCH", tag, "stream", k, cfgs[k], "picture", i, flush=True); break
            else: continue
            break
    except Exception as ex:
        bad += 1; print("EXC", tag, ex, flush=True)
    if case % 50 == 0: print("progress", case, round(time.time() - t0, 1), "pictures per step %.2f" % (pics / max(1, steps)), flush=True)
print("soak_streams done seed", seed, "cases", ncase, "bad", bad, "pictures", pics, "steps", steps)
