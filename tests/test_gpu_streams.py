"""Streams of the shared engine (include/mi355x_h264.h "streams"): pictures of DIFFERENT streams leave as
one lockstep step, every picture with its own QP, picture type, frame_num, idr_pic_id and reference pictures.  Whatever the
batching does, a stream's access units must be the CPU oracle's for that stream's pictures and QP sequence."""
import threading
import numpy as np
import pytest
from media_amd import capi, synth
from oracle_lib import OracleEncoder

pytestmark = pytest.mark.gpu


def test_one_stream_equals_the_oracle_through_qp_changes_and_forced_idr():
    w, h = 320, 240
    s = capi.Stream(w, h, qp=26, gop=5)
    orc = OracleEncoder(w, h, qp=26, gop=5)
    for i, f in enumerate(synth.sequence("s1", w, h, 13)):
        if i in (3, 8):
            s.set_qp(20 + i)
            orc.set_qp(20 + i)
        if i == 7:
            s.force_idr()
        bs, ft = s.encode(f)
        obs, idr = orc.encode(f, force_idr=(i == 7))
        assert bs == obs, "picture %d" % i
        assert (ft == capi.FRAME_IDR) == bool(idr)
        assert s.me_cost() == orc.me_cost()
        for p in range(3):
            assert np.array_equal(s.recon(p), orc.recon(p)), "picture %d plane %d" % (i, p)
    assert s.hub_stats()["pictures"] == 13
    s.close()


@pytest.mark.parametrize("w,h,prof,slices,nstreams,npic", [(320, 240, 66, 0, 6, 12), (176, 144, 100, 3, 5, 9), (176, 144, 66, 0, 12, 8), (640, 368, 77, 0, 4, 7),
                                                           (64, 48, 66, 0, 44, 5)])   # 44 streams: two engines (32 streams each at most), one HIP stream per step
def test_streams_on_threads_each_equal_their_oracle(w, h, prof, slices, nstreams, npic):
    """every stream has its own content, its own GOP length (IDR pictures fall on different ticks: steps mix picture types), its
    own QP walk; twelve streams make steps of eight pictures or more (the pair form of the loop filter, indirect)"""
    kinds = ["s1", "scroll", "split", "cut", "s3", "ramp"]
    streams, want, got = [], [], [[] for _ in range(nstreams)]
    seqs = []
    for k in range(nstreams):
        gop = 3 + (k % 4)
        qp0 = 22 + 3 * (k % 5)
        seqs.append(synth.sequence(kinds[k % len(kinds)], w, h, npic, start=17 * k))
        streams.append(capi.Stream(w, h, qp=qp0, gop=gop, profile_idc=prof, slices=slices))
        orc = OracleEncoder(w, h, qp=qp0, gop=gop, profile_idc=prof, slices=slices)
        exp = []
        for i, f in enumerate(seqs[k]):
            orc.set_qp(min(51, qp0 + (i * (k + 1)) % 7))
            exp.append(orc.encode(f)[0])
        want.append(exp)
        orc.close()
    go = threading.Barrier(nstreams)

    def work(k):
        qp0 = 22 + 3 * (k % 5)
        go.wait()
        for i, f in enumerate(seqs[k]):
            streams[k].set_qp(min(51, qp0 + (i * (k + 1)) % 7))
            got[k].append(streams[k].encode(f)[0])

    ths = [threading.Thread(target=work, args=(k,)) for k in range(nstreams)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for k in range(nstreams):
        for i in range(npic):
            assert got[k][i] == want[k][i], "stream %d picture %d" % (k, i)
    st = streams[0].hub_stats()
    assert st["open_streams"] == min(nstreams, 32), "streams beyond an engine's 32 open a second engine"
    assert st["pictures"] == st["open_streams"] * npic
    assert st["steps"] < st["pictures"], "at least one step carried pictures of two streams"
    for s in streams:
        s.close()


def test_streams_come_and_go_and_geometries_do_not_mix():
    a = capi.Stream(176, 144, qp=30, gop=30)
    b = capi.Stream(320, 240, qp=30, gop=30)          # another geometry: another engine
    fa, fb = synth.sequence("s1", 176, 144, 4), synth.sequence("s1", 320, 240, 4)
    oa, ob = OracleEncoder(176, 144, qp=30, gop=30), OracleEncoder(320, 240, qp=30, gop=30)
    assert a.encode(fa[0])[0] == oa.encode(fa[0])[0] and b.encode(fb[0])[0] == ob.encode(fb[0])[0]
    assert a.hub_stats()["open_streams"] == 1 and b.hub_stats()["open_streams"] == 1
    c = capi.Stream(176, 144, qp=24, gop=30)          # joins a's engine mid-stream, starts with its own IDR picture
    oc = OracleEncoder(176, 144, qp=24, gop=30)
    assert a.hub_stats()["open_streams"] == 2
    for i in range(1, 4):
        assert a.encode(fa[i])[0] == oa.encode(fa[i])[0]
        assert c.encode(fa[i])[0] == oc.encode(fa[i])[0]
        assert b.encode(fb[i])[0] == ob.encode(fb[i])[0]
    a.close()                                          # c goes on alone on the engine a opened
    assert c.encode(fa[0])[0] == oc.encode(fa[0])[0]
    c.close()
    b.close()
    d = capi.Stream(176, 144, qp=30, gop=30)          # a fresh engine after the last stream of the old one left
    od = OracleEncoder(176, 144, qp=30, gop=30)
    assert d.encode(fa[0])[0] == od.encode(fa[0])[0]
    d.close()
