"""Closed-GOP sharding end to end across PROCESSES on the GPU box: two ranks with the HIP encoder (both on the box's one
GPU; gloo carries the barrier, the gather and - in bitrate mode - the rate-control broadcast), the schedule of
media_amd.shard.  The reassembled streams must equal what the CPU oracle produces for the same schedule: the serial
stream under fixed QP, the single-process emulation of the round-wise controller hand-over in bitrate mode."""
import os
import sys
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu
W, H, GOP, N_GOPS, FPS, BITRATE = 320, 192, 4, 5, 30, 600000


class HipEngine:
    """the C ABI behind the engine interface the schedule functions drive (the oracle offers the same)"""

    def __init__(self, w, h, qp, gop):
        from media_amd import capi
        self.enc = capi.Encoder(w, h, qp=qp, gop=gop)
        self.IDR = capi.FRAME_IDR

    def set_qp(self, qp):
        self.enc.set_qp(qp)

    def set_idr_id(self, nxt, step):
        self.enc.set_idr_pic_id(nxt, step)

    def encode(self, f, force_idr=False):
        if force_idr:
            self.enc.force_idr()
        bs, ft = self.enc.encode(f)
        return bs, ft == self.IDR


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from media_amd import shard, synth
    from media_amd.ratecontrol import RateControl
    from test_shard_gloo import _rc_encode_gop
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = synth.sequence("s1", W, H, GOP * N_GOPS)
    # fixed QP: no exchange at all
    eng = HipEngine(W, H, 28, GOP)
    eng.set_idr_id(rank, world)
    fixed = {k: b"".join(eng.encode(f)[0] for f in frames[k * GOP:(k + 1) * GOP]) for k in shard.gops_for_rank(N_GOPS, rank, world)}
    # bitrate mode: the controller state crosses ranks once per round
    eng = HipEngine(W, H, 30, GOP)
    eng.set_idr_id(rank, world)
    rc = RateControl(BITRATE, FPS)
    rated = shard.encode_gops_bitrate(lambda k, c: _rc_encode_gop(frames, GOP, FPS, eng, k, c), N_GOPS, rank, world, rc, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (fixed, rated, rc.state()))
    if rank == 0:
        f_all, r_all = {}, {}
        for f, r, _ in gathered:
            f_all.update(f)
            r_all.update(r)
        q.put((shard.reassemble(f_all), shard.reassemble(r_all), [s for _, _, s in gathered]))
    dist.destroy_process_group()


def test_two_processes_shard_the_gops_of_one_stream():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from media_amd import shard, synth
    from media_amd.ratecontrol import RateControl
    from oracle_lib import OracleEncoder
    from test_shard_gloo import _rc_encode_gop
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    fixed, rated, states = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    frames = synth.sequence("s1", W, H, GOP * N_GOPS)
    serial = OracleEncoder(W, H, qp=28, gop=GOP)
    assert fixed == b"".join(serial.encode(f)[0] for f in frames)
    encs = [OracleEncoder(W, H, qp=30, gop=GOP) for _ in range(world)]
    for r, e in enumerate(encs):
        e.set_idr_id(r, world)
    state = RateControl(BITRATE, FPS).state()
    parts = {}
    for j in range((N_GOPS + world - 1) // world):
        nxt = state
        for r in range(world):
            k = j * world + r
            if k >= N_GOPS:
                continue
            rc = RateControl(BITRATE, FPS)
            rc.set_state(state)
            parts[k] = _rc_encode_gop(frames, GOP, FPS, encs[r], k, rc)
            nxt = rc.state()
        state = nxt
    assert shard.reassemble(parts) == rated
    assert states[0] == states[1] == state
