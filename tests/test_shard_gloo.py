"""N>1 path on CPU: two gloo ranks shard closed GOPs of one stream (here encoded by the
CPU oracle as the stand-in engine, since this box has no GPU), gather them, and the
reassembled stream must equal the serial encode; the clock is the max over ranks."""
import os
import sys
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from media_amd import shard, synth
    from oracle_lib import OracleEncoder
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, gop, n_gops = 64, 48, 3, 4
    frames = synth.sequence("s1", w, h, gop * n_gops)
    mine = {}
    enc = OracleEncoder(w, h, qp=28, gop=gop)
    enc.set_idr_id(rank, world)                     # rank r owns GOPs r, r+world, ... -> idr_pic_id r, r+world, ...
    for k in shard.gops_for_rank(n_gops, rank, world):
        mine[k] = b"".join(enc.encode(f)[0] for f in frames[k * gop:(k + 1) * gop])
    dist.barrier()
    t = shard.max_over_ranks(1.0 + rank, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    if rank == 0:
        parts = {}
        for g in gathered:
            parts.update(g)
        q.put((shard.reassemble(parts), t))
    dist.destroy_process_group()


def test_gop_sharding_two_ranks():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from media_amd import shard, synth
    from oracle_lib import OracleEncoder
    assert shard.gops_for_rank(5, 1, 2) == [1, 3] and shard.streams_for_rank(4, 0, 4) == [0]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    stream, t = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert t == 2.0                                    # max over ranks
    # the sharded stream is the serial stream, byte for byte
    w, h, gop, n_gops = 64, 48, 3, 4
    frames = synth.sequence("s1", w, h, gop * n_gops)
    enc = OracleEncoder(w, h, qp=28, gop=gop)
    serial = b"".join(enc.encode(f)[0] for f in frames)
    assert serial == stream


def _rc_encode_gop(frames, gop, fps, engine, k, rc):
    """one closed GOP in bitrate mode on `engine` (oracle here, the HIP encoder on a GPU box)"""
    out = []
    for i, f in enumerate(frames[k * gop:(k + 1) * gop]):
        engine.set_qp(rc.qp)
        bs, idr = engine.encode(f, force_idr=(i == 0))
        rc.update(len(bs), idr)
        out.append(bs)
    return b"".join(out)


def _rc_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from media_amd import shard, synth
    from media_amd.ratecontrol import RateControl
    from oracle_lib import OracleEncoder
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, gop, n_gops, fps, bitrate = 96, 64, 4, 5, 30, 120000
    frames = synth.sequence("s1", w, h, gop * n_gops)
    enc = OracleEncoder(w, h, qp=30, gop=gop)
    enc.set_idr_id(rank, world)
    rc = RateControl(bitrate, fps)
    mine = shard.encode_gops_bitrate(lambda k, c: _rc_encode_gop(frames, gop, fps, enc, k, c), n_gops, rank, world, rc, dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (mine, rc.state()))
    if rank == 0:
        parts = {}
        for g, _ in gathered:
            parts.update(g)
        q.put((shard.reassemble(parts), [s for _, s in gathered]))
    dist.destroy_process_group()


def test_bitrate_mode_gop_sharding_broadcasts_rc_state():
    """the one exchange step of the path: two ranks shard the GOPs of a bitrate-mode stream and broadcast the
    controller state once per round; the result equals a single-process emulation of the same schedule."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from media_amd import shard, synth
    from media_amd.ratecontrol import RateControl
    from oracle_lib import OracleEncoder
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_rc_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    stream, states = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert states[0] == states[1]                      # every rank ends on the broadcast state
    # single-process emulation: per round every GOP starts from the round's state; the last GOP's end state carries on
    w, h, gop, n_gops, fps, bitrate = 96, 64, 4, 5, 30, 120000
    frames = synth.sequence("s1", w, h, gop * n_gops)
    encs = [OracleEncoder(w, h, qp=30, gop=gop) for _ in range(world)]
    for r, e in enumerate(encs):
        e.set_idr_id(r, world)
    state = RateControl(bitrate, fps).state()
    parts = {}
    for j in range((n_gops + world - 1) // world):
        nxt = state
        for r in range(world):
            k = j * world + r
            if k >= n_gops:
                continue
            rc = RateControl(bitrate, fps)
            rc.set_state(state)
            parts[k] = _rc_encode_gop(frames, gop, fps, encs[r], k, rc)
            nxt = rc.state()
        state = nxt
    assert shard.reassemble(parts) == stream
    assert states[0] == state
    assert state != RateControl(bitrate, fps).state()   # the controller actually moved


def _band_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from media_amd import shard, synth
    from oracle_lib import OracleEncoder
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from media_amd.ratecontrol import RateControl
    w, h, slices, n = 176, 144, 4, 6
    eng = OracleEncoder(w, h, qp=27, gop=4, slices=slices, band_index=rank, band_count=world)
    halo = shard.BandHalo(eng.halo_bytes())
    aus = [shard.encode_picture_bands(eng, f, rank, world, dist, halo) for f in synth.sequence("s1", w, h, n)]
    # bitrate mode: every rank runs its own copy of the controller on the gathered picture sizes
    eng2 = OracleEncoder(w, h, qp=27, gop=4, slices=slices, band_index=rank, band_count=world)
    rc = RateControl(300000, 30)
    aus_rc = [shard.encode_picture_bands(eng2, f, rank, world, dist, halo, rc=rc) for f in synth.sequence("s1", w, h, n)]
    # scene detection in band mode: the bands' motion costs are summed over the ranks, a cut re-codes the picture as an IDR
    eng3 = OracleEncoder(w, h, qp=27, gop=30, slices=slices, band_index=rank, band_count=world)
    aus_cut = [shard.encode_picture_bands(eng3, f, rank, world, dist, halo, scene_detect=True, picture_mbs=(w // 16) * (h // 16), cut_cost_per_mb=1500)
               for f in synth.sequence("cut", w, h, 5)]
    if rank == 0:
        q.put((aus, aus_rc, rc.state(), aus_cut))
    dist.barrier()
    dist.destroy_process_group()


def test_slice_band_sharding_of_one_picture_swaps_halos():
    """SURVEY.md 8e-3 / BASELINE.json configs[4]: the slices of every picture are split over two ranks; after each
    picture the ranks swap two macroblock rows of reconstruction point to point (gloo here, RCCL send/recv over xGMI
    on GPUs) and rank 0 gathers the slice NAL units.  The result must be the stream one encoder with the same number
    of slices makes.  (Engine: the CPU oracle's band mode; tests/test_gpu_parity.py checks that the HIP encoder's band
    mode equals the oracle's through the same export/import calls.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from media_amd import synth
    from oracle_lib import OracleEncoder
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    aus, aus_rc, rc_state, aus_cut = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from media_amd.ratecontrol import RateControl
    w, h, slices, n = 176, 144, 4, 6
    one = OracleEncoder(w, h, qp=27, gop=4, slices=slices)
    for i, f in enumerate(synth.sequence("s1", w, h, n)):
        assert aus[i] == one.encode(f)[0], "picture %d" % i
    # bitrate mode: the same controller driving ONE encoder makes the same stream and ends in the same state
    one = OracleEncoder(w, h, qp=27, gop=4, slices=slices)
    rc = RateControl(300000, 30)
    qps = set()
    for i, f in enumerate(synth.sequence("s1", w, h, n)):
        one.set_qp(rc.qp)
        qps.add(rc.qp)
        bs, idr = one.encode(f)
        rc.update(len(bs), idr)
        assert aus_rc[i] == bs, "bitrate mode, picture %d" % i
    assert rc.state() == tuple(rc_state) and len(qps) > 1      # (the controller did move the QP)
    # scene detection: ONE encoder under the plugin class's rule (mean motion cost > 3000 per macroblock -> the P picture is
    # dropped and coded again as IDR) makes the same stream; the cut of the "cut" content (picture 2) is found
    one = OracleEncoder(w, h, qp=27, gop=30, slices=slices)
    kinds = []
    for i, f in enumerate(synth.sequence("cut", w, h, 5)):
        bs, idr = one.encode(f)
        if not idr and one.me_cost() > 1500 * (w // 16) * (h // 16):   # (the plugin class's rule with a threshold this small content reaches)
            bs, idr = one.encode(f, force_idr=True)
        assert aus_cut[i] == bs, "scene detection, picture %d" % i
        kinds.append(bool(idr))
    assert kinds == [True, False, True, False, False]
