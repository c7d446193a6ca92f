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
