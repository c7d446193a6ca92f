"""Slice-band sharding end to end across PROCESSES on the GPU box: two ranks (both on the box's one GPU, so the transport
is gloo over host buffers instead of RCCL over xGMI - RCCL refuses two ranks on one device), each with the HIP encoder in
band mode, media_amd.shard doing the halo swap and the access-unit gather over torch.distributed.  Rank 0 checks every
access unit against the CPU oracle's stream with the same slices."""
import os
import sys
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    from media_amd import capi, shard, synth
    from media_amd.ratecontrol import RateControl
    dist.init_process_group("gloo", rank=rank, world_size=world)
    w, h, slices, n = 640, 368, 6, 8
    enc = capi.Encoder(w, h, qp=27, gop=5, slices=slices, band_index=rank, band_count=world)
    halo = shard.BandHalo(enc.band_info()[4])          # host tensors: gloo carries them
    frames = synth.sequence("s1", w, h, n)
    aus = [shard.encode_picture_bands(enc, f, rank, world, dist, halo) for f in frames]
    enc.close()
    enc = capi.Encoder(w, h, qp=27, gop=5, slices=slices, band_index=rank, band_count=world)
    rc = RateControl(1500000, 30)
    aus_rc = [shard.encode_picture_bands(enc, f, rank, world, dist, halo, rc=rc) for f in frames]
    enc.close()
    if rank == 0:
        q.put((aus, aus_rc))
    dist.barrier()
    dist.destroy_process_group()


def test_two_processes_share_one_picture():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from media_amd import synth
    from media_amd.ratecontrol import RateControl
    from oracle_lib import OracleEncoder, OracleDecoder
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    aus, aus_rc = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    w, h, slices, n = 640, 368, 6, 8
    frames = synth.sequence("s1", w, h, n)
    one = OracleEncoder(w, h, qp=27, gop=5, slices=slices)
    dec = OracleDecoder()
    for i, f in enumerate(frames):
        assert aus[i] == one.encode(f)[0], "picture %d" % i
        assert dec.decode(aus[i]) == 1
    one = OracleEncoder(w, h, qp=27, gop=5, slices=slices)
    rc = RateControl(1500000, 30)
    for i, f in enumerate(frames):
        one.set_qp(rc.qp)
        bs, idr = one.encode(f)
        rc.update(len(bs), idr)
        assert aus_rc[i] == bs, "bitrate mode, picture %d" % i
