"""Decoder plugin boundary checked against the REFERENCE's own header: oracle/_ref/ref_decoder_header_caller is this build's
caller source (tests/boundary/ref_decoder_header_caller.cpp) compiled with -I/root/reference/video_decoder/include
(VideoDecoder.h used where it lies, never copied) and linked with this build's libVideoDecoder.so.  CPU: no device ->
the object is created, CreateDecoder accepts AVC and refuses HEVC, StartDecoder fails (there is no CPU decoding).  GPU: the
pictures equal the oracle encoder's reconstruction, after the reference's size-change protocol."""
import json
import os
import struct
import subprocess
import numpy as np
import pytest
from media_amd import synth
from oracle_lib import OracleEncoder

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_decoder_header_caller")
REF_HDR = "/root/reference/video_decoder/include/VideoDecoder.h"


def _ensure_built():
    if os.path.exists(REF_HDR):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "_ref"])
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/ref_decoder_header_caller not built and /root/reference absent")


def _run(tmp_path, aus):
    inp, out = os.path.join(str(tmp_path), "in.aus"), os.path.join(str(tmp_path), "out.i420")
    with open(inp, "wb") as f:
        for a in aus:
            f.write(struct.pack("<I", len(a)) + a)
    r = subprocess.run([BIN, inp, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    return rec, open(out, "rb").read()


def test_reference_decoder_header_caller_without_a_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a device is present: the GPU variant of this test runs instead")
    _ensure_built()
    rec, raw = _run(tmp_path, [])
    # VIDEO_DECODER_CREATE_FAIL = 1 for HEVC, DECODE_FAIL = 4 before the start, START_FAIL = 3 for want of a device, PIXEL_FORMAT_YUV_420P = 1
    assert rec == {"create": 0, "hevc": 1, "avc": 0, "init": 0, "port": 0, "out_format": 1, "send_before_start": 4, "start": 3,
                   "destroy_null": 0, "destroy": 0}
    assert raw == b""


@pytest.mark.gpu
def test_reference_decoder_header_caller_decodes(tmp_path):
    if not os.path.exists(BIN):
        pytest.fail("oracle/_ref/ref_decoder_header_caller missing on the GPU box: run __graft_entry__.build() before gpurun")
    w, h = 176, 144
    enc = OracleEncoder(w, h, qp=27, gop=30)
    frames = synth.sequence("s1", w, h, 4)
    aus, recs = [], []
    for f in frames:
        aus.append(enc.encode(f)[0])
        recs.append(np.concatenate([enc.recon(0)[:h, :w].ravel(), enc.recon(1)[: h // 2, : w // 2].ravel(), enc.recon(2)[: h // 2, : w // 2].ravel()]))
    rec, raw = _run(tmp_path, aus)
    assert rec["create"] == 0 and rec["hevc"] == 1 and rec["avc"] == 0 and rec["start"] == 0
    assert rec["send_before_start"] == 4 and rec["underflow"] == 12
    # the first picture: BAD_PIC_SIZE (13) + the size-change event, then delivered once the size is configured; the rest directly
    assert rec["steps"][0] == [0, 13, 0] and rec["steps"][1:] == [[0, 0, 99]] * 3
    assert rec["events"] == 1 and (rec["width"], rec["height"]) == (w, h)
    assert rec["flush"] == 0 and rec["stop"] == 0 and rec["send_after_stop"] == 4 and rec["destroy_null"] == 0 and rec["destroy"] == 0
    fsz = w * h * 3 // 2
    assert len(raw) == 4 * fsz
    for i in range(4):
        assert np.array_equal(np.frombuffer(raw[i * fsz:(i + 1) * fsz], np.uint8), recs[i]), "picture %d" % i
