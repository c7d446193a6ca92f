"""Frame-level rate control of the bitrate mode, host side.

Integer-for-integer the controller of VideoEncoderMI355X::RateControlUpdate
(media_amd/host/VideoEncoderMI355X.cpp), which stands in for OpenH264's RC_BITRATE_MODE
(/root/reference/video_codec/VideoEncoderOpenH264.cpp:274; OpenH264's own model is not
available: PARITY UNPINNED).  Kept in Python as well because the multi-GPU drivers
(media_amd/shard.py) carry its state between ranks: the state is two integers, which is
all that ever crosses xGMI when one stream's closed GOPs are sharded in bitrate mode.
"""

QP_MIN, QP_MAX, QP_START = 12, 48, 30
IDR_WEIGHT = 4   # an IDR picture is budgeted this many P pictures' worth; the budgets of a GOP add up to its share of the rate


class RateControl:
    def __init__(self, bitrate, fps, qp=QP_START, gop=30):
        self.bitrate = int(bitrate)
        self.fps = max(1, int(fps))
        self.gop = max(1, int(gop))
        self.qp = int(qp)
        self.buffer_bits = 0

    # -- the state that crosses ranks --
    def state(self):
        return (self.qp, self.buffer_bits)

    def set_state(self, state):
        self.qp, self.buffer_bits = int(state[0]), int(state[1])

    def update(self, frame_bytes, is_idr):
        """account one coded picture; returns the QP of the next picture"""
        rate = self.bitrate
        target = rate // self.fps
        bits = int(frame_bytes) * 8
        self.buffer_bits = max(self.buffer_bits + bits - target, -rate)
        p_budget = target * self.gop // (self.gop - 1 + IDR_WEIGHT) if self.gop > 1 else target
        budget = IDR_WEIGHT * p_budget if (is_idr and self.gop > 1) else p_budget
        step = 0
        if bits * 2 > budget * 3:
            step = 2
        elif bits * 10 > budget * 11:
            step = 1
        elif bits * 3 < budget * 2:
            step = -2
        elif bits * 10 < budget * 9:
            step = -1
        if self.buffer_bits * 4 > rate:     # a quarter second of debt, then half a second
            step += 1
        if self.buffer_bits * 2 > rate:
            step += 1
        if self.buffer_bits * 4 < -rate:
            step -= 1
        if self.buffer_bits * 2 < -rate:
            step -= 1
        self.qp = min(QP_MAX, max(QP_MIN, self.qp + step))
        return self.qp
