"""Frame-level rate control of the bitrate mode, host side.

Integer-for-integer the controller of VideoEncoderMI355X::RateControlUpdate
(media_amd/host/VideoEncoderMI355X.cpp), which stands in for OpenH264's RC_BITRATE_MODE
(/root/reference/video_codec/VideoEncoderOpenH264.cpp:274; OpenH264's own model is not
available: PARITY UNPINNED).  Kept in Python as well because the multi-GPU drivers
(media_amd/shard.py) carry its state between ranks: the state is five integers, which is
all that ever crosses xGMI when one stream's closed GOPs are sharded in bitrate mode.
"""

QP_MIN, QP_MAX, QP_START = 12, 48, 30


def start_qp(bitrate, fps, width, height):
    """QP of a stream's first picture from its bits per pixel (1 000 x bitrate / (fps x width x height)): 5 Mbit/s at 1080p30
    is 80 -> 30, 1 Mbit/s -> 38, 10 Mbit/s -> 27.  The controller then takes over; this only spares it the first second."""
    mbpp = int(bitrate) * 1000 // max(1, int(fps) * int(width) * int(height))
    for lim, qp in ((200, 24), (100, 27), (50, 30), (25, 34), (12, 38)):
        if mbpp >= lim:
            return qp
    return 42


class RateControl:
    """GOP budget + damped QP steps (round 3; round 2's controller judged every picture against a fixed budget, gave the IDR
    picture a third of what it costs and swung between two QPs).  Integer arithmetic only:
      * a GOP (IDR to IDR) has target * gop bits, less the debt (plus the credit) the virtual buffer carries into it, within
        [1/2, 3/2] of that; the IDR picture is paid out of it, and every P picture is budgeted (what is left) / (pictures left) -
        so a GOP lands on its share of the rate whatever the IDR picture cost;
      * the QP moves by one step when the running mean of the P pictures' bits (3/4 old + 1/4 new: consecutive pictures are
        strongly anti-correlated, a fine picture leaves the next one little to code) is 15 % over or 13 % under the budget of
        the next picture, by two steps beyond 3/2 and 2/3.
    On the 1080p30 S1 content at 5 Mbit/s: 4.94 / 4.99 / 4.98 Mbit/s after 1 / 2 / 3 seconds, QP 25-27 after the first half
    second.  PARITY UNPINNED: OpenH264's own model is not available."""

    def __init__(self, bitrate, fps, qp=QP_START, gop=30):
        self.bitrate = int(bitrate)
        self.fps = max(1, int(fps))
        self.gop = max(1, int(gop))
        self.qp = int(qp)
        self.buffer_bits = 0
        self.gop_left = 0      # bits left for the pictures still to come in this GOP
        self.pics_left = 0     # P pictures still to come in this GOP
        self.mean_p = 0        # running mean of the P pictures' bits (0: none coded yet)

    # -- the state that crosses ranks (five integers) --
    def state(self):
        return (self.qp, self.buffer_bits, self.gop_left, self.pics_left, self.mean_p)

    def set_state(self, state):
        self.qp, self.buffer_bits, self.gop_left, self.pics_left, self.mean_p = (int(x) for x in state)

    def update(self, frame_bytes, is_idr):
        """account one coded picture; returns the QP of the next picture"""
        rate = self.bitrate
        target = rate // self.fps
        bits = int(frame_bytes) * 8
        debt = self.buffer_bits
        self.buffer_bits = max(self.buffer_bits + bits - target, -rate)
        if is_idr or self.pics_left <= 0:
            full = target * self.gop
            budget = min(max(full - debt, full // 2), full * 3 // 2)
            self.gop_left = budget - bits
            self.pics_left = self.gop - 1
            est = self.mean_p if self.mean_p else bits // 9   # (no P picture yet: an IDR picture costs about nine of them)
        else:
            self.gop_left -= bits
            self.pics_left -= 1
            self.mean_p = (self.mean_p * 3 + bits) // 4 if self.mean_p else bits
            est = self.mean_p
        nxt = max(self.gop_left // max(1, self.pics_left), target // 4)
        step = 0
        if est * 2 > nxt * 3:
            step = 2
        elif est * 100 > nxt * 115:
            step = 1
        elif est * 3 < nxt * 2:
            step = -2
        elif est * 100 < nxt * 87:
            step = -1
        self.qp = min(QP_MAX, max(QP_MIN, self.qp + step))
        return self.qp
