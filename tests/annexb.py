"""Test-side Annex-B tools: this build's statement of what the reference's decoder adapter does
with an incoming access unit before it hands it to the hardware - video_decoder/VideoDecoderNetint.cpp:
  * :844-860 FindNalStartCode: first 00 00 01 or 00 00 00 01 in the buffer (-1: none),
  * :794-842 FindNextNonVclNalu: if the first NAL unit of the buffer is NOT a slice (H.264 types 1..5), the number
    of bytes up to the next start code (or the whole buffer) - the adapter walks the leading SPS / PPS this way and
    caches them for its post-flush resume (:770-788); 0 when the first unit is a slice or no start code is found.
The encoder's access units must be digestible by exactly that walk: [SPS][PPS] in front of every IDR picture, slices
after them, 4-byte start codes.  Pure Python, used by CPU and GPU tests."""

START_CODE_MIN = 3


def find_nal_start_code(buf):
    """offset of the first start code (its first zero byte), -1 if there is none"""
    n = len(buf)
    i = 0
    while i + START_CODE_MIN <= n:
        if buf[i] == 0 and buf[i + 1] == 0 and (buf[i + 2] == 1 or (i + 3 < n and buf[i + 2] == 0 and buf[i + 3] == 1)):
            return i
        i += 1
    return -1


def find_next_non_vcl_nalu(buf):
    """(size, nal_type): size of the leading non-VCL NAL unit including its start code, 0 if the buffer starts with a
    slice (types 1..5) or holds no start code; nal_type -1 without a start code"""
    n = len(buf)
    if n <= START_CODE_MIN:
        return 0, -1
    i = find_nal_start_code(buf)
    if i < 0:
        return 0, -1
    i += 4 if buf[i + 2] != 1 else 3
    nal_type = buf[i] & 0x1F
    if 1 <= nal_type <= 5:
        return 0, nal_type
    while True:
        if i + START_CODE_MIN > n:
            return n, nal_type
        if buf[i] == 0 and buf[i + 1] == 0 and buf[i + 2] in (0, 1):
            return i, nal_type
        i += 1


def leading_parameter_sets(au):
    """walk an access unit the way the adapter does: [(nal_type, bytes)] of the non-VCL units in front, rest"""
    out = []
    buf = bytes(au)
    while True:
        size, t = find_next_non_vcl_nalu(buf)
        if size == 0:
            return out, buf
        out.append((t, buf[:size]))
        buf = buf[size:]


def split_nal_units(au):
    """every NAL unit of an access unit as (nal_ref_idc, nal_type, payload without start code)"""
    buf = bytes(au)
    starts = []
    i = 0
    while i + 3 <= len(buf):
        if buf[i] == 0 and buf[i + 1] == 0 and buf[i + 2] == 1:
            starts.append(i + 3)
            i += 3
        else:
            i += 1
    out = []
    for k, s in enumerate(starts):
        e = len(buf) if k + 1 == len(starts) else starts[k + 1] - 3
        while e > s and k + 1 < len(starts) and buf[e - 1] == 0:   # the zero_byte of a following 4-byte start code
            e -= 1
        out.append(((buf[s] >> 5) & 3, buf[s] & 31, buf[s + 1:e]))
    return out
