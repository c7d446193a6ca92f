"""media_amd -- MI355X-native H.264 encode path behind the kunpengcompute/media
VideoCodecApi / VideoEncoder plugin surface.

  csrc/   hand-written HIP kernels for gfx950 + the C ABI (include/mi355x_h264.h)
  host/   C++ host side: VideoCodecApi factory, VideoEncoderMI355X, Property, MediaLog
  capi.py ctypes plumbing over the C ABI (tests, bench)
  synth.py synthetic I420 inputs of SURVEY.md 8(d)
"""
__version__ = "0.1.0"
