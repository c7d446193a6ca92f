/* oracle/h264_rgba.c -- TEST INFRASTRUCTURE ONLY (see oracle/README.md): CPU restatement of the RGBA ingest of
 * include/mi355x_h264.h (mi355x_h264_encode_rgba).
 *
 * No reference ENCODER path takes RGBA (the reference names the layout on its decoder side only,
 * /root/reference/video_decoder/include/VideoDecoder.h:40-47), so there is nothing of the reference to restate: the
 * arithmetic is the published BT.601 studio-swing integer conversion (the form ITU-R BT.601 derived code commonly uses:
 * coefficients 66 129 25 / -38 -74 112 / 112 -94 -18 over 256, offsets 16 / 128), chroma from the rounded mean of each 2x2
 * block.  Pinned by the known answers of the primaries (tests/test_oracle_kat.py): white (235, 128, 128), black (16, 128, 128),
 * red (82, 90, 240), green (144, 54, 34), blue (41, 240, 110). */
#include "h264_oracle.h"

void h264o_rgba_to_i420(const uint8_t *rgba, int stride, int w, int h, uint8_t *i420)
{
    uint8_t *Y = i420, *U = i420 + (size_t)w * h, *V = U + (size_t)(w / 2) * (h / 2);
    for (int by = 0; by < h / 2; by++)
        for (int bx = 0; bx < w / 2; bx++) {
            int sr = 0, sg = 0, sb = 0;
            for (int r = 0; r < 2; r++)
                for (int c = 0; c < 2; c++) {
                    const uint8_t *p = rgba + (size_t)(2 * by + r) * stride + 4 * (size_t)(2 * bx + c);
                    const int R = p[0], G = p[1], B = p[2];
                    Y[(size_t)(2 * by + r) * w + 2 * bx + c] = (uint8_t)(((66 * R + 129 * G + 25 * B + 128) >> 8) + 16);
                    sr += R; sg += G; sb += B;
                }
            const int R = (sr + 2) >> 2, G = (sg + 2) >> 2, B = (sb + 2) >> 2;
            /* (the sums are negative for some inputs: >> is the arithmetic shift, floor, here and on the GPU) */
            U[(size_t)by * (w / 2) + bx] = (uint8_t)(((-38 * R - 74 * G + 112 * B + 128) >> 8) + 128);
            V[(size_t)by * (w / 2) + bx] = (uint8_t)(((112 * R - 94 * G - 18 * B + 128) >> 8) + 128);
        }
}
