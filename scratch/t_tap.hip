#include <hip/hip_runtime.h>
#include <cstdio>
#include "../media_amd/csrc/k_pmb2.h"
using namespace h264;
__global__ void kt(int* out, int wxo)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_w[21 * 28 + 12];
    for (int i = threadIdx.x; i < 21 * 28 + 12; i += 64) s_w[i] = (uint8_t)((i * 37 + 11) & 255);
    __syncthreads();
    const int lane = threadIdx.x, blk = lane >> 2, r = lane & 3, lx = blk_x(blk) * 4, ly = blk_y(blk) * 4 + r;
    const int g = (ly + 2) * 28 + wxo + lx + 2;
    int t[4], u[4];
    htap4(s_w, g - 2, t);
    vtap4(s_w, g - 56, 28, u);
    int bad = 0;
    for (int k = 0; k < 4; k++) {
        const uint8_t* p = s_w + g + k;
        int hr = p[-2] - 5 * p[-1] + 20 * p[0] + 20 * p[1] - 5 * p[2] + p[3];
        int vr = p[-56] - 5 * p[-28] + 20 * p[0] + 20 * p[28] - 5 * p[56] + p[84];
        if (hr != t[k]) bad |= 1;
        if (vr != u[k]) bad |= 2;
    }
    uint32_t a = lds_ld4(s_w, g), b = lds_ld4(s_w, g + 1);
    uint32_t av = avg4(a, b);
    for (int k = 0; k < 4; k++) if (byte_of(av, k) != ((s_w[g + k] + s_w[g + 1 + k] + 1) >> 1)) bad |= 4;
    // transform check: residual rows -> fdct_quad vs scalar fdct
    int d[4] = {(lane * 7) % 23 - 11, (lane * 5) % 19 - 9, (lane * 3) % 17 - 8, (lane * 11) % 29 - 14};
    int full[16];
    for (int rr = 0; rr < 4; rr++) for (int c = 0; c < 4; c++) {
        int l2 = (lane & ~3) | rr;
        int dd[4] = {(l2 * 7) % 23 - 11, (l2 * 5) % 19 - 9, (l2 * 3) % 17 - 8, (l2 * 11) % 29 - 14};
        full[4 * rr + c] = dd[c];
    }
    fdct4x4(full);
    fdct_quad(d, r);
    for (int c = 0; c < 4; c++) if (d[c] != full[4 * r + c]) bad |= 8;
    int full2[16]; for (int i = 0; i < 16; i++) full2[i] = full[i] * 3;
    int e[4] = {d[0] * 3, d[1] * 3, d[2] * 3, d[3] * 3};
    idct4x4(full2);
    idct_quad(e, r);
    for (int c = 0; c < 4; c++) if (e[c] != full2[4 * r + c]) bad |= 16;
    out[lane] = bad;
}
int main()
{
    int* d; hipMalloc(&d, 256);
    for (int wxo = 0; wxo < 4; wxo++) {
        hipLaunchKernelGGL(kt, dim3(1), dim3(64), 0, 0, d, wxo);
        int h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
        int all = 0; for (int i = 0; i < 64; i++) all |= h[i];
        printf("wxo %d bad mask %d (1 htap 2 vtap 4 avg 8 fdct 16 idct)\n", wxo, all);
    }
    return 0;
}
