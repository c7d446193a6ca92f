// tools/ubench_fetch.hip -- calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X) for the ACCESS SHAPES of
// this build's kernels ("settle the FETCH_SIZE x2 question for k_tq's 4 B/lane loads with a known-size
// microbenchmark").
//
// MI355X_MICROARCH.md: FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming read (16 B per lane) and
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern".  Every kernel below
// moves a KNOWN number of bytes exactly once (buffers larger than the 256 MiB Infinity Cache, so nothing is served on-die
// from an earlier pass) in one of the shapes the encoder uses:
//   rd16     : 16 B per lane, 1 KiB contiguous per wave instruction (the guide's reference shape; expected ratio 0.5)
//   rd4_row  : 4 B per lane, 256 B contiguous per wave instruction
//   rd4_tq   : k_tq's luma loads - lane = (macroblock of 4, 4x4 block): per instruction four 64-B segments, one per
//              picture-row group, rows 4 * pitch apart; four instructions cover the 16 rows of four macroblocks
//   rd4_win  : k_me's window loads - lane = (dword column 0..15, row group 0..3), 14 rows per lane: 64-B row segments
//   wr4_tq   : k_tq's reconstruction stores (the rd4_tq shape, stores)
//   wr32_lv  : k_tq's level stores - two 16-B stores per lane, 512 B contiguous per 16 lanes
//   wr16     : 16 B per lane streaming stores (the guide's exact shape for WRITE_SIZE)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes) + `--kernel-trace`; the program prints the
// bytes each kernel moved, tools/summarize_fetch.py divides.  Every load is folded into a checksum stored once per wave, so
// no load can be dropped.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_fetch.bin tools/ubench_fetch.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t r_ = (x); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(r_)); exit(1); } } while (0)

enum { PITCH = 1920, ROWS = 1088 };                   // one 1080p luma plane, as the encoder lays it out
static const size_t PLANE = (size_t)PITCH * ROWS;     // 2 088 960 B
enum { NPL = 192 };                                    // planes per buffer: 401 MB > the 256 MiB Infinity Cache

__global__ __launch_bounds__(64) void rd16(const uint4* __restrict__ src, size_t n16, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    uint32_t acc = 0;
    if (i < n16) { const uint4 v = src[i]; acc = v.x ^ v.y ^ v.z ^ v.w; }
    acc ^= __shfl_xor(acc, 32);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
__global__ __launch_bounds__(64) void rd4_row(const uint32_t* __restrict__ src, size_t n4, uint32_t* out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // four instructions of 256 B each per wave
    uint32_t acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) if (i + 64 * k < n4) acc ^= src[i + 64 * k];
    acc ^= __shfl_xor(acc, 32);
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}
// one wave = four horizontally adjacent macroblocks of one plane (blockIdx.y), all 16 rows: 1 024 B
__device__ __forceinline__ size_t tq_offset(int lane, int r)
{
    const int blk = lane & 15, m = lane >> 4;
    const int bx = (blk & 1) | ((blk >> 1) & 2), by = ((blk >> 1) & 1) | ((blk >> 2) & 2);   // blkIdx -> 4x4 raster
    const int mb4 = blockIdx.x, mbx = (mb4 % (PITCH / 64)) * 4 + m, mby = mb4 / (PITCH / 64);
    return (size_t)blockIdx.y * PITCH * ROWS + (size_t)(16 * mby + 4 * by + r) * PITCH + 16 * mbx + 4 * bx;
}
__global__ __launch_bounds__(64) void rd4_tq(const uint8_t* __restrict__ src, uint32_t* out)
{
    uint32_t acc = 0;
#pragma unroll
    for (int r = 0; r < 4; r++) acc ^= *(const uint32_t*)(src + tq_offset(threadIdx.x, r));
    acc ^= __shfl_xor(acc, 32);
    if (threadIdx.x == 0) out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}
__global__ __launch_bounds__(64) void wr4_tq(uint8_t* dst, uint32_t v)
{
#pragma unroll
    for (int r = 0; r < 4; r++) *(uint32_t*)(dst + tq_offset(threadIdx.x, r)) = v + r;
}
// k_tq's luma accesses since the end of round 3: lane = (macroblock of 8, block row of the pass, block column); a wave = eight
// horizontally adjacent macroblocks, two passes of block rows {0, 1} / {2, 3}: per instruction TWO 128-B segments
__device__ __forceinline__ size_t tq8_offset(int lane, int p, int r)
{
    const int m = lane >> 3, bx = lane & 3, by = 2 * p + ((lane >> 2) & 1);
    const int mb8 = blockIdx.x, mbx = (mb8 % (PITCH / 128)) * 8 + m, mby = mb8 / (PITCH / 128);
    return (size_t)blockIdx.y * PITCH * ROWS + (size_t)(16 * mby + 4 * by + r) * PITCH + 16 * mbx + 4 * bx;
}
__global__ __launch_bounds__(64) void rd4_tq8(const uint8_t* __restrict__ src, uint32_t* out)
{
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int r = 0; r < 4; r++) acc ^= *(const uint32_t*)(src + tq8_offset(threadIdx.x, p, r));
    acc ^= __shfl_xor(acc, 32);
    if (threadIdx.x == 0) out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}
__global__ __launch_bounds__(64) void wr4_tq8(uint8_t* dst, uint32_t v)
{
#pragma unroll
    for (int p = 0; p < 2; p++)
#pragma unroll
        for (int r = 0; r < 4; r++) *(uint32_t*)(dst + tq8_offset(threadIdx.x, p, r)) = v + r;
}
// k_me's window: 56 rows x 64 B at (16 mx - 20, 16 my - 20) rounded down to a dword; one wave per macroblock, interior only.
// Neighbouring macroblocks' windows overlap (each plane byte lies in ~12 windows): the unique bytes per plane are the plane
// itself, so this kernel measures how much of the overlap reaches the memory-side counters, not a 1:1 ratio.
__global__ __launch_bounds__(64) void rd4_win(const uint8_t* __restrict__ src, uint32_t* out)
{
    const int mbw = PITCH / 16, mbh = ROWS / 16;
    // the encoder's XCD-aware map: blocks b and b + 8 share an XCD, each XCD works on one band of the picture
    const int n = mbw * mbh, b = blockIdx.x, per = (n + 7) / 8;
    const int mbi = (b & 7) * per + (b >> 3);
    uint32_t acc = 0;
    if (mbi < n) {
        const int my = mbi / mbw, mx = mbi - my * mbw;
        const int wx0 = 16 * mx - 20, wy0 = 16 * my - 20;
        if (wx0 >= 0 && wx0 + 64 <= PITCH && wy0 >= 0 && wy0 + 56 <= ROWS) {
            const int c = threadIdx.x & 15, rg = threadIdx.x >> 4;
            const uint8_t* rp = src + (size_t)blockIdx.y * PITCH * ROWS + (size_t)(wy0 + rg) * PITCH + wx0 + 4 * c;
#pragma unroll
            for (int t = 0; t < 14; t++) acc ^= *(const uint32_t*)(rp + (size_t)t * 4 * PITCH);
        }
    }
    acc ^= __shfl_xor(acc, 32);
    if (threadIdx.x == 0) out[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
}
__global__ __launch_bounds__(64) void wr32_lv(uint4* dst, size_t n32, uint32_t v)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;   // lane i owns bytes [32 i, 32 i + 32)
    if (i < n32) { dst[2 * i] = make_uint4(v, v + 1, v + 2, v + 3); dst[2 * i + 1] = make_uint4(v + 4, v + 5, v + 6, v + 7); }
}
__global__ __launch_bounds__(64) void wr16(uint4* dst, size_t n16, uint32_t v)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < n16) dst[i] = make_uint4(v, v + 1, v + 2, v + 3);
}
// evicts the caches between two measured kernels: streams a 640 MiB buffer of its own
__global__ __launch_bounds__(256) void flush_caches(uint4* p, size_t n16, uint32_t v)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) { uint4 t = p[i]; t.x += v; p[i] = t; }
}

int main()
{
    const size_t bytes = PLANE * NPL;
    uint8_t *a = nullptr, *b = nullptr;
    uint4* fl = nullptr;
    uint32_t* out = nullptr;
    const size_t flush_bytes = (size_t)640 << 20;
    CK(hipMalloc((void**)&a, bytes + 4096));
    CK(hipMalloc((void**)&b, bytes + 4096));
    CK(hipMalloc((void**)&fl, flush_bytes));
    CK(hipMalloc((void**)&out, (size_t)64 << 20));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    CK(hipMemset(fl, 3, flush_bytes));
    CK(hipDeviceSynchronize());
    auto flush = [&]() { hipLaunchKernelGGL(flush_caches, dim3(4096), dim3(256), 0, 0, fl, flush_bytes / 16, 1u); };
    const int mb4 = (PITCH / 64) * (ROWS / 16), nmb = (PITCH / 16) * (ROWS / 16);
    for (int rep = 0; rep < 3; rep++) {
        flush(); hipLaunchKernelGGL(rd16, dim3((unsigned)(bytes / 16 / 64)), dim3(64), 0, 0, (const uint4*)a, bytes / 16, out);
        flush(); hipLaunchKernelGGL(rd4_row, dim3((unsigned)(bytes / 4 / 256)), dim3(64), 0, 0, (const uint32_t*)a, bytes / 4, out);
        flush(); hipLaunchKernelGGL(rd4_tq, dim3(mb4, NPL), dim3(64), 0, 0, (const uint8_t*)a, out);
        flush(); hipLaunchKernelGGL(rd4_tq8, dim3(mb4 / 2, NPL), dim3(64), 0, 0, (const uint8_t*)a, out);
        flush(); hipLaunchKernelGGL(wr4_tq8, dim3(mb4 / 2, NPL), dim3(64), 0, 0, b, 9u);
        flush(); hipLaunchKernelGGL(rd4_win, dim3(nmb, NPL), dim3(64), 0, 0, (const uint8_t*)a, out);
        flush(); hipLaunchKernelGGL(wr4_tq, dim3(mb4, NPL), dim3(64), 0, 0, b, 7u);
        flush(); hipLaunchKernelGGL(wr32_lv, dim3((unsigned)(bytes / 32 / 64)), dim3(64), 0, 0, (uint4*)b, bytes / 32, 7u);
        flush(); hipLaunchKernelGGL(wr16, dim3((unsigned)(bytes / 16 / 64)), dim3(64), 0, 0, (uint4*)b, bytes / 16, 7u);
    }
    CK(hipDeviceSynchronize());
    // rd4_win: the interior windows' requested bytes (with overlap) and the unique bytes they cover
    long interior = 0;
    for (int my = 0; my < ROWS / 16; my++)
        for (int mx = 0; mx < PITCH / 16; mx++) {
            const int wx0 = 16 * mx - 20, wy0 = 16 * my - 20;
            interior += wx0 >= 0 && wx0 + 64 <= PITCH && wy0 >= 0 && wy0 + 56 <= ROWS;
        }
    printf("{\"bytes\": {\"rd16\": %zu, \"rd4_row\": %zu, \"rd4_tq\": %zu, \"rd4_win_requested\": %zu, \"rd4_win_unique_upper\": %zu, "
           "\"wr4_tq\": %zu, \"wr32_lv\": %zu, \"wr16\": %zu, \"rd4_tq8\": %zu, \"wr4_tq8\": %zu, \"flush_caches_read\": %zu, \"flush_caches_written\": %zu}, \"planes\": %d, \"plane_bytes\": %zu}\n",
           bytes, bytes, bytes, (size_t)interior * 56 * 64 * NPL, bytes, bytes, bytes, bytes, bytes, bytes, flush_bytes, flush_bytes, (int)NPL, PLANE);
    return 0;
}
