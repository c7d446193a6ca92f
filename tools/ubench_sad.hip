#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
template <int MODE>
__global__ void k(unsigned* out, unsigned a, unsigned b, int iters)
{
    unsigned x0 = threadIdx.x * a, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, s = b;
    u64 q0 = x0, q1 = x1, q2 = x2, q3 = x3, w = ((u64)a << 32) | b;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 16; k++) { x0 = __builtin_amdgcn_sad_u8(x0 ^ 1, s, x0); x1 = __builtin_amdgcn_sad_u8(x1 ^ 3, s, x1); x2 = __builtin_amdgcn_sad_u8(x2 ^ 5, s, x2); x3 = __builtin_amdgcn_sad_u8(x3 ^ 7, s, x3); }
        } else if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < 16; k++) { q0 = __builtin_amdgcn_qsad_pk_u16_u8(w ^ q1, s, q0); q1 = __builtin_amdgcn_qsad_pk_u16_u8(w ^ q2, s, q1); q2 = __builtin_amdgcn_qsad_pk_u16_u8(w ^ q3, s, q2); q3 = __builtin_amdgcn_qsad_pk_u16_u8(w ^ q0, s, q3); }
        } else if (MODE == 3) {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 m0 = {x0, x1, x2, x3}, m1 = {x1, x2, x3, x0}, m2 = {x2, x3, x0, x1}, m3 = {x3, x0, x1, x2};
#pragma unroll
            for (int k = 0; k < 16; k++) { m0 = __builtin_amdgcn_mqsad_u32_u8(w ^ q1, s, m0); m1 = __builtin_amdgcn_mqsad_u32_u8(w ^ q2, s, m1); m2 = __builtin_amdgcn_mqsad_u32_u8(w ^ q3, s, m2); m3 = __builtin_amdgcn_mqsad_u32_u8(w ^ q0, s, m3); }
            x0 = m0.x + m0.y + m0.z + m0.w; x1 = m1.x + m1.y + m1.z + m1.w; x2 = m2.x + m2.y + m2.z + m2.w; x3 = m3.x + m3.y + m3.z + m3.w;
        } else if (MODE == 4) {
            typedef short s2 __attribute__((ext_vector_type(2)));
            s2 p0 = __builtin_bit_cast(s2, x0), p1 = __builtin_bit_cast(s2, x1), p2 = __builtin_bit_cast(s2, x2), p3 = __builtin_bit_cast(s2, x3);
#pragma unroll
            for (int k = 0; k < 16; k++) { p0 = p0 + p1; p1 = p1 - p2; p2 = __builtin_elementwise_max(p2, p3); p3 = p3 + p0; }
            x0 = __builtin_bit_cast(unsigned, p0); x1 = __builtin_bit_cast(unsigned, p1); x2 = __builtin_bit_cast(unsigned, p2); x3 = __builtin_bit_cast(unsigned, p3);
        } else if (MODE == 6) {
#pragma unroll
            for (int k = 0; k < 16; k++) { x0 = __builtin_amdgcn_sad_u16(x0 ^ 1, s, x0); x1 = __builtin_amdgcn_sad_u16(x1 ^ 3, s, x1); x2 = __builtin_amdgcn_sad_u16(x2 ^ 5, s, x2); x3 = __builtin_amdgcn_sad_u16(x3 ^ 7, s, x3); }
        } else if (MODE == 7) {
#pragma unroll
            for (int k = 0; k < 16; k++) { x0 = __builtin_amdgcn_lerp(x0 ^ 1, s, x1); x1 = __builtin_amdgcn_lerp(x1 ^ 3, s, x2); x2 = __builtin_amdgcn_lerp(x2 ^ 5, s, x3); x3 = __builtin_amdgcn_lerp(x3 ^ 7, s, x0); }
        } else if (MODE == 8) {
            typedef unsigned short s2 __attribute__((ext_vector_type(2)));
            s2 p0 = __builtin_bit_cast(s2, x0), p1 = __builtin_bit_cast(s2, x1);
#pragma unroll
            for (int k = 0; k < 32; k++) { p0 = p0 + p1; p1 = p1 - p0; }
            x0 = __builtin_bit_cast(unsigned, p0); x1 = __builtin_bit_cast(unsigned, p1);
        } else if (MODE == 5) {
#pragma unroll
            for (int k = 0; k < 16; k++) { x0 = __builtin_amdgcn_perm(x0, x1, s); x1 = __builtin_amdgcn_perm(x1, x2, s); x2 = __builtin_amdgcn_perm(x2, x3, s); x3 = __builtin_amdgcn_perm(x3, x0, s); }
        } else {
#pragma unroll
            for (int k = 0; k < 16; k++) { x0 = x0 * 3 + s; x1 = x1 * 5 + s; x2 = x2 * 7 + s; x3 = x3 * 9 + s; }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + (unsigned)(q0 + q1 + q2 + q3);
}
template <int MODE> void run(const char* name)
{
    unsigned* d; hipMalloc(&d, 4 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 2000, blocks = 256 * 16, threads = 256;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 3u, 5u, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 3u, 5u, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double winstr = (double)blocks * (threads / 64) * iters * 64.0;  // 64 ops per iteration per wave (xor etc. extra)
    printf("%s: %.3f ms, %.2f G wave-instr/s of the measured op (+ 1 xor each)\n", name, ms, winstr / ms / 1e6);
    hipFree(d);
}
int main() { run<2>("v_mad (mul+add)"); run<0>("v_sad_u8"); run<1>("v_qsad_pk_u16_u8"); run<3>("v_mqsad_u32_u8"); run<4>("v_pk_add/sub/max_i16"); run<5>("v_perm_b32"); run<6>("v_sad_u16"); run<7>("v_lerp_u8"); run<8>("dependent v_pk_add/sub chain (64 per iter, no xor)"); return 0; }
