// tools/ubench_issue.hip -- VALU issue-rate controls for gfx950 (MI355X).
//
// Question ("what's weak" 5): is a wave64 VALU instruction issued in 2 or in 4 cycles once a
// SIMD holds several waves, and does that differ between the f32 ops the guide quotes (v_fma_f32: 2 cycles)
// and the integer / packed-16 / byte ops the encoder's kernels are made of?
//
// Every kernel: 256-thread blocks, 16 blocks per CU (=> 16 waves per SIMD requested, occupancy permitting),
// 8 independent dependency chains per lane, 64 instructions of the measured opcode per loop trip written as
// volatile inline asm (the compiler can neither drop nor fuse them); the loop overhead is 3 scalar
// instructions per 64.  Output: G wave-instructions/s and cycles per wave-instruction per SIMD at 2.4 GHz
// (1024 SIMDs).  `waves` sweeps the blocks per CU so the single-wave figure is seen too.
//
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_issue.bin tools/ubench_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP64(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X) REP8(X)

#define DEFK(NAME, ASM3)                                                                                   \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned* out, unsigned a, unsigned b, int iters)     \
    {                                                                                                      \
        unsigned x[8];                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 8; i++) x[i] = threadIdx.x * a + i;                          \
        unsigned s = b, t = a ^ 0x01010101u;                                                               \
        for (int it = 0; it < iters; it++) {                                                               \
            REP64(ASM3)                                                                                    \
        }                                                                                                  \
        unsigned r = 0;                                                                                    \
        _Pragma("unroll") for (int i = 0; i < 8; i++) r += x[i];                                           \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                    \
    }

// three-operand forms: dst = op(dst, s, t) / two-operand: dst = op(dst, s)
#define OP_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_ADDF(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_XOR(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_ADD3(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_LSHLOR(i) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(s));
#define OP_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(x[i]));
#define OP_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_PKADD(i) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_PKSUBI(i) asm volatile("v_pk_sub_i16 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_PKMAD(i) asm volatile("v_pk_mad_u16 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_PKMUL(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_PKMAX(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_PKASHR(i) asm volatile("v_pk_ashrrev_i16 %0, 1, %0" : "+v"(x[i]));
#define OP_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_ALIGN(i) asm volatile("v_alignbyte_b32 %0, %0, %1, 1" : "+v"(x[i]) : "v"(s));
#define OP_SADU8(i) asm volatile("v_sad_u8 %0, %1, %2, %0" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_SADU16(i) asm volatile("v_sad_u16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_LERP(i) asm volatile("v_lerp_u8 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_MADI16(i) asm volatile("v_mad_i32_i16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_DOT2(i) asm volatile("v_dot2_i32_i16 %0, %1, %2, %0" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_DOT4(i) asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_DPPMOV(i) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]));
#define OP_DPPADD(i) asm volatile("v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x[i]) : "v"(s));
#define OP_SATPK(i) asm volatile("v_sat_pk_u8_i16 %0, %0" : "+v"(x[i]));
#define OP_MED3(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_MAXI(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(s));
#define OP_ADDSGPR(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "s"(b));
#define OP_SDWA(i) asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD" : "+v"(x[i]) : "v"(s));
#define OP_PKFMAF(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(sy), "v"(ty));
#define OP_CVTPKU8(i) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(x[i]) : "v"(s));
#define OP_MSAD(i) asm volatile("v_msad_u8 %0, %1, %2, %0" : "+v"(x[i]) : "v"(s), "v"(t));

DEFK(fma_f32, OP_FMA)
DEFK(add_f32, OP_ADDF)
DEFK(add_u32, OP_ADDU)
DEFK(add_u32_sgpr, OP_ADDSGPR)
DEFK(xor_b32, OP_XOR)
DEFK(add3_u32, OP_ADD3)
DEFK(lshl_or_b32, OP_LSHLOR)
DEFK(bfe_u32, OP_BFE)
DEFK(mad_u32_u24, OP_MAD24)
DEFK(mul_lo_u32, OP_MULLO)
DEFK(pk_add_u16, OP_PKADD)
DEFK(pk_sub_i16, OP_PKSUBI)
DEFK(pk_mad_u16, OP_PKMAD)
DEFK(pk_mul_lo_u16, OP_PKMUL)
DEFK(pk_max_i16, OP_PKMAX)
DEFK(pk_ashrrev_i16, OP_PKASHR)
DEFK(perm_b32, OP_PERM)
DEFK(alignbyte_b32, OP_ALIGN)
DEFK(sad_u8, OP_SADU8)
DEFK(msad_u8, OP_MSAD)
DEFK(sad_u16, OP_SADU16)
DEFK(lerp_u8, OP_LERP)
DEFK(mad_i32_i16, OP_MADI16)
DEFK(dot2_i32_i16, OP_DOT2)
DEFK(dot4_u32_u8, OP_DOT4)
DEFK(mov_dpp_quad, OP_DPPMOV)
DEFK(add_u32_dpp_quad, OP_DPPADD)
DEFK(add_u32_sdwa, OP_SDWA)
DEFK(sat_pk_u8_i16, OP_SATPK)
DEFK(med3_i32, OP_MED3)
DEFK(max_i32, OP_MAXI)
DEFK(cndmask_b32, OP_CNDMASK)
DEFK(cvt_pk_u8_f32, OP_CVTPKU8)


#define OP_SUBU(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_SUBREV(i) asm volatile("v_subrev_u32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_AND(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_LSHLC(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(x[i]));
#define OP_LSHRC(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(x[i]));
#define OP_ASHRC(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(x[i]));
#define OP_ASHRV(i) asm volatile("v_ashrrev_i32 %0, %1, %0" : "+v"(x[i]) : "v"(s));
#define OP_MULU24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_MULI24(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_ADDINL(i) asm volatile("v_add_u32 %0, 32, %0" : "+v"(x[i]));
#define OP_ADDLIT(i) asm volatile("v_add_u32 %0, 0x12345, %0" : "+v"(x[i]));
#define OP_MINU(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_MINI(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_MULF(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_FMAC(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_CVTFI(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(x[i]));
#define OP_CVTIF(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(x[i]));
#define OP_CVTUB0(i) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(x[i]));
#define OP_CVTUB3(i) asm volatile("v_cvt_f32_ubyte3 %0, %0" : "+v"(x[i]));
#define OP_FLOOR(i) asm volatile("v_floor_f32 %0, %0" : "+v"(x[i]));
#define OP_ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[i]) : "v"(s));
#define OP_SUBSDWA(i) asm volatile("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_CMPADDC(i) asm volatile("v_cmp_ne_u32 vcc, %1, %0\n v_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(x[i]) : "v"(s) : "vcc");
#define OP_CMP(i) asm volatile("v_cmp_ne_u32 vcc, %1, %0" : : "v"(x[i]), "v"(s) : "vcc");
#define OP_CNDS(i) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "s"(m64));
#define OP_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_MULHI24(i) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_PACK(i) asm volatile("v_pack_b32_f16 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_CVTPKI16(i) asm volatile("v_cvt_pk_i16_i32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_MADI24(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(x[i]) : "v"(s), "v"(t));
#define OP_SUBB(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_XNOR(i) asm volatile("v_xnor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_MAXU(i) asm volatile("v_max_u32 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_MAXI16(i) asm volatile("v_max_i16 %0, %0, %1" : "+v"(x[i]) : "v"(s));
#define OP_ADDU16(i) asm volatile("v_add_u16 %0, %0, %1" : "+v"(x[i]) : "v"(s));

DEFK(sub_u32, OP_SUBU)
DEFK(subrev_u32, OP_SUBREV)
DEFK(and_b32, OP_AND)
DEFK(or_b32, OP_OR)
DEFK(lshlrev_const, OP_LSHLC)
DEFK(lshrrev_const, OP_LSHRC)
DEFK(ashrrev_const, OP_ASHRC)
DEFK(ashrrev_vgpr, OP_ASHRV)
DEFK(mul_u32_u24, OP_MULU24)
DEFK(mul_i32_i24, OP_MULI24)
DEFK(mov_b32, OP_MOV)
DEFK(add_u32_inline, OP_ADDINL)
DEFK(add_u32_literal, OP_ADDLIT)
DEFK(min_u32, OP_MINU)
DEFK(min_i32, OP_MINI)
DEFK(mul_f32, OP_MULF)
DEFK(max_f32, OP_MAXF)
DEFK(fmac_f32, OP_FMAC)
DEFK(cvt_f32_i32, OP_CVTFI)
DEFK(cvt_i32_f32, OP_CVTIF)
DEFK(cvt_f32_ubyte0, OP_CVTUB0)
DEFK(cvt_f32_ubyte3, OP_CVTUB3)
DEFK(floor_f32, OP_FLOOR)
DEFK(and_or_b32, OP_ANDOR)
DEFK(lshl_add_u32, OP_LSHLADD)
DEFK(sub_u32_sdwa_bytes, OP_SUBSDWA)
DEFK(cmp_addc, OP_CMPADDC)
DEFK(cmp_ne_u32, OP_CMP)
DEFK(bfi_b32, OP_BFI)
DEFK(mul_hi_u32_u24, OP_MULHI24)
DEFK(pack_b32_f16, OP_PACK)
DEFK(cvt_pk_i16_i32, OP_CVTPKI16)
DEFK(mad_i32_i24, OP_MADI24)
DEFK(sub_f32, OP_SUBB)
DEFK(xnor_b32, OP_XNOR)
DEFK(max_u32, OP_MAXU)
DEFK(max_i16, OP_MAXI16)
DEFK(add_u16, OP_ADDU16)

__global__ __launch_bounds__(256) void k_cndmask_sgpr(unsigned* out, unsigned a, unsigned b, int iters)
{
    unsigned x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = threadIdx.x * a + i;
    unsigned s = b, t = a ^ 0x01010101u;
    unsigned long long m64 = __ballot((threadIdx.x * a) & 1);
    for (int it = 0; it < iters; it++) { REP64(OP_CNDS) }
    unsigned r = t;
#pragma unroll
    for (int i = 0; i < 8; i++) r += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

__global__ __launch_bounds__(256) void k_pk_fma_f32(unsigned* out, unsigned a, unsigned b, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) y[i] = f2{(float)(threadIdx.x * a + i), 1.0f};
    f2 sy = {1.0f, 0.5f}, ty = {(float)b, 0.25f};
    for (int it = 0; it < iters; it++) { REP64(OP_PKFMAF) }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) r += y[i].x + y[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)r;
}

typedef void (*kfn)(unsigned*, unsigned, unsigned, int);
static unsigned* d_out;

static void run(const char* name, kfn f, int blocks_per_cu)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000, blocks = 256 * blocks_per_cu, threads = 256;
    hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 0, 0, d_out, 3u, 5u, 10);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL(f, dim3(blocks), dim3(threads), 0, 0, d_out, 3u, 5u, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    const double winstr = (double)blocks * (threads / 64) * iters * 64.0;
    const double rate = winstr / best / 1e6;   // G wave-instr/s
    printf("{\"op\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.3f, \"g_wave_instr_per_s\": %.1f, \"cycles_per_wave_instr_per_simd_at_2p4ghz\": %.2f}\n",
           name, blocks_per_cu, best, rate, 1024.0 * 2.4 / rate);
    fflush(stdout);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main(int argc, char** argv)
{
    hipMalloc(&d_out, 256 * 16 * 256 * 4);
    struct { const char* n; kfn f; } ks[] = {
        {"v_fma_f32", k_fma_f32}, {"v_add_f32", k_add_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_add_u32", k_add_u32},
        {"v_add_u32 (sgpr operand)", k_add_u32_sgpr}, {"v_xor_b32", k_xor_b32}, {"v_add3_u32", k_add3_u32},
        {"v_lshl_or_b32", k_lshl_or_b32}, {"v_bfe_u32", k_bfe_u32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mul_lo_u32", k_mul_lo_u32},
        {"v_pk_add_u16", k_pk_add_u16}, {"v_pk_sub_i16", k_pk_sub_i16}, {"v_pk_mad_u16", k_pk_mad_u16}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16},
        {"v_pk_max_i16", k_pk_max_i16}, {"v_pk_ashrrev_i16", k_pk_ashrrev_i16}, {"v_perm_b32", k_perm_b32}, {"v_alignbyte_b32", k_alignbyte_b32},
        {"v_sad_u8", k_sad_u8}, {"v_msad_u8", k_msad_u8}, {"v_sad_u16", k_sad_u16}, {"v_lerp_u8", k_lerp_u8}, {"v_mad_i32_i16", k_mad_i32_i16},
        {"v_dot2_i32_i16", k_dot2_i32_i16}, {"v_dot4_u32_u8", k_dot4_u32_u8}, {"v_mov_b32_dpp quad_perm", k_mov_dpp_quad},
        {"v_add_u32_dpp quad_perm", k_add_u32_dpp_quad}, {"v_add_u32_sdwa", k_add_u32_sdwa}, {"v_sat_pk_u8_i16", k_sat_pk_u8_i16},
        {"v_med3_i32", k_med3_i32}, {"v_max_i32", k_max_i32}, {"v_cndmask_b32", k_cndmask_b32}, {"v_cvt_pk_u8_f32", k_cvt_pk_u8_f32},
        {"v_sub_u32", k_sub_u32}, {"v_subrev_u32", k_subrev_u32}, {"v_and_b32", k_and_b32}, {"v_or_b32", k_or_b32}, {"v_xnor_b32", k_xnor_b32},
        {"v_lshlrev_b32 (const)", k_lshlrev_const}, {"v_lshrrev_b32 (const)", k_lshrrev_const}, {"v_ashrrev_i32 (const)", k_ashrrev_const},
        {"v_ashrrev_i32 (vgpr)", k_ashrrev_vgpr}, {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mul_i32_i24", k_mul_i32_i24}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24},
        {"v_mad_i32_i24", k_mad_i32_i24}, {"v_mov_b32", k_mov_b32}, {"v_add_u32 (inline const)", k_add_u32_inline}, {"v_add_u32 (literal)", k_add_u32_literal},
        {"v_min_u32", k_min_u32}, {"v_max_u32", k_max_u32}, {"v_min_i32", k_min_i32}, {"v_max_i16", k_max_i16}, {"v_add_u16", k_add_u16},
        {"v_mul_f32", k_mul_f32}, {"v_sub_f32", k_sub_f32}, {"v_max_f32", k_max_f32},
        {"v_fmac_f32", k_fmac_f32}, {"v_cvt_f32_i32", k_cvt_f32_i32}, {"v_cvt_i32_f32", k_cvt_i32_f32}, {"v_cvt_f32_ubyte0", k_cvt_f32_ubyte0},
        {"v_cvt_f32_ubyte3", k_cvt_f32_ubyte3}, {"v_floor_f32", k_floor_f32}, {"v_and_or_b32", k_and_or_b32}, {"v_lshl_add_u32", k_lshl_add_u32},
        {"v_sub_u32_sdwa (byte,byte)", k_sub_u32_sdwa_bytes}, {"v_cmp_ne_u32 + v_addc_co_u32 (pair)", k_cmp_addc}, {"v_cmp_ne_u32", k_cmp_ne_u32},
        {"v_cndmask_b32 (sgpr mask)", k_cndmask_sgpr}, {"v_bfi_b32", k_bfi_b32}, {"v_pack_b32_f16", k_pack_b32_f16}, {"v_cvt_pk_i16_i32", k_cvt_pk_i16_i32},
    };
    const int only_full = argc > 1 && !strcmp(argv[1], "--full-only");
    for (auto& k : ks) {
        if (!only_full) { run(k.n, k.f, 1); run(k.n, k.f, 2); }
        run(k.n, k.f, 4);
        run(k.n, k.f, 8);
    }
    hipFree(d_out);
    return 0;
}
