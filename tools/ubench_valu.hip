// ubench_valu.hip -- VALU issue-rate microbenchmark for gfx950 (MI355X): which integer forms issue in 2 cycles
// per wave64 instruction and which in 4 (or more), at 1 / 2 / 4 / 8 waves per SIMD.
//
// Every wave runs ITERS x 8 independent chains of one instruction and brackets the loop with s_memtime (shader
// clock ticks, MI355X_MICROARCH.md); the host also times the launch with HIP events.  Reported per form and
// occupancy:
//   GHz        shader clock while the loop runs = s_memtime ticks / s_memrealtime ticks (100 MHz), mean over waves
//   wave c/i   shader cycles one wave spends per instruction of its own (issue interval seen by a single wave)
//   SIMD c/i   shader cycles per wave-instruction per SIMD from the throughput: clock x wall time x 1024 SIMDs /
//              all wave-instructions of the launch (the chip is loaded evenly: 256 workgroups of 4 waves per step)
//   lane-ops/s chip-wide = waves x 64 x instructions / wall time
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define ITERS 2048
#define CHAINS 8

#define R8(M) M("%0") M("%1") M("%2") M("%3") M("%4") M("%5") M("%6") M("%7")
// %8 = a VGPR operand, %9 = an SGPR operand, %10 = a second VGPR operand
#define M_ADD_E32(r) "v_add_u32_e32 " r ", %8, " r "\n"
#define M_ADD_E64(r) "v_add_u32_e64 " r ", " r ", %8\n"
#define M_SUB_E32(r) "v_sub_u32_e32 " r ", %8, " r "\n"
#define M_AND_E32(r) "v_and_b32_e32 " r ", %8, " r "\n"
#define M_OR_E32(r) "v_or_b32_e32 " r ", %8, " r "\n"
#define M_XOR_E32(r) "v_xor_b32_e32 " r ", %8, " r "\n"
#define M_LSHL_E32(r) "v_lshlrev_b32_e32 " r ", 1, " r "\n"
#define M_LSHR_E32(r) "v_lshrrev_b32_e32 " r ", 1, " r "\n"
#define M_ASHR_E32(r) "v_ashrrev_i32_e32 " r ", 1, " r "\n"
#define M_MIN_U32_E32(r) "v_min_u32_e32 " r ", %8, " r "\n"
#define M_MAX_I32_E32(r) "v_max_i32_e32 " r ", %8, " r "\n"
#define M_MOV(r) "v_mov_b32_e32 " r ", %8\n"
#define M_ADDF_E32(r) "v_add_f32_e32 " r ", %8, " r "\n"
#define M_FMA(r) "v_fma_f32 " r ", " r ", %8, %10\n"
#define M_FMAC(r) "v_fmac_f32_e32 " r ", %8, %10\n"
#define M_CNDMASK(r) "v_cndmask_b32_e32 " r ", %8, " r ", vcc\n"
#define M_MUL_LO(r) "v_mul_lo_u32 " r ", " r ", %8\n"
#define M_MAD24(r) "v_mad_u32_u24 " r ", " r ", %9, %8\n"
#define M_MAD16(r) "v_mad_u32_u16 " r ", " r ", %9, %8\n"
#define M_ADD3(r) "v_add3_u32 " r ", " r ", %8, %10\n"
#define M_LSHL_ADD(r) "v_lshl_add_u32 " r ", " r ", 1, %8\n"
#define M_LSHL_OR(r) "v_lshl_or_b32 " r ", " r ", 16, %8\n"
#define M_AND_OR(r) "v_and_or_b32 " r ", " r ", %8, %10\n"
#define M_BFE(r) "v_bfe_u32 " r ", " r ", 4, 16\n"
#define M_PERM(r) "v_perm_b32 " r ", " r ", %8, %10\n"
#define M_ALIGNBYTE(r) "v_alignbyte_b32 " r ", " r ", %8, 1\n"
#define M_MIN3(r) "v_min3_u32 " r ", " r ", %8, %10\n"
#define M_MED3(r) "v_med3_i32 " r ", " r ", %8, %10\n"
#define M_SAD_U8(r) "v_sad_u8 " r ", %8, %10, " r "\n"
#define M_PK_ADD_U16(r) "v_pk_add_u16 " r ", " r ", %8\n"
#define M_PK_SUB_I16(r) "v_pk_sub_i16 " r ", " r ", %8\n"
#define M_PK_MAX_I16(r) "v_pk_max_i16 " r ", " r ", %8\n"
#define M_PK_MAD_U16(r) "v_pk_mad_u16 " r ", " r ", %8, %10\n"
#define M_PK_LSHL(r) "v_pk_lshlrev_b16 " r ", 1, " r "\n"
#define M_DOT4_I8(r) "v_dot4_i32_i8 " r ", %8, %10, " r "\n"
#define M_DOT4_U8(r) "v_dot4_u32_u8 " r ", %8, %10, " r "\n"
#define M_DOT2_I16(r) "v_dot2_i32_i16 " r ", %8, %10, " r "\n"
#define M_DOT2C_I16(r) "v_dot2c_i32_i16_e32 " r ", %8, %10\n"
#define M_DOT4C_I8(r) "v_dot4c_i32_i8_e32 " r ", %8, %10\n"
#define M_MUL_U24(r) "v_mul_u32_u24_e32 " r ", %8, " r "\n"
#define M_SBFE(r) "v_bfe_i32 " r ", " r ", 8, 8\n"
#define M_ADD_DPP(r) "v_add_u32_dpp " r ", %8, " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define M_MOV_DPP(r) "v_mov_b32_dpp " r ", %8 row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define M_ADD_SDWA(r) "v_add_u32_sdwa " r ", %8, " r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define M_PK_ADD_F32(r) "v_pk_add_f32 %0, %0, %[w]\n v_pk_add_f32 %1, %1, %[w]\n v_pk_add_f32 %2, %2, %[w]\n v_pk_add_f32 %3, %3, %[w]\n"

enum
{
    OP_ADD_E32, OP_ADD_E64, OP_SUB_E32, OP_AND_E32, OP_OR_E32, OP_XOR_E32, OP_LSHL_E32, OP_LSHR_E32, OP_ASHR_E32,
    OP_MIN_U32_E32, OP_MAX_I32_E32, OP_MOV, OP_ADDF_E32, OP_FMA, OP_FMAC, OP_CNDMASK, OP_MUL_LO, OP_MAD24, OP_MAD16,
    OP_ADD3, OP_LSHL_ADD, OP_LSHL_OR, OP_AND_OR, OP_BFE, OP_PERM, OP_ALIGNBYTE, OP_MIN3, OP_MED3, OP_SAD_U8,
    OP_PK_ADD_U16, OP_PK_SUB_I16, OP_PK_MAX_I16, OP_PK_MAD_U16, OP_PK_LSHL, OP_DOT4_I8, OP_DOT4_U8, OP_DOT2_I16,
    OP_ADD_DPP, OP_MOV_DPP, OP_ADD_SDWA, OP_QSAD, OP_MQSAD, OP_DOT2C_I16, OP_DOT4C_I8, OP_MUL_U24, OP_SBFE, OP_COUNT
};

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint64_t *ticks, uint32_t seed)
{
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 99;
    uint32_t a4 = a0 * 7 + 1, a5 = a1 * 5 + 3, a6 = a2 ^ 0x55aa55aau, a7 = a3 + 0x1234567u;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3, q4 = a4, q5 = a5, q6 = a6, q7 = a7;
    const uint32_t f = a0 * 3 + 1, g = a1 * 9 + 5;
    const uint64_t r = ((uint64_t)a1 << 32) | a2;
    const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
#define RUN(M) asm volatile(R8(M) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(f), "s"(1024u), "v"(g) : "vcc")
    for (int i = 0; i < ITERS; i++)
    {
        if (OP == OP_ADD_E32) RUN(M_ADD_E32);
        else if (OP == OP_ADD_E64) RUN(M_ADD_E64);
        else if (OP == OP_SUB_E32) RUN(M_SUB_E32);
        else if (OP == OP_AND_E32) RUN(M_AND_E32);
        else if (OP == OP_OR_E32) RUN(M_OR_E32);
        else if (OP == OP_XOR_E32) RUN(M_XOR_E32);
        else if (OP == OP_LSHL_E32) RUN(M_LSHL_E32);
        else if (OP == OP_LSHR_E32) RUN(M_LSHR_E32);
        else if (OP == OP_ASHR_E32) RUN(M_ASHR_E32);
        else if (OP == OP_MIN_U32_E32) RUN(M_MIN_U32_E32);
        else if (OP == OP_MAX_I32_E32) RUN(M_MAX_I32_E32);
        else if (OP == OP_MOV) RUN(M_MOV);
        else if (OP == OP_ADDF_E32) RUN(M_ADDF_E32);
        else if (OP == OP_FMA) RUN(M_FMA);
        else if (OP == OP_FMAC) RUN(M_FMAC);
        else if (OP == OP_CNDMASK) RUN(M_CNDMASK);
        else if (OP == OP_MUL_LO) RUN(M_MUL_LO);
        else if (OP == OP_MAD24) RUN(M_MAD24);
        else if (OP == OP_MAD16) RUN(M_MAD16);
        else if (OP == OP_ADD3) RUN(M_ADD3);
        else if (OP == OP_LSHL_ADD) RUN(M_LSHL_ADD);
        else if (OP == OP_LSHL_OR) RUN(M_LSHL_OR);
        else if (OP == OP_AND_OR) RUN(M_AND_OR);
        else if (OP == OP_BFE) RUN(M_BFE);
        else if (OP == OP_PERM) RUN(M_PERM);
        else if (OP == OP_ALIGNBYTE) RUN(M_ALIGNBYTE);
        else if (OP == OP_MIN3) RUN(M_MIN3);
        else if (OP == OP_MED3) RUN(M_MED3);
        else if (OP == OP_SAD_U8) RUN(M_SAD_U8);
        else if (OP == OP_PK_ADD_U16) RUN(M_PK_ADD_U16);
        else if (OP == OP_PK_SUB_I16) RUN(M_PK_SUB_I16);
        else if (OP == OP_PK_MAX_I16) RUN(M_PK_MAX_I16);
        else if (OP == OP_PK_MAD_U16) RUN(M_PK_MAD_U16);
        else if (OP == OP_PK_LSHL) RUN(M_PK_LSHL);
        else if (OP == OP_DOT4_I8) RUN(M_DOT4_I8);
        else if (OP == OP_DOT4_U8) RUN(M_DOT4_U8);
        else if (OP == OP_DOT2_I16) RUN(M_DOT2_I16);
        else if (OP == OP_ADD_DPP) RUN(M_ADD_DPP);
        else if (OP == OP_MOV_DPP) RUN(M_MOV_DPP);
        else if (OP == OP_ADD_SDWA) RUN(M_ADD_SDWA);
        else if (OP == OP_DOT2C_I16) RUN(M_DOT2C_I16);
        else if (OP == OP_DOT4C_I8) RUN(M_DOT4C_I8);
        else if (OP == OP_MUL_U24) RUN(M_MUL_U24);
        else if (OP == OP_SBFE) RUN(M_SBFE);
        else if (OP == OP_QSAD)
        {
            q0 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q0); q1 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q1);
            q2 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q2); q3 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q3);
            q4 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q4); q5 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q5);
            q6 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q6); q7 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q7);
        }
        else if (OP == OP_MQSAD)
        {
            q0 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q0); q1 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q1);
            q2 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q2); q3 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q3);
            q4 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q4); q5 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q5);
            q6 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q6); q7 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q7);
        }
    }
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
    const uint64_t q = q0 + q1 + q2 + q3 + q4 + q5 + q6 + q7;
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)q + (uint32_t)(q >> 32);
    if ((threadIdx.x & 63) == 0)
    {
        ticks[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
        ticks[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
    }
}

static const char *k_names[OP_COUNT] = {
    "v_add_u32_e32", "v_add_u32_e64", "v_sub_u32_e32", "v_and_b32_e32", "v_or_b32_e32", "v_xor_b32_e32", "v_lshlrev_b32_e32",
    "v_lshrrev_b32_e32", "v_ashrrev_i32_e32", "v_min_u32_e32", "v_max_i32_e32", "v_mov_b32_e32", "v_add_f32_e32", "v_fma_f32",
    "v_fmac_f32_e32", "v_cndmask_b32_e32", "v_mul_lo_u32", "v_mad_u32_u24", "v_mad_u32_u16", "v_add3_u32", "v_lshl_add_u32",
    "v_lshl_or_b32", "v_and_or_b32", "v_bfe_u32", "v_perm_b32", "v_alignbyte_b32", "v_min3_u32", "v_med3_i32", "v_sad_u8",
    "v_pk_add_u16", "v_pk_sub_i16", "v_pk_max_i16", "v_pk_mad_u16", "v_pk_lshlrev_b16", "v_dot4_i32_i8", "v_dot4_u32_u8",
    "v_dot2_i32_i16", "v_add_u32_dpp", "v_mov_b32_dpp", "v_add_u32_sdwa", "v_qsad_pk_u16_u8", "v_mqsad_pk_u16_u8",
    "v_dot2c_i32_i16_e32", "v_dot4c_i32_i8_e32", "v_mul_u32_u24_e32", "v_bfe_i32",
};

static uint32_t *d_out;
static uint64_t *d_ticks;
static uint64_t h_ticks[2 * 256 * 8 * 4];

template <int OP>
void run_one(int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd; // 256 CUs, one 4-wave workgroup per CU and occupancy step
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    // warm up for about 30 ms so that the clock has settled under this instruction mix
    for (int r = 0; r < 40; r++)
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; r++)
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, 2u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h_ticks, d_ticks, sizeof(uint64_t) * 2 * blocks * 4, hipMemcpyDeviceToHost);
    double mean = 0, real = 0;
    for (int i = 0; i < blocks * 4; i++)
    {
        mean += (double)h_ticks[2 * i];
        real += (double)h_ticks[2 * i + 1];
    }
    const double insts = (double)ITERS * CHAINS;
    const double wall = ms * 1e-3 / reps;
    const double ghz = mean / real * 0.1; // s_memrealtime counts at 100 MHz
    mean /= blocks * 4;
    printf("%-20s waves/SIMD %d  clock %4.2f GHz  wave c/i %5.2f  SIMD c/i %5.2f  %.3e lane-ops/s  (kernel %.1f us)\n", k_names[OP],
           waves_per_simd, ghz, mean / insts, ghz * 1e9 * wall * 1024.0 / (blocks * 4.0 * insts), blocks * 4.0 * 64.0 * insts / wall,
           wall * 1e6);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
}

template <int OP>
void run_all()
{
    for (int w : { 1, 2, 4, 8 })
        run_one<OP>(w);
    if constexpr (OP + 1 < OP_COUNT)
        run_all<OP + 1>();
}

int main()
{
    hipMalloc(&d_out, 256 * 8 * 256 * 4);
    hipMalloc(&d_ticks, sizeof(h_ticks));
    run_all<0>();
    hipFree(d_out);
    hipFree(d_ticks);
    return 0;
}
