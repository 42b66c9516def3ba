// ubench_valu.hip -- issue-rate microbenchmark for the byte-SAD instructions on gfx950.
// Prints lane-ops/s for v_sad_u8, v_qsad_pk_u16_u8, v_pk_add_u16, v_add_u32 (reference rate).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x * 2654435761u + seed, a1 = a0 ^ 0x9e3779b9u, a2 = a0 + 77, a3 = a1 + 99;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    uint32_t f = a0 * 3 + 1;
    uint64_t r = ((uint64_t)a1 << 32) | a2;
    for (int i = 0; i < ITERS; i++)
    {
        if (OP == 0)
        {
            a0 = __builtin_amdgcn_sad_u8(f, (uint32_t)r, a0);
            a1 = __builtin_amdgcn_sad_u8(f, (uint32_t)(r >> 32), a1);
            a2 = __builtin_amdgcn_sad_u8(f, (uint32_t)r, a2);
            a3 = __builtin_amdgcn_sad_u8(f, (uint32_t)(r >> 32), a3);
        }
        else if (OP == 1)
        {
            q0 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q0);
            q1 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q1);
            q2 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q2);
            q3 = __builtin_amdgcn_qsad_pk_u16_u8(r, f, q3);
        }
        else if (OP == 2)
        {
            asm volatile("v_pk_add_u16 %0, %0, %4\n v_pk_add_u16 %1, %1, %4\n v_pk_add_u16 %2, %2, %4\n v_pk_add_u16 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
        }
        else if (OP == 3)
        {
            asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
        }
        else if (OP == 4)
        {
            asm volatile("v_pk_min_u16 %0, %0, %4\n v_pk_min_u16 %1, %1, %4\n v_pk_min_u16 %2, %2, %4\n v_pk_min_u16 %3, %3, %4"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
        }
#define ASM4(OPC, TAIL)                                                                                          \
    asm volatile(OPC " %0, %0, " TAIL "\n " OPC " %1, %1, " TAIL "\n " OPC " %2, %2, " TAIL "\n " OPC " %3, %3, " TAIL  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)                                                         \
                 : "v"(f), "s"(1024u))
        else if (OP == 6)
            ASM4("v_mad_u32_u16", "%5, %4");
        else if (OP == 7)
            ASM4("v_lshl_add_u32", "10, %4");
        else if (OP == 8)
            ASM4("v_min3_u32", "%4, %4");
        else if (OP == 9)
            ASM4("v_min_u32", "%4");
        else if (OP == 10)
            ASM4("v_bfe_u32", "%4, 16");
        else if (OP == 11)
            ASM4("v_add3_u32", "%4, %4");
        else if (OP == 12)
            ASM4("v_mad_u32_u24", "%5, %4");
        else if (OP == 13)
            ASM4("v_and_or_b32", "%4, %4");
        else if (OP == 14)
            ASM4("v_perm_b32", "%4, %4");
        else if (OP == 15)
        {
            asm volatile("v_add_u32_dpp %0, %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_u32_dpp %1, %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_u32_dpp %2, %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                         "v_add_u32_dpp %3, %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
        }
        else if (OP == 16)
        {
            asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                         "v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                         "v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
                         "v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(f));
        }
        else if (OP == 17)
            ASM4("v_mul_lo_u32", "%4");
        else if (OP == 18)
            ASM4("v_pk_mad_u16", "%4, %4");
        else if (OP == 19)
            ASM4("v_dot4_u32_u8", "%4, %0");
        else if (OP == 5)
        {
            q0 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q0);
            q1 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q1);
            q2 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q2);
            q3 = __builtin_amdgcn_mqsad_pk_u16_u8(r, f, q3);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + (uint32_t)(q0 + q1 + q2 + q3) + (uint32_t)((q0 + q1 + q2 + q3) >> 32);
}

template <int OP>
void run(const char *name, uint32_t *d)
{
    const int blocks = 256 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++)
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 2u + r);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = 5.0 * blocks * 256.0 * ITERS * 4.0;
    printf("%-22s %8.3f ms  %.3e lane-ops/s  (%.2f of 7.86e13 full-rate)\n", name, ms, ops / (ms * 1e-3), ops / (ms * 1e-3) / 7.86e13);
}

int main()
{
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    run<3>("v_add_u32", d);
    run<0>("v_sad_u8", d);
    run<1>("v_qsad_pk_u16_u8", d);
    run<5>("v_mqsad_pk_u16_u8", d);
    run<2>("v_pk_add_u16", d);
    run<4>("v_pk_min_u16", d);
    run<6>("v_mad_u32_u16", d);
    run<12>("v_mad_u32_u24", d);
    run<7>("v_lshl_add_u32", d);
    run<8>("v_min3_u32", d);
    run<9>("v_min_u32", d);
    run<10>("v_bfe_u32", d);
    run<11>("v_add3_u32", d);
    run<13>("v_and_or_b32", d);
    run<14>("v_perm_b32", d);
    run<15>("v_add_u32_dpp", d);
    run<16>("v_add_u32_sdwa", d);
    run<17>("v_mul_lo_u32", d);
    run<18>("v_pk_mad_u16", d);
    run<19>("v_dot4_u32_u8", d);
    hipFree(d);
    return 0;
}
