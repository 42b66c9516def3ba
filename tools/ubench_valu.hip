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
    hipFree(d);
    return 0;
}
