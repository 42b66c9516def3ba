// does LDS accept unaligned 4 / 8 / 12 / 16-byte reads on this box?  prints mismatches per width
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
__global__ void k(uint32_t *out)
{
    __shared__ __attribute__((aligned(16))) uint8_t buf[1024];
    for (int i = threadIdx.x; i < 1024; i += 64)
        buf[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const int o = threadIdx.x * 5 + 1; // odd and even offsets, all alignments mod 16
    uint32_t a;
    __builtin_memcpy(&a, buf + o, 4);
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    typedef uint32_t u3 __attribute__((ext_vector_type(3)));
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u2 b; u3 c; u4 d;
    __builtin_memcpy(&b, buf + o, 8);
    __builtin_memcpy(&c, buf + o, 12);
    __builtin_memcpy(&d, buf + o, 16);
    uint32_t *p = out + threadIdx.x * 10;
    p[0] = a; p[1] = b.x; p[2] = b.y; p[3] = c.x; p[4] = c.y; p[5] = c.z; p[6] = d.x; p[7] = d.y; p[8] = d.z; p[9] = d.w;
}
int main()
{
    uint32_t *d, h[640];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad[4] = { 0, 0, 0, 0 };
    for (int t = 0; t < 64; t++)
    {
        uint8_t ref[16];
        for (int i = 0; i < 16; i++) ref[i] = (uint8_t)((t * 5 + 1 + i) * 7 + 3);
        bad[0] += memcmp(&h[t * 10], ref, 4) != 0;
        bad[1] += memcmp(&h[t * 10 + 1], ref, 8) != 0;
        bad[2] += memcmp(&h[t * 10 + 3], ref, 12) != 0;
        bad[3] += memcmp(&h[t * 10 + 6], ref, 16) != 0;
    }
    printf("unaligned LDS reads, mismatching lanes: b32 %d b64 %d b96 %d b128 %d\n", bad[0], bad[1], bad[2], bad[3]);
    return 0;
}
