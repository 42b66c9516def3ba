// fullsearch.hip -- exhaustive integer-pel search of one 64x64 CTU per workgroup (gfx950, wave64).
//
// Reproduces, for each of the CTU's 85 square PUs, the integer stage of motionEstimate with
// X265_FULL_SEARCH (reference encoder/motion.cpp:997-1032): raster order over [mvmin, mvmax],
// cost = sad + mvcost(mv << 2), strict-less update, i.e. the first minimum in raster order.
//
// Structure
//   * The (64+nx-1) x (64+ny-1) reference window and the source CTU are staged in LDS once.
//   * A lane owns one 16x16 block (its four 8x8 blocks) and four adjacent x positions.  Its 16x16
//     source pixels stay in 64 VGPRs.  Walking down the window one row per step, every new row of
//     reference bytes (five aligned dwords per block row) feeds v_qsad_pk_u16_u8: one instruction =
//     four SADs of 4 pixels at four consecutive x positions, accumulated as 4 x u16.  Eight
//     positions (y .. y+7) are in flight per lane in a rotating set of accumulators; one completes
//     per step.
//   * A wave's 16-lane DPP row holds the 16 blocks of one x group.  16x16 sums are plain 32-bit adds of
//     the packed pairs (16x16 SAD <= 65280, no carry between halves).  32x32 sums are reduce-scattered
//     over the quad (two quad_perm exchanges; lane keeps x position lane & 3), 64x64 sums add the four
//     quads with row_ror 4 and 8.
//   * Per PU a lane keeps min((cost << 10) | (y << 2) | (x & 3)) over its four x positions and all y
//     (v_mad_u32_u16 builds a key from a packed u16 SAD, v_min3_u32 folds two keys per instruction);
//     the unsigned minimum is the first minimum in raster order.  At the end the keys are widened to
//     (cost, y, x) and meet in LDS through 64-bit atomic minima.
//   * Columns beyond a multiple of four (the 129th of a +-64 window) are done one at a time with v_sad_u8 in
//     the same streaming form: all 32 lane rows take the column and 1/32 of the window rows each.
//   * Windows narrower than four columns go through a one-position-per-thread v_sad_u8 loop.
#include "kernels.h"
#include "me_device.h"

namespace x265hip {

#define FS_MAXPOS 132
#define FS_WS 200 // window row pitch in bytes (>= 64 + 132 - 1 + 3, multiple of 4)
#define FS_WROWS (64 + FS_MAXPOS - 1 + 8)
#define FS_THREADS 512
#define FS_STAGE 5 // 16-byte pieces of the window per thread: ceil(195 * 13 / 512)

struct Fs2Lds
{
    uint8_t win[FS_WROWS * FS_WS];
    uint8_t fenc[64 * 64];
    unsigned long long best[X265_HIP_PUS_PER_CTU]; // (cost << 16) | (y << 8) | x
    unsigned pass_best[X265_HIP_PUS_PER_CTU];
    unsigned cy[FS_MAXPOS + 24]; // entry y + 8; read one step ahead, so y runs from -7 to ny + 8
};

__device__ __forceinline__ uint4 ld16u(const uint8_t *p)
{
    uint4 v;
    __builtin_memcpy(&v, p, 16);
    return v;
}

__device__ __forceinline__ int fs_s16(int v) { return (int)(int16_t)v; }

// Search keys: (cost << 10) | (y << 2) | (x & 3).  cost < 2^21 (64x64 SAD + mvcost), y < 256, so a key
// fits 32 bits and the unsigned minimum is the first minimum in raster order among the four x
// positions of a lane.
#define FS_KSHIFT 10

// (half of `packed`) * 1024 + add in one instruction
template <int HI>
__device__ __forceinline__ unsigned mad16_key(unsigned packed, unsigned add)
{
    unsigned r;
    if (HI)
        asm("v_mad_u32_u16 %0, %1, %2, %3 op_sel:[1,0,0,0]" : "=v"(r) : "v"(packed), "s"(1u << FS_KSHIFT), "v"(add));
    else
        asm("v_mad_u32_u16 %0, %1, %2, %3" : "=v"(r) : "v"(packed), "s"(1u << FS_KSHIFT), "v"(add));
    return r;
}

__device__ __forceinline__ unsigned min3u(unsigned a, unsigned b, unsigned c)
{
    unsigned r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_get(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

template <int CTRL>
__device__ __forceinline__ unsigned dpp_add(unsigned v)
{
    return v + (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, false);
}

__device__ __forceinline__ unsigned fs_wave_min_u32(unsigned v)
{
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0xB1, 0xf, 0xf, false));  // quad_perm 1,0,3,2
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x4E, 0xf, 0xf, false));  // quad_perm 2,3,0,1
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x141, 0xf, 0xf, false)); // row_half_mirror
    v = min(v, (unsigned)__builtin_amdgcn_update_dpp(-1, (int)v, 0x140, 0xf, 0xf, false)); // row_mirror
    unsigned a = __builtin_amdgcn_readlane((int)v, 0), b = __builtin_amdgcn_readlane((int)v, 16);
    unsigned c = __builtin_amdgcn_readlane((int)v, 32), d = __builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}

__device__ __forceinline__ unsigned long long u64min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }

__global__ __launch_bounds__(FS_THREADS) void fullsearch2_kernel(const DeviceTables *tabs, const x265_hip_ctu_job *jobs,
                                                                 int n, x265_hip_me_result *out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    Fs2Lds &L = *(Fs2Lds *)smem;
    const int id = blockIdx.x;
    if (id >= n)
        return;
    const x265_hip_ctu_job j = jobs[id];
    const int tid = threadIdx.x;
    const PlaneDesc fp = tabs->planes[j.fenc_plane], rp = tabs->planes[j.ref_plane];
    const uint16_t *tab = tabs->costtab[j.qp];
    const int nx = j.mvmax_x - j.mvmin_x + 1, ny = j.mvmax_y - j.mvmin_y + 1;
    const int ww = 64 + nx - 1, wh = 64 + ny - 1;

    // ---- stage source CTU, reference window, per-row cost, result keys --------------------------
    {
        // 16-byte pieces, every load of a thread issued before its first LDS store.  A window row is at
        // most 195 bytes: 13 pieces, the 13th stored as 8 bytes (row pitch 200).  Loads run up to 12 bytes
        // past the window inside the plane row (planes carry 64 bytes of slack after the last row).
        const uint8_t *fsrc = fp.origin + (long)j.ctu_y * fp.stride + j.ctu_x;
        const uint8_t *rsrc = rp.origin + (long)(j.ctu_y + j.mvmin_y) * rp.stride + j.ctu_x + j.mvmin_x;
        const int w16 = (ww + 15) >> 4, npiece = wh * w16; // <= 195 * 13 <= FS_STAGE * FS_THREADS
        uint4 fv = {}, wv[FS_STAGE];
        int wo[FS_STAGE];
        if (tid < 256)
            fv = ld16u(fsrc + (long)(tid >> 2) * fp.stride + 16 * (tid & 3));
#pragma unroll
        for (int k = 0; k < FS_STAGE; k++)
        {
            const int i = tid + k * FS_THREADS;
            wo[k] = -1;
            if (i < npiece)
            {
                const int y = i / w16, c = i - y * w16;
                wv[k] = ld16u(rsrc + (long)y * rp.stride + 16 * c);
                wo[k] = y * FS_WS + 16 * c;
            }
        }
        if (tid < 256)
            *(uint4 *)(L.fenc + 16 * tid) = fv;
#pragma unroll
        for (int k = 0; k < FS_STAGE; k++)
            if (wo[k] >= 0)
            {
                *(uint2 *)(L.win + wo[k]) = make_uint2(wv[k].x, wv[k].y);
                if (wo[k] % FS_WS < 192)
                    *(uint2 *)(L.win + wo[k] + 8) = make_uint2(wv[k].z, wv[k].w);
            }
        if (tid < ny) // y part of the key: (cost_y << 10) | (y << 2)
            L.cy[tid + 8] = ((unsigned)tab_at(tab, j.qmvp_y, fs_s16((j.mvmin_y + tid) << 2)) << FS_KSHIFT) | ((unsigned)tid << 2);
        if (tid < X265_HIP_PUS_PER_CTU)
        {
            L.best[tid] = ~0ull;
            L.pass_best[tid] = 0xffffffffu;
        }
    }
    __syncthreads();

    const int G = min(32, nx >> 2);
    if (G > 0)
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int q = lane & 15;                 // 16x16 block, z-order: bits 3:2 = 32x32 quadrant
        const int g = wave * 4 + (lane >> 4);    // 16-lane row
        const int q32 = q >> 2, sub = q & 3;
        const int bx16 = (q32 & 1) * 32 + (sub & 1) * 16, by16 = (q32 >> 1) * 32 + (sub >> 1) * 16;

        uint32_t f[16][4];
#pragma unroll
        for (int r = 0; r < 16; r++)
        {
            const uint4 v = *(const uint4 *)(L.fenc + (by16 + r) * 64 + bx16);
            f[r][0] = v.x;
            f[r][1] = v.y;
            f[r][2] = v.z;
            f[r][3] = v.w;
        }
        // 32x32 and 64x64 sums are reduce-scattered over the quad: lane keeps x position (lane & 3)
        const int xl = lane & 3;
        const unsigned sh_keep = (xl & 1) * 16, sh_send = 16 - sh_keep;
        const bool keep_b = (xl & 2) != 0;

        uint64_t acc[8][4]; // [slot][TL, TR, BL, BR], 4 x u16 each
#pragma unroll
        for (int k = 0; k < 8; k++)
#pragma unroll
            for (int b = 0; b < 4; b++)
                acc[k][b] = 0;

        // ---- main pass: x positions [0, 4G), one group of four per 16-lane row, all ny rows -------------
        {
        const bool active = g < G; // rows beyond G repeat group G - 1: the same keys twice
        const int xbase = 4 * (active ? g : G - 1), ybase = 0, npos = ny;
        unsigned cx[4]; // x part of the key: (cost_x << 10) | i
#pragma unroll
        for (int i = 0; i < 4; i++)
            cx[i] = ((unsigned)tab_at(tab, j.qmvp_x, fs_s16((j.mvmin_x + xbase + i) << 2)) << FS_KSHIFT) | (unsigned)i;
        const unsigned cxl = cx[xl]; // carries i = xl in its low bits

        unsigned best8[4], best16 = 0xffffffffu, best32 = 0xffffffffu, best64 = 0xffffffffu;
#pragma unroll
        for (int b = 0; b < 4; b++)
            best8[b] = 0xffffffffu;

        const uint32_t *wtop = (const uint32_t *)(L.win + (by16 + ybase) * FS_WS + bx16 + xbase);
        const unsigned *cyp = L.cy + ybase;
        const int S = npos + 7;
        // the rows of step s + 1 are fetched from LDS while step s computes
        uint32_t nt[5], nb[5];
        unsigned ncy = cyp[1];
#pragma unroll
        for (int c = 0; c < 5; c++)
        {
            nt[c] = wtop[c];
            nb[c] = wtop[8 * (FS_WS / 4) + c];
        }
        for (int s0 = 0; s0 < S; s0 += 8)
        {
#pragma unroll
            for (int jj = 0; jj < 8; jj++)
            {
                const int s = s0 + jj;
                const uint32_t t0 = nt[0], t1 = nt[1], t2 = nt[2], t3 = nt[3], t4 = nt[4];
                const uint32_t b0 = nb[0], b1 = nb[1], b2 = nb[2], b3 = nb[3], b4 = nb[4];
                const unsigned cyv = ncy; // cyp[(s - 7) + 8]
                {
                    ncy = cyp[s + 2];
                    const uint32_t *pt = wtop + (s + 1) * (FS_WS / 4); // row <= 48 + 144 + 8 < FS_WROWS
                    const uint32_t *pb = pt + 8 * (FS_WS / 4);
#pragma unroll
                    for (int c = 0; c < 5; c++)
                    {
                        nt[c] = pt[c];
                        nb[c] = pb[c];
                    }
                }
                const uint64_t t01 = ((uint64_t)t1 << 32) | t0, t12 = ((uint64_t)t2 << 32) | t1;
                const uint64_t t23 = ((uint64_t)t3 << 32) | t2, t34 = ((uint64_t)t4 << 32) | t3;
                const uint64_t b01 = ((uint64_t)b1 << 32) | b0, b12 = ((uint64_t)b2 << 32) | b1;
                const uint64_t b23 = ((uint64_t)b3 << 32) | b2, b34 = ((uint64_t)b4 << 32) | b3;
#pragma unroll
                for (int i = 0; i < 8; i++)
                {
                    const int k = (jj - i) & 7; // position y = s - i lives in slot y & 7
                    // a position's first source row (i == 0) restarts its accumulators
                    const uint64_t z0 = i ? acc[k][0] : 0, z1 = i ? acc[k][1] : 0;
                    const uint64_t z2 = i ? acc[k][2] : 0, z3 = i ? acc[k][3] : 0;
                    acc[k][0] = __builtin_amdgcn_qsad_pk_u16_u8(t12, f[i][1], __builtin_amdgcn_qsad_pk_u16_u8(t01, f[i][0], z0));
                    acc[k][1] = __builtin_amdgcn_qsad_pk_u16_u8(t34, f[i][3], __builtin_amdgcn_qsad_pk_u16_u8(t23, f[i][2], z1));
                    acc[k][2] = __builtin_amdgcn_qsad_pk_u16_u8(b12, f[8 + i][1], __builtin_amdgcn_qsad_pk_u16_u8(b01, f[8 + i][0], z2));
                    acc[k][3] = __builtin_amdgcn_qsad_pk_u16_u8(b34, f[8 + i][3], __builtin_amdgcn_qsad_pk_u16_u8(b23, f[8 + i][2], z3));
                }
                // position y = s - 7 is complete in slot (jj + 1) & 7
                const int y = s - 7;
                if (y >= 0 && y < npos)
                {
                    const int k = (jj + 1) & 7;
                    const unsigned kmask = (1u << (16 + FS_KSHIFT)) - 1; // mvcost() wraps at 16 bits
                    unsigned kmv[4];
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        kmv[i] = (cx[i] + cyv) & kmask;
                    unsigned lo[4], hi[4];
#pragma unroll
                    for (int b = 0; b < 4; b++)
                    {
                        lo[b] = (unsigned)acc[k][b];
                        hi[b] = (unsigned)(acc[k][b] >> 32);
                        best8[b] = min3u(best8[b], mad16_key<0>(lo[b], kmv[0]), mad16_key<1>(lo[b], kmv[1]));
                        best8[b] = min3u(best8[b], mad16_key<0>(hi[b], kmv[2]), mad16_key<1>(hi[b], kmv[3]));
                    }
                    // 16x16: the packed pairs add without carry (4 * 16320 < 65536)
                    const unsigned slo = lo[0] + lo[1] + lo[2] + lo[3], shi = hi[0] + hi[1] + hi[2] + hi[3];
                    best16 = min3u(best16, mad16_key<0>(slo, kmv[0]), mad16_key<1>(slo, kmv[1]));
                    best16 = min3u(best16, mad16_key<0>(shi, kmv[2]), mad16_key<1>(shi, kmv[3]));
                    // the four lanes of a quad hold the four 16x16 blocks of one 32x32.  Exchange with
                    // lane ^ 1 (each keeps two of the four x positions), then with lane ^ 2: the lane ends
                    // with the 32x32 sum of x position (lane & 3); row_ror 4 and 8 add the four quads.
                    const unsigned pa = __builtin_amdgcn_ubfe(slo, sh_keep, 16) + dpp_get<0xB1>(__builtin_amdgcn_ubfe(slo, sh_send, 16));
                    const unsigned pb = __builtin_amdgcn_ubfe(shi, sh_keep, 16) + dpp_get<0xB1>(__builtin_amdgcn_ubfe(shi, sh_send, 16));
                    const unsigned s32 = (keep_b ? pb : pa) + dpp_get<0x4E>(keep_b ? pa : pb);
                    const unsigned kmvl = (cxl + cyv) & kmask;
                    best32 = min(best32, (s32 << FS_KSHIFT) + kmvl);
                    const unsigned s64 = dpp_add<0x128>(dpp_add<0x124>(s32));
                    best64 = min(best64, (s64 << FS_KSHIFT) + kmvl);
                }
            }
        }

        if (active)
        {
            // widen to (cost << 16) | (y << 8) | x and meet the other lanes in LDS
            auto widen = [&](unsigned k) {
                return ((unsigned long long)(k >> FS_KSHIFT) << 16) | (((k >> 2) & 255u) << 8) | (unsigned)(xbase + (k & 3u));
            };
#pragma unroll
            for (int b = 0; b < 4; b++)
            {
                const int x8 = bx16 + (b & 1) * 8, y8 = by16 + (b >> 1) * 8;
                atomicMin(&L.best[21 + (y8 >> 3) * 8 + (x8 >> 3)], widen(best8[b]));
            }
            atomicMin(&L.best[5 + (by16 >> 4) * 4 + (bx16 >> 4)], widen(best16));
            atomicMin(&L.best[1 + q32], widen(best32));
            if (q < 4)
                atomicMin(&L.best[0], widen(best64));
        }
        } // main pass

        // ---- columns [4G, nx): one column at a time with v_sad_u8 (same pixel rate as qsad, and a single
        // column is a quarter of a qsad step's work).  Every 16-lane row takes ceil(ny / 32) window rows
        // (slices may overlap: a position evaluated twice yields the same key twice); same rotating
        // accumulators, now one u32 per 8x8 block.
#pragma clang loop unroll(disable)
        for (int xt = 4 * G; xt < nx; xt++)
        {
            const int npos = (ny + 31) >> 5;
            const int ybase = min(g * npos, ny - npos);
            const unsigned cxk = ((unsigned)tab_at(tab, j.qmvp_x, fs_s16((j.mvmin_x + xt) << 2)) << FS_KSHIFT) | (unsigned)(xt & 3);
            const unsigned kmask = (1u << (16 + FS_KSHIFT)) - 1;
            unsigned b8[4], b16 = 0xffffffffu, b32 = 0xffffffffu, b64 = 0xffffffffu;
#pragma unroll
            for (int b = 0; b < 4; b++)
                b8[b] = 0xffffffffu;
            uint32_t a[8][4];
#pragma unroll
            for (int k = 0; k < 8; k++)
#pragma unroll
                for (int b = 0; b < 4; b++)
                    a[k][b] = 0;
            const uint8_t *wtop = L.win + (by16 + ybase) * FS_WS + bx16 + xt; // any byte alignment
            const unsigned *cyp = L.cy + ybase;
            const int S = npos + 7;
            for (int s0 = 0; s0 < S; s0 += 8)
            {
#pragma unroll
                for (int jj = 0; jj < 8; jj++)
                {
                    const int s = s0 + jj;
                    const uint8_t *pt = wtop + s * FS_WS, *pb = pt + 8 * FS_WS;
                    const uint32_t t0 = ld4u(pt), t1 = ld4u(pt + 4), t2 = ld4u(pt + 8), t3 = ld4u(pt + 12);
                    const uint32_t u0 = ld4u(pb), u1 = ld4u(pb + 4), u2 = ld4u(pb + 8), u3 = ld4u(pb + 12);
#pragma unroll
                    for (int i = 0; i < 8; i++)
                    {
                        const int k = (jj - i) & 7;
                        a[k][0] = __builtin_amdgcn_sad_u8(t1, f[i][1], __builtin_amdgcn_sad_u8(t0, f[i][0], i ? a[k][0] : 0u));
                        a[k][1] = __builtin_amdgcn_sad_u8(t3, f[i][3], __builtin_amdgcn_sad_u8(t2, f[i][2], i ? a[k][1] : 0u));
                        a[k][2] = __builtin_amdgcn_sad_u8(u1, f[8 + i][1], __builtin_amdgcn_sad_u8(u0, f[8 + i][0], i ? a[k][2] : 0u));
                        a[k][3] = __builtin_amdgcn_sad_u8(u3, f[8 + i][3], __builtin_amdgcn_sad_u8(u2, f[8 + i][2], i ? a[k][3] : 0u));
                    }
                    const int y = s - 7;
                    if (y >= 0 && y < npos)
                    {
                        const int k = (jj + 1) & 7;
                        const unsigned kmv = (cxk + cyp[y + 8]) & kmask;
#pragma unroll
                        for (int b = 0; b < 4; b++)
                            b8[b] = min(b8[b], (a[k][b] << FS_KSHIFT) + kmv);
                        const unsigned s16sum = a[k][0] + a[k][1] + a[k][2] + a[k][3];
                        b16 = min(b16, (s16sum << FS_KSHIFT) + kmv);
                        const unsigned s32 = dpp_add<0x4E>(dpp_add<0xB1>(s16sum)); // the quad's four 16x16 blocks
                        b32 = min(b32, (s32 << FS_KSHIFT) + kmv);
                        const unsigned s64 = dpp_add<0x128>(dpp_add<0x124>(s32));  // the row's four quads
                        b64 = min(b64, (s64 << FS_KSHIFT) + kmv);
                    }
                }
            }
            auto widen = [&](unsigned k) {
                return ((unsigned long long)(k >> FS_KSHIFT) << 16) | (((k >> 2) & 255u) << 8) | (unsigned)((xt & ~3) + (k & 3u));
            };
#pragma unroll
            for (int b = 0; b < 4; b++)
            {
                const int x8 = bx16 + (b & 1) * 8, y8 = by16 + (b >> 1) * 8;
                atomicMin(&L.best[21 + (y8 >> 3) * 8 + (x8 >> 3)], widen(b8[b]));
            }
            atomicMin(&L.best[5 + (by16 >> 4) * 4 + (bx16 >> 4)], widen(b16));
            if (sub == 0)
                atomicMin(&L.best[1 + q32], widen(b32));
            if (q == 0)
                atomicMin(&L.best[0], widen(b64));
        }
    }
    __syncthreads();

    // ---- windows narrower than four columns: one position per thread, raster order = thread order --
    const int ntail = G == 0 ? nx : 0;
    const int total = ntail * ny;
    for (int base = 0; base < total; base += FS_THREADS)
    {
        const int idx = base + tid;
        const bool live = idx < total;
        const int py = live ? idx / ntail : 0, px = live ? 4 * G + (idx - py * ntail) : 0;
        int mvc = 0;
        if (live)
            mvc = mvcost_q(tab, j.qmvp_x, j.qmvp_y, fs_s16((j.mvmin_x + px) << 2), fs_s16((j.mvmin_y + py) << 2));
        const uint8_t *wbase = L.win + py * FS_WS + px;
        unsigned s64 = 0;
        for (int q32 = 0; q32 < 4; q32++)
        {
            unsigned s32 = 0;
            for (int q16 = 0; q16 < 4; q16++)
            {
                unsigned s16sum = 0;
                const int x16 = (q32 & 1) * 32 + (q16 & 1) * 16, y16 = (q32 >> 1) * 32 + (q16 >> 1) * 16;
#pragma unroll
                for (int q8 = 0; q8 < 4; q8++)
                {
                    const int x8 = x16 + (q8 & 1) * 8, y8 = y16 + (q8 >> 1) * 8;
                    unsigned s8 = 0;
#pragma unroll
                    for (int r = 0; r < 8; r++)
                    {
                        const uint2 fv = *(const uint2 *)(L.fenc + (y8 + r) * 64 + x8);
                        const uint8_t *w = wbase + (y8 + r) * FS_WS + x8;
                        s8 = __builtin_amdgcn_sad_u8(fv.x, ld4u(w), s8);
                        s8 = __builtin_amdgcn_sad_u8(fv.y, ld4u(w + 4), s8);
                    }
                    s16sum += s8;
                    unsigned key = fs_wave_min_u32(live ? ((s8 + mvc) << 9) | tid : 0xffffffffu);
                    if ((tid & 63) == 0)
                        atomicMin(&L.pass_best[21 + (y8 >> 3) * 8 + (x8 >> 3)], key);
                }
                s32 += s16sum;
                unsigned key = fs_wave_min_u32(live ? ((s16sum + mvc) << 9) | tid : 0xffffffffu);
                if ((tid & 63) == 0)
                    atomicMin(&L.pass_best[5 + (y16 >> 4) * 4 + (x16 >> 4)], key);
            }
            s64 += s32;
            unsigned key = fs_wave_min_u32(live ? ((s32 + mvc) << 9) | tid : 0xffffffffu);
            if ((tid & 63) == 0)
                atomicMin(&L.pass_best[1 + q32], key);
        }
        {
            unsigned key = fs_wave_min_u32(live ? ((s64 + mvc) << 9) | tid : 0xffffffffu);
            if ((tid & 63) == 0)
                atomicMin(&L.pass_best[0], key);
        }
        __syncthreads();
        if (tid < X265_HIP_PUS_PER_CTU)
        {
            const unsigned k = L.pass_best[tid];
            if (k != 0xffffffffu)
            {
                const int widx = base + (int)(k & 511);
                const int wy = widx / ntail, wx = 4 * G + (widx - wy * ntail);
                const unsigned long long key64 = ((unsigned long long)(k >> 9) << 16) | ((unsigned)wy << 8) | (unsigned)wx;
                L.best[tid] = u64min(L.best[tid], key64);
            }
            L.pass_best[tid] = 0xffffffffu;
        }
        __syncthreads();
    }

    if (tid < X265_HIP_PUS_PER_CTU)
    {
        const unsigned long long k = L.best[tid];
        x265_hip_me_result r;
        r.cost = (int)(k >> 16);
        r.mvx = (int16_t)((j.mvmin_x + (int)(k & 255)) << 2);
        r.mvy = (int16_t)((j.mvmin_y + (int)((k >> 8) & 255)) << 2);
        out[(long)id * X265_HIP_PUS_PER_CTU + tid] = r;
    }
}

int fullsearch2_setup()
{
    return (int)hipFuncSetAttribute((const void *)fullsearch2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)sizeof(Fs2Lds));
}

void launch_fullsearch2(const DeviceTables *tabs, const x265_hip_ctu_job *jobs, int n, x265_hip_me_result *out,
                        hipStream_t st)
{
    if (n > 0)
        hipLaunchKernelGGL(fullsearch2_kernel, dim3(n), dim3(FS_THREADS), sizeof(Fs2Lds), st, tabs, jobs, n, out);
}

} // namespace x265hip
