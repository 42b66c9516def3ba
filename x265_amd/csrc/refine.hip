// refine.hip -- one sub-pel refinement ring per request, with the interpolation shared through LDS.
//
// Reference: the sub-pel stage of MotionEstimate::motionEstimate (encoder/motion.cpp:1047-1119): each iteration
// measures subpelCompare(ref, bmv + square1[i] * step, cmp) (:1126-1164) for i = 1 .. 4 or 8 around one centre, step 2
// (half-pel ring) or 1 (quarter-pel ring), plus (first half-pel iteration with SATD) the centre itself.  The nine
// positions are a 3 x 3 grid: three horizontal positions x three vertical positions.  luma_hps + luma_vsp
// (common/ipfilter.cpp:117-160, 241-282) is separable and its first stage is exact (no rounding at 8 bits), so
//   * one horizontal pass per horizontal position k gives an int16 plane I_k = sum_t c[fx_k][t] * (p - 128) that all
//     three vertical positions of that column of the grid share;
//   * with the tap row {0,0,0,64,0,0,0,0} a stage is the identity up to scale, and the two-stage formula with it
//     reproduces luma_hpp / luma_vpp / the full-pel copy bit for bit ((64 S + 2048) >> 12 == (S + 32) >> 6), so all
//     nine positions go through the same code whatever their fractions are.
//
// Shape of the kernel.  A wave takes a batch of requests and walks over their cells (<= 16 x 16 pixels of a PU), as
// many cells at a time as fill its 64 lanes and its LDS region (four 8x8 PUs, one 16x16 cell, ...):
//   stage A  items (cell, k, row pair, 4 pixels): 16-byte load of the reference row, 8 v_dot4_i32_i8, the two rows
//            stored as vertical int16 pairs in both pairings -- E[m] = (row 2m, row 2m+1), O[m] = (row 2m+1, row 2m+2) --
//            so that stage B finds (row t, row t+1) in one dword for any t;
//   stage B  items (cell, candidate, strip of 4 x 8 pixels): per output row 4 x 4 v_dot2_i32_i16 on pairs read with
//            ds_read_b128 (taps pre-multiplied by 16: the upper half of the sum is the >> 12), clamp on packed
//            pairs, then SAD or the packed 4x4 Hadamard of me_device.h against the source block; partial sums meet in
//            LDS atomics.
// Per 16x16 cell and ring of 8: about 150 + 330 wave-instructions against 8 x 95 for eight independent cost requests.
#include "kernels.h"
#include "me_device.h"

namespace x265hip {

#define RF_BATCH 16     // requests per batch (at most)
#define RF_SLOTS 8      // cells worked on at a time (at most)
#define RF_PLANE_DW 1664 // filtered planes per wave in dwords: four 8x8 cells (4 x 408) or one 16x16 cell (1200)
#define RF_RAW_DW 544    // raw reference windows per wave in dwords: four 8x8 cells (4 x 17 rows x 8)
#define RF_PARTS 64
#define RF_CHUNK 8
#define RF_MAX_BLOCKS (256 * 4)
#define RF_Z 3 // raw 16-byte loads per lane and group (at most: 136 for four 8x8 cells)

// vertical taps x 16 as int16 pairs (c0, c1) (c2, c3) (c4, c5) (c6, c7)
#define RF_V16(c0, c1, c2, c3, c4, c5, c6, c7) \
    { X265HIP_PK16(16 * (c0), 16 * (c1)), X265HIP_PK16(16 * (c2), 16 * (c3)), X265HIP_PK16(16 * (c4), 16 * (c5)), X265HIP_PK16(16 * (c6), 16 * (c7)) }
__device__ __constant__ __attribute__((aligned(16))) const uint32_t k_vtaps16[4][4] = {
    RF_V16(0, 0, 0, 64, 0, 0, 0, 0),
    RF_V16(-1, 4, -10, 58, 17, -5, 1, 0),
    RF_V16(-1, 4, -11, 40, 40, -11, 4, -1),
    RF_V16(0, 1, -5, 17, 58, -10, 4, -1),
};
// (dx, dy) of square1[0..8] (motion.cpp:71-73) as (dx + 1) | (dy + 1) << 2
__device__ __constant__ const uint8_t k_sq_kj[9] = { 1 | 1 << 2, 1 | 0 << 2, 1 | 2 << 2, 0 | 1 << 2, 2 | 1 << 2, 0 | 0 << 2, 0 | 2 << 2, 2 | 0 << 2, 2 | 2 << 2 };

struct RefineLds
{
    uint32_t plane[RF_PLANE_DW];
    uint32_t raw[RF_RAW_DW + 4]; // a row's last read may touch one dword past its window
    int cost[RF_BATCH * X265_HIP_RING_COSTS];
    const uint8_t *fenc[RF_BATCH]; // PU origin in the source plane
    const uint8_t *ref[RF_BATCH];  // PU origin in the reference plane
    int fstride[RF_BATCH], rstride[RF_BATCH];
    int geom[RF_BATCH]; // log2(cw / 4) | log2(ch / 4) << 2 | ncx << 4 | strips are 8 tall << 12 | metric << 13
    int qmv[RF_BATCH];  // (cy << 16) | (cx & 0xffff)
    int pat[RF_BATCH];  // step | ncand << 4 | centre << 8
    int pre[RF_BATCH + 1]; // exclusive prefix of the requests' cell counts
    // the cells of a group, two sets: the next group is formed and its loads issued while the current one computes
    int s_req[2][RF_SLOTS], s_xy[2][RF_SLOTS], s_plane[2][RF_SLOTS], s_raw[2][RF_SLOTS];
};

// wave-uniform description of a group of cells
struct Group
{
    int size, total_z, total_a, total_b;
    int zpre[RF_SLOTS], apre[RF_SLOTS], bpre[RF_SLOTS]; // exclusive prefixes of the item counts; slots past the group never match
};

__device__ __forceinline__ int rf_s16(int v) { return (int)(int16_t)v; }

__device__ __forceinline__ int scan8(int v) // inclusive scan over lanes 0..7 of a DPP row
{
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    return v;
}

// slot of item i and the slot's first item
__device__ __forceinline__ int slot_of(const int pre[RF_SLOTS], int i, int &first)
{
    int g = 0;
    first = 0;
#pragma unroll
    for (int t = 1; t < RF_SLOTS; t++)
        if (pre[t] <= i)
        {
            g = t;
            first = pre[t];
        }
    return g;
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void refine_kernel(const DeviceTables *tabs, const x265_hip_refine_req *reqs,
                                                                                           int n, int32_t *out, uint32_t *counters, int parity, int rb)
{
    __shared__ RefineLds lds[4];
    RefineLds &L = lds[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;

    // Work per batch is bounded: every wave looks at the same 64 requests spread over the list (multiples of the
    // golden ratio, see cost_body) and halves rb until a batch holds about 1024 stage-B items by their mean.
    if (rb > 1)
    {
        const int si = n >= 64 ? (int)(((unsigned long long)((unsigned)lane * 0x9E3779B1u) * (unsigned long long)n) >> 32) : lane;
        int items = 0;
        if (si < n)
            items = (reqs[si].pu_w >> 2) * ((reqs[si].pu_h + 7) >> 3) * 8;
        for (int o = 32; o; o >>= 1)
            items += __shfl_xor(items, o);
        const int cnt = n < 64 ? n : 64;
        while (rb > 1 && items * rb > 1024 * cnt)
            rb >>= 1;
        rb = uniform(rb);
    }
    const int nbatch = (n + rb - 1) / rb;
    const int parts = min((int)gridDim.x, RF_PARTS), part = blockIdx.x % parts;
    const int nwaves = 4 * (((int)gridDim.x - part + parts - 1) / parts);
    auto batch_of = [&](int j) { return ((j / RF_CHUNK) * parts + part) * RF_CHUNK + j % RF_CHUNK; };
    const int nchunk = (nbatch + RF_CHUNK - 1) / RF_CHUNK;
    const int hi = ((nchunk - part + parts - 1) / parts) * RF_CHUNK;
    uint32_t *const counter = counters + (parity * RF_PARTS + part) * 16;
    if (blockIdx.x == 0 && threadIdx.x < RF_PARTS)
        counters[((parity ^ 1) * RF_PARTS + threadIdx.x) * 16] = 0;
    int b = (blockIdx.x / parts) * 4 + (threadIdx.x >> 6);
    x265_hip_refine_req cur = {};
    if (lane < rb && b < hi && (long)batch_of(b) * rb + lane < n)
        cur = reqs[batch_of(b) * rb + lane];

    while (b < hi)
    {
        int nb = 0;
        if (lane == 0)
            nb = (int)atomicAdd(counter, 1u);
        nb = nwaves + uniform(nb);
        const int base = batch_of(b) * rb;
        const bool have = lane < rb && base + lane < n;
        // ---- per request: addresses, geometry, cell count ------------------------------------------------
        int ncell = 0;
        if (have)
        {
            const PlaneDesc fp = tabs->planes[cur.fenc_plane], rp = tabs->planes[cur.ref_plane];
            const int w = cur.pu_w, h = cur.pu_h;
            const int lcw = (w & 15) == 0 ? 2 : (w & 7) == 0 ? 1 : 0, lch = (h & 15) == 0 ? 2 : (h & 7) == 0 ? 1 : 0;
            const int ncx = w >> (2 + lcw), ncy = h >> (2 + lch);
            const int step = (cur.pattern & X265_HIP_RING_HPEL) ? 2 : 1;
            const int centre = (cur.pattern & X265_HIP_RING_CENTRE) ? 1 : 0;
            const int ncand = ((cur.pattern & X265_HIP_RING_8) ? 8 : 4) + centre;
            ncell = ncx * ncy;
            L.fenc[lane] = fp.origin + (long)cur.pu_y * fp.stride + cur.pu_x;
            L.ref[lane] = rp.origin + (long)cur.pu_y * rp.stride + cur.pu_x;
            L.fstride[lane] = fp.stride;
            L.rstride[lane] = rp.stride;
            L.geom[lane] = lcw | (lch << 2) | (ncx << 4) | ((lch >= 1) << 12) | ((cur.metric & 1) << 13);
            L.qmv[lane] = (int)(((uint32_t)(uint16_t)cur.qmv_y << 16) | (uint16_t)cur.qmv_x);
            L.pat[lane] = step | (ncand << 4) | (centre << 8);
        }
        {
            // exclusive prefix of the cell counts over lanes 0..15 (one DPP row)
            int inc = ncell;
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x111, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x112, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x114, 0xf, 0xf, true);
            inc += __builtin_amdgcn_update_dpp(0, inc, 0x118, 0xf, 0xf, true);
            if (lane < RF_BATCH)
                L.pre[lane + 1] = inc;
            if (lane == 0)
                L.pre[0] = 0;
        }
        for (int i = lane; i < RF_BATCH * X265_HIP_RING_COSTS; i += 64)
            L.cost[i] = 0;
        // next batch's records: in flight during this batch's arithmetic
        {
            const int nid = batch_of(nb) * rb + lane;
            if (lane < rb && nb < hi && nid < n)
                cur = reqs[nid];
        }
        wave_sync();
        const int total_cells = L.pre[RF_BATCH];

        // lane g < 8 describes cell c0 + g; the group is the longest prefix that fits the lanes and the LDS regions
        auto form = [&](int c0, int set, Group &G) {
            int g_req = 0, g_ipc = 0, g_apc = 0, g_zpc = 0, g_dw = 0, g_rdw = 0, g_xy = 0;
            const int ci = c0 + lane;
            const bool cell_ok = lane < RF_SLOTS && ci < total_cells;
            if (cell_ok)
            {
#pragma unroll
                for (int s = 8; s > 0; s >>= 1)
                    if (L.pre[g_req + s] <= ci)
                        g_req += s;
                const int geom = L.geom[g_req], pat = L.pat[g_req];
                const int lcw = geom & 3, lch = (geom >> 2) & 3, ncx = (geom >> 4) & 255;
                const int cw = 4 << lcw, ch = 4 << lch, rows = ch + 9;
                const int q = ci - L.pre[g_req];
                const int qy = (int)(((float)q + 0.5f) * __builtin_amdgcn_rcpf((float)ncx)), qx = q - qy * ncx;
                g_xy = (qx * cw) | ((qy * ch) << 16);
                const int nstrip = (1 << lcw) * (lch == 2 ? 2 : 1);
                const int nload = lcw ? 2 : 1; // 16-byte pieces of a raw row: cw + 11 bytes
                g_ipc = ((pat >> 4) & 15) * nstrip;
                g_apc = 3 * ((rows + 1) >> 1) * (1 << lcw);
                g_zpc = rows * nload;
                g_dw = 3 * rows * cw;
                g_rdw = rows * nload * 4;
            }
            const int sc_ipc = scan8(g_ipc), sc_apc = scan8(g_apc), sc_zpc = scan8(g_zpc), sc_dw = scan8(g_dw), sc_rdw = scan8(g_rdw);
            const bool fits = cell_ok && (lane == 0 || (sc_ipc <= 64 && sc_dw <= RF_PLANE_DW && sc_rdw <= RF_RAW_DW));
            const int gsize = __builtin_ctzll(~__ballot(fits));
            if (lane < gsize)
            {
                L.s_req[set][lane] = g_req;
                L.s_xy[set][lane] = g_xy;
                L.s_plane[set][lane] = sc_dw - g_dw;
                L.s_raw[set][lane] = sc_rdw - g_rdw;
            }
            G.size = gsize;
            G.zpre[0] = G.apre[0] = G.bpre[0] = 0;
#pragma unroll
            for (int g = 1; g < RF_SLOTS; g++)
            {
                G.zpre[g] = g < gsize ? __builtin_amdgcn_readlane(sc_zpc, g - 1) : 0x7fffffff;
                G.apre[g] = g < gsize ? __builtin_amdgcn_readlane(sc_apc, g - 1) : 0x7fffffff;
                G.bpre[g] = g < gsize ? __builtin_amdgcn_readlane(sc_ipc, g - 1) : 0x7fffffff;
            }
            G.total_z = __builtin_amdgcn_readlane(sc_zpc, gsize - 1);
            G.total_a = __builtin_amdgcn_readlane(sc_apc, gsize - 1);
            G.total_b = __builtin_amdgcn_readlane(sc_ipc, gsize - 1);
        };

        // stage 0, first half: this lane's raw 16-byte pieces of the group's reference windows into registers
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 rawv[RF_Z];
        int rawdst[RF_Z];
        auto issue0 = [&](int set, const Group &G) {
#pragma unroll
            for (int z = 0; z < RF_Z; z++)
            {
                const int i = lane + 64 * z;
                rawdst[z] = -1;
                if (i < G.total_z)
                {
                    int first;
                    const int g = slot_of(G.zpre, i, first);
                    const int r = L.s_req[set][g], xy = L.s_xy[set][g], geom = L.geom[r], q = L.qmv[r], step = L.pat[r] & 3;
                    const int lnl = (geom & 3) ? 1 : 0;
                    const int rem = i - first, row = rem >> lnl, hf = rem & lnl;
                    const int cx = (int)(int16_t)q, cy = q >> 16;
                    const int ix0 = rf_s16(cx - step) >> 2, rb_row = (rf_s16(cy - step) >> 2) - 3;
                    const gptr_t p = as_global(L.ref[r]) + (long)((xy >> 16) + rb_row + row) * L.rstride[r] + (xy & 0xffff) + ix0 - 3;
                    rawv[z] = *(const __attribute__((address_space(1))) u32x4 *)(p - ((unsigned)(uintptr_t)p & 3u) + 16 * hf);
                    rawdst[z] = L.s_raw[set][g] + ((row << lnl) + hf) * 4;
                }
            }
        };
        auto store0 = [&]() {
#pragma unroll
            for (int z = 0; z < RF_Z; z++)
                if (rawdst[z] >= 0)
                    *(u32x4 *)(L.raw + rawdst[z]) = rawv[z];
        };

        Group G, GN;
        int c0 = 0, set = 0;
        form(c0, set, G);
        wave_sync();
        issue0(set, G);
        for (;;)
        {
            store0();
            wave_sync();
            // ---- stage A: horizontal pass, raw window -> int16 planes in both vertical pairings ------------------------
            for (int i = lane; i < G.total_a; i += 64)
            {
                int first;
                const int g = slot_of(G.apre, i, first);
                const int r = L.s_req[set][g], xy = L.s_xy[set][g], geom = L.geom[r], q = L.qmv[r], step = L.pat[r] & 3;
                const int lcw = geom & 3, lch = (geom >> 2) & 3, cw = 4 << lcw, rows = (4 << lch) + 9, ne = (rows + 1) >> 1;
                const int lnl = lcw ? 1 : 0;
                const int rem = i - first, xg = rem & ((1 << lcw) - 1), t = rem >> lcw;
                const int k = (int)(((float)t + 0.5f) * __builtin_amdgcn_rcpf((float)ne)), m = t - k * ne;
                const int cx = (int)(int16_t)q, cy = q >> 16;
                const int qx = rf_s16(cx + (k - 1) * step), fx = qx & 3;
                const int ix0 = rf_s16(cx - step) >> 2, rb_row = (rf_s16(cy - step) >> 2) - 3;
                const int rstride = L.rstride[r];
                // low address bits of raw row 0's first wanted byte: the alignment of a row inside its window
                const unsigned a0 = (unsigned)(uintptr_t)L.ref[r] + (unsigned)(((xy >> 16) + rb_row) * rstride + (xy & 0xffff) + ix0 - 3);
                const int r0 = 2 * m, r1 = min(2 * m + 1, rows - 1);
                const uint32_t tlo = k_taps_s8[fx][0], thi = k_taps_s8[fx][1];
                const uint32_t *const rawbase = L.raw + L.s_raw[set][g];
                int a[4], bb[4];
                auto hrow = [&](int row, int o[4]) {
                    const int bo = ((a0 + (unsigned)(row * rstride)) & 3u) + ((qx >> 2) - ix0) + 4 * xg;
                    const uint32_t *w = rawbase + (row << (2 + lnl)) + (bo >> 2);
                    const uint32_t w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
                    const unsigned sh = bo & 3;
                    const uint32_t d0 = __builtin_amdgcn_alignbyte(w1, w0, sh), d1 = __builtin_amdgcn_alignbyte(w2, w1, sh),
                                   d2 = __builtin_amdgcn_alignbyte(w3, w2, sh);
                    tap8x4<0>(d0 ^ 0x80808080u, d1 ^ 0x80808080u, d2 ^ 0x80808080u, tlo, thi, o);
                };
                hrow(r0, a);
                hrow(r1, bb);
                uint32_t *const plane = L.plane + L.s_plane[set][g] + k * rows * cw;
                uint4 e;
                e.x = pack_s16(a[0], bb[0]);
                e.y = pack_s16(a[1], bb[1]);
                e.z = pack_s16(a[2], bb[2]);
                e.w = pack_s16(a[3], bb[3]);
                *(uint4 *)(plane + m * cw + 4 * xg) = e; // E[m]
                // O[m] = (row 2m+1, row 2m+2): this item's odd row is its low half, its even row the high half of O[m-1]
                int16_t *const o = (int16_t *)(plane + ne * cw);
                if (m < (rows >> 1)) // O has rows / 2 entries (rows is odd)
#pragma unroll
                    for (int x = 0; x < 4; x++)
                        o[2 * (m * cw + 4 * xg + x)] = (int16_t)bb[x];
                if (m >= 1)
#pragma unroll
                    for (int x = 0; x < 4; x++)
                        o[2 * ((m - 1) * cw + 4 * xg + x) + 1] = (int16_t)a[x];
            }
            // the next group: formed now, its raw loads fly during stage B
            c0 += G.size;
            const bool more = c0 < total_cells;
            if (more)
            {
                form(c0, set ^ 1, GN);
                wave_sync();
                issue0(set ^ 1, GN);
            }
            else
            {
#pragma unroll
                for (int z = 0; z < RF_Z; z++)
                    rawdst[z] = -1;
                wave_sync();
            }

            // ---- stage B: vertical pass per candidate + comparison -----------------------------------------
            for (int i = lane; i < G.total_b; i += 64)
            {
                int first;
                const int g = slot_of(G.bpre, i, first);
                const int r = L.s_req[set][g], xy = L.s_xy[set][g], geom = L.geom[r], q = L.qmv[r], pat = L.pat[r];
                const int lcw = geom & 3, lch = (geom >> 2) & 3, cw = 4 << lcw, rows = (4 << lch) + 9, ne = (rows + 1) >> 1;
                const int tall = (geom >> 12) & 1, metric = (geom >> 13) & 1, step = pat & 3;
                const int lns = lcw + (lch == 2 ? 1 : 0); // log2(strips per cell)
                const int rem = i - first, c = rem >> lns, st = rem & ((1 << lns) - 1);
                const int sx = st & ((1 << lcw) - 1), sy = st >> lcw;
                const int cand = c + 1 - ((pat >> 8) & 1); // index into square1: 0 is the centre
                const int kj = k_sq_kj[cand], k = kj & 3, j = kj >> 2;
                const int cy = q >> 16;
                const int qy = rf_s16(cy + (j - 1) * step), fy = qy & 3;
                const int rb_row = (rf_s16(cy - step) >> 2) - 3;
                const int t0 = (qy >> 2) - 3 - rb_row + (sy << (2 + tall)); // plane row of output row 0's first tap
                const uint32_t *const plane = L.plane + L.s_plane[set][g] + k * rows * cw + 4 * sx;
                const int pa = t0 >> 1, odd = t0 & 1;
                // rows y = 2u take pairs la[u .. u+3], rows y = 2u + 1 take lb[u .. u+3]
                const uint32_t *const la = plane + ((odd ? ne : 0) + pa) * cw;
                const uint32_t *const lb = plane + ((odd ? 0 : ne) + pa + odd) * cw;
                const uint4 vt = *(const uint4 *)&k_vtaps16[fy][0];
                constexpr int seed = 16 * (2048 + (8192 << 6));
                const int fstride = L.fstride[r];
                const uint8_t *f = L.fenc[r] + (long)((xy >> 16) + (sy << (2 + tall))) * fstride + (xy & 0xffff) + 4 * sx;
                int v = 0;
                auto rows4 = [&](int u0, uint32_t P[4][2]) {
                    // output rows 2 u0 .. 2 u0 + 3 of the strip
                    uint4 wa[5], wb[5];
#pragma unroll
                    for (int z = 0; z < 5; z++)
                    {
                        wa[z] = *(const uint4 *)(la + (u0 + z) * cw);
                        wb[z] = *(const uint4 *)(lb + (u0 + z) * cw);
                    }
#pragma unroll
                    for (int y = 0; y < 4; y++)
                    {
                        const uint4 *w = (y & 1) ? wb : wa;
                        const int u = y >> 1;
                        int s[4];
                        s[0] = sdot2(w[u + 3].x, vt.w, sdot2(w[u + 2].x, vt.z, sdot2(w[u + 1].x, vt.y, sdot2_first<seed>(w[u].x, vt.x))));
                        s[1] = sdot2(w[u + 3].y, vt.w, sdot2(w[u + 2].y, vt.z, sdot2(w[u + 1].y, vt.y, sdot2_first<seed>(w[u].y, vt.x))));
                        s[2] = sdot2(w[u + 3].z, vt.w, sdot2(w[u + 2].z, vt.z, sdot2(w[u + 1].z, vt.y, sdot2_first<seed>(w[u].z, vt.x))));
                        s[3] = sdot2(w[u + 3].w, vt.w, sdot2(w[u + 2].w, vt.z, sdot2(w[u + 1].w, vt.y, sdot2_first<seed>(w[u].w, vt.x))));
                        // upper halves = sums >> 12 (taps were x 16), as int16 pairs; then the clamp to 0..255
                        const s16x2 lo = { 0, 0 }, hi255 = { 255, 255 };
                        s16x2 p0 = as_s2(__builtin_amdgcn_perm((uint32_t)s[1], (uint32_t)s[0], 0x07060302));
                        s16x2 p1 = as_s2(__builtin_amdgcn_perm((uint32_t)s[3], (uint32_t)s[2], 0x07060302));
                        p0 = __builtin_elementwise_min(__builtin_elementwise_max(p0, lo), hi255);
                        p1 = __builtin_elementwise_min(__builtin_elementwise_max(p1, lo), hi255);
                        P[y][0] = as_u32(p0);
                        P[y][1] = as_u32(p1);
                    }
                };
                {
                    uint32_t P[4][2];
                    rows4(0, P);
                    v = tile_metric<true>(f, fstride, P, metric);
                }
                if (tall)
                {
                    uint32_t P[4][2];
                    rows4(2, P);
                    v += tile_metric<true>(f + 4 * fstride, fstride, P, metric);
                }
                atomicAdd(&L.cost[r * X265_HIP_RING_COSTS + cand], v);
            }
            if (!more)
                break;
            G = GN;
            set ^= 1;
        }
        wave_sync();
        // ---- results: nine per request, INT_MAX where the pattern has no candidate ---------------------------------
        for (int i = lane; i < rb * X265_HIP_RING_COSTS; i += 64)
        {
            const int r = i / X265_HIP_RING_COSTS, k = i - r * X265_HIP_RING_COSTS;
            if (base + r < n)
            {
                const int pat = L.pat[r], centre = (pat >> 8) & 1, ncand = (pat >> 4) & 15;
                const bool used = k == 0 ? centre != 0 : k <= ncand - centre;
                out[(long)(base + r) * X265_HIP_RING_COSTS + k] = used ? L.cost[i] : 0x7fffffff;
            }
        }
        wave_sync(); // the next batch rewrites this wave's record
        b = nb;
    }
}

void launch_refine(const DeviceTables *tabs, const x265_hip_refine_req *reqs, int n, int32_t *out, uint32_t *counters, int parity, hipStream_t st)
{
    if (n <= 0)
        return;
    int rb = RF_BATCH;
    while (rb > 1 && (n + rb - 1) / rb < 2048)
        rb >>= 1;
    int blocks = (n + 3) / 4;
    if (blocks > RF_MAX_BLOCKS)
        blocks = RF_MAX_BLOCKS;
    hipLaunchKernelGGL(refine_kernel, dim3(blocks), dim3(256), 0, st, tabs, reqs, n, out, counters, parity, rb);
}

} // namespace x265hip
