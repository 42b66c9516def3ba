// me_team.hip -- MotionEstimate::motionEstimate for small blocks, several calls per wavefront in lockstep.
//
// Reference: encoder/motion.cpp:526-1164, the diamond and hexagon searches (:605-700) with the pre-tests (:551-598)
// and the sub-pel stage (:1039-1119, workload[] :47-57) of a full-resolution reference.
//
// Why.  me_kernel gives a call a whole wave: a round of the search (pre-tests, a hexagon step, the square, a sub-pel
// ring) is a pass of the 64 lanes over (candidate, 4x4 tile) items plus ~165 VALU and ~200 SALU instructions of queue,
// arg-min and state code, and for an 8x8 block (four tiles) that bookkeeping is most of the call: 1200 VALU
// instructions per 8x8 call, of which the arithmetic on pixels is a third.  Here a call gets a *team*
// of L lanes, one lane per 4x4 tile of the block (L = 4: 8x8 and smaller, sixteen calls per wave; L = 16: up to
// 16x16, four calls per wave), and the wave runs the searches of its teams in lockstep:
//   * every lane keeps its tile of the source block in four registers for the whole call (no LDS copy of the block);
//   * a round evaluates the teams' candidate lists one candidate per pass: lane = tile, so a pass is one tile
//     comparison per lane, the tile sums meet in two (L = 4) or four (L = 16) DPP steps, and every lane of the team
//     ends up with the candidate's cost -- no LDS atomics, no queue of costs, the strict-less update in visiting order
//     (COPYn_IF_LT) is a running minimum of (cost << 4 | index);
//   * all search state is per team and lives in vector registers (uniform inside a team), so the state code runs
//     once per round for all teams of the wave; teams in different states take their branches one after the other.
// Teams wait for each other once, between the integer stage and the sub-pel stage, so that the passes of a round run
// one kind of tile arithmetic (full-pel SAD, or the filters) instead of both.
//   * Reference pixels come from a window in LDS.  With this little arithmetic per load the first version (every
//     candidate's rows fetched from the plane) was bound by the vector memory path: one 64-byte line per clock and CU,
//     and a 4x4 tile row is a line of its own -- 11 k line look-ups per sixteen 8x8 calls, 70 us per chunk where the
//     VALU work is 10 us (profiles/r03_team_pmc.txt).  Now a team keeps the (w + 16) x (h + 16) samples around the
//     centre of its search in LDS (24 x 24 bytes for an 8x8 block, 32 x 32 for 16x16), loaded with 12- / 16-byte loads,
//     two per row; the candidates of the hexagon, square and sub-pel rounds -- within +-2 samples of a centre that moves
//     by at most 2 per round -- read their tile rows from there (aligned dwords + v_alignbyte: unaligned LDS reads
//     work but stall, me_device.h), and the window is loaded again only when a round's candidates (with the filters'
//     margins) no longer fit it.  The pre-tests' candidates lie anywhere and are read from the plane.
// Results are those of me_kernel bit for bit (same candidates, same order, same ties); tests/test_gpu_parity.py and
// the encoder captures run through this kernel whenever a list has such jobs (me_job_class, me_shared.h).
#include <stddef.h>
#include <type_traits>

#include "me_shared.h"

namespace x265hip {

#define TQ 16 // candidates per round at most: 3 + X265_HIP_MAX_MVC pre-tests
static_assert(3 + X265_HIP_MAX_MVC <= TQ, "the pre-test list must fit a round");
static_assert(sizeof(x265_hip_me_job) == 80 && offsetof(x265_hip_me_job, pu_x) == 8 && offsetof(x265_hip_me_job, pu_w) == 12 &&
                  offsetof(x265_hip_me_job, mvmin_x) == 16 && offsetof(x265_hip_me_job, qmvp_x) == 24 && offsetof(x265_hip_me_job, merange) == 28 &&
                  offsetof(x265_hip_me_job, qp) == 30 && offsetof(x265_hip_me_job, num_mvc) == 31 && offsetof(x265_hip_me_job, mvc) == 32,
              "team_chunk reads the job record as five 16-byte words");

enum TeamState { TS_PRE, TS_DIA, TS_HEX1, TS_HEXN, TS_SQ, TS_POST_WAIT, TS_HPEL, TS_QPEL, TS_DONE };

#ifndef TEAM_AH1
#define TEAM_AH1 11 // rows in flight in the two-stage filter of the pre-tests (from the plane), single tiles / strips
#endif
#ifndef TEAM_AH2
#define TEAM_AH2 0
#endif
#define TEAM_RY 6 // rows of the window above and below the block at its centre position
// A class of blocks: L lanes per team, a lane owns a cell of CW x CH 4x4 tiles of the block (its source pixels stay in
// registers for the whole call); the window of a team is ROWS rows of PITCH bytes, loaded in units of UNIT bytes.
//   TeamT4   blocks up to 8x8:   4 lanes,  cell = one tile            (16 calls per wave)
//   TeamT8   blocks up to 16x16: 16 lanes, cell = one tile            ( 4 calls per wave).  With TEAM_T8_STRIPS: 8 lanes,
//            cell = a strip of 4 x 8 (15 filtered rows per 8 rows of output instead of 2 x 11), which needs the registers
//            of two waves per SIMD and measured slower
//   TeamT32  blocks up to 32x32: 16 lanes, cell = 8 x 8 (two strips)  ( 4 calls per wave)
// PREWIN: the window is loaded around the predictor in front of the pre-tests (measured on the lists of a 4K picture x 3
// references: 8x8 0.347 -> 0.339 ms, 16x16 0.303 -> 0.322 ms, 32x32 the same: on for the smallest class only)
struct TeamT4
{
    enum { L = 4, CW = 1, CH = 1, PITCH = 24, ROWS = 8 + 2 * TEAM_RY, UNIT = 12, CLS = MC_T4, PREWIN = 1 };
};
#ifndef TEAM_T8_STRIPS
#define TEAM_T8_STRIPS 0 // measured: strips at two waves per SIMD 0.323 ms for the 16x16 calls of a 4K picture, tiles at three 0.294
#endif
#if TEAM_T8_STRIPS
struct TeamT8
{
    enum { L = 8, CW = 1, CH = 2, PITCH = 32, ROWS = 16 + 2 * TEAM_RY, UNIT = 16, CLS = MC_T8, PREWIN = 0 };
};
#else
struct TeamT8 // (16 lanes, a tile each)
{
    enum { L = 16, CW = 1, CH = 1, PITCH = 32, ROWS = 16 + 2 * TEAM_RY, UNIT = 16, CLS = MC_T8, PREWIN = 0 };
};
#endif
struct TeamT32
{
    enum { L = 16, CW = 2, CH = 2, PITCH = 48, ROWS = 32 + 2 * TEAM_RY, UNIT = 16, CLS = MC_T32, PREWIN = 0 };
};
template <typename T>
struct __attribute__((aligned(16))) TeamLds
{
    int cand[64 / T::L][TQ]; // candidate positions of the round, quarter-pel, (y << 16) | (x & 0xffff)
    int cost[64 / T::L][TQ]; // the pre-tests' costs (the other rounds keep a running minimum)
    uint8_t win[64 / T::L][T::PITCH * T::ROWS];
    uint8_t pad[16]; // a 12-byte row read at the window's last bytes runs into this
};

__device__ __forceinline__ int pack_mv(int x, int y) { return (int)(((uint32_t)y << 16) | (uint32_t)(uint16_t)x); }

// sum over the L lanes of a team; every lane gets it
template <int L>
__device__ __forceinline__ int team_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false); // quad_perm 1,0,3,2
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false); // quad_perm 2,3,0,1
    if (L >= 8)
        v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false); // row_half_mirror
    if (L >= 16)
        v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false); // row_mirror
    return v;
}

// SAD or SATD of this lane's source cell (CW x CH tiles, f[cx][row]) against the reference at `r` (rows `stride` apart;
// PT: plane memory or the LDS window, whose pitch is a constant) at fraction (xf, yf).  One path per pass of the wave,
// the cheapest that serves all its lanes.  A single tile (CW = CH = 1): full-pel, one-dimensional filter (a full-pel
// position goes through it with the tap row {0,0,0,64,...}; ft: the source tile by columns), two-stage filter
// (ditto).  Larger cells: strips of CH tiles through ref_strip (full-pel / horizontal / vertical / two-stage, the last
// for all lanes of a pass as soon as one of them needs it).
template <typename PT, int AH, int CW, int CH>
__device__ __forceinline__ int team_cell(const uint8_t *r, int stride, int xf, int yf, const uint32_t (&f)[CW][4 * CH], const uint32_t ft[4], int metric)
{
    const bool anyfrac = __ballot((xf | yf) != 0) != 0, anydiag = __ballot(xf != 0 && yf != 0) != 0;
    if (!anyfrac)
    {
        int v = 0;
#pragma unroll
        for (int cx = 0; cx < CW; cx++)
        {
            uint32_t rr[4 * CH];
#pragma unroll
            for (int y = 0; y < 4 * CH; y++)
                rr[y] = ld4g((PT)r + (y * stride + 4 * cx));
            if (metric == X265_HIP_SAD)
            {
                uint32_t a = 0;
#pragma unroll
                for (int y = 0; y < 4 * CH; y++)
                    a = __builtin_amdgcn_sad_u8(f[cx][y], rr[y], a);
                v += (int)a;
            }
            else
            {
#pragma unroll
                for (int t = 0; t < CH; t++)
                {
                    uint32_t pix[4][2];
#pragma unroll
                    for (int y = 0; y < 4; y++)
                    {
                        pix[y][0] = __builtin_amdgcn_perm(0, rr[4 * t + y], 0x0c010c00);
                        pix[y][1] = __builtin_amdgcn_perm(0, rr[4 * t + y], 0x0c030c02);
                    }
                    v += tile_metric_rows(&f[cx][4 * t], pix, X265_HIP_SATD);
                }
            }
        }
        return v;
    }
    if constexpr (CW == 1 && CH == 1)
    {
        if (!anydiag)
        {
            uint32_t pix[4][2], fs[4];
            ref_tile_1d<PT>(r, stride, xf, yf, pix); // a vertical position comes back transposed
#pragma unroll
            for (int y = 0; y < 4; y++)
                fs[y] = xf == 0 ? ft[y] : f[0][y];
            return tile_metric_rows(fs, pix, metric);
        }
    }
    int v = 0;
#pragma unroll
    for (int cx = 0; cx < CW; cx++)
    {
        uint32_t pix[4 * CH][2];
        ref_strip<CH, false, AH, PT>(r + 4 * cx, stride, xf, yf, pix, anydiag);
#pragma unroll
        for (int t = 0; t < CH; t++)
            v += tile_metric_rows(&f[cx][4 * t], &pix[4 * t], metric);
    }
    return v;
}

// The searches of one chunk of a part's list: entries chunk * G .. chunk * G + G - 1, one per team.
template <typename T>
__device__ __forceinline__ void team_chunk(TeamLds<T> &lds, const int *pat, const DeviceTables *tabs, const x265_hip_me_job *jobs, const uint32_t *list,
                                           uint32_t count, uint32_t chunk, x265_hip_me_result *out, int out_stride)
{
    constexpr int L = T::L, CW = T::CW, CH = T::CH, G = 64 / L, PITCH = T::PITCH, UNIT = T::UNIT;
    const int lane = threadIdx.x & 63, team = lane / L, tl = lane % L;
    const uint32_t e = chunk * G + team;
    const bool has_job = e < count;
    // a team without a job (the ragged end of a list) goes through the set-up with the chunk's first job, so that all
    // its addresses are valid ones, never has a candidate and writes no result
    const uint32_t id = list[has_job ? e : chunk * G];
    const gptr_t jp = as_global((const uint8_t *)(jobs + id));
    const uint4 h0 = ld16g(jp), h1 = ld16g(jp + 16);
    const uint4 m0 = ld16g(jp + 32), m1 = ld16g(jp + 48), m2 = ld16g(jp + 64);
    const int pu_x = (int)(int16_t)h0.z, pu_y = (int)h0.z >> 16;
    const int w = (int)(h0.w & 0xff), h = (int)((h0.w >> 8) & 0xff), method = (int)((h0.w >> 16) & 0xff), subme = (int)(h0.w >> 24);
    const int lox = (int)(int16_t)h1.x, loy = (int)h1.x >> 16, hix = (int)(int16_t)h1.y, hiy = (int)h1.y >> 16; // full-pel, inclusive
    const int mvpx = (int)(int16_t)h1.z, mvpy = (int)h1.z >> 16;
    const int merange = (int)(int16_t)h1.w, qp = (int)((h1.w >> 16) & 0xff), nmvc = (int)(h1.w >> 24);
    const PlaneDesc fp = tabs->planes[(int)h0.x], rp = tabs->planes[(int)h0.y];
    const uint16_t *const tab = tabs->costtab[qp];
    const int stride = rp.stride;

    // this lane's cell
    const int cells_x = w / (4 * CW), ncell = cells_x * (h / (4 * CH));
    const bool has_tile = tl < ncell;
    const int tu = has_tile ? tl : 0;
    const int ty = (int)(((float)tu + 0.5f) * __builtin_amdgcn_rcpf((float)cells_x)), tx = tu - ty * cells_x; // exact: tu < 16, cells_x <= 16
    const int cell_x = 4 * CW * tx, cell_y = 4 * CH * ty;
    uint32_t frow[CW][4 * CH], ft[4]; // the source pixels (setSourcePU, motion.cpp:110-126); a single tile also by columns
    {
        const gptr_t f = as_global(fp.origin) + ((long)(pu_y + cell_y) * fp.stride + pu_x + cell_x);
#pragma unroll
        for (int cx = 0; cx < CW; cx++)
#pragma unroll
            for (int y = 0; y < 4 * CH; y++)
                frow[cx][y] = ld4g(f + (y * fp.stride + 4 * cx));
#pragma unroll
        for (int y = 0; y < 4; y++)
            ft[y] = frow[0][y];
        transpose4x4_u8(ft);
    }
    const uint8_t *const pbase = rp.origin + ((long)pu_y * stride + pu_x); // the block's position in the reference plane
    const int toff = cell_y * stride + cell_x;                             // this lane's cell inside the block, plane
    const int ltoff = cell_y * PITCH + cell_x;                             // and window

    // workload[subme] (motion.cpp:47-57)
    const int wsh = 4 * (subme & 7);
    const int hpel_iters = (int)(workload_column(0) >> wsh) & 15, hpel_dirs = (int)(workload_column(1) >> wsh) & 15;
    const int qpel_iters = (int)(workload_column(2) >> wsh) & 15, qpel_dirs = (int)(workload_column(3) >> wsh) & 15;
    const int hsatd = (int)(workload_column(4) >> wsh) & 15;

#define IN_RANGE(x, y) ((x) >= lox && (x) <= hix && (y) >= loy && (y) <= hiy)
    // clipped quarter-pel predictor (motion.cpp:552; MV::clipped, mv.h:89-94) and its full-pel rounding (mv.h:76)
    const int qlox = s16(lox << 2), qloy = s16(loy << 2), qhix = s16(hix << 2), qhiy = s16(hiy << 2);
    const int pmx = max(min(mvpx, qhix), qlox), pmy = max(min(mvpy, qhiy), qloy);
    const int fpx = s16(s16(pmx + 2) >> 2), fpy = s16(s16(pmy + 2) >> 2);
    const bool sub = ((pmx | pmy) & 3) != 0, nz = (pmx | pmy) != 0;

    // ---- the window.  It holds the samples [wx, wx + PITCH) x [wy, wy + h + 2 TEAM_RY) relative to the block's
    // position; loads stay inside the samples the job may touch at all (what the host checks of a job, check_job in
    // api.hip: the search range and the zero vector, plus the steps the search may take outside them, plus the filters'
    // margins): a window at the edge of that region is shifted sideways into it and repeats its outermost rows instead
    // of reading outside the plane's buffer; no candidate of the search uses the repeated rows.
    const int wrows = h + 2 * TEAM_RY, rx = (PITCH - w) >> 1;
    int wx = 0, wy = 0; // its origin
    bool wok = false; // the window holds something
    const int over = (method == X265_HIP_DIA ? 1 : 3) + (int)((0x22122111u >> wsh) & 15); // k_int_over + (k_qsteps + 3) / 4
    const int gx0 = min(lox, 0) - over - 6, gx1 = max(hix, 0) + over + w + 16 - PITCH; // first / last start of a window row
    const int gy0 = min(loy, 0) - over - 3, gy1 = max(hiy, 0) + over + h + 3;          // first / last row
    auto load_window = [&](int ccx, int ccy, bool mine) {
        // units u = UPR row + half, dealt to the lanes of the team; a lane's loads are in flight together, half of them
        // at a time (registers)
        constexpr int UPR = PITCH / UNIT, NU = (UPR * T::ROWS + L - 1) / L, NH = (NU + 1) / 2;
        // (the region is at least PITCH wide for the blocks of a class: w + 26 samples, me_job_class)
        const int nwx = min(max(ccx - rx, gx0), gx1), nwy = ccy - TEAM_RY;
#pragma unroll
        for (int k0 = 0; k0 < NU; k0 += NH)
        {
            uint32_t v[NH][4];
#pragma unroll
            for (int k = 0; k < NH; k++)
            {
                const int u = tl + (k0 + k) * L, row = u / UPR, half = u - row * UPR;
                const int ax = nwx + half * UNIT, ay = min(max(nwy + row, gy0), gy1);
                // every lane loads (a lane without a unit, or of a team that keeps its window, from a valid address
                // all the same): behind a branch per unit the loads waited for each other
                const gptr_t src = as_global(pbase) + (ay * stride + ax);
                if (UNIT == 12)
                {
                    ld12g(src, v[k][0], v[k][1], v[k][2]);
                    v[k][3] = 0;
                }
                else
                {
                    const uint4 q = ld16g(src);
                    v[k][0] = q.x, v[k][1] = q.y, v[k][2] = q.z, v[k][3] = q.w;
                }
            }
#pragma unroll
            for (int k = 0; k < NH; k++)
            {
                const int u = tl + (k0 + k) * L, row = u / UPR, half = u - row * UPR;
                if (mine && k0 + k < NU && row < wrows)
                {
                    __attribute__((address_space(3))) uint32_t *dst = (__attribute__((address_space(3))) uint32_t *)&lds.win[team][row * PITCH + half * UNIT];
#pragma unroll
                    for (int i = 0; i < UNIT / 4; i++)
                        dst[i] = v[k][i];
                }
            }
        }
        if (mine)
        {
            wx = nwx;
            wy = nwy;
            wok = true;
        }
    };
    // whether the candidates of a ring around (ccx, ccy) (full-pel) with integer parts within `reach` of it and the
    // margins (ml, mr) of their footprints lie in the window
    auto fits = [&](int ccx, int ccy, int reach, int ml, int mr) {
        return wok && ccx - reach - ml >= wx && ccx + reach + w + mr <= wx + PITCH && ccy - reach - ml >= wy && ccy + reach + h + mr <= wy + wrows;
    };

    // ---- the pre-tests' list (motion.cpp:551-598).  Slots are by kind, so that the passes of the round are of one
    // kind for all teams: 0 the predictor (no mv rate), 1 its full-pel rounding (if the predictor is fractional), 2 .. the
    // candidate vectors that are neither zero nor the predictor, in list order ("m != bestpre", :584, is resolved when the
    // costs are consumed), and last, in slot TQ - 1, the zero vector (if the predictor is not zero): the others mostly
    // lie near the predictor and share a window, the zero vector needs one of its own
    constexpr int ZSLOT = 2;
        uint32_t vmask = has_job ? (1u | (sub ? 2u : 0u) | (nz ? 1u << ZSLOT : 0u)) : 0u; // the round's candidates, a bit per slot
    if (tl == 0)
    {
        lds.cand[team][0] = pack_mv(pmx, pmy);
        lds.cand[team][1] = pack_mv(s16(fpx << 2), s16(fpy << 2));
        lds.cand[team][ZSLOT] = 0;
    }
    int npre = 3; // first slot behind the candidate vectors
    {
        const uint32_t mw[X265_HIP_MAX_MVC] = { m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w, m2.x, m2.y, m2.z, m2.w };
#pragma unroll
        for (int i = 0; i < X265_HIP_MAX_MVC; i++)
            if (i < nmvc)
            {
                const int mx = max(min((int)(int16_t)mw[i], qhix), qlox), my = max(min((int)mw[i] >> 16, qhiy), qloy);
                if ((mx | my) && !(mx == pmx && my == pmy))
                {
                    if (tl == 0)
                        lds.cand[team][npre] = pack_mv(mx, my);
                    if (has_job)
                        vmask |= 1u << npre;
                    npre++;
                }
            }
    }

    int st = has_job ? TS_PRE : TS_DONE;
    int bcost = 0, bx = 0, by = 0; // best so far: full-pel during the integer stage, quarter-pel after it
    int bprecost = 0, bp = pack_mv(pmx, pmy); // bestpre
    int cnt = 0, dir = 0, sub_it = 0, metric = X265_HIP_SAD;
    bool cf = false; // the round's first candidate re-measures the centre: its cost replaces bcost (motion.cpp:1082-1086, 1103-1105)
    bool need_win = false; // the next round's candidates do not fit the window: load it around (ncx, ncy)
    int ncx = 0, ncy = 0;

    // the next round: [the centre,] centre + pat[first + i] * mul for i < count, quarter-pel after << sh (int16 wrap
    // as in the reference's MV arithmetic); lanes of the team write the slots
    auto ring = [&](int cx, int cy, int first, int count, int mul, int sh, bool centre) {
        const int off = centre ? 1 : 0;
        if (centre && tl == 0)
            lds.cand[team][0] = pack_mv(cx, cy);
        for (int i = tl; i < count; i += L)
        {
            const int p = pat[first + i];
            lds.cand[team][off + i] = pack_mv(s16(s16(cx + pat_dx(p) * mul) << sh), s16(s16(cy + pat_dy(p) * mul) << sh));
        }
        vmask = (1u << (count + off)) - 1u;
        cf = centre;
        // integer rings: points within 2 of the centre, no margins; sub-pel rings: integer parts within 1 of the
        // centre's, the filters' 3 samples before and 4 behind (a 12-byte row read takes one more)
        ncx = sh ? cx : cx >> 2;
        ncy = sh ? cy : cy >> 2;
        need_win = sh ? !fits(ncx, ncy, 2, 0, 0) : !fits(ncx, ncy, 1, 3, 4);
    };

    if (T::PREWIN)
    {
        // The window is loaded around the predictor's full-pel rounding before the pre-tests: the predictor, its rounding
        // and the candidate vectors near them are measured out of it (the others, the zero vector mostly, from the
        // plane), and the integer stage starts in it when the predictor wins.
        need_win = has_job;
        ncx = fpx;
        ncy = fpy;
    }
    int guard = 0;
    while (__ballot(st != TS_DONE) != 0 && ++guard < 8192)
    {
        if (__ballot(need_win && vmask != 0u) != 0)
            load_window(ncx, ncy, need_win && vmask != 0u);
        need_win = false;
        wave_sync(); // the lists, the windows
        // ---- evaluate: pass c takes candidate c of every team that has one -------------------------------------------
        int key = 0x7fffffff; // min over the round of (cost << 4) | index: the first minimum in visiting order
        const bool pre = st == TS_PRE;
        typedef const __attribute__((address_space(3))) uint8_t *lptr_t;
        // BitCost::mvcost (bitcost.h:45)
        auto mv_rate = [&](int qx, int qy) { return (int)(uint16_t)(tab_at(tab, mvpx, qx) + tab_at(tab, mvpy, qy)); };
        if (__ballot(vmask != 0u && !(st >= TS_DIA && st <= TS_SQ)) == 0)
        {
            // A round of the integer stage for every team that has candidates: full-pel positions, SAD, out of the
            // windows.  Straight-line code over groups of four slots; slots without a candidate read the window's first
            // bytes and are not used.
            const lptr_t lbase = (lptr_t)&lds.win[team][0] + (ltoff - wy * PITCH - wx); // this lane's cell at the zero vector, were it in the window
#pragma unroll
            for (int g = 0; g < 2; g++)
            {
                if (g && __ballot((vmask >> 4) != 0u) == 0)
                    break;
                typedef int i32x4 __attribute__((ext_vector_type(4)));
                const i32x4 pv = *(const i32x4 *)&lds.cand[team][4 * g];
                const int pos[4] = { pv.x, pv.y, pv.z, pv.w };
                // the rates are loads of device memory: all in flight before the first is used
                int rate[4];
                if constexpr (CW * CH > 1)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        rate[j] = mv_rate((int)(int16_t)pos[j], pos[j] >> 16);
                if constexpr (CW * CH == 1)
                {
                    // single tiles: the four candidates' rows are all asked for before the first is used
                    uint32_t rr[4][4];
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        const bool act = ((vmask >> (4 * g + j)) & 1u) != 0;
                        const int qx = (int)(int16_t)pos[j], qy = pos[j] >> 16;
                        rate[j] = mv_rate(qx, qy);
                        const ldsa_t r((const uint8_t *)(act ? lbase + ((qy >> 2) * PITCH + (qx >> 2)) : (lptr_t)&lds.win[team][0]));
#pragma unroll
                        for (int y = 0; y < 4; y++)
                            rr[j][y] = ld4g(r + y * PITCH);
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        uint32_t a = 0;
#pragma unroll
                        for (int y = 0; y < 4; y++)
                            a = __builtin_amdgcn_sad_u8(frow[0][y], rr[j][y], a);
                        const int cost = team_sum<L>(has_tile ? (int)a : 0) + rate[j];
                        if ((vmask >> (4 * g + j)) & 1u)
                            key = min(key, (cost << 4) | (4 * g + j));
                    }
                }
                else
#pragma unroll
                for (int j = 0; j < 4; j++)
                {
                    const bool act = ((vmask >> (4 * g + j)) & 1u) != 0;
                    const int qx = (int)(int16_t)pos[j], qy = pos[j] >> 16;
                    const ldsa_t r((const uint8_t *)(act ? lbase + ((qy >> 2) * PITCH + (qx >> 2)) : (lptr_t)&lds.win[team][0]));
                    uint32_t a = 0;
#pragma unroll
                    for (int cx = 0; cx < CW; cx++)
#pragma unroll
                        for (int y = 0; y < 4 * CH; y++)
                            a = __builtin_amdgcn_sad_u8(frow[cx][y], ld4g(r + (y * PITCH + 4 * cx)), a);
                    const int cost = team_sum<L>(has_tile ? (int)a : 0) + rate[j];
                    if (act)
                        key = min(key, (cost << 4) | (4 * g + j));
                    __builtin_amdgcn_sched_barrier(0); // (one candidate's window reads at a time: all four at once is 64 .. 128 registers)
                }
            }
        }
        else
        for (int c = 0; c < TQ; c++)
        {
            if (__ballot((vmask >> c) != 0u) == 0)
                break;
            const bool act = ((vmask >> c) & 1u) != 0;
            if (__ballot(act) == 0)
                continue;
            const int pos = act ? lds.cand[team][c] : 0;
            // (all teams of a wave are in the pre-tests together) one path per pass: with PREWIN the plane's unless every
            // team's candidate of this slot lies in its window with the filters' margins
            const bool from_plane = __ballot(pre) != 0 && (!T::PREWIN || __ballot(act && !fits((int)(int16_t)pos >> 2, pos >> 18, 0, 3, 4)) != 0);
            int v = 0, seed = 0;
            if (act)
            {
                const int qx = (int)(int16_t)pos, qy = pos >> 16;
                const int rate = mv_rate(qx, qy);
                if (from_plane)
                    v = team_cell<gptr_t, CH == 1 ? TEAM_AH1 : TEAM_AH2, CW, CH>(pbase + (toff + (qy >> 2) * stride + (qx >> 2)), stride, qx & 3, qy & 3, frow, ft, metric);
                else
                {
                    const lptr_t lbase = (lptr_t)&lds.win[team][0] + (ltoff - wy * PITCH - wx);
                    v = team_cell<ldsa_t, 0, CW, CH>((const uint8_t *)(lbase + ((qy >> 2) * PITCH + (qx >> 2))), PITCH, qx & 3, qy & 3, frow, ft, metric);
                }
                if (!has_tile)
                    v = 0;
                seed = (pre && c == 0) ? 0 : rate; // the predictor's own test has no rate (:559)
            }
            const int cost = team_sum<L>(v) + seed;
            if (act)
            {
                if (pre)
                {
                    if (tl == 0)
                        lds.cost[team][c] = cost;
                }
                else if (c == 0 && cf)
                    bcost = cost;
                else
                    key = min(key, (cost << 4) | (c - (cf ? 1 : 0)));
            }
        }
        wave_sync(); // the pre-tests' costs

        // ---- the teams' state code -----------------------------------------------------------------------------------
        const bool better = (key >> 4) < bcost; // the round's first minimum beats the centre (strict)
        const int kk = key & 15;
        if (st == TS_PRE)
        {
            bprecost = lds.cost[team][0];
            bx = fpx;
            by = fpy;
            bcost = bprecost;
            if (sub)
                bcost = lds.cost[team][1]; // full-pel start re-measured with its mv rate (:562-567)
            if (nz)
            {
                const int cz = lds.cost[team][ZSLOT]; // zero vector (:570-578)
                if (cz < bcost)
                {
                    bcost = cz;
                    bx = by = 0;
                }
            }
            const int nslot = 32 - __clz((int)vmask);
            for (int k = 3; k < nslot; k++)
            {
                const int pos = lds.cand[team][k];
                if (pos == bp)
                    continue;
                const int cc = lds.cost[team][k];
                if (cc < bprecost) // candidates compete with bprecost only (:592)
                {
                    bprecost = cc;
                    bp = pos;
                }
            }
            if (method == X265_HIP_DIA)
            {
                cnt = merange;
                ring(bx, by, PAT_SQ + 1, 4, 1, 2, false); // (0,-1) (0,1) (-1,0) (1,0)
                st = TS_DIA;
            }
            else
            {
                ring(bx, by, PAT_HEX + 1, 6, 1, 2, false); // hex2[1..6]: directions 0..5
                st = TS_HEX1;
            }
        }
        else if (st == TS_DIA)
        {
            // radius-1 diamond, the centre wins ties (motion.cpp:605-626)
            bool go_on = false;
            if (better)
            {
                bcost = key >> 4;
                const int p = pat[PAT_SQ + 1 + kk];
                bx = s16(bx + pat_dx(p));
                by = s16(by + pat_dy(p));
                go_on = --cnt != 0 && IN_RANGE(bx, by);
            }
            if (go_on)
                ring(bx, by, PAT_SQ + 1, 4, 1, 2, false);
            else
            {
                st = TS_POST_WAIT;
                vmask = 0;
            }
        }
        else if (st == TS_HEX1 || st == TS_HEXN)
        {
            // radius-2 hexagon, then half hexagons in the direction of the move (motion.cpp:628-683)
            bool go_on = false;
            if (better)
            {
                bcost = key >> 4;
                if (st == TS_HEX1)
                {
                    dir = kk;
                    cnt = (merange >> 1) - 1;
                }
                else
                {
                    dir += kk - 1;
                    dir = dir < 0 ? dir + 6 : (dir >= 6 ? dir - 6 : dir); // mod6m1
                    cnt--;
                }
                const int p = pat[PAT_HEX + dir + 1];
                bx = s16(bx + pat_dx(p));
                by = s16(by + pat_dy(p));
                go_on = cnt > 0 && IN_RANGE(bx, by);
            }
            if (go_on)
            {
                ring(bx, by, PAT_HEX + dir, 3, 1, 2, false);
                st = TS_HEXN;
            }
            else
            {
                ring(bx, by, PAT_SQ + 1, 8, 1, 2, false); // square refine (:687-698)
                st = TS_SQ;
            }
        }
        else if (st == TS_SQ)
        {
            if (better)
            {
                bcost = key >> 4;
                const int p = pat[PAT_SQ + 1 + kk];
                bx = s16(bx + pat_dx(p));
                by = s16(by + pat_dy(p));
            }
            st = TS_POST_WAIT;
            vmask = 0;
        }
        else if (st == TS_HPEL || st == TS_QPEL)
        {
            // a sub-pel ring around a fixed centre, one move at its end (motion.cpp:1090-1118)
            if (better)
            {
                bcost = key >> 4;
                const int p = pat[PAT_SQ + 1 + kk], mul = st == TS_HPEL ? 2 : 1;
                bx = s16(bx + pat_dx(p) * mul);
                by = s16(by + pat_dy(p) * mul);
            }
            sub_it++;
            if (st == TS_HPEL)
            {
                if (sub_it < hpel_iters)
                    ring(bx, by, PAT_SQ + 1, hpel_dirs, 2, 0, false);
                else
                {
                    // half-pel used SAD: the winner is re-measured with SATD in front of the quarter-pel ring (:1103-1105),
                    // whose positions do not depend on that cost: one round
                    sub_it = 0;
                    metric = X265_HIP_SATD;
                    if (!hsatd || qpel_iters > 0)
                    {
                        ring(bx, by, PAT_SQ + 1, qpel_iters > 0 ? qpel_dirs : 0, 1, 0, !hsatd);
                        st = TS_QPEL;
                    }
                    else
                    {
                        st = TS_DONE;
                        vmask = 0;
                    }
                }
            }
            else if (sub_it < qpel_iters)
                ring(bx, by, PAT_SQ + 1, qpel_dirs, 1, 0, false);
            else
            {
                st = TS_DONE;
                vmask = 0;
            }
        }
        // after the integer stage (motion.cpp:1039-1088), once every team of the wave is there
        if (__ballot(st < TS_POST_WAIT) == 0 && st == TS_POST_WAIT)
        {
            if (bprecost < bcost) // pre-test winner bypasses the integer result
            {
                bx = (int)(int16_t)bp;
                by = bp >> 16;
                bcost = bprecost;
            }
            else
            {
                bx = s16(bx << 2);
                by = s16(by << 2);
            }
            if (!bcost)
            {
                bcost = mvcost_q(tab, mvpx, mvpy, bx, by); // zero residual: rate only (:1049-1054)
                st = TS_DONE;
            }
            else
            {
                sub_it = 0;
                metric = hsatd ? X265_HIP_SATD : X265_HIP_SAD;
                // with hpel_satd the SATD re-measure of the centre heads the first ring (:1082-1086)
                ring(bx, by, PAT_SQ + 1, hpel_dirs, 2, 0, hsatd != 0);
                st = TS_HPEL;
            }
        }
    }
#undef IN_RANGE
    if (has_job && tl == 0)
    {
        x265_hip_me_result *o = (x265_hip_me_result *)((char *)out + (size_t)id * out_stride);
        o->cost = bcost;
        o->mvx = (int16_t)bx;
        o->mvy = (int16_t)by;
    }
    wave_sync(); // the next chunk rewrites the lists
}

template <typename T>
__device__ __forceinline__ void team_lists(TeamLds<T> &lds, const int *pat, const DeviceTables *tabs, const x265_hip_me_job *jobs, uint32_t *route,
                                           int parity, int cap, x265_hip_me_result *out, int out_stride)
{
    constexpr int G = 64 / T::L, cls = T::CLS;
    const int p = blockIdx.x & (ME_PARTS - 1), waves = gridDim.x / ME_PARTS;
    const uint32_t count = min(*route_append(route, parity, cls, p), (uint32_t)cap);
    const uint32_t nchunk = (count + G - 1) / G;
    const uint32_t *const list = route_list(route, cap, cls, p);
    uint32_t k = blockIdx.x / ME_PARTS;
    while (k < nchunk)
    {
        team_chunk<T>(lds, pat, tabs, jobs, list, count, k, out, out_stride);
        k = 0;
        if ((threadIdx.x & 63) == 0)
            k = (uint32_t)waves + atomicAdd(route_handout(route, parity, cls, p), 1u);
        k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
    }
}

// gridDim.x is a multiple of ME_PARTS; a wave works through its part of the list of class A, then of class B (one launch
// for two classes of equal register needs: the shorter calls of the second fill the end of the first), and leaves when
// the hand-out counters have passed the ends of the lists.  The class of 8x8 cells takes more registers per lane (its
// source cell and filtered rows) and has a kernel of its own at a lower occupancy.
template <typename A, typename B, int WAVES>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(WAVES))) void me_team_kernel(const DeviceTables *tabs, const x265_hip_me_job *jobs,
                                                                                            uint32_t *route, int parity, int cap,
                                                                                            x265_hip_me_result *out, int out_stride)
{
    __shared__ union TeamLdsBoth
    {
        TeamLds<A> a;
        TeamLds<B> b;
    } lds; // one list after the other
    __shared__ int pat[PAT_N];
    if (threadIdx.x < PAT_N)
        pat[threadIdx.x] = k_pat[threadIdx.x];
    wave_sync();
    team_lists<A>(lds.a, pat, tabs, jobs, route, parity, cap, out, out_stride);
    if constexpr (!std::is_same<A, B>::value)
        team_lists<B>(lds.b, pat, tabs, jobs, route, parity, cap, out, out_stride);
}

#ifndef ME_T4_WAVES
#define ME_T4_WAVES 3
#endif
#ifndef ME_T32_WAVES
#define ME_T32_WAVES 2
#endif

template <typename A, typename B, int WAVES>
static void launch_team_classes(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n, uint32_t *route, int parity, int cap, x265_hip_me_result *out,
                                int out_stride, hipStream_t st)
{
    constexpr int G = 64 / (A::L > B::L ? A::L : B::L);
    const int want = (n + ME_PARTS * G - 1) / (ME_PARTS * G), most = 256 * 4 * WAVES / ME_PARTS;
    const int grid = ME_PARTS * (want < 1 ? 1 : want > most ? most : want);
    hipLaunchKernelGGL((me_team_kernel<A, B, WAVES>), dim3(grid), dim3(64), 0, st, tabs, jobs, route, parity, cap, out, out_stride);
}

// The classes of single tiles first and the class of 8x8 cells behind them (measured on the ME-stage list of a 4K
// picture, profiles/r03_team_order.txt: 0.956 ms; the 8x8 cells first 1.003, on the side stream behind me_big_kernel
// 1.005, in me_rest_kernel instead 0.971): me_big_kernel runs beside the first of these kernels, and its waves of 128
// registers share a SIMD with waves of 168 better than with waves of 224.
void launch_me_team(const DeviceTables *tabs, const x265_hip_me_job *jobs, int n, uint32_t *route, int parity, int cap, x265_hip_me_result *out,
                    int out_stride, hipStream_t st, bool with_t32)
{
    launch_team_classes<TeamT8, TeamT4, ME_T4_WAVES>(tabs, jobs, n, route, parity, cap, out, out_stride, st);
    if (with_t32)
        launch_team_classes<TeamT32, TeamT32, ME_T32_WAVES>(tabs, jobs, n, route, parity, cap, out, out_stride, st);
}

} // namespace x265hip
