// tron_minimax.hip — the reference's depth-2 Minimax/Voronoi opponent (tron/minimax.py:57-278,
// MinimaxPlayer(2, "voronoi") at util.py:82-83 and ACKTR.py:13) for N boards at once.
//
// One wavefront per board, one lane per board ROW: every lane keeps its row as bit masks (bit c =
// column c), so the flood fills of get_shortest_path (minimax.py:63-86) become level-synchronous
// bitboard expansions — left/right are shifts, up/down one cross-lane shuffle each.  The reference
// walks an OrderedSet queue of (x, y, l) tuples; on a grid (bipartite) that queue never holds two
// different l for one cell, so it is a plain breadth-first search and the distances agree exactly
// (tests/test_gpu_minimax.py checks every fixture board and random soups against the literal
// oracle).  Both players' fills advance together and each new cell is classified on the spot:
//
//   get_voronoi_value (minimax.py:88-125), for the leaf map after my move a and the opponent's b:
//     +1 per empty cell I reach strictly sooner, -1 per cell the opponent reaches strictly sooner,
//     -1 per empty cell neither reaches (1 + 1 > 0), +1 per enemy-body cell (-3 + -3 < 0); my own
//     bodies (-2 == the opponent's start marker), walls and both start cells are skipped.
//
// The tree (minimax_search, minimax.py:216-275) at depth 2: my unblocked moves, under each the
// opponent's moves that are unblocked or land on my new head ("crash": the 10 is overwritten, so
// argmax falls back to the first empty cell in column-major order); child value = min over its
// leaves (0 if the opponent has no move at all), root picks the max, ties by random.choice, a
// boxed-in root by random.randint(1,4).  The "alpha-beta" clause (minimax.py:256-262) cannot fire
// at depth 2 (the root's minimax action is still 0 while its children run).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tron_hip.h"
#include "tron_device.hpp"
#include "tron_minimax.hpp"

namespace {

using namespace tron;

constexpr int MM_BLOCK = 256;
constexpr int MM_WAVES = MM_BLOCK / 64;

// whole-wave shifts by one lane on the DPP path (no LDS crossbar round trip): wave_shr:1 hands lane
// r the value of lane r-1, wave_shl:1 that of lane r+1; the lane with no source keeps `old` = 0
__device__ __forceinline__ uint32_t dpp_from_above(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ uint32_t dpp_from_below(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, false);
}
template <typename M>
__device__ __forceinline__ M from_row_above(M v)                     // lane r <- lane r-1
{
    if constexpr (sizeof(M) == 8)
        return (M)(((uint64_t)dpp_from_above((uint32_t)(v >> 32)) << 32) | dpp_from_above((uint32_t)v));
    else
        return (M)dpp_from_above((uint32_t)v);
}
template <typename M>
__device__ __forceinline__ M from_row_below(M v)                     // lane r <- lane r+1
{
    if constexpr (sizeof(M) == 8)
        return (M)(((uint64_t)dpp_from_below((uint32_t)(v >> 32)) << 32) | dpp_from_below((uint32_t)v));
    else
        return (M)dpp_from_below((uint32_t)v);
}
// sums / minima over the GROUP consecutive lanes that hold one board (every lane gets the result)
template <int GROUP>
__device__ __forceinline__ int group_sum(int v)
{
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <int GROUP>
__device__ __forceinline__ uint32_t group_min_u32(uint32_t v)
{
#pragma unroll
    for (int o = GROUP / 2; o > 0; o >>= 1) {
        const uint32_t w = (uint32_t)__shfl_xor((int)v, o);
        v = w < v ? w : v;
    }
    return v;
}
template <typename M>
__device__ __forceinline__ int popc(M v)
{
    if constexpr (sizeof(M) == 8) return __popcll((unsigned long long)v);
    else return __popc((uint32_t)v);
}
template <typename M>
__device__ __forceinline__ int ctz(M v)          // v != 0
{
    if constexpr (sizeof(M) == 8) return __ffsll((long long)v) - 1;
    else return __ffs((int)v) - 1;
}
// first set cell in the reference's argmax order: game_map[x][y] scanned x-major = column-major here
template <typename M, int GROUP>
__device__ __forceinline__ uint32_t first_colmajor(M mask, int row)      // (col << 8) | row, or ~0u
{
    const uint32_t key = mask ? (((uint32_t)ctz(mask) << 8) | (uint32_t)row) : 0xFFFFFFFFu;
    return group_min_u32<GROUP>(key);
}
template <typename M>
__device__ __forceinline__ M bit_at(int row, int r, int c) { return row == r ? (M)((M)1 << c) : (M)0; }

// run of set bits of `m` starting next to bit c, towards higher (dir=+1) or lower (dir=-1) bits
template <typename M>
__device__ __forceinline__ int run_from(M m, int c, int dir)
{
    constexpr int B = (int)sizeof(M) * 8;
    if (dir > 0) {
        if (c + 1 >= B) return 0;
        const M t = (M)(~(m >> (c + 1)));             // zeros shift in from the top: the run always ends
        return ctz(t);
    }
    if (c <= 0) return 0;
    const M t = (M)(~(m << (B - c)));                  // bit c-1 at the top
    if constexpr (sizeof(M) == 8) return __clzll((long long)t);
    else return __clz((int)t);
}

// Minimax.distance_walls (minimax.py:128-147) on the leaf's empty-cell mask; gbase = first lane of the board
template <typename M, int GROUP>
__device__ __forceinline__ int distance_walls(M open, int gbase, int r, int c)
{
    // the row's mask at lane gbase + r -> every lane of the board; the column as a mask over its rows
    const M row = (M)(sizeof(M) == 8
                          ? (((uint64_t)(uint32_t)__shfl((int)(uint32_t)((uint64_t)open >> 32), gbase + r) << 32) |
                             (uint32_t)__shfl((int)(uint32_t)open, gbase + r))
                          : (uint64_t)(uint32_t)__shfl((int)(uint32_t)open, gbase + r));
    uint64_t col = __ballot((open >> c) & 1) >> gbase;
    if constexpr (GROUP < 64) col &= (1ull << GROUP) - 1ull;
    const int up = run_from<uint64_t>(col, r, -1), down = run_from<uint64_t>(col, r, +1);
    const int right = run_from<M>(row, c, +1), left = run_from<M>(row, c, -1);
    return 4 + up + right + down + left;
}

struct Leaf {
    int s1r, s1c, s2r, s2c;
};

// get_voronoi_value on one leaf per board of the wave; `open` = cells equal to 1 in the leaf map, x3 =
// enemy bodies.  Boards with active == false take no part (their frontier is empty from the start).  Rows
// of neighbouring boards never leak into each other: the first lane of a board is its border row and the
// last one is a border row or beyond the board, so `& open` clears whatever the wave-wide shift brings in.
template <typename M, int GROUP>
__device__ __forceinline__ int voronoi(M open, M x3, int row, Leaf L, bool active)
{
    M f1 = active ? bit_at<M>(row, L.s1r, L.s1c) : (M)0, f2 = active ? bit_at<M>(row, L.s2r, L.s2c) : (M)0;
    M vis1 = f1, vis2 = f2;
    int acc = popc<M>(x3);
    while (__ballot((f1 | f2) != 0)) {
        const M e1 = (M)((f1 << 1) | (f1 >> 1) | from_row_above<M>(f1) | from_row_below<M>(f1));
        const M e2 = (M)((f2 << 1) | (f2 >> 1) | from_row_above<M>(f2) | from_row_below<M>(f2));
        const M n1 = e1 & open & ~vis1, n2 = e2 & open & ~vis2;
        acc += popc<M>(n1 & ~vis2 & ~n2) - popc<M>(n2 & ~vis1 & ~n1);
        vis1 |= n1;
        vis2 |= n2;
        f1 = n1;
        f2 = n2;
    }
    acc -= popc<M>(open & ~vis1 & ~vis2);
    return group_sum<GROUP>(acc);
}

// GROUP lanes per board (one lane per row): 64 / GROUP boards share a wave.  Everything that was
// wave-uniform for one board per wave is per-board here and lives in vector registers; the tree is walked
// by all boards of the wave together, each skipping (by predicate) the moves it does not have.
template <typename M, int GROUP>
__global__ __launch_bounds__(MM_BLOCK) void k_minimax(MinimaxSrc src, int n, int S, int mode, int8_t *__restrict__ out_actions,
                                                     int32_t *__restrict__ out_values, int8_t *__restrict__ out_expanded)
{
    extern __shared__ int8_t mm_lds[];
    constexpr int BPW = 64 / GROUP;                                  // boards per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane / GROUP, row = lane % GROUP, gbase = grp * GROUP;
    const int board = (blockIdx.x * MM_WAVES + wave) * BPW + grp;
    const int G = S * S;
    int8_t *tile = mm_lds + (size_t)(wave * BPW + grp) * (size_t)((G + 15) & ~15);
    const bool have = board < n;

    if (have) {
        if (src.codes) {
            const int8_t *p = src.codes + (size_t)board * src.stride;
            for (int i = row; i < G; i += GROUP) tile[i] = p[i];
        } else {
            const int8_t *p = src.grid + (size_t)board * (size_t)G;
            for (int i = row; i < G; i += GROUP) tile[i] = code1(p[i], src.player == 2);
        }
    }
    __syncthreads();

    // my row as masks
    M E = 0, X3 = 0, H10 = 0, Hm10 = 0;
    if (have && row < S) {
        const int8_t *rp = tile + row * S;
        for (int c = 0; c < S; ++c) {
            const int v = rp[c];
            const M b = (M)((M)1 << c);
            E |= (v == 1) ? b : (M)0;
            X3 |= (v == -3) ? b : (M)0;
            H10 |= (v == 10) ? b : (M)0;
            Hm10 |= (v == -10) ? b : (M)0;
        }
    }
    const uint32_t kme = first_colmajor<M, GROUP>(H10, row), kop = first_colmajor<M, GROUP>(Hm10, row);
    int r0 = (int)(kme & 0xFFu), c0 = (int)(kme >> 8), r1 = (int)(kop & 0xFFu), c1 = (int)(kop >> 8);
    const bool valid = have && kme != 0xFFFFFFFFu && kop != 0xFFFFFFFFu && r0 >= 1 && c0 >= 1 && r0 <= S - 2 &&
                       c0 <= S - 2 && r1 >= 1 && c1 >= 1 && r1 <= S - 2 && c1 <= S - 2;
    bool done = false;
    uint32_t rnd = 0u;
    if (have) {
        if (src.st4) {
            const uint4 st = src.st4[board];
            done = (st.y & META_DONE) != 0;
            uint32_t x[4];
            philox4x32_10((uint32_t)board, st.w, RNG_MINIMAX, (uint32_t)src.player, src.seed, src.stream, x);
            rnd = x[0];
        } else if (src.rnd) {
            rnd = src.rnd[board];
        }
    }
    // no live pair of heads: the reference's behaviour is index arithmetic on junk; such a board sits the
    // search out (its lanes keep the wave's shuffles company) and reports move -1
    const bool live = valid && !done;
    if (!live) r0 = c0 = r1 = c1 = 1;                               // any in-range cell: nothing is used from it

    // actions 1..4 = UP, RIGHT, DOWN, LEFT (minimax.py:290-297); kept 0-based here
    auto d_row = [](int a) { return a == 0 ? -1 : a == 2 ? 1 : 0; };
    auto d_col = [](int a) { return a == 1 ? 1 : a == 3 ? -1 : 0; };
    int values[4] = {0, 0, 0, 0};
    uint32_t expanded = 0u;
#pragma unroll 1
    for (int a = 0; a < 4; ++a) {
        const int tr = r0 + d_row(a), tc = c0 + d_col(a);
        const bool root_ok = live && tile[tr * S + tc] == 1;     // get_blocked at the root (minimax.py:170-205)
        if (root_ok) expanded |= 1u << a;
        // the opponent's options on the map after my move: 2 bits per move (0 free, 1 blocked, 2 crash)
        uint32_t blocked = 0u;
        bool any_free = false;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int ur = r1 + d_row(b), uc = c1 + d_col(b);
            const int v = (ur == tr && uc == tc) ? 10 : (ur == r0 && uc == c0) ? -1 : (int)tile[ur * S + uc];
            blocked |= ((v == 1) ? 0u : (v == 10) ? 2u : 1u) << (2 * b);
            any_free |= v == 1;
        }
        const bool child_ok = root_ok && any_free;                // all_blocked: the child keeps value 0
        if (!__ballot(child_ok)) continue;
        int best = 0;
        bool first = true;
#pragma unroll 1
        for (int b = 0; b < 4; ++b) {
            const uint32_t bl = (blocked >> (2 * b)) & 3u;
            const bool act = child_ok && bl != 1u;
            if (!__ballot(act)) continue;
            const int ur = r1 + d_row(b), uc = c1 + d_col(b);
            const M open = E & ~bit_at<M>(row, tr, tc) & ~bit_at<M>(row, ur, uc);
            Leaf L{tr, tc, ur, uc};
            const uint32_t k = first_colmajor<M, GROUP>(open, row);
            if (bl == 2u) {                                       // crash: no 10 left, argmax = first 1, else the corner wall
                L.s1r = (k == 0xFFFFFFFFu) ? 0 : (int)(k & 0xFFu);
                L.s1c = (k == 0xFFFFFFFFu) ? 0 : (int)(k >> 8);
            }
            int v;
            if (mode == TRON_MINIMAX_DISTWALL)
                v = distance_walls<M, GROUP>(open, gbase, L.s1r, L.s1c) - distance_walls<M, GROUP>(open, gbase, L.s2r, L.s2c);
            else
                v = voronoi<M, GROUP>(open, X3, row, L, act);
            if (act) {
                if (first || v < best) best = v;
                first = false;
            }
        }
        if (child_ok) {
#pragma unroll
            for (int k = 0; k < 4; ++k) values[k] = (k == a) ? best : values[k];
        }
    }

    int action;
    if (expanded == 0u) {
        action = (int)(((uint64_t)rnd * 4u) >> 32);               // random.randint(1, 4) - 1
    } else {
        int best = 0;
        bool first = true;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if ((expanded >> a) & 1u) {
                if (first || values[a] > best) best = values[a];
                first = false;
            }
        uint32_t tie = 0u;
        int nt = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (((expanded >> a) & 1u) && values[a] == best) {
                tie |= 1u << a;
                ++nt;
            }
        int pick = (int)(((uint64_t)rnd * (uint32_t)nt) >> 32);   // random.choice(minimax_acts)
        action = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if ((tie >> a) & 1u) {
                if (pick == 0) action = a;
                --pick;
            }
    }
    if (have && row == 0) {
        out_actions[board] = (int8_t)(live ? action : -1);
        if (out_expanded) out_expanded[board] = (int8_t)(live ? expanded : 0u);
        if (out_values)
            for (int a = 0; a < 4; ++a) out_values[(size_t)board * 4 + a] = live ? values[a] : 0;
    }
}

}  // namespace

int launch_minimax(const MinimaxSrc &src, int n, int S, int mode, int8_t *out_actions, int32_t *out_values,
                   int8_t *out_expanded, hipStream_t stream)
{
    if (n <= 0) return TRON_OK;
    const int group = S <= 16 ? 16 : S <= 32 ? 32 : 64;               // lanes per board: one per row
    const int per_block = MM_WAVES * (64 / group);
    const size_t smem = (size_t)per_block * (size_t)((S * S + 15) & ~15);
    const dim3 grid((unsigned)((n + per_block - 1) / per_block)), block(MM_BLOCK);
    if (group == 16)
        hipLaunchKernelGGL((k_minimax<uint32_t, 16>), grid, block, smem, stream, src, n, S, mode, out_actions, out_values,
                           out_expanded);
    else if (group == 32)
        hipLaunchKernelGGL((k_minimax<uint32_t, 32>), grid, block, smem, stream, src, n, S, mode, out_actions, out_values,
                           out_expanded);
    else
        hipLaunchKernelGGL((k_minimax<uint64_t, 64>), grid, block, smem, stream, src, n, S, mode, out_actions, out_values,
                           out_expanded);
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

extern "C" int tron_minimax_codes(const int8_t *codes, int64_t n, int32_t side, int32_t depth, int32_t mode,
                                  const uint32_t *draws, int8_t *out_actions, int32_t *out_values, int8_t *out_expanded,
                                  void *stream)
{
    if (!codes || !out_actions || n < 0 || n > 0x7FFFFFFF || side < 3 || side > 64) return TRON_ERR_BAD_ARG;
    if (mode != TRON_MINIMAX_VORONOI && mode != TRON_MINIMAX_DISTWALL) return TRON_ERR_BAD_ARG;
    if (depth != 2) return TRON_ERR_UNSUPPORTED;          // the reference only ever builds MinimaxPlayer(2, ...)
    MinimaxSrc src{};
    src.codes = codes;
    src.stride = (size_t)side * (size_t)side;
    src.player = 1;
    src.rnd = draws;
    return launch_minimax(src, (int)n, side, mode, out_actions, out_values, out_expanded,
                          reinterpret_cast<hipStream_t>(stream));
}
