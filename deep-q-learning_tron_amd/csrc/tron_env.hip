// gfx950 kernels + C ABI for the vectorised TRON env (include/tron_hip.h).
//
// Data layout.  Everything is env-major: grid[N][G] int8, obs[N][2][...],
// one uint4 of hot state words per env.  A 256-thread workgroup owns a tile of E consecutive
// envs, cut into per-env 16-byte chunks (CPE = ceil(G/16) per env, the last one possibly
// short) so a chunk never spans envs; chunk i of the tile sits in LDS slot i.
//
// k_tile, one launch = one Game.step for every env + both players' observations:
//   1  all threads: chunk loads HBM -> registers -> LDS (6 loads in flight per thread).
//      In the shadow of that load wave 0, ONE ENV PER LANE, does all the random-number
//      work the step can need: the Philox block for actions / slide uniforms and,
//      speculatively, the start of the game after next (both need only the counters);
//   2  barrier; wave 0 plays the move against the LDS tile — reads its 2-4 cells, writes
//      its <=6 cells, marks their chunks in an LDS bitmask — and leaves its results
//      (new state words, done/winner/reward, restart words) in LDS records.  It issues
//      no global store: under store back-pressure each one would stall the lone wave;
//   3  barrier; waves 1-3 write the records out (one array each), then ALL threads
//      stream: LDS chunk (or the fresh-board template + heads for a restarted env) ->
//      v_perm byte LUT -> two coalesced 16-byte stores, plus a predicated write-back of
//      the grid chunk when it is dirty or its env restarted.
// HBM sees one coalesced read of the grid, one coalesced write of both observation
// planes, ~4 dirty 16-byte chunks per env, and ~100 bytes of state/outputs per env.
// A lone wave retires about one instruction per five cycles, so the serial section
// between the barriers is LDS-only and ~200 instructions; everything heavier is off it.
//
// Kernels in this file:
//   k_tile / tile_step      board-owning layout (grid[N][G] + caller's obs): every mode, format, side
//   k_obs / obs_tile        observation-is-state (mode None, int8 codes, even side): the caller's
//                           attached obs buffer is the env state; read G, write 2G per env-step
//   k_obs_roll, k_tile_roll tron_rollout_random: the same per-tile step, but each workgroup steps
//                           its own tile up to TRON_ROLLOUT_CHUNK times in ONE launch
//   k_inc                   TRON_STEP_INCREMENTAL: writes only the touched cells + restarted boards
//   k_reset, k_obs_reset, k_obs_to_grid, k_obs_planes, k_get_state, k_encode_codes, k_pop_up, ...
//                           resets, read-back and stateless encodes
#include "tron_device.hpp"
#include "tron_minimax.hpp"
#include "../../include/tron_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <new>
#include <type_traits>

using namespace tron;

namespace {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int DK = 6;            // 16-byte loads per thread in flight per batch

struct StepOut {
    int8_t *done;
    int8_t *winner;
    float *reward;
    unsigned long long *totals;
};

// Diagnostic build only (-DTRON_STAMPS): out.totals is then a stamp buffer [blocks][2][8] of
// s_memrealtime ticks (100 MHz) for wave 0 and wave 1; never enabled in the shipped library.
#ifdef TRON_STAMPS
#define STAMP(slot)                                                                               \
    do {                                                                                          \
        if (out.totals && (tid == 0 || tid == 64))                                                \
            out.totals[((size_t)blockIdx.x * 2 + (tid >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif

struct __attribute__((packed, aligned(4))) U4A4 {   // 16 bytes at 4-byte alignment
    uint32_t x, y, z, w;
};

// ---- global access of one chunk: 16-byte ops when G % 4 == 0, bytes otherwise ----
template <bool ALIGNED>
__device__ __forceinline__ uint4 load_chunk(const int8_t *p)
{
    if (ALIGNED) {
        const U4A4 v = *reinterpret_cast<const U4A4 *>(p);          // allocation is padded: over-read is safe
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 16; ++j) w[j >> 2] |= (uint32_t)(uint8_t)p[j] << ((j & 3) * 8);
    return make_uint4(w[0], w[1], w[2], w[3]);
}
template <bool ALIGNED>
__device__ __forceinline__ void store_chunk(int8_t *p, int nb, const uint32_t w[4])
{
    if (ALIGNED) {
        if (nb == 16) {
            *reinterpret_cast<U4A4 *>(p) = U4A4{w[0], w[1], w[2], w[3]};
        } else {
            for (int j = 0; j < (nb >> 2); ++j) reinterpret_cast<uint32_t *>(p)[j] = w[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < nb) p[j] = (int8_t)(w[j >> 2] >> ((j & 3) * 8));
    }
}

// one chunk (nb valid cells of one env starting at cell c) -> both players' observations
template <int FMT, bool ALIGNED>
__device__ __forceinline__ void store_chunk_obs(void *__restrict__ obs, size_t env, int G, uint32_t c, int nb,
                                                const uint32_t w[4], float p4)
{
    if (FMT == TRON_OBS_CODES_I8) {
        int8_t *o1 = reinterpret_cast<int8_t *>(obs) + env * 2u * G + c;
        const uint32_t c1[4] = {codes4(w[0], false), codes4(w[1], false), codes4(w[2], false), codes4(w[3], false)};
        const uint32_t c2[4] = {codes4(w[0], true), codes4(w[1], true), codes4(w[2], true), codes4(w[3], true)};
        store_chunk<ALIGNED>(o1, nb, c1);
        store_chunk<ALIGNED>(o1 + G, nb, c2);
    } else if (FMT == TRON_OBS_PLANES3_F32 || FMT == TRON_OBS_PLANES4_F32) {
        constexpr int CH = (FMT == TRON_OBS_PLANES3_F32) ? 3 : 4;
        float *ob = reinterpret_cast<float *>(obs) + env * 2u * CH * G + c;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                float *dst = ob + (size_t)(p * CH + ch) * G;
                const uint32_t bits = (ch < 3) ? plane_bits(ch, p != 0) : 0u;
                if (ALIGNED) {                                       // G % 4 == 0: 16-byte aligned rows of 4 cells
                    for (int j = 0; j < (nb >> 2); ++j)
                        reinterpret_cast<float4 *>(dst)[j] =
                            (ch == 3) ? make_float4(p4, p4, p4, p4)
                                      : make_float4(plane_val(bits, w[j]), plane_val(bits, w[j] >> 8),
                                                    plane_val(bits, w[j] >> 16), plane_val(bits, w[j] >> 24));
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j)
                        if (j < nb) dst[j] = (ch == 3) ? p4 : plane_val(bits, w[j >> 2] >> ((j & 3) * 8));
                }
            }
        }
    }
}

// state words of one env, loaded ahead of the tile
// chunk index -> env within the tile: i / cpe by multiplication with ceil(2^32 / cpe); that constant
// does not fit 32 bits for cpe == 1 (boards of at most 16 cells), where the quotient is i itself
__device__ __forceinline__ uint32_t chunk_env(uint32_t i, uint32_t cpe, uint32_t cpe_magic)
{
    return cpe == 1u ? i : __umulhi(i, cpe_magic);
}

// Game.get_rate (game.py:100-102) as a table: "temper" mode compares a float32 uniform with the float64 rate of
// (degree, weight); for every (degree in [-30, 30], weight in [40, 101]) — the ranges Game.__init__ draws from
// (game.py:83,87) — the table holds the largest float32 t with (double)t <= rate, so `u <= t` decides exactly what
// `(double)u <= rate` decides, without two float64 divisions per player in the one-env-per-lane section of the step.
// Filled once per process by tron_create (host arithmetic, same operation order); values assigned from outside the
// ranges (tron_set_weight_degree) take the float64 path.
constexpr int RATE_DEG = 61, RATE_W = 62;
__device__ float g_rate_thr[RATE_DEG * RATE_W];

struct EnvRegs {
    uint32_t pos, meta, eplen, tick;           // st4
    uint32_t envp, episode, nstart, nenvp;     // rs4
    uint32_t act;                              // a0 | a1 << 8 when the caller supplies actions
    float u0, u1;                              // slide uniforms when the caller supplies them
    double slide;
};

// result records left in LDS by the move (read by waves 1-3 and by the stream)
enum { RES_STEPPED = 1u, RES_DONE = 2u, RES_STORE_ST = 4u, RES_RESET = 8u };

// ------------------------------------------------------------------ the move --
// One lane = one env, against its LDS copy g: Game.next_frame + Game.step (game.py:149-277).
// LDS-only, one read round trip.  Leaves: rec_st (new st4), rec_out {flags | winner<<4, reward1,
// reward2, restart word}.  (Pre-fetching the target cells from HBM before the barrier was
// measured too: +20 VGPRs held across the tile load cost a workgroup per CU and lost.)
__device__ inline void lane_move(const Params &P, unsigned char *g, const EnvRegs &R, const int a[2], const float u[2],
                                 uint32_t flags, uint32_t *dirty, uint32_t chunk0, uint4 &rec_st, uint4 &rec_out)
{
    const int S = P.S, W = P.W;
    uint32_t m = R.meta;
    int r[2] = {(int)(int8_t)(R.pos), (int)(int8_t)(R.pos >> 16)};
    int c[2] = {(int)(int8_t)(R.pos >> 8), (int)(int8_t)(R.pos >> 24)};
    bool done = (m & META_DONE) != 0;
    int winner = (int)((m >> 4) & 3u);
    float rw0 = 0.0f, rw1 = 0.0f;
    uint32_t res = 0u;
    rec_st = make_uint4(R.pos, R.meta, R.eplen, R.tick);

    // "temper": both players' slide thresholds, requested before anything else so that they arrive under the cell reads
    float thr[2] = {0.0f, 0.0f};
    bool thr_ok[2] = {false, false};
    if (P.mode == TRON_MODE_TEMPER) {
        const uint32_t di = (uint32_t)((int)(int8_t)(R.envp >> 16) + 30);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t wi = ((R.envp >> (8 * p)) & 0xFFu) - 40u;
            thr_ok[p] = di < (uint32_t)RATE_DEG && wi < (uint32_t)RATE_W;
            thr[p] = g_rate_thr[thr_ok[p] ? di * RATE_W + wi : 0u];
        }
    }
    if (!done) {
        res |= RES_STEPPED;
        const bool sliding = (P.mode != TRON_MODE_NONE);
        // The <=4 cells the move can look at — first target n[p] and slide landing s[p] of each
        // player (player.py:124-132, game.py:163-178) — are read from the LDS copy in ONE round
        // trip; the write-then-read dependencies between the players are resolved in registers.
        int dr[2], dc[2], n[2], sl[2], tn[2], ts[2];
        bool inb[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            dr[p] = (a[p] == 0) ? -1 : (a[p] == 2) ? 1 : 0;   // UP / DOWN
            dc[p] = (a[p] == 1) ? 1 : (a[p] == 3) ? -1 : 0;   // RIGHT / LEFT
            const int nr = r[p] + dr[p], nc = c[p] + dc[p];
            inb[p] = nr >= 0 && nc >= 0 && nr < W && nc < W;
            n[p] = cell_index(S, nr, nc);
            sl[p] = inb[p] ? cell_index(S, nr + dr[p], nc + dc[p]) : n[p];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            tn[p] = (int)(int8_t)g[n[p]];
            ts[p] = sliding ? (int)(int8_t)g[sl[p]] : (int)TRON_WALL;
        }
        // cells written so far, in program order; a read sees the latest write to that cell
        int cells[6], vals[6];
        cells[0] = cell_index(S, r[0], c[0]); vals[0] = TRON_P1_BODY;     // game.py:155-156: heads -> bodies first
        cells[1] = cell_index(S, r[1], c[1]); vals[1] = TRON_P2_BODY;
#pragma unroll
        for (int k = 2; k < 6; ++k) { cells[k] = cells[k & 1]; vals[k] = vals[k & 1]; }
        auto tile_at = [&](int idx, int before, int upto) {
            int v = before;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k < upto && cells[k] == idx) v = vals[k];
            return v;
        };

        // game.py:158-178 — advance, optional slide, player order
        int f[2], tf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f[p] = n[p];
            tf[p] = tn[p];
            int nr = r[p] + dr[p], nc = c[p] + dc[p];
            // the uniform is consulted only for an in-bounds EMPTY target (game.py:164-165)
            if (sliding && inb[p] && tile_at(n[p], tn[p], 2 + p) == TRON_EMPTY) {
                bool slides;                                             // game.py:169: random.random() <= rate
                if (P.mode == TRON_MODE_ICE) {
                    slides = (double)u[p] <= R.slide;
                } else if (thr_ok[p]) {
                    slides = u[p] <= thr[p];
                } else {
                    slides = (double)u[p] <= get_rate((int)(int8_t)(R.envp >> 16), (int)((R.envp >> (8 * p)) & 0xFFu));
                }
                if (slides) {
                    cells[2 + p] = n[p];
                    vals[2 + p] = (p == 0) ? TRON_P1_SLIDE : TRON_P2_SLIDE;
                    f[p] = sl[p];
                    tf[p] = ts[p];
                    nr += dr[p];
                    nc += dc[p];
                }
            }
            r[p] = nr;
            c[p] = nc;
        }

        // game.py:205-214 — collisions in player order; the head is written in every branch
        // (an out-of-bounds head lands on the border WALL cell)
        uint32_t alive = m & 3u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool oob = r[p] < 0 || c[p] < 0 || r[p] >= W || c[p] >= W;
            if (oob || tile_at(f[p], tf[p], 4 + p) != TRON_EMPTY)
                alive &= ~(1u << p);
            cells[4 + p] = f[p];
            vals[4 + p] = (p == 0) ? TRON_P1_HEAD : TRON_P2_HEAD;
        }
        // the writes, in program order (same-lane LDS writes keep their order); fire and forget
#pragma unroll
        for (int k = 0; k < 6; ++k) g[cells[k]] = (unsigned char)vals[k];

        // game.py:264-275 — done / winner (same cell => draw)
        const int n_alive = (int)(alive & 1u) + (int)((alive >> 1) & 1u);
        if (n_alive <= 1) {
            if (n_alive == 1 && (r[0] != r[1] || c[0] != c[1]))
                winner = (alive & 1u) ? 1 : 2;
            done = true;
        }

        // rewards: util.py:87-94 / DDQN.py:289-305 / DQN.py:224-241
        if (!done) {
            rw0 = rw1 = P.r_index ? (float)R.eplen : P.r_step;
        } else if (winner == 0) {
            rw0 = rw1 = P.r_draw;
        } else {
            rw0 = (winner == 1) ? P.r_win : P.r_lose;
            rw1 = (winner == 2) ? P.r_win : P.r_lose;
        }

        if (!(done && (flags & TRON_STEP_AUTORESET))) {
            // the stream writes these chunks back to the grid
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const uint32_t ci = chunk0 + (uint32_t)(cells[k] >> 4);
                atomicOr(&dirty[ci >> 5], 1u << (ci & 31u));
            }
        }
        rec_st = make_uint4(pack_pos(r[0], c[0], r[1], c[1]),
                            alive | (done ? META_DONE : 0u) | ((uint32_t)winner << 4) | ((uint32_t)(a[0] + 1) << 8) |
                                ((uint32_t)(a[1] + 1) << 12),
                            R.eplen + 1u, R.tick + 1u);
        res |= RES_STORE_ST;
    }
    if (done) res |= RES_DONE;

    uint32_t restart = 0u;
    if (done && (flags & TRON_STEP_AUTORESET)) {                    // ACKTR.py:307-310
        // the game being started was drawn at the previous restart (rs4.nstart / .nenvp)
        rec_st = make_uint4(R.nstart, META_ALIVE0 | META_ALIVE1, 0u, rec_st.w);
        res |= RES_STORE_ST | RES_RESET;
        const int h1 = cell_index(S, (int)(int8_t)(R.nstart), (int)(int8_t)(R.nstart >> 8));
        const int h2 = cell_index(S, (int)(int8_t)(R.nstart >> 16), (int)(int8_t)(R.nstart >> 24));
        restart = 0x80000000u | (uint32_t)h1 | ((uint32_t)h2 << 14);
    }
    rec_out = make_uint4(res | ((uint32_t)winner << 4), __float_as_uint(rw0), __float_as_uint(rw1), restart);
}

// ---------------------------------------------------------------- the kernel --
// One tile of E envs through one step (or one encode); shared by k_tile (one tile per workgroup per
// launch) and k_tile_roll (the persistent rollout, see k_obs_roll).
template <int FMT, bool DO_STEP, bool ALIGNED>
__device__ __forceinline__ void tile_step(const Params &P, int E, uint32_t cpe, uint32_t cpe_magic,
                                          const int8_t *__restrict__ actions, const float *__restrict__ uniforms,
                                          uint32_t flags, void *__restrict__ obs, const StepOut &out, int tile_idx,
                                          unsigned char *smem)
{
    const int G = P.G;
    uint4 *tile = reinterpret_cast<uint4 *>(smem);                  // [E*cpe]
    uint4 *tmpl = tile + (size_t)E * cpe;                           // [cpe] fresh board as chunks
    uint4 *rec_st = tmpl + cpe;                                     // [E] new st4
    uint4 *rec_out = rec_st + E;                                    // [E] flags / rewards / restart word
    uint4 *rec_rs = rec_out + E;                                    // [E] new rs4 (restarted envs)
    float *plane4 = reinterpret_cast<float *>(rec_rs + E);          // [E]
    uint32_t *dirty = reinterpret_cast<uint32_t *>(plane4 + E);     // [ceil(E*cpe/32)] chunk bitmask

    const int tid = threadIdx.x;
    const int e0 = tile_idx * E;
    const int ne = min(E, P.N - e0);
    const int env = e0 + tid;
    const bool mine = tid < ne;
    const uint32_t nchunks = (uint32_t)ne * cpe;
    const int8_t *gtile = P.grid + (size_t)e0 * G;
    const bool sliding = (P.mode != TRON_MODE_NONE);
    const bool autoreset = (flags & TRON_STEP_AUTORESET) != 0u, nonrev = (flags & TRON_STEP_NONREVERSING) != 0u;
    const bool w0 = DO_STEP && tid < WAVE;
    constexpr bool PLANES_BY_ROW = ALIGNED && (FMT == TRON_OBS_PLANES3_F32 || FMT == TRON_OBS_PLANES4_F32);

    STAMP(0);
    // ---- 1: state words first (their latency hides under the tile load), then the tile
    EnvRegs R{};
    if (w0 && mine) {
        const uint4 st = P.st4[env];
        R.pos = st.x; R.meta = st.y; R.eplen = st.z; R.tick = st.w;
        if (autoreset || sliding) {
            const uint4 rs = P.rs4[env];
            R.envp = rs.x; R.episode = rs.y; R.nstart = rs.z; R.nenvp = rs.w;
        }
        if (actions) R.act = reinterpret_cast<const uint16_t *>(actions)[env];
        if (sliding) {
            if (uniforms) {
                const float2 uu = reinterpret_cast<const float2 *>(uniforms)[env];
                R.u0 = uu.x;
                R.u1 = uu.y;
            }
            R.slide = P.slide[env];
        }
    }
    if (FMT == TRON_OBS_PLANES4_F32 && mine) plane4[tid] = (float)degree_slide(P.slide[env]);   // game.py:124-132
    if (DO_STEP && autoreset)
        for (uint32_t d = (uint32_t)tid; d < cpe * 16u; d += BLOCK)
            reinterpret_cast<int8_t *>(tmpl)[d] = (d < (uint32_t)G) ? P.fresh[d] : (int8_t)0;

    int a[2] = {0, 0};
    float u[2] = {R.u0, R.u1};
    for (uint32_t base = 0; base < nchunks; base += DK * BLOCK) {
        uint4 v[DK];
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            const uint32_t i = base + (uint32_t)tid + (uint32_t)k * BLOCK;
            const uint32_t le = chunk_env(i, cpe, cpe_magic);
            const uint32_t c = (i - le * cpe) * 16u;
            if (i < nchunks) v[k] = load_chunk<ALIGNED>(gtile + (size_t)le * G + c);
        }
        if (base == 0u && w0) {
            // the random-number work, in the shadow of the tile load (it needs the counters only)
            for (uint32_t d = (uint32_t)tid; d < (nchunks + 31u) / 32u; d += WAVE) dirty[d] = 0u;
            if (mine) {
                const bool have_actions = actions != nullptr, have_uniforms = uniforms != nullptr;
                if (!have_actions || (sliding && !have_uniforms)) {
                    uint32_t x[4];
                    philox4x32_10((uint32_t)env, R.tick, RNG_STEP, 0u, P.seed, P.stream, x);
                    a[0] = draw_action(x[0], (R.meta >> 8) & 0xFu, nonrev);
                    a[1] = draw_action(x[1], (R.meta >> 12) & 0xFu, nonrev);
                    if (!have_uniforms) {
                        u[0] = (float)(x[2] >> 8) * (1.0f / 16777216.0f);
                        u[1] = (float)(x[3] >> 8) * (1.0f / 16777216.0f);
                    }
                }
                if (have_actions) {
                    a[0] = (int)(R.act & 3u);
                    a[1] = (int)((R.act >> 8) & 3u);
                }
                if (autoreset) {       // speculative: stored only if this env restarts in this launch
                    const NewGame ng = make_game(P.seed, P.stream, P.W, P.fair, (uint32_t)env, R.episode + 1u);
                    rec_rs[tid] = make_uint4(R.nenvp, R.episode + 1u, pack_pos(ng.r1, ng.c1, ng.r2, ng.c2),
                                             pack_envp(ng.w0, ng.w1, ng.degree));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            const uint32_t i = base + (uint32_t)tid + (uint32_t)k * BLOCK;
            if (i < nchunks) tile[i] = v[k];
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    if (DO_STEP) {
        // ---- 2: the move, wave 0, one env per lane, LDS only
        if (w0) {
            uint4 rs = make_uint4(0u, 0u, 0u, 0u), ro = make_uint4(0u, 0u, 0u, 0u);
            if (mine)
                lane_move(P, reinterpret_cast<unsigned char *>(tile + (size_t)tid * cpe), R, a, u, flags, dirty,
                          (uint32_t)tid * cpe, rs, ro);
            if (tid < E) {
                rec_st[tid] = rs;
                rec_out[tid] = ro;
            }
        }
        STAMP(3);
        __syncthreads();
        STAMP(4);

        // ---- 3a: waves 1-3 write the records out, one kind of array each (lane = env)
        const int wave = tid >> 6, lane = tid & 63;
        if (wave >= 1 && lane < ne) {
            const uint4 ro = rec_out[lane];
            const int genv = e0 + lane;
            if (wave == 1) {
                if (ro.x & RES_STORE_ST) P.st4[genv] = rec_st[lane];
                if (ro.x & RES_RESET) P.rs4[genv] = rec_rs[lane];
            } else if (wave == 2) {
                if (out.done) out.done[genv] = (int8_t)((ro.x & RES_DONE) != 0u);
                if (out.winner) out.winner[genv] = (int8_t)((ro.x >> 4) & 3u);
            } else {
                if (out.reward)
                    reinterpret_cast<float2 *>(out.reward)[genv] = make_float2(__uint_as_float(ro.y), __uint_as_float(ro.z));
            }
        }
#ifndef TRON_STAMPS
        if (out.totals && wave == 3) {
            // {env_steps, p1_wins, p2_wins, draws}: one atomic per counter per workgroup
            const uint32_t f = lane < ne ? rec_out[lane].x : 0u;
            const int wn = ((f & RES_STEPPED) && (f & RES_DONE)) ? (int)((f >> 4) & 3u) : -1;
            const unsigned long long bs = __ballot((f & RES_STEPPED) != 0u);
            const unsigned long long b1 = __ballot(wn == 1), b2 = __ballot(wn == 2), b0 = __ballot(wn == 0);
            if (lane == 0) {
                if (bs) atomicAdd(&out.totals[0], (unsigned long long)__popcll(bs));
                if (b1) atomicAdd(&out.totals[1], (unsigned long long)__popcll(b1));
                if (b2) atomicAdd(&out.totals[2], (unsigned long long)__popcll(b2));
                if (b0) atomicAdd(&out.totals[3], (unsigned long long)__popcll(b0));
            }
        }
#endif
    }

    // ---- 3b: the stream
    for (uint32_t i = (uint32_t)tid; i < nchunks; i += BLOCK) {
        const uint32_t le = chunk_env(i, cpe, cpe_magic);
        const uint32_t k = i - le * cpe;
        const uint32_t c = k * 16u;
        const int nb = min(16, G - (int)c);                               // valid cells in this chunk
        uint4 t = tile[i];
        bool wb = false;
        if (DO_STEP) {
            const uint32_t ri = rec_out[le].w;
            if (ri >> 31) {                                                // restarted env: fresh board + heads
                t = tmpl[k];
                const uint32_t d1 = (ri & 0x3FFFu) - c, d2 = ((ri >> 14) & 0x3FFFu) - c;
                const uint32_t v1 = (uint32_t)TRON_P1_HEAD << ((d1 & 3u) * 8u);   // EMPTY is 0: OR the head in
                const uint32_t v2 = (uint32_t)TRON_P2_HEAD << ((d2 & 3u) * 8u);   // game.py:90-91
                t.x |= (d1 < 4u ? v1 : 0u) | (d2 < 4u ? v2 : 0u);
                t.y |= (d1 - 4u < 4u ? v1 : 0u) | (d2 - 4u < 4u ? v2 : 0u);
                t.z |= (d1 - 8u < 4u ? v1 : 0u) | (d2 - 8u < 4u ? v2 : 0u);
                t.w |= (d1 - 12u < 4u ? v1 : 0u) | (d2 - 12u < 4u ? v2 : 0u);
                wb = true;
            } else {
                wb = ((dirty[i >> 5] >> (i & 31u)) & 1u) != 0u;
            }
        }
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
        if (wb) store_chunk<ALIGNED>(P.grid + (size_t)(e0 + le) * G + c, nb, w);
        if (PLANES_BY_ROW) {
            if (DO_STEP && wb) tile[i] = t;                                // the plane pass reads the tile from LDS
        } else {
            store_chunk_obs<FMT, ALIGNED>(obs, (size_t)(e0 + le), G, c, nb, w,
                                          (FMT == TRON_OBS_PLANES4_F32) ? plane4[le] : 0.0f);
        }
    }
    if (PLANES_BY_ROW) {
        // f32 planes, 16-byte aligned rows: one wave per env, one lane per 4 cells, so every store
        // instruction writes 64 x 16 contiguous bytes of one plane (per-chunk stores would leave each
        // lane its own 64-byte run: a quarter of every line per instruction)
        constexpr int CH = (FMT == TRON_OBS_PLANES4_F32) ? 4 : 3;
        __syncthreads();
        const int wave = tid >> 6, lane = tid & 63, quads = G >> 2;
        for (int le = wave; le < ne; le += BLOCK / WAVE) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(tile + (size_t)le * cpe);
            float4 *ob = reinterpret_cast<float4 *>(reinterpret_cast<float *>(obs) + (size_t)(e0 + le) * 2u * CH * G);
            const float p4 = (FMT == TRON_OBS_PLANES4_F32) ? plane4[le] : 0.0f;
            for (int j = lane; j < quads; j += WAVE) {
                const uint32_t wd = src[j];
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int ch = 0; ch < CH; ++ch) {
                        const uint32_t bits = (ch < 3) ? plane_bits(ch, p != 0) : 0u;
                        ob[(size_t)(p * CH + ch) * quads + j] =
                            (ch == 3) ? make_float4(p4, p4, p4, p4)
                                      : make_float4(plane_val(bits, wd), plane_val(bits, wd >> 8),
                                                    plane_val(bits, wd >> 16), plane_val(bits, wd >> 24));
                    }
            }
        }
    }
    STAMP(7);
}

template <int FMT, bool DO_STEP, bool ALIGNED>
__global__ __launch_bounds__(BLOCK) void k_tile(Params P, int E, uint32_t cpe, uint32_t cpe_magic,
                                                const int8_t *__restrict__ actions,
                                                const float *__restrict__ uniforms, uint32_t flags,
                                                void *__restrict__ obs, StepOut out, int tile0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    tile_step<FMT, DO_STEP, ALIGNED>(P, E, cpe, cpe_magic, actions, uniforms, flags, obs, out, (int)blockIdx.x + tile0, smem);
}

typedef __attribute__((address_space(4))) const unsigned char kernarg_t;    // the kernel-argument segment (constant address space: scalar loads)
__device__ __forceinline__ void load_params(Params &p, kernarg_t *q)       // Params is every rollout kernel's first argument
{
    // one block copy straight from the constant address space (wide scalar loads).  Measured alternatives, same box: word by
    // word, or through a generic pointer, the rollout is 8 % SLOWER than with the parameters simply kept live.
    __builtin_memcpy(&p, q, sizeof(Params));
}

// persistent rollout on the board-owning layout (every mode / format / side): see k_obs_roll
template <int FMT, bool ALIGNED>
__global__ __launch_bounds__(BLOCK) void k_tile_roll(Params P, int E, uint32_t cpe, uint32_t cpe_magic, uint32_t flags,
                                                     void *__restrict__ obs, StepOut out, int k_steps, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long acc[4];           // see k_obs_roll
    StepOut lo = out;
    if (out.totals) {
        if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
        lo.totals = acc;
        __syncthreads();
    }
    // (Params stays live across the loop here: re-read per step as in k_obs_roll, this kernel goes from 97-106 to 135-141 VGPRs —
    // three waves per SIMD instead of four — and temper mode from 2.58 to 2.23 G env-steps/s)
    for (int s = 0; s < k_steps; ++s)
        for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
            tile_step<FMT, true, ALIGNED>(P, E, cpe, cpe_magic, nullptr, nullptr, flags, obs, lo, t, smem);
            __syncthreads();
        }
    if (out.totals && threadIdx.x < 4 && acc[threadIdx.x]) atomicAdd(&out.totals[threadIdx.x], acc[threadIdx.x]);
}

// ------------------------------------------------- observation-is-state kernel --
// Mode None, int8 code observations, even board side.  There are no slide tiles in this mode,
// so the player-1 code plane (map.py:67-81) is a lossless image of the board.  The caller
// attaches its [N][2][G] observation buffer once (tron_attach_obs_state) and that buffer IS the
// env state: a step reads the player-1 plane (G bytes per env) and rewrites both planes (2G) —
// exactly the algorithmic traffic, no dirty-chunk write-back, no rewrite of restarted boards.
// Same three phases as k_tile; differences: the tile holds player-1 codes, the move works in
// code space (EMPTY is 1; bodies -2 / -3; heads 10 / -10), the player-2 plane is the
// swap_codes4 LUT of the player-1 plane, and wave 1 draws the speculative next start so it runs
// beside wave 0's action Philox instead of after it.
__device__ inline void lane_move_codes(const Params &P, unsigned char *g, const EnvRegs &R, const int a[2],
                                       uint32_t flags, uint4 &rec_st, uint4 &rec_out)
{
    const int S = P.S, W = P.W;
    uint32_t m = R.meta;
    int r[2] = {(int)(int8_t)(R.pos), (int)(int8_t)(R.pos >> 16)};
    int c[2] = {(int)(int8_t)(R.pos >> 8), (int)(int8_t)(R.pos >> 24)};
    bool done = (m & META_DONE) != 0;
    int winner = (int)((m >> 4) & 3u);
    float rw0 = 0.0f, rw1 = 0.0f;
    uint32_t res = 0u;
    rec_st = make_uint4(R.pos, R.meta, R.eplen, R.tick);

    if (!done) {
        res |= RES_STEPPED;
        constexpr int C_EMPTY = 1, C_P1_BODY = -2, C_P2_BODY = -3, C_P1_HEAD = 10, C_P2_HEAD = -10;
        int old[2], f[2], tf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            old[p] = cell_index(S, r[p], c[p]);
            r[p] += (a[p] == 0) ? -1 : (a[p] == 2) ? 1 : 0;          // UP / DOWN   (player.py:124-132)
            c[p] += (a[p] == 1) ? 1 : (a[p] == 3) ? -1 : 0;          // RIGHT / LEFT
            f[p] = cell_index(S, r[p], c[p]);
        }
        tf[0] = (int)(int8_t)g[f[0]];                                  // one LDS round trip for both targets
        tf[1] = (int)(int8_t)g[f[1]];
        // game.py:155-156 — heads turn into bodies BEFORE anyone moves: a target that is either
        // old head is a body by now
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            if (f[p] == old[0]) tf[p] = C_P1_BODY;
            if (f[p] == old[1]) tf[p] = C_P2_BODY;
        }
        if (f[1] == f[0]) tf[1] = C_P1_HEAD;                           // game.py:205-214: P2 tests after P1's head is down
        uint32_t alive = m & 3u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool oob = r[p] < 0 || c[p] < 0 || r[p] >= W || c[p] >= W;
            if (oob || tf[p] != C_EMPTY) alive &= ~(1u << p);
        }
        // writes in the reference's order: bodies, then P1's head, then P2's (an out-of-bounds
        // head lands on the border WALL cell; a same-cell head-on leaves P2's head)
        g[old[0]] = (unsigned char)C_P1_BODY;
        g[old[1]] = (unsigned char)C_P2_BODY;
        g[f[0]] = (unsigned char)C_P1_HEAD;
        g[f[1]] = (unsigned char)C_P2_HEAD;

        // game.py:264-275 — done / winner (same cell => draw)
        const int n_alive = (int)(alive & 1u) + (int)((alive >> 1) & 1u);
        if (n_alive <= 1) {
            if (n_alive == 1 && (r[0] != r[1] || c[0] != c[1]))
                winner = (alive & 1u) ? 1 : 2;
            done = true;
        }
        // rewards: util.py:87-94 / DDQN.py:289-305 / DQN.py:224-241
        if (!done) {
            rw0 = rw1 = P.r_index ? (float)R.eplen : P.r_step;
        } else if (winner == 0) {
            rw0 = rw1 = P.r_draw;
        } else {
            rw0 = (winner == 1) ? P.r_win : P.r_lose;
            rw1 = (winner == 2) ? P.r_win : P.r_lose;
        }
        rec_st = make_uint4(pack_pos(r[0], c[0], r[1], c[1]),
                            alive | (done ? META_DONE : 0u) | ((uint32_t)winner << 4) | ((uint32_t)(a[0] + 1) << 8) |
                                ((uint32_t)(a[1] + 1) << 12),
                            R.eplen + 1u, R.tick + 1u);
        res |= RES_STORE_ST;
    }
    if (done) res |= RES_DONE;

    uint32_t restart = 0u;
    if (done && (flags & TRON_STEP_AUTORESET)) {                      // ACKTR.py:307-310
        rec_st = make_uint4(R.nstart, META_ALIVE0 | META_ALIVE1, 0u, rec_st.w);
        res |= RES_STORE_ST | RES_RESET;
        const int h1 = cell_index(S, (int)(int8_t)(R.nstart), (int)(int8_t)(R.nstart >> 8));
        const int h2 = cell_index(S, (int)(int8_t)(R.nstart >> 16), (int)(int8_t)(R.nstart >> 24));
        restart = 0x80000000u | (uint32_t)h1 | ((uint32_t)h2 << 14);
    }
    rec_out = make_uint4(res | ((uint32_t)winner << 4), __float_as_uint(rw0), __float_as_uint(rw1), restart);
}

// ---- the sliding modes ("ice", "temper") on the observation-is-state layout ---------------------------------------------
// A slide leaves a P1_slide / P2_slide tile behind (game.py:163-178) that Map.color shows as that player's body
// (map.py:67-81): the dynamics never tell the two apart (a cell is EMPTY or it is not), only the board image does
// (tron_get_grid).  So the player-1 code plane carries the game here too, and the slide marks go to a per-env LOG —
// entry = cell | player << 15, appended by the step that makes the mark (2 bytes, only when somebody slides), their
// number in st4.meta bits 16-29 (a restart rewrites meta: the log empties by itself) — which tron_get_grid replays:
// a logged cell that still holds its player's body code is a slide tile.  The board-owning layout wrote every
// dirty 16-byte chunk of the board back as a partial line instead: 153 MB per step against 121 (profiles/r04_temper_pmc.txt).
constexpr uint32_t SLIDE_CNT_SHIFT = 16u, SLIDE_CNT_MASK = 0x3FFFu;      // (a mark takes a cell and so does the head behind it: <= W W / 2 marks; 14 bits cover every side the 15-bit cell index allows)
__device__ __forceinline__ uint16_t *slide_log(const Params &P)
{
    return reinterpret_cast<uint16_t *>(reinterpret_cast<char *>(P.slide) + (((size_t)P.N * 8u + 255u) & ~(size_t)255u));
}
__host__ __device__ __forceinline__ int slide_log_len(int W) { return W * W; }   // a mark takes a cell of its own

// lane_move (the board-owning layout's move, above) in code space; marks out: (cell + 1) of player 1's slide mark | (cell + 1)
// << 14 of player 2's, 0 = none (rec_out.w when the env does not restart: a restarting env's marks die with its board)
__device__ inline void lane_move_codes_slide(const Params &P, unsigned char *g, const EnvRegs &R, const int a[2], const float u[2],
                                             uint32_t flags, uint4 &rec_st, uint4 &rec_out)
{
    constexpr int C_EMPTY = 1, C_WALL = -1, C_P1_BODY = -2, C_P2_BODY = -3, C_P1_HEAD = 10, C_P2_HEAD = -10;
    const int S = P.S, W = P.W;
    uint32_t m = R.meta;
    int r[2] = {(int)(int8_t)(R.pos), (int)(int8_t)(R.pos >> 16)};
    int c[2] = {(int)(int8_t)(R.pos >> 8), (int)(int8_t)(R.pos >> 24)};
    bool done = (m & META_DONE) != 0;
    int winner = (int)((m >> 4) & 3u);
    float rw0 = 0.0f, rw1 = 0.0f;
    uint32_t res = 0u, marks = 0u;
    rec_st = make_uint4(R.pos, R.meta, R.eplen, R.tick);

    float thr[2] = {0.0f, 0.0f};
    bool thr_ok[2] = {false, false};
    if (P.mode == TRON_MODE_TEMPER) {
        const uint32_t di = (uint32_t)((int)(int8_t)(R.envp >> 16) + 30);
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const uint32_t wi = ((R.envp >> (8 * p)) & 0xFFu) - 40u;
            thr_ok[p] = di < (uint32_t)RATE_DEG && wi < (uint32_t)RATE_W;
            thr[p] = g_rate_thr[thr_ok[p] ? di * RATE_W + wi : 0u];
        }
    }
    if (!done) {
        res |= RES_STEPPED;
        int dr[2], dc[2], n[2], sl[2], tn[2], ts[2];
        bool inb[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            dr[p] = (a[p] == 0) ? -1 : (a[p] == 2) ? 1 : 0;   // UP / DOWN
            dc[p] = (a[p] == 1) ? 1 : (a[p] == 3) ? -1 : 0;   // RIGHT / LEFT
            const int nr = r[p] + dr[p], nc = c[p] + dc[p];
            inb[p] = nr >= 0 && nc >= 0 && nr < W && nc < W;
            n[p] = cell_index(S, nr, nc);
            sl[p] = inb[p] ? cell_index(S, nr + dr[p], nc + dc[p]) : n[p];
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            tn[p] = (int)(int8_t)g[n[p]];
            ts[p] = (int)(int8_t)g[sl[p]];
        }
        int cells[6], vals[6];
        cells[0] = cell_index(S, r[0], c[0]); vals[0] = C_P1_BODY;       // game.py:155-156: heads -> bodies first
        cells[1] = cell_index(S, r[1], c[1]); vals[1] = C_P2_BODY;
#pragma unroll
        for (int k = 2; k < 6; ++k) { cells[k] = cells[k & 1]; vals[k] = vals[k & 1]; }
        auto code_at = [&](int idx, int before, int upto) {
            int v = before;
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (k < upto && cells[k] == idx) v = vals[k];
            return v;
        };
        int f[2], tf[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            f[p] = n[p];
            tf[p] = tn[p];
            int nr = r[p] + dr[p], nc = c[p] + dc[p];
            if (inb[p] && code_at(n[p], tn[p], 2 + p) == C_EMPTY) {       // the uniform is consulted only for an in-bounds EMPTY target (game.py:164-165)
                bool slides;                                             // game.py:169: random.random() <= rate
                if (P.mode == TRON_MODE_ICE) {
                    slides = (double)u[p] <= R.slide;
                } else if (thr_ok[p]) {
                    slides = u[p] <= thr[p];
                } else {
                    slides = (double)u[p] <= get_rate((int)(int8_t)(R.envp >> 16), (int)((R.envp >> (8 * p)) & 0xFFu));
                }
                if (slides) {
                    cells[2 + p] = n[p];
                    vals[2 + p] = (p == 0) ? C_P1_BODY : C_P2_BODY;      // the slide tile, as Map.color shows it
                    marks |= (uint32_t)(n[p] + 1) << (14 * p);
                    f[p] = sl[p];
                    tf[p] = ts[p];
                    nr += dr[p];
                    nc += dc[p];
                }
            }
            r[p] = nr;
            c[p] = nc;
        }
        uint32_t alive = m & 3u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const bool oob = r[p] < 0 || c[p] < 0 || r[p] >= W || c[p] >= W;
            if (oob || code_at(f[p], tf[p], 4 + p) != C_EMPTY)
                alive &= ~(1u << p);
            cells[4 + p] = f[p];
            vals[4 + p] = (p == 0) ? C_P1_HEAD : C_P2_HEAD;
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) g[cells[k]] = (unsigned char)vals[k];

        const int n_alive = (int)(alive & 1u) + (int)((alive >> 1) & 1u);
        if (n_alive <= 1) {
            if (n_alive == 1 && (r[0] != r[1] || c[0] != c[1]))
                winner = (alive & 1u) ? 1 : 2;
            done = true;
        }
        if (!done) {
            rw0 = rw1 = P.r_index ? (float)R.eplen : P.r_step;
        } else if (winner == 0) {
            rw0 = rw1 = P.r_draw;
        } else {
            rw0 = (winner == 1) ? P.r_win : P.r_lose;
            rw1 = (winner == 2) ? P.r_win : P.r_lose;
        }
        const uint32_t cnt = ((m >> SLIDE_CNT_SHIFT) & SLIDE_CNT_MASK) + ((marks & 0x3FFFu) ? 1u : 0u) + ((marks >> 14) ? 1u : 0u);
        rec_st = make_uint4(pack_pos(r[0], c[0], r[1], c[1]),
                            alive | (done ? META_DONE : 0u) | ((uint32_t)winner << 4) | ((uint32_t)(a[0] + 1) << 8) |
                                ((uint32_t)(a[1] + 1) << 12) | (cnt << SLIDE_CNT_SHIFT),
                            R.eplen + 1u, R.tick + 1u);
        res |= RES_STORE_ST;
    }
    if (done) res |= RES_DONE;

    uint32_t restart = 0u;
    if (done && (flags & TRON_STEP_AUTORESET)) {                      // ACKTR.py:307-310
        rec_st = make_uint4(R.nstart, META_ALIVE0 | META_ALIVE1, 0u, rec_st.w);
        res |= RES_STORE_ST | RES_RESET;
        const int h1 = cell_index(S, (int)(int8_t)(R.nstart), (int)(int8_t)(R.nstart >> 8));
        const int h2 = cell_index(S, (int)(int8_t)(R.nstart >> 16), (int)(int8_t)(R.nstart >> 24));
        restart = 0x80000000u | (uint32_t)h1 | ((uint32_t)h2 << 14);
    }
    rec_out = make_uint4(res | ((uint32_t)winner << 4), __float_as_uint(rw0), __float_as_uint(rw1), restart ? restart : marks);
}

// One tile of E envs through one step; `smem` is the workgroup's dynamic LDS.  Shared by k_obs (one
// tile per workgroup per launch) and k_obs_roll (workgroups that keep stepping their own tiles).
// keep_tile (persistent rollout, TRON_ROLLOUT_RESIDENT): the tile's LDS copy is left exactly as the next step needs
// it (restarted boards are written back to it); have_tile: it already is, so the tile is not loaded again.
template <bool DO_STEP, bool SLIDING = false>
__device__ __forceinline__ void obs_tile(const Params &P, int E, uint32_t cpe, uint32_t cpe_magic,
                                         const int8_t *__restrict__ actions, uint32_t flags, const StepOut &out,
                                         int tile_idx, unsigned char *smem, bool have_tile = false, bool keep_tile = false,
                                         const float *__restrict__ uniforms = nullptr)
{
    const int G = P.G;
    uint4 *tile = reinterpret_cast<uint4 *>(smem);                  // [E*cpe] player-1 codes
    uint4 *tmpl = tile + (size_t)E * cpe;                           // [cpe] fresh board as player-1 codes
    uint4 *rec_st = tmpl + cpe;                                     // [E]
    uint4 *rec_out = rec_st + E;                                    // [E]
    uint4 *rec_rs = rec_out + E;                                    // [E] new rs4 (restarted envs)
    uint4 *rs_in = rec_rs + E;                                      // [E] rs4 as loaded by wave 1

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int e0 = tile_idx * E;
    const int ne = min(E, P.N - e0);
    const uint32_t nchunks = (uint32_t)ne * cpe;
    int8_t *otile = P.obs_state + (size_t)e0 * 2u * G;              // this tile's [ne][2][G] planes
    const bool autoreset = (flags & TRON_STEP_AUTORESET) != 0u, nonrev = (flags & TRON_STEP_NONREVERSING) != 0u;
    const bool mine = lane < ne;
    const int env = e0 + lane;

    STAMP(0);
    // ---- 1: state words (wave 0: st4, wave 1: rs4), then the tile
    EnvRegs R{};
    uint4 rs = make_uint4(0u, 0u, 0u, 0u);
    if (DO_STEP && mine) {
        if (wave == 0) {
            const uint4 st = P.st4[env];
            R.pos = st.x; R.meta = st.y; R.eplen = st.z; R.tick = st.w;
            if (actions) R.act = reinterpret_cast<const uint16_t *>(actions)[env];
            if (SLIDING) {                                            // the slide's rate: this env's scalars (game.py:83-88,100-112)
                R.envp = P.rs4[env].x;
                R.slide = P.slide[env];
                if (uniforms) {
                    const float2 uu = reinterpret_cast<const float2 *>(uniforms)[env];
                    R.u0 = uu.x;
                    R.u1 = uu.y;
                }
            }
        } else if (wave == 1 && autoreset) {
            rs = P.rs4[env];
        }
    }
    if (DO_STEP && autoreset)
        for (uint32_t d = (uint32_t)tid; d < cpe * 16u; d += BLOCK)
            reinterpret_cast<int8_t *>(tmpl)[d] = (d < (uint32_t)G) ? (P.fresh[d] == TRON_EMPTY ? (int8_t)1 : (int8_t)-1) : (int8_t)0;

    int a[2] = {0, 0};
    float u[2] = {R.u0, R.u1};
    for (uint32_t base = 0; base < nchunks; base += DK * BLOCK) {
        uint4 v[DK];
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            const uint32_t i = base + (uint32_t)tid + (uint32_t)k * BLOCK;
            const uint32_t le = chunk_env(i, cpe, cpe_magic);
            const uint32_t c = (i - le * cpe) * 16u;
            if (i < nchunks && !have_tile) v[k] = load_chunk<true>(otile + (size_t)le * 2u * G + c);   // player-1 plane
        }
        if (base == 0u && DO_STEP && mine) {
            // random-number work in the shadow of the tile load: wave 0 the actions (and slide uniforms), wave 1 the next start
            if (wave == 0) {
                if (!actions || (SLIDING && !uniforms)) {
                    uint32_t x[4];
                    philox4x32_10((uint32_t)env, R.tick, RNG_STEP, 0u, P.seed, P.stream, x);
                    a[0] = draw_action(x[0], (R.meta >> 8) & 0xFu, nonrev);
                    a[1] = draw_action(x[1], (R.meta >> 12) & 0xFu, nonrev);
                    if (SLIDING && !uniforms) {
                        u[0] = (float)(x[2] >> 8) * (1.0f / 16777216.0f);
                        u[1] = (float)(x[3] >> 8) * (1.0f / 16777216.0f);
                    }
                }
                if (actions) {
                    a[0] = (int)(R.act & 3u);
                    a[1] = (int)((R.act >> 8) & 3u);
                }
            } else if (wave == 1 && autoreset) {
                rs_in[lane] = rs;
                const NewGame ng = make_game(P.seed, P.stream, P.W, P.fair, (uint32_t)env, rs.y + 1u);
                rec_rs[lane] = make_uint4(rs.w, rs.y + 1u, pack_pos(ng.r1, ng.c1, ng.r2, ng.c2),
                                          pack_envp(ng.w0, ng.w1, ng.degree));
            }
        }
#pragma unroll
        for (int k = 0; k < DK; ++k) {
            const uint32_t i = base + (uint32_t)tid + (uint32_t)k * BLOCK;
            if (i < nchunks && !have_tile) tile[i] = v[k];
        }
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    if (DO_STEP) {
        // ---- 2: the move, wave 0, one env per lane, LDS only
        if (wave == 0) {
            uint4 rst = make_uint4(0u, 0u, 0u, 0u), ro = make_uint4(0u, 0u, 0u, 0u);
            if (mine) {
                if (autoreset) R.nstart = rs_in[lane].z;
                if (SLIDING) lane_move_codes_slide(P, reinterpret_cast<unsigned char *>(tile + (size_t)lane * cpe), R, a, u, flags, rst, ro);
                else lane_move_codes(P, reinterpret_cast<unsigned char *>(tile + (size_t)lane * cpe), R, a, flags, rst, ro);
            }
            if (lane < E) {
                rec_st[lane] = rst;
                rec_out[lane] = ro;
            }
        }
        STAMP(3);
        __syncthreads();
        STAMP(4);

        // ---- 3a: waves 1-3 write the records out (lane = env)
        if (wave >= 1 && mine) {
            const uint4 ro = rec_out[lane];
            if (wave == 1) {
                if (ro.x & RES_STORE_ST) P.st4[env] = rec_st[lane];
                if (ro.x & RES_RESET) P.rs4[env] = rec_rs[lane];
            } else if (wave == 2) {
                if (out.done) out.done[env] = (int8_t)((ro.x & RES_DONE) != 0u);
                if (out.winner) out.winner[env] = (int8_t)((ro.x >> 4) & 3u);
            } else {
                if (out.reward)
                    reinterpret_cast<float2 *>(out.reward)[env] = make_float2(__uint_as_float(ro.y), __uint_as_float(ro.z));
                if (SLIDING && ro.w && !(ro.w >> 31)) {               // this step's slide marks go to the env's log (see lane_move_codes_slide)
                    const uint32_t m0 = ro.w & 0x3FFFu, m1 = ro.w >> 14;
                    uint32_t at = ((rec_st[lane].y >> SLIDE_CNT_SHIFT) & SLIDE_CNT_MASK) - (m0 ? 1u : 0u) - (m1 ? 1u : 0u);
                    uint16_t *lg = slide_log(P) + (size_t)env * slide_log_len(P.W);
                    if (m0) lg[at++] = (uint16_t)(m0 - 1u);
                    if (m1) lg[at] = (uint16_t)((m1 - 1u) | 0x8000u);
                }
            }
        }
#ifndef TRON_STAMPS
        if (out.totals && wave == 3) {
            const uint32_t f = mine ? rec_out[lane].x : 0u;
            const int wn = ((f & RES_STEPPED) && (f & RES_DONE)) ? (int)((f >> 4) & 3u) : -1;
            const unsigned long long bs = __ballot((f & RES_STEPPED) != 0u);
            const unsigned long long b1 = __ballot(wn == 1), b2 = __ballot(wn == 2), b0 = __ballot(wn == 0);
            if (lane == 0) {
                if (bs) atomicAdd(&out.totals[0], (unsigned long long)__popcll(bs));
                if (b1) atomicAdd(&out.totals[1], (unsigned long long)__popcll(b1));
                if (b2) atomicAdd(&out.totals[2], (unsigned long long)__popcll(b2));
                if (b0) atomicAdd(&out.totals[3], (unsigned long long)__popcll(b0));
            }
        }
#endif
    } else {
        return;                                                        // nothing to re-encode: the planes are the state
    }

    // ---- 3b: the stream: both planes of every chunk
    for (uint32_t i = (uint32_t)tid; i < nchunks; i += BLOCK) {
        const uint32_t le = chunk_env(i, cpe, cpe_magic);
        const uint32_t k = i - le * cpe;
        const uint32_t c = k * 16u;
        const int nb = min(16, G - (int)c);
        uint4 t = tile[i];
        const uint32_t ri = rec_out[le].w;
        if (ri >> 31) {                                                // restarted env: fresh board + heads
            t = tmpl[k];
            const uint32_t d1 = (ri & 0x3FFFu) - c, d2 = ((ri >> 14) & 0x3FFFu) - c;
            // the head cells are EMPTY (code 1) in the template: XOR turns 1 into 10 / -10 (game.py:90-91)
            const uint32_t v1 = (uint32_t)(0x01 ^ 0x0A) << ((d1 & 3u) * 8u), v2 = (uint32_t)(0x01 ^ 0xF6) << ((d2 & 3u) * 8u);
            t.x ^= (d1 < 4u ? v1 : 0u) ^ (d2 < 4u ? v2 : 0u);
            t.y ^= (d1 - 4u < 4u ? v1 : 0u) ^ (d2 - 4u < 4u ? v2 : 0u);
            t.z ^= (d1 - 8u < 4u ? v1 : 0u) ^ (d2 - 8u < 4u ? v2 : 0u);
            t.w ^= (d1 - 12u < 4u ? v1 : 0u) ^ (d2 - 12u < 4u ? v2 : 0u);
            if (keep_tile) tile[i] = t;
        }
        const uint32_t w1[4] = {t.x, t.y, t.z, t.w};
        const uint32_t w2[4] = {swap_codes4(t.x), swap_codes4(t.y), swap_codes4(t.z), swap_codes4(t.w)};
        int8_t *o1 = otile + (size_t)le * 2u * G + c;
        store_chunk<true>(o1, nb, w1);
        store_chunk<true>(o1 + G, nb, w2);
    }
    STAMP(7);
}

template <bool DO_STEP>
__global__ __launch_bounds__(BLOCK) void k_obs(Params P, int E, uint32_t cpe, uint32_t cpe_magic,
                                               const int8_t *__restrict__ actions, uint32_t flags, StepOut out,
                                               int tile0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    obs_tile<DO_STEP>(P, E, cpe, cpe_magic, actions, flags, out, (int)blockIdx.x + tile0, smem);
}
// ... and in the sliding modes (own kernels: mode None's stay as they were, instruction for instruction)
__global__ __launch_bounds__(BLOCK) void k_obs_slide(Params P, int E, uint32_t cpe, uint32_t cpe_magic, const int8_t *__restrict__ actions,
                                                     const float *__restrict__ uniforms, uint32_t flags, StepOut out, int tile0)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    obs_tile<true, true>(P, E, cpe, cpe_magic, actions, flags, out, (int)blockIdx.x + tile0, smem, false, false, uniforms);
}

// The random-action rollout as ONE launch for k_steps steps (tron_rollout_random): envs never interact,
// so a workgroup can step its own tiles k_steps times without waiting for anybody else — there is no
// drain of the whole chip between steps, the load phase of one workgroup overlaps the store phase of
// its neighbours across step boundaries.  Workgroup w owns tiles w, w + gridDim.x, ...; every step
// still reads its tile's state from memory and rewrites both observation planes (the same work and
// the same results, bit for bit, as k_steps launches of k_obs).  Every workgroup runs a fixed trip
// count, so the grid always drains.
__global__ __launch_bounds__(BLOCK) void k_obs_roll(Params P, int E, uint32_t cpe, uint32_t cpe_magic, uint32_t flags,
                                                   StepOut out, int k_steps, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    // the {steps, wins, wins, draws} counters are summed in LDS over the whole launch and flushed once:
    // 4 global atomics per tile per step on the same four words serialise in L2 (98 us per step measured)
    __shared__ unsigned long long acc[4];
    StepOut lo = out;
    if (out.totals) {
        if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
        lo.totals = acc;
        __syncthreads();
    }
    // TRON_ROLLOUT_RESIDENT (one tile per workgroup): the board never leaves LDS between the steps of a launch — it is
    // read from memory once, every step still writes both observation planes (which also are the state in memory)
    const bool resident = (flags & TRON_ROLLOUT_RESIDENT) != 0u && (int)gridDim.x == ntiles;
    flags &= ~TRON_ROLLOUT_RESIDENT;
    // The step's parameters are re-read from the kernel-argument segment at every step (scalar loads through a pointer the
    // compiler cannot see through): kept live across the loop, the 26 words of Params pushed the kernel to 63 SGPR spills —
    // 122 v_readlane per step; re-read, 33 / 24 (and 108 VGPRs instead of 93: four waves per SIMD instead of five), and the
    // rollout is 3.5 % faster at 64 steps per launch, 0.5 - 1.5 % at 20 (same box, A/B, scripts/env_kernarg_ab.sh).
    kernarg_t *kp = (kernarg_t *)__builtin_amdgcn_kernarg_segment_ptr();       // (Params is the first argument)
    for (int s = 0; s < k_steps; ++s)
        for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
            kernarg_t *q = kp;
            asm volatile("" : "+s"(q));
            Params Pl;
            load_params(Pl, q);
            obs_tile<true>(Pl, E, cpe, cpe_magic, nullptr, flags, lo, t, smem, resident && s > 0, resident);
            __syncthreads();        // the tile's LDS is reused; this step's state words are visible to the next
        }
    if (out.totals && threadIdx.x < 4 && acc[threadIdx.x]) atomicAdd(&out.totals[threadIdx.x], acc[threadIdx.x]);
}

// load_params reads sizeof(Params) bytes from offset 0 of the kernel-argument segment: that is Params only while it is the
// kernel's FIRST parameter (arguments are laid out in declaration order from offset 0) and a plain block of bytes.
template <class F> struct first_kernel_arg;
template <class R, class A0, class... A> struct first_kernel_arg<R (*)(A0, A...)> { typedef A0 type; };
static_assert(std::is_same<first_kernel_arg<decltype(&k_obs_roll)>::type, Params>::value,
              "k_obs_roll re-reads Params from kernarg offset 0: Params must stay its first parameter");
static_assert(std::is_trivially_copyable<Params>::value && alignof(Params) <= 8 && sizeof(Params) % 4 == 0,
              "Params is block-copied from the kernel-argument segment with scalar loads");

// the sliding modes' persistent rollout: k_obs_roll's loop on obs_tile<true, true>
#ifndef TRON_SLIDE_KERNARG_REREAD
#define TRON_SLIDE_KERNARG_REREAD 1   // as k_obs_roll: Params re-read from the kernel-argument segment per step (69 -> 41 SGPR spills; temper 0.750 -> 0.757, 0.720 -> 0.734 at 20 steps per launch)
#endif
__global__ __launch_bounds__(BLOCK) void k_obs_roll_slide(Params P, int E, uint32_t cpe, uint32_t cpe_magic, uint32_t flags,
                                                         StepOut out, int k_steps, int ntiles)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long acc[4];           // see k_obs_roll
    StepOut lo = out;
    if (out.totals) {
        if (threadIdx.x < 4) acc[threadIdx.x] = 0ull;
        lo.totals = acc;
        __syncthreads();
    }
    const bool resident = (flags & TRON_ROLLOUT_RESIDENT) != 0u && (int)gridDim.x == ntiles;
    flags &= ~TRON_ROLLOUT_RESIDENT;
#if TRON_SLIDE_KERNARG_REREAD
    kernarg_t *kp = (kernarg_t *)__builtin_amdgcn_kernarg_segment_ptr();       // (Params is the first argument: see k_obs_roll)
#endif
    for (int s = 0; s < k_steps; ++s)
        for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
#if TRON_SLIDE_KERNARG_REREAD
            kernarg_t *q = kp;
            asm volatile("" : "+s"(q));
            Params Pl;
            load_params(Pl, q);
            obs_tile<true, true>(Pl, E, cpe, cpe_magic, nullptr, flags, lo, t, smem, resident && s > 0, resident);
#else
            obs_tile<true, true>(P, E, cpe, cpe_magic, nullptr, flags, lo, t, smem, resident && s > 0, resident);
#endif
            __syncthreads();
        }
    if (out.totals && threadIdx.x < 4 && acc[threadIdx.x]) atomicAdd(&out.totals[threadIdx.x], acc[threadIdx.x]);
}
static_assert(std::is_same<first_kernel_arg<decltype(&k_obs_roll_slide)>::type, Params>::value, "k_obs_roll_slide may re-read Params from kernarg offset 0");

// ------------------------------------------------------------ incremental step --
// Observation-is-state, TRON_STEP_INCREMENTAL: the attached planes already hold the previous
// observation, and a move changes at most 4 cells per plane, so only those cells are written;
// only boards that restart are rewritten in full.  No tile, no LDS staging: ONE ENV PER LANE
// (wave 0 of each 64-env workgroup) reads its state words and two target cells, decides, and
// stores 8 bytes; the four waves then share the rewrites of the restarting boards (coalesced
// 16-byte stores from an LDS template).
// HBM traffic per env-step is ~100 B + 2G per restart instead of 3G — this is a different
// contract from "both planes written every step" and is reported under its own label.
template <int DUMMY>
__global__ __launch_bounds__(BLOCK) void k_inc(Params P, uint32_t cpe, const int8_t *__restrict__ actions,
                                               uint32_t flags, StepOut out)
{
    // 256 threads per 64 envs: wave 0 decides (lane = env), wave 1 draws the speculative next start,
    // then all four waves share the rewrites of the restarting boards.
    __shared__ uint4 tmpl[640];                                   // fresh board as codes (same for both players)
    __shared__ uint4 rec_rs[WAVE];                                // new rs4 of an env, should it restart
    __shared__ uint32_t rec_nstart[WAVE];                         // its cached start positions (rs4.z)
    __shared__ uint32_t rec_heads[WAVE];                          // restart word: 0x80000000 | head1 | head2 << 14
    const int G = P.G, S = P.S, W = P.W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int env = blockIdx.x * WAVE + lane;
    const bool mine = env < P.N;
    const bool autoreset = (flags & TRON_STEP_AUTORESET) != 0u, nonrev = (flags & TRON_STEP_NONREVERSING) != 0u;

    uint4 st = make_uint4(0u, META_DONE, 0u, 0u), rs = make_uint4(0u, 0u, 0u, 0u);
    uint32_t act = 0u;
    if (mine && wave == 0) {
        st = P.st4[env];
        if (actions) act = reinterpret_cast<const uint16_t *>(actions)[env];
    }
    if (mine && wave == 1 && autoreset) rs = P.rs4[env];
    if (autoreset)
        for (uint32_t d = (uint32_t)tid; d < cpe * 16u; d += BLOCK)
            reinterpret_cast<int8_t *>(tmpl)[d] = (d < (uint32_t)G) ? (P.fresh[d] == TRON_EMPTY ? (int8_t)1 : (int8_t)-1) : (int8_t)0;
    if (wave == 1 && autoreset) {
        rec_nstart[lane] = rs.z;
        if (mine) {
            const NewGame ng = make_game(P.seed, P.stream, W, P.fair, (uint32_t)env, rs.y + 1u);
            rec_rs[lane] = make_uint4(rs.w, rs.y + 1u, pack_pos(ng.r1, ng.c1, ng.r2, ng.c2), pack_envp(ng.w0, ng.w1, ng.degree));
        }
    }

    bool restart = false, done = false, stepped = false;
    int winner = 0;
    uint4 new_st = st;
    if (wave == 0) {
        int8_t *o1 = P.obs_state + (size_t)(mine ? env : 0) * 2u * G, *o2 = o1 + G;
        uint32_t m = st.y;
        int r[2] = {(int)(int8_t)(st.x), (int)(int8_t)(st.x >> 16)};
        int c[2] = {(int)(int8_t)(st.x >> 8), (int)(int8_t)(st.x >> 24)};
        done = (m & META_DONE) != 0;
        stepped = mine && !done;
        winner = (int)((m >> 4) & 3u);
        float rw0 = 0.0f, rw1 = 0.0f;
        if (stepped) {
            int a[2];
            if (!actions) {
                uint32_t x[4];
                philox4x32_10((uint32_t)env, st.w, RNG_STEP, 0u, P.seed, P.stream, x);
                a[0] = draw_action(x[0], (m >> 8) & 0xFu, nonrev);
                a[1] = draw_action(x[1], (m >> 12) & 0xFu, nonrev);
            } else {
                a[0] = (int)(act & 3u);
                a[1] = (int)((act >> 8) & 3u);
            }
            int old[2], f[2], tf[2];
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                old[p] = cell_index(S, r[p], c[p]);
                r[p] += (a[p] == 0) ? -1 : (a[p] == 2) ? 1 : 0;          // player.py:124-132
                c[p] += (a[p] == 1) ? 1 : (a[p] == 3) ? -1 : 0;
                f[p] = cell_index(S, r[p], c[p]);
            }
            tf[0] = o1[f[0]];
            tf[1] = o1[f[1]];
#pragma unroll
            for (int p = 0; p < 2; ++p) {                                // game.py:155-156: heads are bodies by now
                if (f[p] == old[0]) tf[p] = -2;
                if (f[p] == old[1]) tf[p] = -3;
            }
            if (f[1] == f[0]) tf[1] = 10;                                // game.py:205-214: P2 tests after P1's head is down
            uint32_t alive = m & 3u;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const bool oob = r[p] < 0 || c[p] < 0 || r[p] >= W || c[p] >= W;
                if (oob || tf[p] != 1) alive &= ~(1u << p);
            }
            const int n_alive = (int)(alive & 1u) + (int)((alive >> 1) & 1u);
            if (n_alive <= 1) {                                          // game.py:264-275
                if (n_alive == 1 && (r[0] != r[1] || c[0] != c[1])) winner = (alive & 1u) ? 1 : 2;
                done = true;
            }
            if (!done) {
                rw0 = rw1 = P.r_index ? (float)st.z : P.r_step;
            } else if (winner == 0) {
                rw0 = rw1 = P.r_draw;
            } else {
                rw0 = (winner == 1) ? P.r_win : P.r_lose;
                rw1 = (winner == 2) ? P.r_win : P.r_lose;
            }
            new_st = make_uint4(pack_pos(r[0], c[0], r[1], c[1]),
                                alive | (done ? META_DONE : 0u) | ((uint32_t)winner << 4) | ((uint32_t)(a[0] + 1) << 8) |
                                    ((uint32_t)(a[1] + 1) << 12),
                                st.z + 1u, st.w + 1u);
            if (!(done && autoreset)) {
                // the four cells, reference order: bodies, P1's head, P2's head — in both planes
                o1[old[0]] = -2; o2[old[0]] = -3;
                o1[old[1]] = -3; o2[old[1]] = -2;
                o1[f[0]] = 10;   o2[f[0]] = -10;
                o1[f[1]] = -10;  o2[f[1]] = 10;
                P.st4[env] = new_st;
            }
        }
        if (mine) {
            if (out.done) out.done[env] = (int8_t)done;
            if (out.winner) out.winner[env] = (int8_t)winner;
            if (out.reward) reinterpret_cast<float2 *>(out.reward)[env] = make_float2(rw0, rw1);
        }
        if (out.totals) {
            const int wn = (stepped && done) ? winner : -1;
            const unsigned long long bs = __ballot(stepped);
            const unsigned long long b1 = __ballot(wn == 1), b2 = __ballot(wn == 2), b0 = __ballot(wn == 0);
            if (lane == 0) {
                if (bs) atomicAdd(&out.totals[0], (unsigned long long)__popcll(bs));
                if (b1) atomicAdd(&out.totals[1], (unsigned long long)__popcll(b1));
                if (b2) atomicAdd(&out.totals[2], (unsigned long long)__popcll(b2));
                if (b0) atomicAdd(&out.totals[3], (unsigned long long)__popcll(b0));
            }
        }
        restart = mine && done && autoreset;
    }
    __syncthreads();                                                 // template, rec_rs, rec_nstart are in LDS
    if (wave == 0) {
        uint32_t hw = 0u;
        if (restart) {                                               // ACKTR.py:307-310; start cached one restart ago
            const uint32_t ns = rec_nstart[lane];
            P.st4[env] = make_uint4(ns, META_ALIVE0 | META_ALIVE1, 0u, new_st.w);
            P.rs4[env] = rec_rs[lane];
            hw = 0x80000000u | (uint32_t)cell_index(S, (int)(int8_t)(ns), (int)(int8_t)(ns >> 8)) |
                 ((uint32_t)cell_index(S, (int)(int8_t)(ns >> 16), (int)(int8_t)(ns >> 24)) << 14);
        }
        rec_heads[lane] = hw;
    }
    if (!autoreset) return;
    __syncthreads();

    // ---- restarting boards: both planes from the template, dealt round-robin to the four waves
    const uint32_t my_hw = rec_heads[lane];
    unsigned long long rmask = __ballot((my_hw >> 31) != 0u);
    int nth = 0;
    while (rmask) {
        const int e = __ffsll((long long)rmask) - 1;
        rmask &= rmask - 1;
        if ((nth++ & 3) != wave) continue;
        const uint32_t hw = rec_heads[e];
        const uint32_t a1 = hw & 0x3FFFu, a2 = (hw >> 14) & 0x3FFFu;
        int8_t *q = P.obs_state + (size_t)(blockIdx.x * WAVE + e) * 2u * G;
        for (uint32_t j = (uint32_t)lane; j < 2u * cpe; j += WAVE) {      // chunk k of plane pl
            const uint32_t pl = j >= cpe ? 1u : 0u, k = j - pl * cpe, cc = k * 16u;
            uint4 t = tmpl[k];
            const uint32_t d1 = a1 - cc, d2 = a2 - cc;
            // template head cells are EMPTY (1): XOR to own head 10 / enemy head -10 per plane (game.py:90-91)
            const uint32_t x1 = (uint32_t)(0x01 ^ (pl ? 0xF6 : 0x0A)) << ((d1 & 3u) * 8u);
            const uint32_t x2 = (uint32_t)(0x01 ^ (pl ? 0x0A : 0xF6)) << ((d2 & 3u) * 8u);
            t.x ^= (d1 < 4u ? x1 : 0u) ^ (d2 < 4u ? x2 : 0u);
            t.y ^= (d1 - 4u < 4u ? x1 : 0u) ^ (d2 - 4u < 4u ? x2 : 0u);
            t.z ^= (d1 - 8u < 4u ? x1 : 0u) ^ (d2 - 8u < 4u ? x2 : 0u);
            t.w ^= (d1 - 12u < 4u ? x1 : 0u) ^ (d2 - 12u < 4u ? x2 : 0u);
            const uint32_t w[4] = {t.x, t.y, t.z, t.w};
            store_chunk<true>(q + (size_t)pl * G + cc, min(16, G - (int)cc), w);
        }
    }
}

// observation-is-state helpers: board images / other formats from the attached planes
__global__ void k_obs_to_grid(Params P, int8_t *__restrict__ grid_out)
{
    const size_t total = (size_t)P.N * P.G;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t e = i / (size_t)P.G, cidx = i - e * (size_t)P.G;
        grid_out[i] = tile_of_code(P.obs_state[e * 2u * P.G + cidx]);
    }
}
// ... and the slide tiles (sliding modes): a logged cell that still holds its player's body code is that player's slide tile
__global__ void k_obs_grid_marks(Params P, int8_t *__restrict__ grid_out)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= P.N) return;
    const uint32_t cnt = (P.st4[env].y >> SLIDE_CNT_SHIFT) & SLIDE_CNT_MASK;
    const uint16_t *lg = slide_log(P) + (size_t)env * slide_log_len(P.W);
    const int8_t *o = P.obs_state + (size_t)env * 2u * P.G;
    for (uint32_t k = 0; k < cnt; ++k) {
        const uint32_t e = lg[k], cell = e & 0x7FFFu, pl = e >> 15;
        if (o[cell] == (pl ? (int8_t)-3 : (int8_t)-2)) grid_out[(size_t)env * P.G + cell] = pl ? TRON_P2_SLIDE : TRON_P1_SLIDE;
    }
}
// attaching to boards that already hold slide tiles (steps were made on the board-owning layout): their log, from the grid
__global__ void k_obs_attach_marks(Params P)
{
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= P.N) return;
    const int8_t *g = P.grid + (size_t)env * P.G;
    uint16_t *lg = slide_log(P) + (size_t)env * slide_log_len(P.W);
    uint32_t cnt = 0u;
    for (int i = 0; i < P.G; ++i) {
        const int8_t t = g[i];
        if (t == TRON_P1_SLIDE || t == TRON_P2_SLIDE) lg[cnt++] = (uint16_t)((uint32_t)i | (t == TRON_P2_SLIDE ? 0x8000u : 0u));
    }
    uint4 st = P.st4[env];
    st.y = (st.y & ~(SLIDE_CNT_MASK << SLIDE_CNT_SHIFT)) | (cnt << SLIDE_CNT_SHIFT);
    P.st4[env] = st;
}
__global__ void k_obs_reset(Params P, const int8_t *__restrict__ mask)
{
    // after k_reset wrote P.grid for the masked envs: re-derive their two planes from it
    const int env = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (env >= P.N) return;
    if (mask && !mask[env]) return;
    const int8_t *g = P.grid + (size_t)env * P.G;
    int8_t *o = P.obs_state + (size_t)env * 2u * P.G;
    for (int i = lane; i < P.G; i += 64) {
        o[i] = code1(g[i], false);
        o[P.G + i] = code1(g[i], true);
    }
}
// planes (util.pop_up [+ prob_map plane]) of every env from the attached code planes
__global__ void k_obs_planes(Params P, int channels, float *__restrict__ out)
{
    const size_t cells = (size_t)P.G, total = (size_t)P.N * 2u * cells;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t k = i / cells, cidx = i - k * cells;          // k = env*2 + player
        const int v = P.obs_state[i];
        float *o = out + k * (size_t)channels * cells + cidx;
        o[0] = (v == -1) ? 1.0f : 0.0f;
        o[cells] = (v == -2) ? 1.0f : (v == 10) ? 10.0f : 0.0f;
        o[2 * cells] = (v == -3) ? 1.0f : (v == -10) ? 10.0f : 0.0f;
        if (channels == 4) o[3 * cells] = (float)degree_slide(P.slide[k >> 1]);
    }
}

// ------------------------------------------------------------- small kernels --
// make_game / Game.__init__ for masked envs: one wave per env.
__global__ __launch_bounds__(BLOCK) void k_reset(Params P, const int8_t *__restrict__ mask,
                                                 const int8_t *__restrict__ start, const int16_t *__restrict__ weight,
                                                 const int16_t *__restrict__ degree)
{
    const int env = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (env >= P.N) return;
    if (mask && !mask[env]) return;
    const uint32_t epi = P.rs4[env].y;
    NewGame ng;
    if (start) {
        ng.r1 = start[4 * env]; ng.c1 = start[4 * env + 1]; ng.r2 = start[4 * env + 2]; ng.c2 = start[4 * env + 3];
        uint32_t x[4];
        philox4x32_10((uint32_t)env, epi, RNG_INIT, 0u, P.seed, P.stream, x);
        ng.w0 = randint_u32(x[0], 40, 101);                              // game.py:83
        ng.w1 = randint_u32(x[1], 40, 101);
        ng.degree = randint_u32(x[2], -30, 30);                          // game.py:87
    } else {
        ng = make_game(P.seed, P.stream, P.W, P.fair, (uint32_t)env, epi);
    }
    if (weight) { ng.w0 = weight[2 * env]; ng.w1 = weight[2 * env + 1]; }
    if (degree) ng.degree = degree[env];
    const int h1 = cell_index(P.S, ng.r1, ng.c1), h2 = cell_index(P.S, ng.r2, ng.c2);
    int8_t *g = P.grid + (size_t)env * P.G;
    for (int i = lane; i < P.G; i += 64) {
        int8_t v = P.fresh[i];
        if (i == h1) v = TRON_P1_HEAD;                                   // game.py:90-91, pps order
        if (i == h2) v = TRON_P2_HEAD;
        g[i] = v;
    }
    if (lane == 0) {
        const uint32_t tick = P.st4[env].w;
        P.st4[env] = make_uint4(pack_pos(ng.r1, ng.c1, ng.r2, ng.c2), META_ALIVE0 | META_ALIVE1, 0u, tick);
        const NewGame nx = make_game(P.seed, P.stream, P.W, P.fair, (uint32_t)env, epi + 1u);   // the next autoreset's game
        P.rs4[env] = make_uint4(pack_envp(ng.w0, ng.w1, ng.degree), epi + 1u, pack_pos(nx.r1, nx.c1, nx.r2, nx.c2),
                                pack_envp(nx.w0, nx.w1, nx.degree));
    }
}

__global__ void k_fresh(int8_t *fresh, int S)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S) return;
    const int r = i / S, c = i - r * S;
    fresh[i] = (r == 0 || r == S - 1 || c == 0 || c == S - 1) ? TRON_WALL : TRON_EMPTY;   // map.py:5-6,48
}

__global__ void k_fill_f64(double *dst, double v, const double *src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src ? src[i] : v;
}

__global__ void k_set_wd(Params P, const int16_t *weight, const int16_t *degree)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.N) return;
    uint32_t ep = P.rs4[i].x;
    if (weight) ep = (ep & 0xFFFF0000u) | (uint32_t)(uint8_t)weight[2 * i] | ((uint32_t)(uint8_t)weight[2 * i + 1] << 8);
    if (degree) ep = (ep & 0xFF00FFFFu) | ((uint32_t)(uint8_t)(int8_t)degree[i] << 16);
    P.rs4[i].x = ep;
}

__global__ void k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, const int8_t *src8,
                         int8_t *dst8, size_t nbytes)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
    if (blockIdx.x == 0)
        for (size_t b = n16 * 16 + threadIdx.x; b < nbytes; b += blockDim.x) dst8[b] = src8[b];
}

__global__ void k_get_state(Params P, int8_t *pos, int8_t *alive, int8_t *dir, int8_t *done, int8_t *winner,
                            int16_t *weight, int16_t *degree, double *slide, uint32_t *counters)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.N) return;
    const uint4 st = P.st4[i], rs = P.rs4[i];
    const uint32_t pw = st.x, m = st.y, ep = rs.x;
    if (pos) reinterpret_cast<uint32_t *>(pos)[i] = pw;
    if (alive) { alive[2 * i] = (int8_t)(m & 1u); alive[2 * i + 1] = (int8_t)((m >> 1) & 1u); }
    if (dir) { dir[2 * i] = (int8_t)((m >> 8) & 7u); dir[2 * i + 1] = (int8_t)((m >> 12) & 7u); }
    if (done) done[i] = (int8_t)((m >> 2) & 1u);
    if (winner) winner[i] = (int8_t)((m >> 4) & 3u);
    if (weight) { weight[2 * i] = (int16_t)(ep & 0xFFu); weight[2 * i + 1] = (int16_t)((ep >> 8) & 0xFFu); }
    if (degree) degree[i] = (int16_t)(int8_t)(ep >> 16);
    if (slide) slide[i] = P.slide[i];
    if (counters) { counters[3 * i] = st.w; counters[3 * i + 1] = rs.y; counters[3 * i + 2] = st.z; }
}

// Map.state_for_player on arbitrary tile images (map.py:67-84)
__global__ void k_encode_codes(const int8_t *__restrict__ tiles, size_t nbytes, int player_is_2, int8_t *__restrict__ out)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * 16;
    for (size_t b = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16; b < nbytes; b += stride) {
        if (b + 16 <= nbytes) {
            const uint4 t = *reinterpret_cast<const uint4 *>(tiles + b);
            *reinterpret_cast<uint4 *>(out + b) = make_uint4(codes4(t.x, player_is_2), codes4(t.y, player_is_2),
                                                             codes4(t.z, player_is_2), codes4(t.w, player_is_2));
        } else {
            for (size_t j = b; j < nbytes; ++j) out[j] = code1(tiles[j], player_is_2);
        }
    }
}

// util.pop_up on code planes (util.py:11-37): (wall, my, enemy)
__global__ void k_pop_up(const int8_t *__restrict__ codes, size_t n, int cells, float *__restrict__ out)
{
    const size_t total = n * (size_t)cells;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const size_t k = i / (size_t)cells, cidx = i - k * (size_t)cells;
        const int v = codes[i];
        float *o = out + k * 3 * (size_t)cells + cidx;
        o[0] = (v == -1) ? 1.0f : 0.0f;
        o[cells] = (v == -2) ? 1.0f : (v == 10) ? 10.0f : 0.0f;
        o[2 * (size_t)cells] = (v == -3) ? 1.0f : (v == -10) ? 10.0f : 0.0f;
    }
}

}  // namespace

// ---------------------------------------------------------------- host side --
struct tron_env {
    Params P;
    int device;
    int E;                    // envs per workgroup tile
    uint32_t cpe, cpe_magic;  // 16-byte chunks per env, ceil(2^32 / cpe)
    size_t smem;              // dynamic LDS bytes
    bool aligned;             // G % 4 == 0: 16-byte global accesses
    void *blob;               // one allocation behind all state arrays
    hipStream_t side;         // TRON_ROLLOUT_TWO_STREAMS: second launch stream + fork/join events, created on first use
    hipEvent_t fork, join;
    int part0, nparts;        // slice of the tiles the next launch covers (0, 1 = all of them)
    int roll_E;               // envs per tile of the persistent rollout (0: not chosen yet), see roll_tile_envs
};

namespace {

inline hipStream_t S_(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

// hipFuncSetAttribute acts on the current device only: remember, per kernel, which devices it was
// prepared on (one bit per device ordinal) instead of a per-process flag.
inline void allow_big_lds(const void *kern, int device, uint64_t &prepared)
{
    const uint64_t bit = 1ull << ((unsigned)device & 63u);
    if (prepared & bit) return;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
        (void)hipGetLastError();
    prepared |= bit;
}

// tiles [tile0, tile0 + blocks) of the handle's current part (all tiles unless a *_part entry point set one)
inline void part_tiles(const tron_env *h, int &tile0, int &blocks)
{
    const int ntiles = (h->P.N + h->E - 1) / h->E;
    const int np = h->nparts > 0 ? h->nparts : 1;
    tile0 = (int)((long long)ntiles * h->part0 / np);
    blocks = (int)((long long)ntiles * (h->part0 + 1) / np) - tile0;
}

template <int FMT, bool DO_STEP, bool ALIGNED>
int launch_one(tron_env *h, const int8_t *actions, const float *uniforms, uint32_t flags, void *obs, StepOut out,
               hipStream_t st)
{
    auto kern = k_tile<FMT, DO_STEP, ALIGNED>;
    static uint64_t prepared = 0;    // per instantiation, one bit per device
    allow_big_lds(reinterpret_cast<const void *>(kern), h->device, prepared);
    int tile0, blocks;
    part_tiles(h, tile0, blocks);
    if (blocks > 0)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), h->smem, st, h->P, h->E, h->cpe, h->cpe_magic, actions,
                           uniforms, flags, obs, out, tile0);
    return launch_status();
}

template <int FMT, bool ALIGNED>
int launch_roll_one(tron_env *h, int k_steps, uint32_t flags, void *obs, StepOut out, hipStream_t st)
{
    auto kern = k_tile_roll<FMT, ALIGNED>;
    static uint64_t prepared = 0;    // per instantiation, one bit per device
    allow_big_lds(reinterpret_cast<const void *>(kern), h->device, prepared);
    const int ntiles = (h->P.N + h->E - 1) / h->E;
    hipLaunchKernelGGL(kern, dim3(ntiles), dim3(BLOCK), h->smem, st, h->P, h->E, h->cpe, h->cpe_magic, flags, obs, out,
                       k_steps, ntiles);
    return launch_status();
}

int launch_roll_fmt(tron_env *h, int fmt, int k_steps, uint32_t flags, void *obs, StepOut out, hipStream_t st)
{
#define TRON_CASE(F)                                                                        \
    case F:                                                                                 \
        return h->aligned ? launch_roll_one<F, true>(h, k_steps, flags, obs, out, st)       \
                          : launch_roll_one<F, false>(h, k_steps, flags, obs, out, st);
    switch (fmt) {
        TRON_CASE(TRON_OBS_NONE)
        TRON_CASE(TRON_OBS_CODES_I8)
        TRON_CASE(TRON_OBS_PLANES3_F32)
        TRON_CASE(TRON_OBS_PLANES4_F32)
    default:
        return TRON_ERR_BAD_ARG;
    }
#undef TRON_CASE
}

template <bool DO_STEP>
int launch_fmt(tron_env *h, int fmt, const int8_t *a, const float *u, uint32_t flags, void *obs, StepOut out,
               hipStream_t st)
{
#define TRON_CASE(F)                                                                          \
    case F:                                                                                   \
        return h->aligned ? launch_one<F, DO_STEP, true>(h, a, u, flags, obs, out, st)        \
                          : launch_one<F, DO_STEP, false>(h, a, u, flags, obs, out, st);
    switch (fmt) {
        TRON_CASE(TRON_OBS_NONE)
        TRON_CASE(TRON_OBS_CODES_I8)
        TRON_CASE(TRON_OBS_PLANES3_F32)
        TRON_CASE(TRON_OBS_PLANES4_F32)
    default:
        return TRON_ERR_BAD_ARG;
    }
#undef TRON_CASE
}

template <bool DO_STEP>
int launch_obs(tron_env *h, const int8_t *actions, uint32_t flags, StepOut out, hipStream_t st, const float *uniforms = nullptr)
{
    int tile0, blocks;
    part_tiles(h, tile0, blocks);
    const size_t smem = ((size_t)h->E + 1u) * h->cpe * 16u + 4u * (size_t)h->E * 16u;
    if (DO_STEP && h->P.mode != TRON_MODE_NONE) {                       // the sliding modes' kernel
        static uint64_t prepared_s = 0;
        allow_big_lds(reinterpret_cast<const void *>(k_obs_slide), h->device, prepared_s);
        if (blocks > 0)
            hipLaunchKernelGGL(k_obs_slide, dim3(blocks), dim3(BLOCK), smem, st, h->P, h->E, h->cpe, h->cpe_magic, actions, uniforms, flags,
                               out, tile0);
        return launch_status();
    }
    auto kern = k_obs<DO_STEP>;
    static uint64_t prepared = 0;    // per instantiation, one bit per device
    allow_big_lds(reinterpret_cast<const void *>(kern), h->device, prepared);
    if (blocks > 0)
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), smem, st, h->P, h->E, h->cpe, h->cpe_magic, actions, flags, out,
                           tile0);
    return launch_status();
}

inline int obs_planes(tron_env *h, int fmt, void *obs, hipStream_t st)
{
    const int ch = (fmt == TRON_OBS_PLANES3_F32) ? 3 : 4;
    hipLaunchKernelGGL(k_obs_planes, dim3(4096), dim3(256), 0, st, h->P, ch, reinterpret_cast<float *>(obs));
    return launch_status();
}

inline bool bad_handle(tron_handle h)
{
    if (!h) return true;
    int dev = -1;
    return hipGetDevice(&dev) != hipSuccess || dev != h->device;
}

}  // namespace

extern "C" {

int tron_abi_version(void) { return TRON_ABI_VERSION; }

const char *tron_strerror(int status)
{
    switch (status) {
    case TRON_OK: return "ok";
    case TRON_ERR_BAD_ARG: return "bad argument";
    case TRON_ERR_NO_DEVICE: return "no HIP device, or the handle's device is not current";
    case TRON_ERR_ALLOC: return "device allocation failed";
    case TRON_ERR_LAUNCH: return "kernel launch failed";
    case TRON_ERR_UNSUPPORTED: return "not supported by this build";
    default: return "unknown status";
    }
}

int tron_synchronize(void *stream)
{
    const hipError_t e = hipStreamSynchronize(S_(stream));
    if (e == hipSuccess) return TRON_OK;
    (void)hipGetLastError();
    return TRON_ERR_LAUNCH;
}

int tron_create(int32_t n_envs, int32_t W, int32_t mode, int32_t fair, uint32_t seed, uint32_t rng_stream,
                tron_handle *out)
{
    if (!out) return TRON_ERR_BAD_ARG;
    *out = nullptr;
    if (n_envs < 1 || W < 2 || W > 96 || mode < TRON_MODE_NONE || mode > TRON_MODE_TEMPER) return TRON_ERR_BAD_ARG;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }

    {   // the slide-threshold table of "temper" mode (g_rate_thr): once per process and device
        static uint64_t filled = 0;
        if (!(filled & (1ull << (dev & 63)))) {
            static float tab[RATE_DEG * RATE_W];
            for (int d = 0; d < RATE_DEG; ++d)
                for (int w = 0; w < RATE_W; ++w) {
                    const double a = (double)((d - 30) - 30) * 0.6;      // game.py:100-102, operation by operation
                    const double b = -a / 100.0;
                    const double c = (double)(70 - (w + 40)) / 100.0;
                    const double rate = b - c;
                    float t = (float)rate;
                    if ((double)t > rate) t = nextafterf(t, -INFINITY);
                    tab[d * RATE_W + w] = t;
                }
            if (hipMemcpyToSymbol(HIP_SYMBOL(g_rate_thr), tab, sizeof(tab)) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_ALLOC; }
            filled |= 1ull << (dev & 63);
        }
    }
    tron_env *h = new (std::nothrow) tron_env();
    if (!h) return TRON_ERR_ALLOC;
    Params &P = h->P;
    P.N = n_envs; P.W = W; P.S = W + 2; P.G = P.S * P.S; P.mode = mode; P.fair = fair ? 1 : 0;
    P.seed = seed; P.stream = rng_stream;
    P.r_step = -1.0f; P.r_win = 100.0f; P.r_lose = -100.0f; P.r_draw = 0.0f; P.r_index = 0;   // DDQN.py:289-305
    h->device = dev;
    h->side = nullptr; h->fork = nullptr; h->join = nullptr; h->part0 = 0; h->nparts = 1;
    h->aligned = (P.G % 4) == 0;
    h->cpe = ((uint32_t)P.G + 15u) / 16u;
    h->cpe_magic = (uint32_t)((0x100000000ull + h->cpe - 1) / h->cpe);   // exact i / cpe for i < 2^32 / cpe

    // tile size: keep the LDS tile near 24 KB so ~6 workgroups share a CU
    int E = (int)((24u * 1024u) / (h->cpe * 16u));
    E = E >= 64 ? 64 : E >= 32 ? 32 : E >= 16 ? 16 : E >= 8 ? 8 : 4;
    if (const char *s = getenv("TRON_TILE_ENVS")) {
        const int v = atoi(s);
        if (v >= 1 && v <= 64 && ((size_t)v + 1u) * h->cpe * 16u + 8192u <= 160u * 1024u) E = v;
    }
    h->E = E;
    h->smem = ((size_t)E + 1u) * h->cpe * 16u + 3u * (size_t)E * 16u + (size_t)E * 4u + 4u * (((size_t)E * h->cpe + 31u) / 32u + 1u) + 16u;

    // one blob: grid (padded for 16-byte over-read) + state words + fresh template
    const size_t N = (size_t)n_envs;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    // (sliding modes: the slide-mark log of the observation-is-state layout sits right behind the slide rates: slide_log())
    const size_t log_bytes = P.mode != TRON_MODE_NONE ? N * (size_t)slide_log_len(P.W) * sizeof(uint16_t) : 0;
    const size_t o_grid = 0, o_st4 = align(o_grid + N * P.G + 64), o_rs4 = align(o_st4 + 16 * N),
                 o_slide = align(o_rs4 + 16 * N), o_log = align(o_slide + 8 * N), o_fresh = align(o_log + log_bytes),
                 total = align(o_fresh + (size_t)P.G + 16);
    char *blob = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&blob), total) != hipSuccess) {
        (void)hipGetLastError();
        delete h;
        return TRON_ERR_ALLOC;
    }
    h->blob = blob;
    P.grid = reinterpret_cast<int8_t *>(blob + o_grid);
    P.st4 = reinterpret_cast<uint4 *>(blob + o_st4);
    P.rs4 = reinterpret_cast<uint4 *>(blob + o_rs4);
    P.slide = reinterpret_cast<double *>(blob + o_slide);
    P.obs_state = nullptr;
    P.fresh = reinterpret_cast<const int8_t *>(blob + o_fresh);
    if (hipMemsetAsync(blob, 0, total, nullptr) != hipSuccess) { (void)hipGetLastError(); }
    hipLaunchKernelGGL(k_fresh, dim3((P.G + 255) / 256), dim3(256), 0, nullptr, const_cast<int8_t *>(P.fresh), P.S);
    hipLaunchKernelGGL(k_fill_f64, dim3((n_envs + 255) / 256), dim3(256), 0, nullptr, P.slide, 0.15,
                       (const double *)nullptr, n_envs);                                     // config.py:31 slide
    if (launch_status() != TRON_OK || hipStreamSynchronize(nullptr) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(blob);
        delete h;
        return TRON_ERR_LAUNCH;
    }
    *out = h;
    return TRON_OK;
}

int tron_destroy(tron_handle h)
{
    if (!h) return TRON_ERR_BAD_ARG;
    if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
    if (h->fork) (void)hipEventDestroy(h->fork);
    if (h->join) (void)hipEventDestroy(h->join);
    (void)hipFree(h->blob);
    delete h;
    return TRON_OK;
}

int tron_info(tron_handle h, int32_t *n_envs, int32_t *W, int32_t *G, int32_t *mode)
{
    if (!h) return TRON_ERR_BAD_ARG;
    if (n_envs) *n_envs = h->P.N;
    if (W) *W = h->P.W;
    if (G) *G = h->P.G;
    if (mode) *mode = h->P.mode;
    return TRON_OK;
}

int tron_set_reward(tron_handle h, float step, float win, float lose, float draw, int32_t step_is_index)
{
    if (!h) return TRON_ERR_BAD_ARG;
    h->P.r_step = step; h->P.r_win = win; h->P.r_lose = lose; h->P.r_draw = draw; h->P.r_index = step_is_index ? 1 : 0;
    return TRON_OK;
}

int tron_set_slide(tron_handle h, double slide, const double *slide_dev, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_fill_f64, dim3((h->P.N + 255) / 256), dim3(256), 0, S_(stream), h->P.slide, slide, slide_dev,
                       h->P.N);
    return launch_status();
}

int tron_set_weight_degree(tron_handle h, const int16_t *weight, const int16_t *degree, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_set_wd, dim3((h->P.N + 255) / 256), dim3(256), 0, S_(stream), h->P, weight, degree);
    return launch_status();
}

int tron_reset(tron_handle h, const int8_t *env_mask, const int8_t *start_pos, const int16_t *weight,
               const int16_t *degree, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    const int per = BLOCK / 64;
    hipLaunchKernelGGL(k_reset, dim3((h->P.N + per - 1) / per), dim3(BLOCK), 0, S_(stream), h->P, env_mask, start_pos,
                       weight, degree);
    if (h->P.obs_state)     // observation-is-state: the masked envs' planes are re-derived from their fresh boards
        hipLaunchKernelGGL(k_obs_reset, dim3((h->P.N + per - 1) / per), dim3(BLOCK), 0, S_(stream), h->P, env_mask);
    return launch_status();
}

int tron_attach_obs_state(tron_handle h, int8_t *obs_codes, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (!obs_codes || (reinterpret_cast<uintptr_t>(obs_codes) & 15u)) return TRON_ERR_BAD_ARG;
    if (!h->aligned || h->P.G >= 0x3FFF) return TRON_ERR_UNSUPPORTED;                 // (cell + 1 travels in 14 bits: restart word, slide marks)
    if (h->P.obs_state) return TRON_ERR_BAD_ARG;                                      // already attached
    h->P.obs_state = obs_codes;
    const int per = BLOCK / 64;     // derive the planes from the boards as they are now
    hipLaunchKernelGGL(k_obs_reset, dim3((h->P.N + per - 1) / per), dim3(BLOCK), 0, S_(stream), h->P,
                       (const int8_t *)nullptr);
    if (h->P.mode != TRON_MODE_NONE)  // slide tiles are not codable (Map.color shows them as bodies): they go to the log (lane_move_codes_slide)
        hipLaunchKernelGGL(k_obs_attach_marks, dim3((h->P.N + 255) / 256), dim3(256), 0, S_(stream), h->P);
    return launch_status();
}

int tron_step_encode(tron_handle h, const int8_t *actions, const float *uniforms, uint32_t flags, int32_t obs_fmt,
                     void *obs, int8_t *out_done, int8_t *out_winner, float *out_reward, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if ((obs_fmt != TRON_OBS_NONE) != (obs != nullptr)) return TRON_ERR_BAD_ARG;
    if (flags & ~(TRON_STEP_AUTORESET | TRON_STEP_INCREMENTAL | TRON_STEP_NONREVERSING)) return TRON_ERR_BAD_ARG;
    if ((flags & TRON_STEP_INCREMENTAL) && (!h->P.obs_state || h->P.mode != TRON_MODE_NONE)) return TRON_ERR_UNSUPPORTED;
    StepOut out{out_done, out_winner, out_reward, nullptr};
    if (h->P.obs_state) {
        if (obs_fmt == TRON_OBS_CODES_I8 && obs != h->P.obs_state) return TRON_ERR_BAD_ARG;   // the attached buffer is the output
        if (obs_fmt < TRON_OBS_NONE || obs_fmt > TRON_OBS_PLANES4_F32) return TRON_ERR_BAD_ARG;
        int rc;
        if (flags & TRON_STEP_INCREMENTAL) {
            hipLaunchKernelGGL((k_inc<0>), dim3((h->P.N + WAVE - 1) / WAVE), dim3(BLOCK), 0, S_(stream), h->P, h->cpe,
                               actions, flags, out);
            rc = launch_status();
        } else {
            rc = launch_obs<true>(h, actions, flags, out, S_(stream), uniforms);
        }
        if (rc != TRON_OK || obs_fmt == TRON_OBS_NONE || obs_fmt == TRON_OBS_CODES_I8) return rc;
        return obs_planes(h, obs_fmt, obs, S_(stream));
    }
    return launch_fmt<true>(h, obs_fmt, actions, uniforms, flags, obs, out, S_(stream));
}

int tron_step(tron_handle h, const int8_t *actions, const float *uniforms, uint32_t flags, int8_t *out_done,
              int8_t *out_winner, float *out_reward, void *stream)
{
    return tron_step_encode(h, actions, uniforms, flags, TRON_OBS_NONE, nullptr, out_done, out_winner, out_reward,
                            stream);
}

int tron_part_range(tron_handle h, int32_t part, int32_t nparts, int32_t *first_env, int32_t *n_envs)
{
    if (!h || nparts < 1 || part < 0 || part >= nparts) return TRON_ERR_BAD_ARG;
    const int ntiles = (h->P.N + h->E - 1) / h->E;
    const long long t0 = (long long)ntiles * part / nparts, t1 = (long long)ntiles * (part + 1) / nparts;
    const long long e0 = t0 * h->E, e1 = t1 * h->E < h->P.N ? t1 * h->E : h->P.N;
    if (first_env) *first_env = (int32_t)e0;
    if (n_envs) *n_envs = (int32_t)(e1 - e0);
    return TRON_OK;
}

int tron_step_encode_part(tron_handle h, int32_t part, int32_t nparts, const int8_t *actions, const float *uniforms,
                          uint32_t flags, int32_t obs_fmt, void *obs, int8_t *out_done, int8_t *out_winner,
                          float *out_reward, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (nparts < 1 || part < 0 || part >= nparts) return TRON_ERR_BAD_ARG;
    if (flags & TRON_STEP_INCREMENTAL) return TRON_ERR_UNSUPPORTED;
    if (h->P.obs_state && (obs_fmt == TRON_OBS_PLANES3_F32 || obs_fmt == TRON_OBS_PLANES4_F32)) return TRON_ERR_UNSUPPORTED;
    h->part0 = part;
    h->nparts = nparts;
    const int rc = tron_step_encode(h, actions, uniforms, flags, obs_fmt, obs, out_done, out_winner, out_reward, stream);
    h->part0 = 0;
    h->nparts = 1;
    return rc;
}

int tron_encode(tron_handle h, int32_t obs_fmt, void *obs, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (obs_fmt == TRON_OBS_NONE || !obs) return TRON_ERR_BAD_ARG;
    StepOut out{nullptr, nullptr, nullptr, nullptr};
    if (h->P.obs_state) {
        if (obs_fmt == TRON_OBS_CODES_I8) {
            if (obs == h->P.obs_state) return TRON_OK;            // the attached planes are always current
            const size_t nbytes = (size_t)h->P.N * 2u * h->P.G;
            if (reinterpret_cast<uintptr_t>(obs) & 15u) return TRON_ERR_BAD_ARG;
            hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, S_(stream), reinterpret_cast<const uint4 *>(h->P.obs_state),
                               reinterpret_cast<uint4 *>(obs), nbytes / 16, h->P.obs_state, reinterpret_cast<int8_t *>(obs),
                               nbytes);
            return launch_status();
        }
        if (obs_fmt != TRON_OBS_PLANES3_F32 && obs_fmt != TRON_OBS_PLANES4_F32) return TRON_ERR_BAD_ARG;
        return obs_planes(h, obs_fmt, obs, S_(stream));
    }
    return launch_fmt<false>(h, obs_fmt, nullptr, nullptr, 0u, obs, out, S_(stream));
}

int tron_get_grid(tron_handle h, int8_t *grid_out, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (!grid_out) return TRON_ERR_BAD_ARG;
    if (h->P.obs_state) {
        hipLaunchKernelGGL(k_obs_to_grid, dim3(2048), dim3(256), 0, S_(stream), h->P, grid_out);
        if (h->P.mode != TRON_MODE_NONE)
            hipLaunchKernelGGL(k_obs_grid_marks, dim3((h->P.N + 255) / 256), dim3(256), 0, S_(stream), h->P, grid_out);
        return launch_status();
    }
    const size_t nbytes = (size_t)h->P.N * h->P.G;
    const bool aligned = (reinterpret_cast<uintptr_t>(grid_out) & 15u) == 0;
    const size_t n16 = aligned ? nbytes / 16 : 0;
    const int blocks = (int)((n16 / 256 < 2048 ? n16 / 256 : 2048) + 1);
    hipLaunchKernelGGL(k_copy16, dim3(blocks), dim3(256), 0, S_(stream), reinterpret_cast<const uint4 *>(h->P.grid),
                       reinterpret_cast<uint4 *>(grid_out), n16, h->P.grid, grid_out, nbytes);
    return launch_status();
}

int tron_get_state(tron_handle h, int8_t *pos, int8_t *alive, int8_t *dir, int8_t *done, int8_t *winner,
                   int16_t *weight, int16_t *degree, double *slide, uint32_t *counters, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_get_state, dim3((h->P.N + 255) / 256), dim3(256), 0, S_(stream), h->P, pos, alive, dir, done,
                       winner, weight, degree, slide, counters);
    return launch_status();
}

namespace {

// k_steps steps of the observation-is-state kernel as persistent launches (k_obs_roll) of at most
// TRON_ROLLOUT_CHUNK steps.  One tile per workgroup: with more workgroups than the chip holds at once the
// late ones start as the early ones finish their steps — measured best at 65 536 x 24x24 (22.1 us per
// step; 23.5-24.0 us with 1024-1536 workgroups walking several tiles each; 27.2 us with one launch per
// step).  TRON_ROLL_E / TRON_ROLL_GRID / TRON_ROLL_CHUNK override tile size, grid and steps per launch.
// Tile size of the persistent rollout.  One tile per workgroup, and the chip holds `slots` workgroups at once
// (occupancy x CUs: 5 x 256 at 24x24), so the launch runs in ceil(ntiles / slots) rounds and the last one should be
// full: at 65 536 envs 32-env tiles are 2 048 workgroups = 1.6 rounds (the tail runs 3 per CU, latency-bound), 26-env
// tiles are 2 521 = 1.97 rounds — 21.4 instead of 21.9 us per step (gpurun sweep, round 2).  Picks the E in
// [3/4 E0, E0] with the fullest last round (E0 = the per-step kernels' tile); small batches that fit in one round
// keep E0.  (With this file's current k_obs_roll — occupancy 4 — 32-env tiles are exactly two rounds and win the sweep.)
int roll_tile_envs(const tron_env *h)
{
    const int E0 = h->E;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, h->device) != hipSuccess) { (void)hipGetLastError(); return E0; }
    int best = E0;
    double best_fill = -1.0;
    for (int E = E0; E >= (3 * E0 + 3) / 4 && E >= 1; --E) {
        const size_t smem = ((size_t)E + 1u) * h->cpe * 16u + 4u * (size_t)E * 16u;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_obs_roll, BLOCK, smem) != hipSuccess || per_cu < 1) {
            (void)hipGetLastError();
            continue;
        }
        const double slots = (double)per_cu * prop.multiProcessorCount;
        const double rounds = ((h->P.N + E - 1) / E) / slots;
        if (rounds <= 1.0) return E == E0 ? E0 : best;                  // everything resident at once: nothing to balance
        const double fill = rounds / (double)(long long)(rounds + 0.999999);
        if (fill > best_fill + 0.02) { best_fill = fill; best = E; }    // prefer the larger tile unless clearly fuller
    }
    return best;
}

int rollout_persistent(tron_env *h, int32_t k_steps, uint32_t flags, StepOut out, hipStream_t st)
{
    static int env_e = 0, env_grid = 0, chunk = TRON_ROLLOUT_CHUNK;
    static bool probed = false;
    if (!probed) {
        if (const char *v = getenv("TRON_ROLL_E")) env_e = atoi(v);
        if (const char *v = getenv("TRON_ROLL_GRID")) env_grid = atoi(v);
        if (const char *v = getenv("TRON_ROLL_CHUNK")) chunk = atoi(v) > 0 ? atoi(v) : TRON_ROLLOUT_CHUNK;
        probed = true;
    }
    static uint64_t prepared = 0;
    allow_big_lds(reinterpret_cast<const void *>(k_obs_roll), h->device, prepared);
    if (!h->roll_E) h->roll_E = roll_tile_envs(h);
    // the resident variant is store-bound and measured best on the per-step tile (3.51 vs 3.21 G env-steps/s at 26)
    // ... and so is a call that fits ONE launch (k_steps <= chunk): measured at 65 536 x 24x24, a single 20- / 64-step launch
    // runs 2.71-2.74 / 2.92 G env-steps/s on 32-env tiles against 2.64-2.65 / 2.84 on 26-env ones, while five 64-step
    // launches back to back run 3.00 against 3.05 — the full last round pays off when the next launch follows at once
    const int E = env_e > 0 ? env_e : ((flags & TRON_ROLLOUT_RESIDENT) || k_steps <= chunk) ? h->E : h->roll_E;
    const size_t smem = ((size_t)E + 1u) * h->cpe * 16u + 4u * (size_t)E * 16u;
    if (smem > 160u * 1024u) return TRON_ERR_BAD_ARG;
    const int ntiles = (h->P.N + E - 1) / E;
    const int grid = (env_grid > 0 && env_grid < ntiles) ? env_grid : ntiles;
    const bool sliding = h->P.mode != TRON_MODE_NONE;
    if (sliding) {
        static uint64_t prepared_s = 0;
        allow_big_lds(reinterpret_cast<const void *>(k_obs_roll_slide), h->device, prepared_s);
    }
    for (int left = k_steps; left > 0; left -= chunk) {
        if (sliding)
            hipLaunchKernelGGL(k_obs_roll_slide, dim3(grid), dim3(BLOCK), smem, st, h->P, E, h->cpe, h->cpe_magic, flags, out,
                               left < chunk ? left : chunk, ntiles);
        else
            hipLaunchKernelGGL(k_obs_roll, dim3(grid), dim3(BLOCK), smem, st, h->P, E, h->cpe, h->cpe_magic, flags, out,
                               left < chunk ? left : chunk, ntiles);
        if (launch_status() != TRON_OK) return TRON_ERR_LAUNCH;
    }
    return TRON_OK;
}

int rollout_launches(tron_env *h, int32_t k_steps, uint32_t flags, int32_t obs_fmt, void *obs, StepOut out,
                     hipStream_t st)
{
    static const bool env_per_step = getenv("TRON_ROLL_PER_STEP") != nullptr;   // A/B switch: one launch per step
    const bool two_streams = (flags & TRON_ROLLOUT_TWO_STREAMS) != 0u;
    if (!(h->P.obs_state && !(flags & (TRON_ROLLOUT_PER_STEP | TRON_ROLLOUT_TWO_STREAMS)))) flags &= ~TRON_ROLLOUT_RESIDENT;   // persistent obs-is-state launches only
    const bool per_step = env_per_step || two_streams || (flags & TRON_ROLLOUT_PER_STEP) != 0u;
    flags &= ~(TRON_ROLLOUT_PER_STEP | TRON_ROLLOUT_TWO_STREAMS);
    if (h->P.obs_state && !per_step && k_steps > 1) {
        const int rc = rollout_persistent(h, k_steps, flags, out, st);
        if (rc != TRON_OK) return rc;
        k_steps = 0;
        if (obs_fmt == TRON_OBS_PLANES3_F32 || obs_fmt == TRON_OBS_PLANES4_F32) return obs_planes(h, obs_fmt, obs, st);
        return TRON_OK;
    }
    flags &= ~TRON_ROLLOUT_RESIDENT;                               // only k_obs_roll knows it
    if (!h->P.obs_state && !per_step && k_steps > 1) {             // board-owning layout: same idea, k_tile_roll
        for (int left = k_steps; left > 0; left -= TRON_ROLLOUT_CHUNK) {
            const int rc = launch_roll_fmt(h, obs_fmt, left < TRON_ROLLOUT_CHUNK ? left : TRON_ROLLOUT_CHUNK, flags, obs, out, st);
            if (rc != TRON_OK) return rc;
        }
        return TRON_OK;
    }
    if (two_streams && k_steps > 0) {
        // The env batch as two independent halves, each a launch sequence of its own on its own stream: a half's
        // step s+1 only depends on its own step s, so the drain of one half's launch overlaps the ramp-up of the
        // other's.  Fork / join events order both against what came before and comes after on `st`.
        if (!h->side) {
            if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess ||
                hipEventCreateWithFlags(&h->fork, hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&h->join, hipEventDisableTiming) != hipSuccess) {
                (void)hipGetLastError();
                return TRON_ERR_ALLOC;
            }
        }
        if (hipEventRecord(h->fork, st) != hipSuccess || hipStreamWaitEvent(h->side, h->fork, 0) != hipSuccess) {
            (void)hipGetLastError();
            return TRON_ERR_LAUNCH;
        }
        h->nparts = 2;
        int rc = TRON_OK;
        for (int k = 0; k < k_steps && rc == TRON_OK; ++k)
            for (int part = 0; part < 2 && rc == TRON_OK; ++part) {
                h->part0 = part;
                hipStream_t s2 = part ? h->side : st;
                rc = h->P.obs_state ? launch_obs<true>(h, nullptr, flags, out, s2)
                                    : launch_fmt<true>(h, obs_fmt, nullptr, nullptr, flags, obs, out, s2);
            }
        h->part0 = 0;
        h->nparts = 1;
        if (hipEventRecord(h->join, h->side) != hipSuccess || hipStreamWaitEvent(st, h->join, 0) != hipSuccess) {
            (void)hipGetLastError();
            return TRON_ERR_LAUNCH;
        }
        if (rc != TRON_OK) return rc;
    } else {
        for (int k = 0; k < k_steps; ++k) {
            const int rc = h->P.obs_state ? launch_obs<true>(h, nullptr, flags, out, st)
                                          : launch_fmt<true>(h, obs_fmt, nullptr, nullptr, flags, obs, out, st);
            if (rc != TRON_OK) return rc;
        }
    }
    if (h->P.obs_state && (obs_fmt == TRON_OBS_PLANES3_F32 || obs_fmt == TRON_OBS_PLANES4_F32) && k_steps > 0)
        return obs_planes(h, obs_fmt, obs, st);
    return TRON_OK;
}

}  // namespace

int tron_rollout_random(tron_handle h, int32_t k_steps, uint32_t flags, int32_t obs_fmt, void *obs,
                        unsigned long long *totals, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (k_steps < 0 || (obs_fmt != TRON_OBS_NONE) != (obs != nullptr)) return TRON_ERR_BAD_ARG;
    if (flags & ~(TRON_STEP_NONREVERSING | TRON_ROLLOUT_PER_STEP | TRON_ROLLOUT_TWO_STREAMS | TRON_ROLLOUT_RESIDENT))
        return TRON_ERR_BAD_ARG;   // autoreset is implied
    StepOut out{nullptr, nullptr, nullptr, totals};
    if (h->P.obs_state && obs_fmt == TRON_OBS_CODES_I8 && obs != h->P.obs_state) return TRON_ERR_BAD_ARG;
    return rollout_launches(h, k_steps, flags | TRON_STEP_AUTORESET, obs_fmt, obs, out, S_(stream));
}

int tron_minimax_actions(tron_handle h, int32_t player, int32_t depth, int32_t mode, int8_t *out_actions,
                         int32_t *out_values, int8_t *out_expanded, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (!out_actions || (player != 1 && player != 2)) return TRON_ERR_BAD_ARG;
    if (mode != TRON_MINIMAX_VORONOI && mode != TRON_MINIMAX_DISTWALL) return TRON_ERR_BAD_ARG;
    if (depth != 2 || h->P.S > 64) return TRON_ERR_UNSUPPORTED;
    MinimaxSrc src{};
    if (h->P.obs_state) {       // the attached planes are the boards: player p's plane is its observation
        src.codes = h->P.obs_state + (size_t)(player - 1) * (size_t)h->P.G;
        src.stride = 2u * (size_t)h->P.G;
    } else {
        src.grid = h->P.grid;
    }
    src.player = player;
    src.st4 = h->P.st4;
    src.seed = h->P.seed;
    src.stream = h->P.stream;
    return launch_minimax(src, h->P.N, h->P.S, mode, out_actions, out_values, out_expanded, S_(stream));
}

int tron_encode_codes(const int8_t *tiles, int64_t n, int32_t cells, int32_t player, int8_t *codes_out, void *stream)
{
    if (!tiles || !codes_out || n < 0 || cells < 1 || (player != 1 && player != 2)) return TRON_ERR_BAD_ARG;
    const size_t nbytes = (size_t)n * (size_t)cells;
    if (nbytes == 0) return TRON_OK;
    if ((reinterpret_cast<uintptr_t>(tiles) | reinterpret_cast<uintptr_t>(codes_out)) & 15u) return TRON_ERR_BAD_ARG;
    size_t blocks = (nbytes / 16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_encode_codes, dim3((unsigned)blocks), dim3(256), 0, S_(stream), tiles, nbytes, player == 2,
                       codes_out);
    return launch_status();
}

int tron_pop_up(const int8_t *codes, int64_t n, int32_t cells, float *planes_out, void *stream)
{
    if (!codes || !planes_out || n < 0 || cells < 1) return TRON_ERR_BAD_ARG;
    if (n == 0) return TRON_OK;
    size_t blocks = ((size_t)n * cells + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_pop_up, dim3((unsigned)blocks), dim3(256), 0, S_(stream), codes, (size_t)n, cells, planes_out);
    return launch_status();
}

}  // extern "C"
