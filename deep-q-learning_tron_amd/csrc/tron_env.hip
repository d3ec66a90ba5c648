// gfx950 kernels + C ABI for the vectorised TRON env (include/tron_hip.h).
//
// Layout.  Everything is env-major: grid[N][G] int8, obs[N][2][...].  One
// workgroup owns a tile of E consecutive envs:
//   A  stream the tile's E*G grid bytes HBM -> LDS with 16-byte loads (coalesced:
//      the tile is one contiguous, 16-byte aligned span because E % 16 == 0);
//   B  wave 0 plays the move: ONE ENV PER LANE, neighbourhood reads and trail
//      writes hit the LDS copy; the <=6 dirty cells go back to HBM as bytes;
//   B2 finished envs (autoreset) get a fresh board: one wave per env rewrites
//      its G bytes in LDS and HBM;
//   C  every thread encodes 16 output bytes per iteration from the LDS tile
//      (v_perm_b32 as an 8-entry byte LUT) and stores them with 16-byte stores.
// HBM traffic per env-step: read G, write 2G (codes) — the algorithmic minimum.
#include "tron_device.hpp"
#include "../../include/tron_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <new>

using namespace tron;

namespace {

constexpr int BLOCK = 256;

struct StepOut {
    int8_t *done;
    int8_t *winner;
    float *reward;
    unsigned long long *totals;
};

__device__ __forceinline__ int cell_index(int S, int r, int c) { return (r + 1) * S + (c + 1); }

// ------------------------------------------------------------------ phase B --
// One lane = one env.  g points at this env's G bytes in LDS.
// Follows Game.next_frame + Game.step (game.py:149-277); see DESIGN.md §4 for
// the rule list.  res.rinfo is the reset word for phase B2 (0 = no reset).
struct LaneResult {
    uint32_t rinfo;
    int stepped, done, winner;
};
__device__ inline LaneResult lane_step(const Params &P, unsigned char *g, int env, const int8_t *actions,
                                     const float *uniforms, uint32_t flags, const StepOut &out)
{
    const int S = P.S, W = P.W;
    const uint32_t pw = P.pos[env];
    uint32_t m = P.meta[env];
    int r[2] = {(int)(int8_t)(pw), (int)(int8_t)(pw >> 16)};
    int c[2] = {(int)(int8_t)(pw >> 8), (int)(int8_t)(pw >> 24)};
    bool done = (m & META_DONE) != 0;
    int winner = (int)((m >> 4) & 3u);
    float rw0 = 0.0f, rw1 = 0.0f;
    int8_t *ggrid = P.grid + (size_t)env * P.G;
    const bool stepped = !done;

    if (!done) {
        const uint32_t tick = P.tick[env];
        const uint32_t eplen = P.eplen[env];
        int a[2];
        float u[2] = {0.0f, 0.0f};
        const bool sliding = (P.mode != TRON_MODE_NONE);
        if (!actions || (sliding && !uniforms)) {
            uint32_t x[4];
            philox4x32_10((uint32_t)env, tick, RNG_STEP, 0u, P.seed, P.stream, x);
            a[0] = (int)(x[0] & 3u);
            a[1] = (int)(x[1] & 3u);
            u[0] = (float)(x[2] >> 8) * (1.0f / 16777216.0f);
            u[1] = (float)(x[3] >> 8) * (1.0f / 16777216.0f);
        }
        if (actions) {
            const uint16_t aw = reinterpret_cast<const uint16_t *>(actions)[env];
            a[0] = (int)(aw & 3u);
            a[1] = (int)((aw >> 8) & 3u);
        }
        if (sliding && uniforms) {
            const float2 uu = reinterpret_cast<const float2 *>(uniforms)[env];
            u[0] = uu.x;
            u[1] = uu.y;
        }

        int dirty[6];
        // game.py:155-156 — both heads turn into bodies before anyone moves
        dirty[0] = cell_index(S, r[0], c[0]);
        dirty[1] = cell_index(S, r[1], c[1]);
        g[dirty[0]] = (unsigned char)TRON_P1_BODY;
        g[dirty[1]] = (unsigned char)TRON_P2_BODY;
        dirty[2] = dirty[0];
        dirty[3] = dirty[1];

        // game.py:158-178 — advance (player.py:124-132), optional slide, player order
        const uint32_t ep = P.envp[env];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int dr = (a[p] == 0) ? -1 : (a[p] == 2) ? 1 : 0;   // UP / DOWN
            const int dc = (a[p] == 1) ? 1 : (a[p] == 3) ? -1 : 0;   // RIGHT / LEFT
            int nr = r[p] + dr, nc = c[p] + dc;
            if (sliding) {
                if (nr >= 0 && nc >= 0 && nr < W && nc < W) {
                    const int idx = cell_index(S, nr, nc);
                    if (g[idx] == (unsigned char)TRON_EMPTY) {
                        const double rate = (P.mode == TRON_MODE_ICE)
                                                ? P.slide[env]
                                                : get_rate((int)(int8_t)(ep >> 16), (int)((ep >> (8 * p)) & 0xFFu));
                        if ((double)u[p] <= rate) {                 // game.py:169
                            g[idx] = (unsigned char)(p == 0 ? TRON_P1_SLIDE : TRON_P2_SLIDE);
                            dirty[2 + p] = idx;
                            nr += dr;
                            nc += dc;
                        }
                    }
                }
            }
            r[p] = nr;
            c[p] = nc;
        }

        // game.py:205-214 — collisions in player order; head written in every branch
        uint32_t alive = m & 3u;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int idx = cell_index(S, r[p], c[p]);
            const bool oob = r[p] < 0 || c[p] < 0 || r[p] >= W || c[p] >= W;
            if (oob || g[idx] != (unsigned char)TRON_EMPTY)
                alive &= ~(1u << p);
            g[idx] = (unsigned char)(p == 0 ? TRON_P1_HEAD : TRON_P2_HEAD);
            dirty[4 + p] = idx;
        }

        // game.py:264-275 — done / winner
        const int n_alive = (int)(alive & 1u) + (int)((alive >> 1) & 1u);
        if (n_alive <= 1) {
            if (n_alive == 1 && (r[0] != r[1] || c[0] != c[1]))
                winner = (alive & 1u) ? 1 : 2;
            done = true;
        }

        // rewards: util.py:87-94 / DDQN.py:289-305 / DQN.py:224-241
        if (!done) {
            rw0 = rw1 = P.r_index ? (float)eplen : P.r_step;
        } else if (winner == 0) {
            rw0 = rw1 = P.r_draw;
        } else {
            rw0 = (winner == 1) ? P.r_win : P.r_lose;
            rw1 = (winner == 2) ? P.r_win : P.r_lose;
        }

        m = alive | (done ? META_DONE : 0u) | ((uint32_t)winner << 4) | ((uint32_t)(a[0] + 1) << 8) |
            ((uint32_t)(a[1] + 1) << 12);
        P.tick[env] = tick + 1u;
        P.eplen[env] = eplen + 1u;

        const bool will_reset = done && (flags & TRON_STEP_AUTORESET);
        if (!will_reset) {
            // final values of the touched cells (order-free: duplicates store the same byte)
#pragma unroll
            for (int k = 0; k < 6; ++k)
                ggrid[dirty[k]] = (int8_t)g[dirty[k]];
            P.pos[env] = (uint32_t)(uint8_t)r[0] | ((uint32_t)(uint8_t)c[0] << 8) |
                         ((uint32_t)(uint8_t)r[1] << 16) | ((uint32_t)(uint8_t)c[1] << 24);
            P.meta[env] = m;
        }
    }

    if (out.done) out.done[env] = (int8_t)done;
    if (out.winner) out.winner[env] = (int8_t)winner;
    if (out.reward) reinterpret_cast<float2 *>(out.reward)[env] = make_float2(rw0, rw1);

    uint32_t rinfo = 0u;
    if (done && (flags & TRON_STEP_AUTORESET)) {                    // ACKTR.py:307-310
        const uint32_t epi = P.episode[env];
        const NewGame ng = make_game(P, (uint32_t)env, epi);
        P.pos[env] = (uint32_t)(uint8_t)ng.r1 | ((uint32_t)(uint8_t)ng.c1 << 8) |
                     ((uint32_t)(uint8_t)ng.r2 << 16) | ((uint32_t)(uint8_t)ng.c2 << 24);
        P.meta[env] = META_ALIVE0 | META_ALIVE1;
        P.envp[env] = (uint32_t)ng.w0 | ((uint32_t)ng.w1 << 8) | ((uint32_t)(uint8_t)(int8_t)ng.degree << 16);
        P.episode[env] = epi + 1u;
        P.eplen[env] = 0u;
        rinfo = 0x80000000u | (uint32_t)cell_index(S, ng.r1, ng.c1) | ((uint32_t)cell_index(S, ng.r2, ng.c2) << 14);
    }
    return LaneResult{rinfo, (int)stepped, (int)done, winner};
}

// ---------------------------------------------------------------- the kernel --
template <int FMT, bool FAST, bool DO_STEP>
__global__ __launch_bounds__(BLOCK) void k_step_encode(Params P, int E, const int8_t *__restrict__ actions,
                                                       const float *__restrict__ uniforms, uint32_t flags,
                                                       void *__restrict__ obs, StepOut out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int G = P.G;
    const int tile_cap = (E * G + 15) & ~15;
    unsigned char *tile = smem;
    uint32_t *rinfo = reinterpret_cast<uint32_t *>(smem + tile_cap);   // [E]
    float *plane4 = reinterpret_cast<float *>(rinfo + E);              // [E]

    const int tid = threadIdx.x;
    const int e0 = blockIdx.x * E;
    const int ne = min(E, P.N - e0);
    const size_t gbase = (size_t)e0 * G;
    const int nbytes = ne * G;

    // ---- A: tile HBM -> LDS (the grid allocation is padded, over-read is safe)
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(P.grid + gbase);
        uint4 *dst = reinterpret_cast<uint4 *>(tile);
        const int nch = (nbytes + 15) >> 4;
        for (int i = tid; i < nch; i += 4 * BLOCK) {
            uint4 v0, v1, v2, v3;
            const int i1 = i + BLOCK, i2 = i + 2 * BLOCK, i3 = i + 3 * BLOCK;
            v0 = src[i];
            if (i1 < nch) v1 = src[i1];
            if (i2 < nch) v2 = src[i2];
            if (i3 < nch) v3 = src[i3];
            dst[i] = v0;
            if (i1 < nch) dst[i1] = v1;
            if (i2 < nch) dst[i2] = v2;
            if (i3 < nch) dst[i3] = v3;
        }
    }
    if (FMT == TRON_OBS_PLANES4_F32 && tid < ne)
        plane4[tid] = (float)degree_slide(P.slide[e0 + tid]);           // game.py:124-132
    __syncthreads();

    if (DO_STEP) {
        // ---- B: one env per lane (wave 0 covers E <= 64 envs)
        if (tid < E) {
            LaneResult lr{0u, 0, 0, 0};
            if (tid < ne)
                lr = lane_step(P, tile + tid * G, e0 + tid, actions, uniforms, flags, out);
            rinfo[tid] = lr.rinfo;
            if (out.totals) {
                // {env_steps, p1_wins, p2_wins, draws}: one atomic per counter per workgroup
                const int wn = (lr.stepped && lr.done) ? lr.winner : -1;
                const unsigned long long bs = __ballot(lr.stepped != 0);
                const unsigned long long b1 = __ballot(wn == 1), b2 = __ballot(wn == 2), b0 = __ballot(wn == 0);
                if (tid == 0) {
                    if (bs) atomicAdd(&out.totals[0], (unsigned long long)__popcll(bs));
                    if (b1) atomicAdd(&out.totals[1], (unsigned long long)__popcll(b1));
                    if (b2) atomicAdd(&out.totals[2], (unsigned long long)__popcll(b2));
                    if (b0) atomicAdd(&out.totals[3], (unsigned long long)__popcll(b0));
                }
            }
        }
        __syncthreads();

        // ---- B2: fresh boards for finished envs, one wave per env
        if (flags & TRON_STEP_AUTORESET) {
            const int wave = tid >> 6, lane = tid & 63;
            for (int e = wave; e < ne; e += BLOCK / 64) {
                const uint32_t ri = rinfo[e];
                if (!(ri >> 31)) continue;                               // wave-uniform
                const int h1 = (int)(ri & 0x3FFFu), h2 = (int)((ri >> 14) & 0x3FFFu);
                if (FAST) {
                    const int D = G >> 2;
                    const uint32_t *fresh32 = reinterpret_cast<const uint32_t *>(P.fresh);
                    uint32_t *t32 = reinterpret_cast<uint32_t *>(tile) + e * D;
                    uint32_t *g32 = reinterpret_cast<uint32_t *>(P.grid + gbase) + (size_t)e * D;
                    for (int d = lane; d < D; d += 64) {
                        uint32_t v = fresh32[d];
                        if ((h1 >> 2) == d) v |= (uint32_t)TRON_P1_HEAD << ((h1 & 3) * 8);   // EMPTY is 0
                        if ((h2 >> 2) == d) v |= (uint32_t)TRON_P2_HEAD << ((h2 & 3) * 8);
                        t32[d] = v;
                        g32[d] = v;
                    }
                } else {
                    unsigned char *t8 = tile + e * G;
                    int8_t *g8 = P.grid + gbase + (size_t)e * G;
                    for (int i = lane; i < G; i += 64) {
                        int8_t v = P.fresh[i];
                        if (i == h1) v = TRON_P1_HEAD;
                        if (i == h2) v = TRON_P2_HEAD;
                        t8[i] = (unsigned char)v;
                        g8[i] = v;
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- C: encode from the LDS tile
    if (FMT == TRON_OBS_CODES_I8) {
        if (FAST) {
            const uint32_t D = (uint32_t)G >> 2;
            const uint32_t *tile32 = reinterpret_cast<const uint32_t *>(tile);
            const uint32_t total = (uint32_t)ne * 2u * D;                  // output dwords
            uint32_t *o32 = reinterpret_cast<uint32_t *>(obs) + (size_t)e0 * 2u * D;
            for (uint32_t od = (uint32_t)tid * 4u; od < total; od += 4u * BLOCK) {
                uint32_t q = __umulhi(od, P.d_magic);                      // plane = od / D
                uint32_t cd = od - q * D;
                uint32_t rr[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t t = (od + j < total) ? tile32[(q >> 1) * D + cd] : 0u;
                    rr[j] = codes4(t, (q & 1u) != 0u);
                    if (++cd == D) { cd = 0u; ++q; }
                }
                if (od + 4u <= total) {
                    *reinterpret_cast<uint4 *>(o32 + od) = make_uint4(rr[0], rr[1], rr[2], rr[3]);
                } else {
                    for (uint32_t j = 0; od + j < total; ++j) o32[od + j] = rr[j];
                }
            }
        } else {
            const uint32_t total = (uint32_t)ne * 2u * (uint32_t)G;        // output bytes
            int8_t *o8 = reinterpret_cast<int8_t *>(obs) + (size_t)e0 * 2u * G;
            for (uint32_t ob = (uint32_t)tid * 16u; ob < total; ob += 16u * BLOCK) {
                uint32_t q = ob / (uint32_t)G, cc = ob - q * (uint32_t)G;
                uint32_t rr[4] = {0u, 0u, 0u, 0u};
                for (uint32_t j = 0; j < 16u && ob + j < total; ++j) {
                    const int t = (int)(int8_t)tile[(q >> 1) * (uint32_t)G + cc];
                    rr[j >> 2] |= (uint32_t)(uint8_t)code1(t, (q & 1u) != 0u) << ((j & 3u) * 8u);
                    if (++cc == (uint32_t)G) { cc = 0u; ++q; }
                }
                if (ob + 16u <= total) {
                    *reinterpret_cast<uint4 *>(o8 + ob) = make_uint4(rr[0], rr[1], rr[2], rr[3]);
                } else {
                    for (uint32_t j = 0; ob + j < total; ++j) o8[ob + j] = (int8_t)(rr[j >> 2] >> ((j & 3u) * 8u));
                }
            }
        }
    } else if (FMT == TRON_OBS_PLANES3_F32 || FMT == TRON_OBS_PLANES4_F32) {
        constexpr uint32_t CH = (FMT == TRON_OBS_PLANES3_F32) ? 3u : 4u;
        if (FAST) {
            const uint32_t D = (uint32_t)G >> 2;
            const uint32_t *tile32 = reinterpret_cast<const uint32_t *>(tile);
            const uint32_t total = (uint32_t)ne * 2u * CH * D;             // output float4s
            float4 *o4 = reinterpret_cast<float4 *>(obs) + (size_t)e0 * 2u * CH * D;
            for (uint32_t i = (uint32_t)tid; i < total; i += BLOCK) {
                const uint32_t pl = __umulhi(i, P.d_magic);                // plane = i / D
                const uint32_t cd = i - pl * D;
                const uint32_t q = pl / CH, ch = pl - q * CH;
                const uint32_t e = q >> 1;
                float4 v;
                if (ch == 3u) {
                    const float f = plane4[e];
                    v = make_float4(f, f, f, f);
                } else {
                    const uint32_t t = tile32[e * D + cd];
                    const uint32_t bits = plane_bits((int)ch, (q & 1u) != 0u);
                    v = make_float4(plane_val(bits, t), plane_val(bits, t >> 8), plane_val(bits, t >> 16),
                                    plane_val(bits, t >> 24));
                }
                o4[i] = v;
            }
        } else {
            const uint32_t total = (uint32_t)ne * 2u * CH * (uint32_t)G;   // output floats
            float *o = reinterpret_cast<float *>(obs) + (size_t)e0 * 2u * CH * G;
            for (uint32_t i = (uint32_t)tid; i < total; i += BLOCK) {
                const uint32_t pl = i / (uint32_t)G, cc = i - pl * (uint32_t)G;
                const uint32_t q = pl / CH, ch = pl - q * CH;
                const uint32_t e = q >> 1;
                o[i] = (ch == 3u) ? plane4[e]
                                  : plane_val(plane_bits((int)ch, (q & 1u) != 0u), (uint32_t)tile[e * (uint32_t)G + cc]);
            }
        }
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
    const uint32_t epi = P.episode[env];
    NewGame ng;
    if (start) {
        ng.r1 = start[4 * env]; ng.c1 = start[4 * env + 1]; ng.r2 = start[4 * env + 2]; ng.c2 = start[4 * env + 3];
        uint32_t x[4];
        philox4x32_10((uint32_t)env, epi, RNG_INIT, 0u, P.seed, P.stream, x);
        ng.w0 = randint_u32(x[0], 40, 101);                              // game.py:83
        ng.w1 = randint_u32(x[1], 40, 101);
        ng.degree = randint_u32(x[2], -30, 30);                          // game.py:87
    } else {
        ng = make_game(P, (uint32_t)env, epi);
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
        P.pos[env] = (uint32_t)(uint8_t)ng.r1 | ((uint32_t)(uint8_t)ng.c1 << 8) | ((uint32_t)(uint8_t)ng.r2 << 16) |
                     ((uint32_t)(uint8_t)ng.c2 << 24);
        P.meta[env] = META_ALIVE0 | META_ALIVE1;
        P.envp[env] = (uint32_t)(uint8_t)ng.w0 | ((uint32_t)(uint8_t)ng.w1 << 8) |
                      ((uint32_t)(uint8_t)(int8_t)ng.degree << 16);
        P.episode[env] = epi + 1u;
        P.eplen[env] = 0u;
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
    const uint32_t pw = P.pos[i], m = P.meta[i], ep = P.envp[i];
    if (pos) reinterpret_cast<uint32_t *>(pos)[i] = pw;
    if (alive) { alive[2 * i] = (int8_t)(m & 1u); alive[2 * i + 1] = (int8_t)((m >> 1) & 1u); }
    if (dir) { dir[2 * i] = (int8_t)((m >> 8) & 7u); dir[2 * i + 1] = (int8_t)((m >> 12) & 7u); }
    if (done) done[i] = (int8_t)((m >> 2) & 1u);
    if (winner) winner[i] = (int8_t)((m >> 4) & 3u);
    if (weight) { weight[2 * i] = (int16_t)(ep & 0xFFu); weight[2 * i + 1] = (int16_t)((ep >> 8) & 0xFFu); }
    if (degree) degree[i] = (int16_t)(int8_t)(ep >> 16);
    if (slide) slide[i] = P.slide[i];
    if (counters) { counters[3 * i] = P.tick[i]; counters[3 * i + 1] = P.episode[i]; counters[3 * i + 2] = P.eplen[i]; }
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
    int E;          // envs per workgroup tile
    size_t smem;    // dynamic LDS bytes
    bool fast;      // G % 4 == 0
    void *blob;     // one allocation behind all state arrays
};

namespace {

inline hipStream_t S_(void *s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH;
}

template <int FMT, bool FAST, bool DO_STEP>
int launch_one(tron_env *h, const int8_t *actions, const float *uniforms, uint32_t flags, void *obs, StepOut out,
               hipStream_t st)
{
    auto kern = k_step_encode<FMT, FAST, DO_STEP>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                160 * 1024) != hipSuccess)
            (void)hipGetLastError();
        attr_done = true;
    }
    const int blocks = (h->P.N + h->E - 1) / h->E;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(BLOCK), h->smem, st, h->P, h->E, actions, uniforms, flags, obs, out);
    return launch_status();
}

template <bool DO_STEP>
int launch_fmt(tron_env *h, int fmt, const int8_t *a, const float *u, uint32_t flags, void *obs, StepOut out,
               hipStream_t st)
{
#define TRON_CASE(F)                                                                          \
    case F:                                                                                   \
        return h->fast ? launch_one<F, true, DO_STEP>(h, a, u, flags, obs, out, st)           \
                       : launch_one<F, false, DO_STEP>(h, a, u, flags, obs, out, st);
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

int tron_create(int32_t n_envs, int32_t W, int32_t mode, int32_t fair, uint32_t seed, uint32_t rng_stream,
                tron_handle *out)
{
    if (!out) return TRON_ERR_BAD_ARG;
    *out = nullptr;
    if (n_envs < 1 || W < 2 || W > 96 || mode < TRON_MODE_NONE || mode > TRON_MODE_TEMPER) return TRON_ERR_BAD_ARG;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }

    tron_env *h = new (std::nothrow) tron_env();
    if (!h) return TRON_ERR_ALLOC;
    Params &P = h->P;
    P.N = n_envs; P.W = W; P.S = W + 2; P.G = P.S * P.S; P.mode = mode; P.fair = fair ? 1 : 0;
    P.seed = seed; P.stream = rng_stream;
    P.r_step = -1.0f; P.r_win = 100.0f; P.r_lose = -100.0f; P.r_draw = 0.0f; P.r_index = 0;   // DDQN.py:289-305
    h->device = dev;
    h->fast = (P.G % 4) == 0;
    const uint32_t D = h->fast ? (uint32_t)P.G / 4u : (uint32_t)P.G;
    P.d_magic = (uint32_t)((0x100000000ull + D - 1) / D);

    // tile size: E % 16 == 0 keeps every tile span 16-byte aligned for any G
    int E = (64 * P.G <= 48 * 1024) ? 64 : (32 * P.G <= 64 * 1024) ? 32 : 16;
    if (const char *s = getenv("TRON_TILE_ENVS")) {
        const int v = atoi(s);
        if ((v == 16 || v == 32 || v == 64) && (size_t)v * P.G + 8u * v + 16 <= 160u * 1024) E = v;
    }
    h->E = E;
    h->smem = (((size_t)E * P.G + 15) & ~(size_t)15) + 8u * (size_t)E;

    // one blob: grid (padded for 16-byte over-read) + SoA words + fresh template
    const size_t N = (size_t)n_envs;
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_grid = 0, o_pos = align(o_grid + N * P.G + 64), o_meta = align(o_pos + 4 * N),
                 o_envp = align(o_meta + 4 * N), o_slide = align(o_envp + 4 * N), o_tick = align(o_slide + 8 * N),
                 o_epi = align(o_tick + 4 * N), o_len = align(o_epi + 4 * N), o_fresh = align(o_len + 4 * N),
                 total = align(o_fresh + (size_t)P.G + 16);
    char *blob = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&blob), total) != hipSuccess) {
        (void)hipGetLastError();
        delete h;
        return TRON_ERR_ALLOC;
    }
    h->blob = blob;
    P.grid = reinterpret_cast<int8_t *>(blob + o_grid);
    P.pos = reinterpret_cast<uint32_t *>(blob + o_pos);
    P.meta = reinterpret_cast<uint32_t *>(blob + o_meta);
    P.envp = reinterpret_cast<uint32_t *>(blob + o_envp);
    P.slide = reinterpret_cast<double *>(blob + o_slide);
    P.tick = reinterpret_cast<uint32_t *>(blob + o_tick);
    P.episode = reinterpret_cast<uint32_t *>(blob + o_epi);
    P.eplen = reinterpret_cast<uint32_t *>(blob + o_len);
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

int tron_reset(tron_handle h, const int8_t *env_mask, const int8_t *start_pos, const int16_t *weight,
               const int16_t *degree, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    const int per = BLOCK / 64;
    hipLaunchKernelGGL(k_reset, dim3((h->P.N + per - 1) / per), dim3(BLOCK), 0, S_(stream), h->P, env_mask, start_pos,
                       weight, degree);
    return launch_status();
}

int tron_step_encode(tron_handle h, const int8_t *actions, const float *uniforms, uint32_t flags, int32_t obs_fmt,
                     void *obs, int8_t *out_done, int8_t *out_winner, float *out_reward, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if ((obs_fmt != TRON_OBS_NONE) != (obs != nullptr)) return TRON_ERR_BAD_ARG;
    if (flags & ~TRON_STEP_AUTORESET) return TRON_ERR_BAD_ARG;
    StepOut out{out_done, out_winner, out_reward, nullptr};
    return launch_fmt<true>(h, obs_fmt, actions, uniforms, flags, obs, out, S_(stream));
}

int tron_step(tron_handle h, const int8_t *actions, const float *uniforms, uint32_t flags, int8_t *out_done,
              int8_t *out_winner, float *out_reward, void *stream)
{
    return tron_step_encode(h, actions, uniforms, flags, TRON_OBS_NONE, nullptr, out_done, out_winner, out_reward,
                            stream);
}

int tron_encode(tron_handle h, int32_t obs_fmt, void *obs, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (obs_fmt == TRON_OBS_NONE || !obs) return TRON_ERR_BAD_ARG;
    StepOut out{nullptr, nullptr, nullptr, nullptr};
    return launch_fmt<false>(h, obs_fmt, nullptr, nullptr, 0u, obs, out, S_(stream));
}

int tron_get_grid(tron_handle h, int8_t *grid_out, void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (!grid_out) return TRON_ERR_BAD_ARG;
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

int tron_rollout_random(tron_handle h, int32_t k_steps, int32_t obs_fmt, void *obs, unsigned long long *totals,
                        void *stream)
{
    if (bad_handle(h)) return h ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (k_steps < 0 || (obs_fmt != TRON_OBS_NONE) != (obs != nullptr)) return TRON_ERR_BAD_ARG;
    StepOut out{nullptr, nullptr, nullptr, totals};
    for (int k = 0; k < k_steps; ++k) {
        const int rc = launch_fmt<true>(h, obs_fmt, nullptr, nullptr, TRON_STEP_AUTORESET, obs, out, S_(stream));
        if (rc != TRON_OK) return rc;
    }
    return TRON_OK;
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
