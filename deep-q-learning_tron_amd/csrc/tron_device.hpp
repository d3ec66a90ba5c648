// Device-side building blocks of the gfx950 TRON env path.
// Rules restated from the reference are cited as file:line into
// Deep-Q-learning_TRON/; nothing here is shared with the CPU checker.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tron {

// ---- env state held by a handle, all in HBM ---------------------------------
// Hot words are packed one uint4 per env so a step costs one 16-byte load and one
// 16-byte store of state; the restart words are read every step, written on restart.
struct Params {
    int32_t N, W, S, G;        // envs, board side, S = W+2, G = S*S cells
    int32_t mode, fair;        // game.py:86 mode; util.py:48 "fair" start placement
    uint32_t seed, stream;     // Philox key
    float r_step, r_win, r_lose, r_draw;
    int32_t r_index;           // DQN.py:224-225: non-terminal reward = step index
    int8_t *grid;              // [N][G] Tile values (map.py:9-17), storage [row+1][col+1] (map.py:86-92)
    uint4 *st4;                // [N] {pos, meta, eplen, tick}
                               //   pos  = r1 | c1<<8 | r2<<16 | c2<<24 (int8 each; game.py:36-41)
                               //   meta = alive0 | alive1<<1 | done<<2 | winner<<4 | dir0<<8 | dir1<<12
                               //   eplen = steps in the current game; tick = steps since create (RNG counter)
    uint4 *rs4;                // [N] {envp, episode, nstart, nenvp}
                               //   envp = weight0 | weight1<<8 | (int8)degree<<16 (game.py:83,87)
                               //   episode = games started (RNG counter)
                               //   nstart / nenvp = pos / envp of the NEXT game (make_game at `episode`)
    double *slide;             // [N] game.py:88
    int8_t *obs_state;         // observation-is-state mode (mode None): caller-attached [N][2][G] code planes
                               //   whose player-1 plane IS the board; `grid` is then unused
    const int8_t *fresh;       // [G] empty board with WALL border (map.py:45-48)
};

enum { RNG_STEP = 0, RNG_RESET = 2, RNG_INIT = 3, RNG_MINIMAX = 4 };
enum { META_ALIVE0 = 1u, META_ALIVE1 = 2u, META_DONE = 4u };

__device__ __forceinline__ uint32_t pack_pos(int r1, int c1, int r2, int c2)
{
    return (uint32_t)(uint8_t)r1 | ((uint32_t)(uint8_t)c1 << 8) | ((uint32_t)(uint8_t)r2 << 16) |
           ((uint32_t)(uint8_t)c2 << 24);
}
__device__ __forceinline__ uint32_t pack_envp(int w0, int w1, int degree)
{
    return (uint32_t)(uint8_t)w0 | ((uint32_t)(uint8_t)w1 << 8) | ((uint32_t)(uint8_t)(int8_t)degree << 16);
}
__device__ __forceinline__ int cell_index(int S, int r, int c) { return (r + 1) * S + (c + 1); }

// ---- Philox-4x32-10 (Salmon et al., SC'11) ----------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4])
{
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// random.randint(a, b) over one u32: a + floor(u * (b-a+1) / 2^32)
__device__ __forceinline__ int randint_u32(uint32_t u, int a, int b)
{
    return a + (int)__umulhi(u, (uint32_t)(b - a + 1));
}

struct NewGame {
    int r1, c1, r2, c2, w0, w1, degree;
};

// One Philox word -> one player's action when the caller supplies none.  Default: uniform over the
// four headings (DQN.py:64, DDQN.py:110).  nonreversing (TRON_STEP_NONREVERSING, the secondary
// synthetic policy of SURVEY.md §8(d)): uniform over the three headings that do not reverse the
// player's last move (`last` = its Direction value 1..4 from the meta word, 0 before its first move).
__device__ __forceinline__ int draw_action(uint32_t x, uint32_t last, bool nonreversing)
{
    if (nonreversing && last) return (int)((last - 1u + 3u + __umulhi(x, 3u)) & 3u);
    return (int)(x & 3u);
}

// util.make_game start placement + Game.__init__ draws (util.py:46-84, game.py:83,87) over the
// reset stream of (env, episode): u32 number n is word n%4 of Philox(ctr = {env, episode,
// RNG_RESET, n/4}).  (x, y) of the reference are (row, col).  Only player 1 is re-drawn on a
// clash (util.py:76-78); bounded at 16 redraw rounds for the GPU.  Written as one loop over the
// draw index with a phase variable so the Philox body exists once in the code.
__device__ inline NewGame make_game(uint32_t seed, uint32_t stream, int W, int fair, uint32_t env, uint32_t episode)
{
    NewGame g{0, 0, 0, 0, 0, 0, 0};
    int lb1x = 0, lb1y = 0, lb2x = 0, lb2y = 0;
    int ub1x = W - 1, ub1y = W - 1, ub2x = W - 1, ub2y = W - 1;
    int py = 0, rounds = 0;
    bool have_p2 = false;
    uint32_t b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    // phases: 0 point_y, 1 point_x (fair only); 2 x1, 3 y1, 4 x2, 5 y2; 7 weight0, 8 weight1, 9 degree
    int phase = fair ? 0 : 2;
    for (int n = 0; phase < 10; ++n) {
        const int j = n & 3;
        if (j == 0) {
            uint32_t o[4];
            philox4x32_10(env, episode, RNG_RESET, (uint32_t)(n >> 2), seed, stream, o);
            b0 = o[0]; b1 = o[1]; b2 = o[2]; b3 = o[3];
        }
        const uint32_t u = j == 0 ? b0 : j == 1 ? b1 : j == 2 ? b2 : b3;
        bool check = false;
        if (phase == 0) {                                   // util.py:49
            py = randint_u32(u, 0, W - 1);
            phase = 1;
        } else if (phase == 1) {                            // util.py:50-62
            const int px = randint_u32(u, 0, W - 1);
            lb1x = max(0, px - 1); ub1x = min(W - 1, px + 1);
            lb1y = max(0, py - 1); ub1y = min(W - 1, py + 1);
            lb2x = W - 1 - ub1x;   ub2x = W - 1 - lb1x;
            lb2y = W - 1 - ub1y;   ub2y = W - 1 - lb1y;
            phase = 2;
        } else if (phase == 2) {                            // util.py:70 / :77
            g.r1 = randint_u32(u, lb1x, ub1x);
            phase = 3;
        } else if (phase == 3) {                            // util.py:71 / :78
            g.c1 = randint_u32(u, lb1y, ub1y);
            phase = 4;
            check = have_p2;
        } else if (phase == 4) {                            // util.py:73
            g.r2 = randint_u32(u, lb2x, ub2x);
            phase = 5;
        } else if (phase == 5) {                            // util.py:74
            g.c2 = randint_u32(u, lb2y, ub2y);
            have_p2 = true;
            check = true;
        } else if (phase == 7) {                            // game.py:83
            g.w0 = randint_u32(u, 40, 101);
            phase = 8;
        } else if (phase == 8) {
            g.w1 = randint_u32(u, 40, 101);
            phase = 9;
        } else {                                            // game.py:87
            g.degree = randint_u32(u, -30, 30);
            phase = 10;
        }
        if (check) {                                        // util.py:76: while x1 == x2 and y1 == y2
            if (g.r1 == g.r2 && g.c1 == g.c2) {
                if (rounds++ == 16) {
                    g.r1 = (g.r1 == ub1x) ? lb1x : g.r1 + 1;
                    phase = 7;
                } else {
                    phase = 2;
                }
            } else {
                phase = 7;
            }
        }
    }
    return g;
}

// Game.get_rate(player) in float64, same operation order as game.py:100-102.
__device__ __forceinline__ double get_rate(int degree, int weight)
{
    const double a = __dmul_rn((double)(degree - 30), 0.6);
    const double b = __ddiv_rn(-a, 100.0);
    const double c = __ddiv_rn((double)(70 - weight), 100.0);
    return __dsub_rn(b, c);
}

// Game.get_degree_silde(): (-slide*100)*(10/6)+30, game.py:110-112.
__device__ __forceinline__ double degree_slide(double slide)
{
    const double t = __dmul_rn(-slide, 100.0);
    const double k = 10.0 / 6.0;
    return __dadd_rn(__dmul_rn(t, k), 30.0);
}

// ---- observation codes (Map.color, map.py:67-81) as a v_perm_b32 byte LUT ----
// index = tile & 7: EMPTY 0, P1_BODY 1, P1_HEAD 2, P2_BODY 3, P2_HEAD 4,
// P1_slide 5, P2_slide 6, WALL (-1) 7.
constexpr uint32_t pack4(int a, int b, int c, int d)
{
    return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) |
           ((uint32_t)(uint8_t)d << 24);
}
constexpr uint32_t CODE_LO_P1 = pack4(1, -2, 10, -3), CODE_HI_P1 = pack4(-10, -2, -3, -1);
constexpr uint32_t CODE_LO_P2 = pack4(1, -3, -10, -2), CODE_HI_P2 = pack4(10, -3, -2, -1);

// four tiles -> four codes for `player_is_2`
__device__ __forceinline__ uint32_t codes4(uint32_t tiles, bool player_is_2)
{
    const uint32_t sel = tiles & 0x07070707u;
    return player_is_2 ? __builtin_amdgcn_perm(CODE_HI_P2, CODE_LO_P2, sel)
                       : __builtin_amdgcn_perm(CODE_HI_P1, CODE_LO_P1, sel);
}
// player-2 codes from player-1 codes: bodies -2 <-> -3, heads 10 <-> -10, EMPTY 1 and WALL -1 stay
// (map.py:67-81).  The six codes have distinct low nibbles (1, F, E, D, A, 6), so this is a
// 16-entry byte LUT: two v_perm halves selected by bit 3.
constexpr uint32_t SWAP_LO = pack4(0, 1, 0, 0) | 0u, SWAP_LO_HI = pack4(0, 0, 10, 0);     // idx 0-3 | 4-7 (6 -> 10)
constexpr uint32_t SWAP_HI_LO = pack4(0, 0, -10, 0), SWAP_HI_HI = pack4(0, -2, -3, -1);   // idx 8-11 (A -> -10) | 12-15 (D->-2, E->-3, F->-1)
__device__ __forceinline__ uint32_t swap_codes4(uint32_t c)
{
    const uint32_t sel = c & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(SWAP_LO_HI, SWAP_LO, sel);       // low nibble 0..7
    const uint32_t hi = __builtin_amdgcn_perm(SWAP_HI_HI, SWAP_HI_LO, sel);    // low nibble 8..15
    const uint32_t m = ((c >> 3) & 0x01010101u) * 0xFFu;                       // 0xFF where bit 3 is set
    return (hi & m) | (lo & ~m);
}
// Tile value (map.py:9-17) from a player-1 code, mode None (no slide tiles): low nibble LUT
// 1->EMPTY 0, F->WALL -1, E->P1_BODY 1, D->P2_BODY 3, A->P1_HEAD 2, 6->P2_HEAD 4.
__device__ __forceinline__ int8_t tile_of_code(int code)
{
    switch (code & 15) {
    case 0x1: return 0;
    case 0xF: return -1;
    case 0xE: return 1;
    case 0xD: return 3;
    case 0xA: return 2;
    default: return 4;
    }
}

__device__ __forceinline__ int8_t code1(int tile, bool player_is_2)
{
    const uint32_t sh = (uint32_t)(tile & 3) * 8u;
    const bool hi = (tile & 4) != 0;
    const uint32_t w = player_is_2 ? (hi ? CODE_HI_P2 : CODE_LO_P2) : (hi ? CODE_HI_P1 : CODE_LO_P1);
    return (int8_t)(w >> sh);
}

// ---- pop_up planes (util.py:11-37): 2 bits per tile index, 0 / 1 / 2(=10.0) ----
// channel order (wall, my, enemy).
constexpr uint32_t lut2(int i0, int i1, int i2, int i3, int i4, int i5, int i6, int i7)
{
    return (uint32_t)i0 | (i1 << 2) | (i2 << 4) | (i3 << 6) | (i4 << 8) | (i5 << 10) | (i6 << 12) | (i7 << 14);
}
constexpr uint32_t PLANE_WALL = lut2(0, 0, 0, 0, 0, 0, 0, 1);
constexpr uint32_t PLANE_P1 = lut2(0, 1, 2, 0, 0, 1, 0, 0);   // P1 body/slide 1, P1 head 10
constexpr uint32_t PLANE_P2 = lut2(0, 0, 0, 1, 2, 0, 1, 0);   // P2 body/slide 1, P2 head 10

__device__ __forceinline__ uint32_t plane_bits(int channel, bool player_is_2)
{
    if (channel == 0) return PLANE_WALL;
    const bool mine = (channel == 1);
    return (mine != player_is_2) ? PLANE_P1 : PLANE_P2;
}
__device__ __forceinline__ float plane_val(uint32_t bits, uint32_t tile)
{
    const uint32_t v = (bits >> ((tile & 7u) * 2u)) & 3u;
    return v == 0u ? 0.0f : (v == 1u ? 1.0f : 10.0f);
}

}  // namespace tron
