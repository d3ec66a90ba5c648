/*
 * oracle/tron_oracle.c — CPU restatement of the reference TRON env path.
 * TEST INFRASTRUCTURE ONLY (see tron_oracle.h).  Parity: PINNED by
 * tests/golden/ .npz files, which were produced by running the reference.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off; no FMA contraction so
 * the float64 expressions round exactly like CPython's).
 */
#include "tron_oracle.h"
#include <string.h>

/* ---------------------------------------------------------------- map ---- */

/* map.py:5-6,45-48 — (W+2)x(W+2) image, WALL on the border, EMPTY inside. */
void orc_map_init(int8_t *grid, int W)
{
    int S = W + 2;
    for (int r = 0; r < S; ++r)
        for (int c = 0; c < S; ++c)
            grid[r * S + c] = (r == 0 || r == S - 1 || c == 0 || c == S - 1) ? ORC_WALL : ORC_EMPTY;
}

/* map.py:86-92 — interior coordinate (i,j) lives at storage [i+1][j+1]. */
static inline int8_t *cell(int8_t *grid, int S, int i, int j)
{
    return &grid[(i + 1) * S + (j + 1)];
}

/* game.py:43-58 — PositionPlayer.body()/head()/slide() for player index p. */
static inline int8_t body_tile(int p)  { return p == 0 ? ORC_P1_BODY  : ORC_P2_BODY; }
static inline int8_t head_tile(int p)  { return p == 0 ? ORC_P1_HEAD  : ORC_P2_HEAD; }
static inline int8_t slide_tile(int p) { return p == 0 ? ORC_P1_SLIDE : ORC_P2_SLIDE; }

/* game.py:71-91 — fresh map, both heads written in pps order. */
void orc_game_init(int8_t *grid, int W, const int8_t start[4])
{
    int S = W + 2;
    orc_map_init(grid, W);
    *cell(grid, S, start[0], start[1]) = head_tile(0);
    *cell(grid, S, start[2], start[3]) = head_tile(1);
}

/* ------------------------------------------------------------- scalars --- */

/* game.py:96-102 — get_rate(): -((degree-30)*0.6)/100 [- (70-weight)/100]. */
double orc_get_rate(int degree, int weight, int has_player)
{
    double a = (double)(degree - 30) * 0.6;
    double b = (-a) / 100.0;
    if (!has_player)
        return b;
    double c = (double)(70 - weight) / 100.0;
    return b - c;
}

/* game.py:110-112 — get_degree_silde(): (-slide*100)*(10/6)+30. */
double orc_degree_slide(double slide)
{
    double t = (-slide) * 100.0;
    double k = 10.0 / 6.0;
    return t * k + 30.0;
}

/* ---------------------------------------------------------------- step --- */

/* player.py:107-132 — action a in 0..3 -> UP, RIGHT, DOWN, LEFT on (row, col). */
static const int DR[4] = { -1, 0, 1, 0 };
static const int DC[4] = { 0, 1, 0, -1 };

int orc_step(int W, int mode, int8_t *grid, int8_t pos[4], int8_t alive[2], int8_t dir[2],
             int8_t *done, int8_t *winner, const int8_t act[2], const float u[2],
             double slide, const int16_t weight[2], int degree, int8_t consumed[2])
{
    int S = W + 2;
    if (*done)
        return -1;

    /* game.py:155-156 — both current heads become bodies before anyone moves */
    for (int p = 0; p < 2; ++p)
        *cell(grid, S, pos[2 * p], pos[2 * p + 1]) = body_tile(p);

    /* game.py:158-178 — advance, optional slide, in player order */
    for (int p = 0; p < 2; ++p) {
        int a = act[p] & 3;
        int r = pos[2 * p] + DR[a], c = pos[2 * p + 1] + DC[a];
        dir[p] = (int8_t)(a + 1);                       /* Direction value, player.py:4-8 */
        consumed[p] = 0;
        if (mode == ORC_MODE_ICE || mode == ORC_MODE_TEMPER) {
            /* short-circuit: the uniform is drawn only for an in-bounds EMPTY target */
            if (r >= 0 && c >= 0 && r < W && c < W && *cell(grid, S, r, c) == ORC_EMPTY) {
                double rate = (mode == ORC_MODE_ICE) ? slide : orc_get_rate(degree, weight[p], 1);
                consumed[p] = 1;
                if ((double)u[p] <= rate) {             /* game.py:169 */
                    *cell(grid, S, r, c) = slide_tile(p);
                    r += DR[a];
                    c += DC[a];
                }
            }
        }
        pos[2 * p] = (int8_t)r;
        pos[2 * p + 1] = (int8_t)c;
    }

    /* game.py:205-214 — collisions in player order; the head tile is written in
     * all three branches (an out-of-bounds head lands on the border WALL cell). */
    for (int p = 0; p < 2; ++p) {
        int r = pos[2 * p], c = pos[2 * p + 1];
        if (r < 0 || c < 0 || r >= W || c >= W)
            alive[p] = 0;
        else if (*cell(grid, S, r, c) != ORC_EMPTY)
            alive[p] = 0;
        *cell(grid, S, r, c) = head_tile(p);
    }

    /* game.py:264-275 — done / winner */
    int n_alive = (alive[0] != 0) + (alive[1] != 0);
    if (n_alive <= 1) {
        if (n_alive == 1) {
            if (pos[0] != pos[2] || pos[1] != pos[3])
                *winner = alive[0] ? 1 : 2;
        }
        *done = 1;
    }
    return 0;
}

/* -------------------------------------------------------------- encode --- */

/* map.py:67-81 — color(t, p): EMPTY->1, WALL->-1, own body/slide->-2,
 * enemy body/slide->-3, own head->10, enemy head->-10.  Index = tile & 7
 * (WALL=-1 -> 7). */
static const int8_t CODE_LUT[2][8] = {
    /* EMPTY P1B  P1H  P2B  P2H  P1S  P2S  WALL */
    { 1, -2, 10, -3, -10, -2, -3, -1 },   /* player 1 */
    { 1, -3, -10, -2, 10, -3, -2, -1 },   /* player 2 */
};

/* map.py:83-84 — the apply()/.T pair cancels: codes[r][c] = color(data[r][c]). */
void orc_state_for_player(const int8_t *grid, int G, int player, int8_t *codes)
{
    const int8_t *lut = CODE_LUT[player == 2];
    for (int i = 0; i < G; ++i)
        codes[i] = lut[grid[i] & 7];
}

/* util.py:11-37 — planes ordered (wall, my, enemy); heads carry 10. */
void orc_pop_up(const int8_t *codes, int G, float *planes3)
{
    float *wall = planes3, *my = planes3 + G, *ener = planes3 + 2 * G;
    for (int i = 0; i < G; ++i) {
        int v = codes[i];
        wall[i] = (v == -1) ? 1.0f : 0.0f;
        my[i]   = (v == -2) ? 1.0f : (v == 10) ? 10.0f : 0.0f;
        ener[i] = (v == -3) ? 1.0f : (v == -10) ? 10.0f : 0.0f;
    }
}

/* --------------------------------------------------------------- reset --- */

static inline int randint_u32(uint32_t u, int a, int b)
{
    return a + (int)(((uint64_t)u * (uint64_t)(uint32_t)(b - a + 1)) >> 32);
}
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

#define ORC_MAX_REDRAW 16

/* util.py:46-84 + game.py:83,87.  (x, y) are (row, col). */
int orc_make_game(int W, int fair, const uint32_t *s, int8_t start[4], int16_t weight[2], int16_t *degree)
{
    int n = 0;
    int lb1x = 0, lb1y = 0, lb2x = 0, lb2y = 0;
    int ub1x = W - 1, ub1y = W - 1, ub2x = W - 1, ub2y = W - 1;
    if (fair) {                                     /* util.py:48-62 */
        int py = randint_u32(s[n++], 0, W - 1);
        int px = randint_u32(s[n++], 0, W - 1);
        lb1x = imax(0, px - 1); ub1x = imin(W - 1, px + 1);
        lb1y = imax(0, py - 1); ub1y = imin(W - 1, py + 1);
        lb2x = W - 1 - ub1x;    ub2x = W - 1 - lb1x;
        lb2y = W - 1 - ub1y;    ub2y = W - 1 - lb1y;
    }
    int x1 = randint_u32(s[n++], lb1x, ub1x);       /* util.py:70-74 */
    int y1 = randint_u32(s[n++], lb1y, ub1y);
    int x2 = randint_u32(s[n++], lb2x, ub2x);
    int y2 = randint_u32(s[n++], lb2y, ub2y);
    int rounds = 0;
    while (x1 == x2 && y1 == y2) {                  /* util.py:76-78: only P1 is redrawn */
        if (rounds++ == ORC_MAX_REDRAW) {           /* bounded for the GPU; never hit in practice */
            x1 = (x1 == ub1x) ? lb1x : x1 + 1;
            break;
        }
        x1 = randint_u32(s[n++], lb1x, ub1x);
        y1 = randint_u32(s[n++], lb1y, ub1y);
    }
    start[0] = (int8_t)x1; start[1] = (int8_t)y1; start[2] = (int8_t)x2; start[3] = (int8_t)y2;
    weight[0] = (int16_t)randint_u32(s[n++], 40, 101);   /* game.py:83 */
    weight[1] = (int16_t)randint_u32(s[n++], 40, 101);
    *degree   = (int16_t)randint_u32(s[n++], -30, 30);   /* game.py:87 */
    return n;
}

/* util.py:87-94 (get_reward), DDQN.py:289-305, DQN.py:224-241, ACKTR.py:294-317 */
void orc_rewards(const orc_reward_t *rw, int done, int winner, uint32_t step_index, float out[2])
{
    if (!done) {
        float s = rw->step_is_index ? (float)step_index : rw->step;
        out[0] = s; out[1] = s;
    } else if (winner == 0) {
        out[0] = rw->draw; out[1] = rw->draw;
    } else if (winner == 1) {
        out[0] = rw->win; out[1] = rw->lose;
    } else {
        out[0] = rw->lose; out[1] = rw->win;
    }
}

/* -------------------------------------------------------------- philox --- */
/* Philox-4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11).  Published algorithm; checked against its KAT vectors. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* ----------------------------------------------------------------- vec --- */

#define ORC_RESET_BLOCKS 12   /* 48 u32 >= 2 + 4 + 2*16 + 3 */

void orc_vec_reset_env(orc_vec_t *v, int i)
{
    int G = (v->W + 2) * (v->W + 2);
    uint32_t stream[4 * ORC_RESET_BLOCKS];
    uint32_t key[2] = { v->seed, v->stream };
    for (uint32_t b = 0; b < ORC_RESET_BLOCKS; ++b) {
        uint32_t ctr[4] = { (uint32_t)i, v->episode[i], ORC_RNG_RESET, b };
        orc_philox4x32_10(ctr, key, &stream[4 * b]);
    }
    int8_t start[4];
    orc_make_game(v->W, v->fair, stream, start, &v->weight[2 * i], &v->degree[i]);
    orc_game_init(&v->grid[(size_t)i * G], v->W, start);
    memcpy(&v->pos[4 * i], start, 4);
    v->alive[2 * i] = v->alive[2 * i + 1] = 1;
    v->dir[2 * i] = v->dir[2 * i + 1] = 0;
    v->done[i] = 0;
    v->winner[i] = 0;
    v->eplen[i] = 0;
    v->episode[i] += 1;
}

/* Game(w, h, pps) with explicit start positions; Game.__init__'s weight/degree draws
 * (game.py:83,87) come from Philox(ctr = {env, episode, ORC_RNG_INIT, 0}) unless given. */
void orc_vec_init_env(orc_vec_t *v, int i, const int8_t start[4], const int16_t *weight, const int16_t *degree)
{
    int G = (v->W + 2) * (v->W + 2);
    uint32_t key[2] = { v->seed, v->stream };
    uint32_t ctr[4] = { (uint32_t)i, v->episode[i], ORC_RNG_INIT, 0 }, x[4];
    orc_philox4x32_10(ctr, key, x);
    v->weight[2 * i]     = weight ? weight[0] : (int16_t)randint_u32(x[0], 40, 101);
    v->weight[2 * i + 1] = weight ? weight[1] : (int16_t)randint_u32(x[1], 40, 101);
    v->degree[i]         = degree ? *degree   : (int16_t)randint_u32(x[2], -30, 30);
    orc_game_init(&v->grid[(size_t)i * G], v->W, start);
    memcpy(&v->pos[4 * i], start, 4);
    v->alive[2 * i] = v->alive[2 * i + 1] = 1;
    v->dir[2 * i] = v->dir[2 * i + 1] = 0;
    v->done[i] = 0;
    v->winner[i] = 0;
    v->eplen[i] = 0;
    v->episode[i] += 1;
}

static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n < 1 ? 1 : n; }

void orc_vec_step(orc_vec_t *v, const int8_t *actions, const float *uniforms, int flags,
                  int8_t *obs_codes, int8_t *out_done, int8_t *out_winner, float *out_reward)
{
    int G = (v->W + 2) * (v->W + 2);
    uint32_t key[2] = { v->seed, v->stream };
    /* envs are independent (each Game owns its map, ACKTR.py:183,289): one env per OpenMP iteration */
#pragma omp parallel for schedule(static) num_threads(orc_threads)
    for (int i = 0; i < v->N; ++i) {
        int8_t *grid = &v->grid[(size_t)i * G];
        int8_t act[2];
        float u[2];
        int8_t consumed[2];
        if (!actions || !uniforms) {
            uint32_t ctr[4] = { (uint32_t)i, v->tick[i], ORC_RNG_ACTION, 0 }, x[4];
            orc_philox4x32_10(ctr, key, x);
            act[0] = (int8_t)(x[0] & 3); act[1] = (int8_t)(x[1] & 3);
            if (flags & ORC_STEP_NONREVERSING)
                /* secondary synthetic policy: uniform over the three headings that do
                 * not reverse the player's last one; a player that has not moved yet draws from all four */
                for (int p = 0; p < 2; ++p)
                    if (v->dir[2 * i + p])
                        act[p] = (int8_t)((v->dir[2 * i + p] - 1 + 3 + (int)(((uint64_t)x[p] * 3u) >> 32)) & 3);
            u[0] = (float)(x[2] >> 8) * (1.0f / 16777216.0f);
            u[1] = (float)(x[3] >> 8) * (1.0f / 16777216.0f);
        }
        if (actions)  { act[0] = actions[2 * i];  act[1] = actions[2 * i + 1]; }
        if (uniforms) { u[0] = uniforms[2 * i];   u[1] = uniforms[2 * i + 1]; }

        if (!v->done[i]) {
            uint32_t idx = v->eplen[i];
            orc_step(v->W, v->mode, grid, &v->pos[4 * i], &v->alive[2 * i], &v->dir[2 * i],
                     &v->done[i], &v->winner[i], act, u, v->slide[i], &v->weight[2 * i],
                     v->degree[i], consumed);
            v->eplen[i] += 1;
            v->tick[i] += 1;
            if (out_reward)
                orc_rewards(&v->reward, v->done[i], v->winner[i], idx, &out_reward[2 * i]);
        } else if (out_reward) {
            /* stepping a finished env without autoreset is a no-op */
            out_reward[2 * i] = out_reward[2 * i + 1] = 0.0f;
        }
        if (out_done)   out_done[i] = v->done[i];
        if (out_winner) out_winner[i] = v->winner[i];
        if ((flags & ORC_STEP_AUTORESET) && v->done[i])
            orc_vec_reset_env(v, i);                 /* ACKTR.py:307-310 */
        if (obs_codes) {
            orc_state_for_player(grid, G, 1, &obs_codes[((size_t)i * 2 + 0) * G]);
            orc_state_for_player(grid, G, 2, &obs_codes[((size_t)i * 2 + 1) * G]);
        }
    }
}
