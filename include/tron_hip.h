/*
 * include/tron_hip.h — C ABI of libtron_hip.so, the MI355X (gfx950) TRON env path.
 *
 * The reference (ckawoalt/Deep-Q-Learning_TRON) has no FFI of its own: its hot
 * path is a set of Python classes.  Each entry point below names the reference
 * interface it replaces (paths relative to Deep-Q-learning_TRON/).  The Python
 * host layer in deep-q-learning_tron_amd/tron/ binds these with ctypes and
 * re-creates the reference's class surface on top.
 *
 * Conventions
 *  - every function returns 0 (TRON_OK) or a negative tron_status; none throws,
 *    none allocates after tron_create / tron_replay_create;
 *  - all buffers are CALLER-OWNED DEVICE pointers (e.g. torch tensors), plain
 *    pointers and sizes, no framework types; the handle owns only env state;
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls
 *    are asynchronous on it; one host thread per handle;
 *  - boards are square, side W in [2, 96]; G = (W+2)*(W+2) cells incl. border;
 *  - env-major layouts: grid [N][G] int8, observations [N][2][...] (player 1,
 *    then player 2 of each env), so a [N*2, C, W+2, W+2] tensor view is free.
 */
#ifndef TRON_HIP_H
#define TRON_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TRON_ABI_VERSION 13

typedef enum {
    TRON_OK = 0,
    TRON_ERR_BAD_ARG = -1,      /* NULL handle / out-of-range size, mode, format */
    TRON_ERR_NO_DEVICE = -2,    /* no HIP device / wrong device current          */
    TRON_ERR_ALLOC = -3,        /* hipMalloc failed in a *_create                */
    TRON_ERR_LAUNCH = -4,       /* kernel launch rejected (hipGetLastError)      */
    TRON_ERR_UNSUPPORTED = -5   /* valid request this build does not implement   */
} tron_status;

/* Tile codes stored in the grid — tron/map.py:9-17 (Tile) */
enum { TRON_WALL = -1, TRON_EMPTY = 0, TRON_P1_BODY = 1, TRON_P1_HEAD = 2, TRON_P2_BODY = 3,
       TRON_P2_HEAD = 4, TRON_P1_SLIDE = 5, TRON_P2_SLIDE = 6 };

/* Game mode — tron/game.py:71,86,163 (mode=None | "ice" | "temper") */
enum { TRON_MODE_NONE = 0, TRON_MODE_ICE = 1, TRON_MODE_TEMPER = 2 };

/* Observation formats written by tron_step_encode / tron_encode */
enum {
    TRON_OBS_NONE = 0,
    TRON_OBS_CODES_I8 = 1,     /* [N][2][G] int8: Map.state_for_player codes, tron/map.py:67-84       */
    TRON_OBS_PLANES3_F32 = 2,  /* [N][2][3][G] f32: util.pop_up planes (wall,my,enemy), util.py:11-37 */
    TRON_OBS_PLANES4_F32 = 3   /* + 4th plane Game.prob_map(), game.py:124-132,297                    */
};

/* tron_step_encode flags */
#define TRON_STEP_AUTORESET 1u /* ACKTR.py:294-314: finished env -> fresh make_game, obs = new game */
#define TRON_STEP_INCREMENTAL 2u /* observation-is-state only: update the attached planes in place — write just
                                  * the <=4 cells a move touches and the boards that restart, instead of
                                  * rewriting both planes.  Same results, different traffic contract.       */
#define TRON_STEP_NONREVERSING 4u /* when actions == NULL: each player draws uniformly from the three headings
                                   * that do not reverse its last move (all four before its first move) instead
                                   * of from all four — the longer-episode synthetic policy of SURVEY.md §8(d) */

#define TRON_ROLLOUT_CHUNK 64     /* steps per persistent rollout launch (tron_rollout_random) */
#define TRON_ROLLOUT_PER_STEP 8u  /* tron_rollout_random flag: one launch per step instead (for A/B measurements) */
#define TRON_ROLLOUT_RESIDENT 32u  /* tron_rollout_random flag, attached observation buffer only: within a persistent launch
                                    * the boards stay in LDS from step to step instead of being read back from the
                                    * observation buffer each step — same results; HBM traffic per env-step drops from
                                    * 3G + 32 to 2G + 32 bytes (the two observation planes are still written every step).
                                    * Ignored (plain behaviour) where it does not apply.                              */
#define TRON_ROLLOUT_TWO_STREAMS 16u /* tron_rollout_random flag: one launch per step and per HALF of the envs, the two
                                      * halves on two streams (the handle owns the second one), so one half's launch
                                      * drains while the other's ramps up — the launch pattern of a caller that pipelines
                                      * two env halves against its policy with tron_step_encode_part */

typedef struct tron_env *tron_handle;

/* --- lifetime ---------------------------------------------------------------
 * Replaces: constructing N `Game(width, height, pps, mode, slide_pram)` objects
 * (tron/game.py:71-91), e.g. the `envs = [make_game(...)]*16` list of
 * ACKTR.py:183.  `fair` selects make_game(mode="fair") start placement
 * (util.py:48-62).  (seed, rng_stream) key the Philox-4x32-10 counter RNG used
 * wherever the reference calls `random` (it never seeds; rng_stream = rank).
 * Envs start un-initialised: call tron_reset before stepping.                */
int tron_create(int32_t n_envs, int32_t W, int32_t mode, int32_t fair,
                uint32_t seed, uint32_t rng_stream, tron_handle *out);
int tron_destroy(tron_handle h);

/* Reward tables — util.get_reward (util.py:87-94), DDQN.py:289-305,
 * DQN.py:224-241, ACKTR.py:294-317.  step_is_index!=0: non-terminal reward is
 * the 0-based step index of the episode (DQN.py:224-225).  Default = DDQN.   */
int tron_set_reward(tron_handle h, float step, float win, float lose, float draw, int32_t step_is_index);

/* `slide_pram` of Game.__init__ (game.py:88); default config.slide = 0.15.
 * slide_dev: optional device double[N] for a per-env value (play.py:76-98
 * sweeps it); NULL => broadcast `slide`.                                     */
int tron_set_slide(tron_handle h, double slide, const double *slide_dev, void *stream);

/* Assign Game.weight / Game.degree (game.py:83,87) of the envs after construction, as user
 * code may (`game.weight = [...]`).  weight int16[N][2] / degree int16[N]; either may be NULL. */
int tron_set_weight_degree(tron_handle h, const int16_t *weight, const int16_t *degree, void *stream);

/* --- reset -------------------------------------------------------------------
 * Replaces: util.make_game (util.py:46-84) + Game.__init__ (game.py:71-91).
 * env_mask  int8[N] or NULL (= all): which envs to (re)start.
 * start_pos int8[N][4] (row1,col1,row2,col2) or NULL: explicit `pps` positions
 *           as in Game(w,h,[PositionPlayer(1,..,[x1,y1]),PositionPlayer(2,..)]);
 *           NULL => drawn like make_game (P1-only redraw on clash).
 * weight    int16[N][2] / degree int16[N] or NULL: explicit Game.weight /
 *           Game.degree (game.py:83,87); NULL => drawn (randint(40,101) x2,
 *           randint(-30,30)).                                                */
int tron_reset(tron_handle h, const int8_t *env_mask, const int8_t *start_pos,
               const int16_t *weight, const int16_t *degree, void *stream);

/* Observation-is-state storage (even W, TRON_OBS_CODES_I8; every mode).  The player-1 code plane carries the game — a cell is
 * EMPTY or it is not — so the caller's observation buffer int8[N][2][G] can BE the env state: each step reads the player-1 plane
 * and rewrites both planes — the algorithmic 3G bytes per env-step, with no separate board write-back.  In mode None the plane
 * is a lossless image of the board; in ice / temper a slide tile shows as its player's body (map.py:67-81), so the env keeps the
 * slide tiles in a per-env log (2 bytes appended per slide) that tron_get_grid replays: the board image stays exact.
 * After attaching, the buffer belongs to the env until tron_destroy: read it, never write it;
 * pass it (or NULL with TRON_OBS_NONE) as `obs` to tron_step_encode.  tron_get_grid, tron_encode
 * and tron_reset keep working.  Returns TRON_ERR_UNSUPPORTED for odd W (TRON_STEP_INCREMENTAL: mode None only). */
int tron_attach_obs_state(tron_handle h, int8_t *obs_codes, void *stream);

/* --- step (+ observation encode), the hot path --------------------------------
 * Replaces: Game.step(a1, a2) -> (next_p1, next_p2, done) for every env
 * (tron/game.py:149-277), Map.state_for_player (map.py:83-84) and, for the
 * PLANES formats, util.pop_up / Game.prob_map.
 * actions   int8[N][2] in 0..3 (player.py:107-118) or NULL => Philox i.i.d.
 * uniforms  f32[N][2]: the value player p's `random.random()` would return
 *           (game.py:169), used only in ice/temper; NULL => Philox.
 * obs       per obs_fmt, or NULL with TRON_OBS_NONE.
 * out_done  int8[N]; out_winner int8[N] (0 = None, 1, 2); out_reward f32[N][2];
 *           each may be NULL.  With TRON_STEP_AUTORESET they describe the game
 *           that just finished while obs already shows the new one.
 * A finished env that is stepped without autoreset is left untouched.        */
int tron_step_encode(tron_handle h, const int8_t *actions, const float *uniforms, uint32_t flags,
                     int32_t obs_fmt, void *obs, int8_t *out_done, int8_t *out_winner,
                     float *out_reward, void *stream);
/* The same step for ONE SLICE of the envs: part `part` of `nparts` equal slices of the handle's env tiles
 * (tron_part_range gives the env range).  All buffers stay full-size and are indexed by the global env
 * number, so a caller can run slices as independent pipelines on different streams: while slice A is in the
 * env kernel, the policy network evaluates slice B's observations (envs never interact: ACKTR.py:183,285-289).
 * Not with TRON_STEP_INCREMENTAL, nor with f32 planes on an attached observation buffer.                    
 * A handle (and the library's per-device launch caches) is for ONE host thread at a time: the slice is passed through
 * the handle, so two threads stepping two slices of one handle race; run the slices from one thread on two streams. */
int tron_step_encode_part(tron_handle h, int32_t part, int32_t nparts, const int8_t *actions,
                          const float *uniforms, uint32_t flags, int32_t obs_fmt, void *obs, int8_t *out_done,
                          int8_t *out_winner, float *out_reward, void *stream);
int tron_part_range(tron_handle h, int32_t part, int32_t nparts, int32_t *first_env, int32_t *n_envs);
/* Same without an observation (Game.next_frame, game.py:149-252).            */
int tron_step(tron_handle h, const int8_t *actions, const float *uniforms, uint32_t flags,
              int8_t *out_done, int8_t *out_winner, float *out_reward, void *stream);
/* Encode the current state only: game.map().state_for_player(p) / pop_up of
 * it (DDQN.py:243-255, game.py:294-304).                                      */
int tron_encode(tron_handle h, int32_t obs_fmt, void *obs, void *stream);

/* K random-action steps with autoreset on `stream` (the synthetic rollout of
 * BASELINE.json).  With an attached observation buffer (tron_attach_obs_state) the
 * steps run as persistent launches of at most TRON_ROLLOUT_CHUNK steps each: envs
 * never interact, so every workgroup steps its own envs without a chip-wide drain
 * between steps — same results, bit for bit, as one launch per step (the other
 * storage modes / formats do launch per step).  flags: 0 or TRON_STEP_NONREVERSING.
 * totals u64[4] (device, may be NULL) accumulates {env_steps, p1_wins,
 * p2_wins, draws}.                                                            */
int tron_rollout_random(tron_handle h, int32_t k_steps, uint32_t flags, int32_t obs_fmt, void *obs,
                        unsigned long long *totals, void *stream);

/* --- state read-back (parity dumps, the scalar Game facade) --------------------
 * Replaces: Game.history[-1].map / Map.array() (map.py:60-61), PositionPlayer
 * .position/.alive (game.py:36-41), Game.winner/.done/.weight/.degree/.slide.
 * Any pointer may be NULL.  pos int8[N][4]; alive int8[N][2]; dir int8[N][2]
 * (Direction value 1..4 of the last move, 0 = none; player.py:4-8);
 * done int8[N]; winner int8[N]; weight int16[N][2]; degree int16[N];
 * slide f64[N]; counters u32[N][3] = {tick, episode, eplen}.                 */
int tron_get_grid(tron_handle h, int8_t *grid_out, void *stream);
int tron_get_state(tron_handle h, int8_t *pos, int8_t *alive, int8_t *dir, int8_t *done,
                   int8_t *winner, int16_t *weight, int16_t *degree, double *slide,
                   uint32_t *counters, void *stream);
int tron_info(tron_handle h, int32_t *n_envs, int32_t *W, int32_t *G, int32_t *mode);

/* --- stateless encoders (the Map / pop_up facade on arbitrary tile images) -----
 * tron_encode_codes: Map.state_for_player(player) on n images of `cells` tiles
 *   (map.py:67-84); player 1 or 2.
 * tron_pop_up: util.pop_up on n code planes -> [n][3][cells] f32 (util.py:11-37). */
int tron_encode_codes(const int8_t *tiles, int64_t n, int32_t cells, int32_t player,
                      int8_t *codes_out, void *stream);
int tron_pop_up(const int8_t *codes, int64_t n, int32_t cells, float *planes_out, void *stream);

/* --- device replay memory -------------------------------------------------------
 * Replaces: DDQN.ReplayBuffer (DDQN.py:167-203: deque(maxlen) + random.sample +
 * np.vstack + .to(device)) and DQN.ReplayMemory (DQN.py:81-132).  A ring of
 * `capacity` transitions in HBM; states are stored as int8 code planes
 * (2*cells + 8 bytes per slot) and expanded to f32 planes when sampled.       */
typedef struct tron_replay *tron_replay_handle;
int tron_replay_create(int64_t capacity, int32_t cells, uint32_t seed, uint32_t rng_stream,
                       tron_replay_handle *out);
int tron_replay_destroy(tron_replay_handle r);
/* Append n transitions (ReplayBuffer.add, DDQN.py:186-189), oldest overwritten.
 * state/next_state int8[n][cells] codes; action int8[n]; reward f32[n]; done int8[n].
 * Row i lands in slot (head + i) mod capacity: the ring order is deterministic. */
int tron_replay_push(tron_replay_handle r, int64_t n, const int8_t *state, const int8_t *action,
                     const float *reward, const int8_t *next_state, const int8_t *done,
                     void *stream);
/* The `state` rows of the NEXT tron_replay_push, written ahead of it (same ring position, nothing advanced): a trainer
 * whose observation buffer IS the env state (tron_attach_obs_state) saves s here before the step overwrites it and
 * then calls tron_replay_push(..., state = NULL, ...) with a, r, s', done — no clone of the observations.          */
int tron_replay_push_states(tron_replay_handle r, int64_t n, const int8_t *state, void *stream);
/* Uniform sample of `batch` DISTINCT slots (random.sample, DDQN.py:191-200) at any
 * batch <= size: slot j = pi(j) for a Philox-keyed permutation pi of the filled slots,
 * a fresh one per call.  Written as pop_up planes: states/next_states f32[batch][channels][cells]
 * (channels 3, or 4 with the constant `plane4` value), actions i64[batch],
 * rewards f32[batch], dones f32[batch].  Needs size >= batch.                  */
int tron_replay_sample(tron_replay_handle r, int32_t batch, int32_t channels, float plane4,
                       float *states, int64_t *actions, float *rewards, float *next_states,
                       float *dones, void *stream);
/* The same draw (random.sample semantics, one fresh permutation per call) with the states left as they are stored:
 * states / next_states int8[batch][cells] observation codes (4-byte aligned) — what conv1's TRON_CONV_IN_CODES staging
 * and tron_conv1_px16 read; 1 byte per cell instead of 12 or 16.                                                   */
int tron_replay_sample_codes(tron_replay_handle r, int32_t batch, int8_t *states, int64_t *actions, float *rewards,
                             int8_t *next_states, float *dones, void *stream);
int tron_replay_size(tron_replay_handle r, int64_t *size, int64_t *capacity);
/* Checkpointing the ring (SURVEY 8(f)4 "replay head"; the reference keeps its deque in host memory and never saves it,
 * DDQN.py:326).  The cursor is the write head, the fill level and the sampler's call counter (the Philox counter of the next
 * draw): restoring it makes a resumed run push to the same slots and draw the same batches.  export / import copy the slots
 * [first, first + n) (no wrap: first + n <= capacity) of all five arrays between the ring and caller-owned DEVICE buffers:
 * states / next_states int8[n][cells], actions int8[n], rewards f32[n], dones int8[n]; any of them may be NULL (skipped). */
int tron_replay_get_cursor(tron_replay_handle r, int64_t *head, int64_t *size, uint32_t *sample_calls);
int tron_replay_set_cursor(tron_replay_handle r, int64_t head, int64_t size, uint32_t sample_calls);
int tron_replay_export(tron_replay_handle r, int64_t first, int64_t n, int8_t *states, int8_t *next_states,
                       int8_t *actions, float *rewards, int8_t *dones, void *stream);
int tron_replay_import(tron_replay_handle r, int64_t first, int64_t n, const int8_t *states, const int8_t *next_states,
                       const int8_t *actions, const float *rewards, const int8_t *dones, void *stream);

/* C f32[M][N] = A B^T + bias[n] on the split-f16 matrix cores (operands split in two halves, three MFMAs per slab, f32
 * accumulation: relative error ~1e-6), f32 in and out — the dense products of the training-path head (conv7 in its dense
 * form: Net/activations.py::_PoolConv7, DQNNet.py:52-55).  A: f32[M][K], or (a_transposed) given as f32[K][M]; B: f32[N][K],
 * or (b_transposed) given as f32[K][N]; N % 64 == 0, and K % 64 == 0 unless both operands are transposed ones (those are
 * zero-padded).  bias may be NULL.  a_scale (may be NULL): one f32 on the device, a power of two A is multiplied by on its
 * way into f16 and C divided by (gradient operands).  workspace: tron_gemm_f16x3_workspace(M, N, K) bytes (0: unsupported). */
int tron_gemm_f16x3(const float *A, int32_t a_transposed, const float *B, int32_t b_transposed, const float *bias,
                    const float *a_scale, float *C, int64_t M, int32_t N, int64_t K, void *workspace, void *stream);
int64_t tron_gemm_f16x3_workspace(int64_t M, int32_t N, int64_t K);

/* ---- the trainer's small device-side steps around the network, one launch each (csrc/tron_dqn.hip) ---------------
 * tron_ddqn_td_loss: the Double-DQN loss of DDQN.py:129-146 and its gradient at the local net's Q-values:
 *   y = rewards + gamma * q_target_next[b][argmax_a q_local_next[b][a]] * (1 - dones),  loss = mean (q[b][actions[b]] - y)^2,
 *   grad_q[b][a] = 2 (q[b][a] - y) / batch at a = actions[b], else 0.  q / q_*_next / grad_q f32[batch][4] (16-byte aligned),
 *   actions i64[batch], rewards / dones f32[batch], loss f32[1]; sums in a fixed order.
 * tron_eps_greedy: DDQN.py:105-110 for n observations: actions[i] = u_i <= *epsilon ? uniform{0..3} : greedy[i]; epsilon is
 *   read on the device; draws are Philox-4x32-10 keyed (seed, stream_id) at counter (i / 4, call).
 * tron_eps_schedule: DDQN.py:313-315 per env step: state4 = {games, cycles, decays, decays_max} (i64 on the device):
 *   games += count(done != 0); cycles = games / games_per_cycle; decays = min(decays + new cycles, decays_max);
 *   epsilon = eps0 * rate ^ decays, written as f64 and f32 — nothing is read back.                                  */
/* out4 (f32[4] on the device, zeroed by the caller, 16-byte aligned) <- {scale, max |x|, scratch, scratch}: scale = the power
 * of two that brings max |x| into [2^(target_exp-1), 2^target_exp) (1 for an all-zero x; the exponent clamped to +-60) —
 * the device-side scale the split-f16 kernels take for gradient operands (in_scale / a_scale / grad_absmax), one launch.   */
int tron_absmax_pow2(const float *x, int64_t n, int32_t target_exp, float *out4, void *stream);
/* An nn.Linear layer's parameter gradients (DQNNet.py:24-31's fc1 / fc2 / actor1 / actor2 in loss.backward(), DDQN.py:148):
 * grad_weight f32[out][in] = grad_out^T input, grad_bias f32[out] (may be NULL) = column sums of grad_out; grad_out
 * f32[batch][out], input f32[batch][in].  Exact f32 FMAs; the batch is split over workgroups and the slices added in a fixed
 * order (deterministic).  workspace: tron_linear_wgrad_workspace(batch, out, in) bytes (0: not supported).               */
int tron_linear_wgrad(const float *grad_out, const float *input, int64_t batch, int32_t out_features, int32_t in_features,
                      float *grad_weight, float *grad_bias, void *workspace, void *stream);
int64_t tron_linear_wgrad_workspace(int64_t batch, int32_t out_features, int32_t in_features);
/* optim.Adam's step (DDQN.py:52,149-150: no weight decay, no amsgrad) and Agent.soft_update (DDQN.py:153-165) for n parameter
 * tensors in one launch per TRON_ADAM_MAX_TENSORS tensors.  For tensor k (numel[k] f32 elements, device pointers in HOST arrays):
 *   m <- m + (1 - beta1)(g - m);  v <- beta2 v + (1 - beta2) g g;
 *   w <- w - lr / (1 - beta1^steps[k]) * m / (sqrt(v) / sqrt(1 - beta2^steps[k]) + eps)      (steps[k] >= 1: the count INCLUDING this step)
 *   target <- tau w + (1 - tau) target                                                     (targets == NULL or targets[k] == NULL: skipped)
 * grads[k] == NULL: that tensor takes no Adam step (its soft update still runs).  The bias corrections are computed in double on the host. */
#define TRON_ADAM_MAX_TENSORS 32
int tron_adam_soft_update(int32_t n, float *const *params, const float *const *grads, float *const *exp_avg, float *const *exp_avg_sq,
                          float *const *targets, const int64_t *numel, const double *steps, double lr, double beta1, double beta2,
                          double eps, double tau, void *stream);
int tron_ddqn_td_loss(const float *q, const int64_t *actions, const float *rewards, const float *dones,
                      const float *q_local_next, const float *q_target_next, float gamma, int64_t batch, float *loss,
                      float *grad_q, void *stream);
int tron_eps_greedy(const int8_t *greedy, int64_t n, const float *epsilon, uint32_t seed, uint32_t stream_id, uint64_t call,
                    int8_t *actions, void *stream);
int tron_eps_schedule(const int8_t *done, int64_t n, int64_t *state4, int64_t games_per_cycle, double eps0, double rate,
                      double *epsilon_out, float *epsilon_out_f32, void *stream);
/* The slots drawn by the last tron_replay_sample, i64[batch] (tests, logging). */
int tron_replay_indices(tron_replay_handle r, int32_t batch, int64_t *indices_out, void *stream);

/* ---- Minimax/Voronoi opponent (tron/minimax.py; MinimaxPlayer(2, "voronoi") at util.py:82-83,
 * ACKTR.py:13,286-287) ------------------------------------------------------------------------ */
enum { TRON_MINIMAX_VORONOI = 0, TRON_MINIMAX_DISTWALL = 1 };   /* minimax.py:228-233 */
/* MinimaxPlayer.action for `player` (1|2) in every env (minimax.py:284-297 on
 * game.map().state_for_player(player), as game.py:181 calls it): out_actions i8[N] in the env's
 * action coding 0..3 = UP, RIGHT, DOWN, LEFT (the reference's 1..4 minus one), ready to be a column
 * of tron_step_encode's actions.  Finished envs get -1.  random.choice among equal moves /
 * random.randint for a boxed-in head draw from Philox (env, tick, purpose 4, player).
 * out_values i32[N][4] / out_expanded i8[N] (bit a = root move a was searched) may be NULL.
 * Only depth 2 exists in the reference's call sites; other depths -> TRON_ERR_UNSUPPORTED.
 * Boards up to 62x62.                                                                       */
int tron_minimax_actions(tron_handle h, int32_t player, int32_t depth, int32_t mode, int8_t *out_actions,
                         int32_t *out_values, int8_t *out_expanded, void *stream);
/* The same search on arbitrary observation-code images i8[n][side][side] (each must hold one +10
 * and one -10 head inside its border, as every live game's does; others get -1).  draws u32[n]:
 * choice -> ties[mulhi(u, len)], randint(1,4) -> 1 + mulhi(u, 4); NULL = all zero.          */
int tron_minimax_codes(const int8_t *codes, int64_t n, int32_t side, int32_t depth, int32_t mode,
                       const uint32_t *draws, int8_t *out_actions, int32_t *out_values, int8_t *out_expanded,
                       void *stream);

/* ---- K-FAC helper of the ACKTR path (Net/kfac.py:28-38 `_extract_patches`; its TODO at kfac.py:9-12
 * asks for this kernel) -------------------------------------------------------------------------- */
/* x f32[batch][channels][height][width] -> out f32[batch*OH*OW][channels*kh*kw], row = (sample, oy, ox),
 * column = c*kh*kw + i*kw + j, value x[n][c][oy*stride-pad+i][ox*stride-pad+j] or 0 outside:
 * F.unfold(x, (kh,kw), padding=pad, stride=stride).transpose(1,2).reshape(-1, C*kh*kw) in one launch.
 * channels*kh*(width+2*pad)*4 bytes must fit 64 KB of LDS (else TRON_ERR_UNSUPPORTED).           */
int tron_extract_patches(const float *x, int64_t batch, int32_t channels, int32_t height, int32_t width,
                         int32_t kh, int32_t kw, int32_t pad, int32_t stride, float *out, void *stream);
/* K-FAC's input factor of one layer and batch (kfac.py:41-58 `compute_cov_a`): gram f32[d][d] = scale * P^T P, overwritten,
 * P = the patch matrix above ([batch*OH*OW][d = channels*kh*kw]) — never materialised as f32: its transpose is written
 * as split f16 (v / 64 = hi + lo 2^-11) in passes of <= 512 MB and multiplied on the f16 matrix cores (three MFMAs per
 * slab, f32 accumulation, relative error ~1e-6), tiles on and above the diagonal only, K split over workgroups, sums in a
 * fixed order.  tron_kfac_gram: the same for a Linear layer's input a f32[rows][d] (gram = scale * a^T a).  d <= 8192;
 * workspace: tron_kfac_*_workspace(...) bytes (0 = not supported), 16-byte aligned.
 * The output-gradient factors (kfac.py:61-76 `compute_cov_g`) are the same products of gradient tensors — a convolution's
 * g f32[batch][cout][OH][OW] is the 1x1 "patch matrix" [batch*OH*OW][cout] — whose entries sit far below f16's range:
 * in_scale (may be NULL) points at ONE f32 on the device, a power of two the input is multiplied by on its way into f16
 * and the result divided by squared (the caller derives it from max |g| without reading it back).                */
int tron_kfac_patch_gram(const float *x, int64_t batch, int32_t channels, int32_t height, int32_t width, int32_t kh,
                         int32_t kw, int32_t pad, int32_t stride, float scale, const float *in_scale, float *gram,
                         void *workspace, void *stream);
int64_t tron_kfac_patch_gram_workspace(int64_t batch, int32_t channels, int32_t height, int32_t width, int32_t kh,
                                       int32_t kw, int32_t pad, int32_t stride);
int tron_kfac_gram(const float *a, int64_t rows, int32_t d, float scale, const float *in_scale, float *gram, void *workspace,
                   void *stream);
int64_t tron_kfac_gram_workspace(int64_t rows, int32_t d);

/* ---- the nets' activation (Net/ACNet.py:56-57: x * tanh(softplus(x))) as one pass each way --------- */
/* y[i] = mish(x[i]); f32, 16-byte aligned buffers, n elements.                                        */
int tron_mish_fwd(const float *x, float *y, int64_t n, void *stream);
/* grad_x[i] = grad_y[i] * mish'(x[i]).                                                                 */
int tron_mish_bwd(const float *x, const float *grad_y, float *grad_x, int64_t n, void *stream);

/* y_pre[n][c][i] += bias[c] (+ residual[n][c][i]); out = mish(y_pre): the bias add, residual add and
 * activation after a convolution (Net/DQNNet.py:33-63) in one pass.  hw % 4 == 0, else UNSUPPORTED.     */
int tron_bias_mish_fwd(float *y_pre, const float *bias, const float *residual, float *out, int64_t batch,
                       int32_t channels, int32_t hw, void *stream);

/* Backward of that pass: grad_pre = grad_out * mish'(y_pre) and bias_grad[c] = sum over batch and positions of
 * grad_pre — one launch plus a tiny fixed-order finish (deterministic).  scratch: f32[channels * 128], caller-owned:
 * [0, channels*64) per-block bias sums, [channels*64, channels*128) per-block maxima of |grad_pre| (what
 * tron_conv3x3_wgrad takes as grad_absmax, n_absmax = channels * 64).                                              */
int tron_bias_mish_bwd(const float *y_pre, const float *grad_out, float *grad_pre, float *bias_grad, float *scratch,
                       int64_t batch, int32_t channels, int32_t hw, void *stream);

/* ---- the CNN's 3x3 convolutions on the matrix cores (Net/DQNNet.py:10-17,33-50 conv1..conv6; the same
 * stacks in Net/ACNet.py) ------------------------------------------------------------------------------ */
/* out[b][co][y][x] = act(bias[co] + residual[b][co][y][x] + sum_{ci,ky,kx} W[co][ci][ky][kx] *
 * in[b][ci][y+ky-1][x+kx-1]) for a batch of side x side images, NCHW f32: F.conv2d(padding=1) + bias + optional
 * residual + optional mish in one launch (arithmetic: `math`, below).
 * in_fmt TRON_CONV_IN_CODES: `in` is int8 observation codes [batch][side*side] (Map.state_for_player, map.py:67-84)
 * and the input channels are util.pop_up's planes (wall, my, enemy; util.py:11-37) built on the fly, plus the
 * constant `plane4` (Game.prob_map, game.py:124-132) when cin == 4 — conv1 straight from the env's output.
 * Otherwise `in` is f32[batch][cin][side][side], cin 3 or 4 (conv1 on its planes) or a multiple of 8.  weight is the nn.Conv2d parameter as it
 * is, f32[cout][cin][3][3] (the kernel reorders it while staging: no packed copy that could go stale).  pre_out (may be NULL) receives the
 * value before the activation (what a backward pass needs).  bias / residual may be NULL.
 * Supported: side 12 or 26 (10x10 / 24x24 boards), cout 32 or 64; anything else TRON_ERR_UNSUPPORTED.
 * All buffers 16-byte aligned.                                                                             */
/* math selects the arithmetic.  TRON_CONV_F32: v_mfma_f32_16x16x4_f32, bit-for-bit an f32 fma chain.
 * TRON_CONV_F16X3: every operand split into two f16 halves (v = hi + lo 2^-11) and three
 * v_mfma_f32_16x16x32_f16 per k-slab (hi*hi, hi*lo, lo*hi; f32 accumulation) — 5x less matrix-pipe time at an
 * error of 2^-22 per product instead of 2^-24 (still within 1e-5 on the Q-values, tests/test_gpu_conv.py).
 * Shapes the split kernel has no instantiation for (cin not a multiple of 16) silently use the f32 kernel.  workspace: caller-owned device
 * scratch of at least tron_conv3x3_workspace(cin, cout) bytes for the split weights (written afresh by every call,
 * so nothing cached can go stale); only TRON_CONV_F16X3 needs it (NULL there means: use the f32 kernel).     */
enum { TRON_CONV_F32 = 0, TRON_CONV_F16X3 = 1, TRON_CONV_F16X3_PRESPLIT = 3 };
/* in_fmt: what `in` holds.  TRON_CONV_IN_SPLIT16 / out_split (may be NULL) chain layers of the split kernel without
 * re-splitting: out_split receives the layer's output as the operand halves the next layer stages — per image
 * [16-channel chunk][hi | lo][pixel][16 ci] f16, batch * cout * side * side * 4 bytes like the f32 tensor — and a
 * layer given that image as `in` copies it into LDS 16 bytes at a time.  `out` may then be NULL (an inner layer whose
 * f32 value nobody reads).  Only with TRON_CONV_F16X3.                                                          */
enum { TRON_CONV_IN_F32 = 0, TRON_CONV_IN_CODES = 1, TRON_CONV_IN_SPLIT16 = 2 };
int tron_conv3x3_fwd(const void *in, int32_t in_fmt, const float *weight, const float *bias,
                     const float *residual, float *out, float *pre_out, int64_t batch, int32_t cin,
                     int32_t cout, int32_t side, float plane4, int32_t apply_mish, int32_t math, void *workspace,
                     void *out_split, void *stream);
int64_t tron_conv3x3_workspace(int32_t cin, int32_t cout);
/* A forward pass over several layers can split all their weights in ONE launch and hand each layer its workspace with
 * math = TRON_CONV_F16X3_PRESPLIT (same arithmetic; the per-call split kernel is skipped): weights[k] f32[couts[k]]
 * [cins[k]][3][3] -> workspaces[k] (>= tron_conv3x3_workspace(cins[k], couts[k]) bytes, 16-byte aligned), n <= 8 layers.
 * The arrays are host arrays.  Still nothing cached: the caller splits again whenever it forwards again.            */
int tron_conv3x3_split_weights(const float *const *weights, const int32_t *cins, const int32_t *couts,
                               void *const *workspaces, int32_t n, void *stream);

/* ---- the same convolutions, weight-stationary, for gradient-free forwards (csrc/tron_conv_ws.hip) ----------------
 * The policy forward over 2N observations per env step (DDQN.py:90-110) and the two target forwards of a learn step
 * (DDQN.py:129-142) run conv1..conv6 (DQNNet.py:33-50) as a chain whose activations never take the f32 NCHW form:
 * PX16 = per image [hi | lo][channel octet][pixel][8 channels] f16, the value scaled by 2^-6 and split as
 * TRON_CONV_F16X3 splits it (v / 64 = hi + lo 2^-11): tron_px16_bytes(batch, channels, side) bytes (4 per element).
 * tron_conv1_px16: conv1 + bias + mish from the env's int8 observation codes [batch][side*side] (the planes of
 *   util.pop_up, util.py:11-37, and the constant fourth plane `plane4` when cin == 4, game.py:124-132, are implied by
 *   the codes: a table sum over the nine taps) -> PX16 with 32 channels.  weight f32[32][cin][3][3], cin 3 or 4;
 *   side 12, 26 or 34.
 * tron_conv3x3_ws_fwd: out = act(conv3x3(in, padding 1) + bias + residual): in / residual / out PX16 (residual
 *   and bias may be NULL, residual has cout channels); out_f32 / pre_f32 (may be NULL) also receive the result / the
 *   pre-activation as f32[batch][cout][side][side]; out_px16 may be NULL when one of them is given.  wfrag: the layer's
 *   weights as MFMA fragments, written by tron_conv3x3_ws_split_weights (weights[k] f32[couts[k]][cins[k]][3][3] ->
 *   workspaces[k], >= tron_conv3x3_ws_workspace(cin, cout) bytes; host arrays, n <= 8 layers, one launch; nothing is
 *   cached: a forward pass splits again).  Same arithmetic and accuracy as TRON_CONV_F16X3.
 *   Supported: side 12 or 26, (cin, cout) in {(32,32), (32,64), (64,64)}; anything else TRON_ERR_UNSUPPORTED.
 * tron_px16_to_f32: a PX16 image -> f32[batch][channels][side][side].
 * All buffers 16-byte aligned.                                                                                       */
int64_t tron_px16_bytes(int64_t batch, int32_t channels, int32_t side);
int tron_conv1_px16(const int8_t *codes, const float *weight, const float *bias, int32_t cin, float plane4,
                    int64_t batch, int32_t side, void *out_px16, void *stream);
int64_t tron_conv3x3_ws_workspace(int32_t cin, int32_t cout);
int tron_conv3x3_ws_split_weights(const float *const *weights, const int32_t *cins, const int32_t *couts,
                                  void *const *workspaces, int32_t n, void *stream);
int tron_conv3x3_ws_fwd(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16,
                        void *out_px16, float *out_f32, float *pre_f32, int64_t batch, int32_t cin, int32_t cout,
                        int32_t side, int32_t apply_mish, void *stream);
int tron_px16_to_f32(const void *in_px16, float *out, int64_t batch, int32_t channels, int32_t side, void *stream);
/* tron_px16_from_f32: the inverse — f32[batch][channels][side][side] (channels % 8 == 0) -> a PX16 image (tron_px16_bytes). */
int tron_px16_from_f32(const float *x, void *out_px16, int64_t batch, int32_t channels, int32_t side, void *stream);
/* K-FAC's input factor of a 3x3 / pad 1 / stride 1 convolution (kfac.py:28-58) from the layer's input as a PX16 image — what the
 * weight-stationary training chain holds anyway: gram f32[9 C][9 C] = scale * P^T P in the reference's patch order (channel major),
 * bitwise symmetric, without the patch matrix (csrc/tron_kfac_px.hip).  channels 32 or 64, side 12, 26 or 34;
 * workspace: tron_kfac_gram_px16_workspace(batch, channels, side) bytes (0: shape not covered).                                */
int64_t tron_kfac_gram_px16_workspace(int64_t batch, int32_t channels, int32_t side);
int tron_kfac_gram_px16(const void *x_px16, int64_t batch, int32_t channels, int32_t side, float scale, float *gram, void *workspace,
                        void *stream);

/* The weight gradient of the same convolutions (loss.backward() through conv1..conv6, DDQN.py:148):
 * grad_weight[co][ci][ky][kx] = sum over b, y, x of grad_pre[b][co][y][x] * in[b][ci][y+ky-1][x+kx-1], overwritten (not
 * accumulated), f32[cout][cin][3][3] like nn.Conv2d's weight.grad.  in: f32[batch][cin][side][side] (the layer's input),
 * grad_pre: f32[batch][cout][side][side] (the gradient at the convolution's output, before bias / activation — what
 * tron_bias_mish_bwd writes).  Split-f16 matrix-core arithmetic (TRON_CONV_F16X3's), f32 accumulation, sums in a fixed
 * order (deterministic).  grad_absmax: n_absmax per-block maxima of |grad_pre| on the device (tron_bias_mish_bwd leaves
 * them in its scratch) used to scale the gradient into f16's normal range, or NULL: a pre-pass finds the maximum.
 * Supported: side 12 with cin 3, 4, 32 or 64 and cout 32 or 64; side 26 or 34 (24x24 / 32x32 boards) with (cin, cout) in {(32,32),
 * (32,64), (64,64)}; side 26 or 34 with cin 3 or 4 and cout 32 (conv1 there: plain f32 FMAs, grad_absmax unused); anything else
 * TRON_ERR_UNSUPPORTED.  workspace: at least
 * tron_conv3x3_wgrad_workspace(cin, cout) bytes; all buffers 16-byte aligned.                                    */
int tron_conv3x3_wgrad(const float *in, const float *grad_pre, const float *grad_absmax, int32_t n_absmax,
                       float *grad_weight, int64_t batch, int32_t cin, int32_t cout, int32_t side, void *workspace,
                       void *stream);
int64_t tron_conv3x3_wgrad_workspace(int32_t cin, int32_t cout);
/* ... and the input gradient: grad_in[b][ci][y][x] = sum over co, ky, kx of grad_pre[b][co][y-ky+1][x-kx+1] *
 * weight[co][ci][ky][kx] — the same convolution run on the gradient with the weight's channel axes swapped and its taps
 * reversed (the kernel reads the forward layer's weight f32[cout][cin][3][3] that way; no transposed copy).  Same
 * arithmetic and scaling as the weight gradient (grad_absmax NULL: the activations' fixed scale).  side 12 or 26, cin 32
 * or 64, cout 16..64 in steps of 16; workspace: tron_conv3x3_workspace(cout, cin) bytes.                          */
int tron_conv3x3_dgrad(const float *grad_pre, const float *weight, const float *grad_absmax, int32_t n_absmax,
                       float *grad_in, int64_t batch, int32_t cin, int32_t cout, int32_t side, void *workspace,
                       void *stream);
/* The same input gradient carried through the activation of the layer below in the same launch — what autograd does
 * between two mish(conv(.)) layers of DQNNet.py:33-50 as three passes (convolution backward, the sum with the gradient
 * arriving over a residual connection, mish backward + bias sum):
 *   grad_pre_below = (dgrad(grad_pre, weight) + extra_grad) * mish'(pre_below)      f32[batch][cin][side][side]
 *   bias_grad_below[c] = sum over b, y, x of grad_pre_below (fixed order: deterministic), absmax_below[c] = max |grad_pre_below|
 * (the grad_absmax / n_absmax = cin the next tron_conv3x3_dgrad* / _wgrad call of the chain takes).  extra_grad (may be
 * NULL) and pre_below are laid out like the output.  side 12 or 26, cin and cout 32 or 64; workspace:
 * tron_conv3x3_dgrad_mish_workspace(batch, cin, cout, side) bytes (split weights + per-workgroup partial sums; 0 = shape
 * not supported), 16-byte aligned like every tensor.                                                              */
int tron_conv3x3_dgrad_mish(const float *grad_pre, const float *weight, const float *grad_absmax, int32_t n_absmax,
                            const float *extra_grad, const float *pre_below, float *grad_pre_below, float *bias_grad_below,
                            float *absmax_below, int64_t batch, int32_t cin, int32_t cout, int32_t side, void *workspace,
                            void *stream);
int64_t tron_conv3x3_dgrad_mish_workspace(int64_t batch, int32_t cin, int32_t cout, int32_t side);

/* ---- the rest of the DQN net after the 3x3 trunk (Net/DQNNet.py:52-63) ------------------------------------------
 * q = actor2(mish(actor1(mish(fc2(mish(fc1(flatten(mish(conv7(pool(x)))))))))))  for gradient-free forwards
 * (dropout is the identity in eval mode): avg-pool 3/2/1, conv7 (7x7, stride 2, pad 3) as the dense
 * [64*6*6] -> [64*3*3] map it is on 6x6 planes, and the three linear layers, all on the f16 matrix cores with
 * every operand split in two halves (TRON_CONV_F16X3's arithmetic), actor2 in f32.  trunk_out: f32[batch][64][side]
 * [side] (conv6's output).  Weights are the nn.Module parameters as they are (conv7_w f32[64][64][7][7], fc1_w
 * f32[256][576], fc2_w f32[128][256], actor1_w f32[64][128], actor2_w f32[4][64]); they are split into `workspace`
 * afresh by every call.  q_out f32[batch][4] and / or greedy_out int8[batch] (first maximum, as torch.argmax).
 * side 12 (10x10 boards: flatten 64*3*3 = what the reference's fc1 expects) or 26 (24x24 boards: pooled 13x13, conv7 as
 * an implicit GEMM over its 49 taps, flatten 64*7*7: fc1_w f32[256][3136]); other sides TRON_ERR_UNSUPPORTED.
 * workspace: at least tron_dqn_head_workspace(batch, side) bytes, 16-byte aligned.                               */
int tron_dqn_head_fwd(const float *trunk_out, int64_t batch, int32_t side, const float *conv7_w,
                      const float *conv7_b, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                      const float *fc2_b, const float *actor1_w, const float *actor1_b, const float *actor2_w,
                      const float *actor2_b, void *workspace, float *q_out, int8_t *greedy_out, void *stream);
int64_t tron_dqn_head_workspace(int64_t batch, int32_t side);
/* The same with the trunk's output handed over as the PX16 image conv6 of the weight-stationary chain writes
 * (tron_conv3x3_ws_fwd; 64 channels): the pooling reads it directly, no f32 tensor in between.  Same workspace size. */
int tron_dqn_head_fwd_px16(const void *trunk_px16, int64_t batch, int32_t side, const float *conv7_w,
                           const float *conv7_b, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                           const float *fc2_b, const float *actor1_w, const float *actor1_b, const float *actor2_w,
                           const float *actor2_b, void *workspace, float *q_out, int8_t *greedy_out, void *stream);
/* 10x10 boards (side 12): conv6 and the pooling as ONE launch (csrc/tron_conv_ws_pool.hip; DQNNet.py:48-52: mish(conv6(x) + res),
 * AvgPool2d(3, 2, 1)) — conv6's 64 x 12 x 12 output stays in LDS, what reaches memory are the pooled rows conv7's GEMM reads:
 * `pooled` = tron_pooled12_bytes(batch) bytes, [hi rows | lo rows], a row = 64 x 6 x 6 split f16.  in_px16, res_px16: PX16
 * images [batch][64][12][12]; wfrag: conv6's fragment image (tron_conv3x3_ws_split_weights).  Same bits as
 * tron_conv3x3_ws_fwd followed by the pooling inside tron_dqn_head_fwd_px16.  tron_dqn_head_fwd_pooled: the head from those rows
 * (side must be 12; workspace tron_dqn_head_workspace(batch, 12)).                                                        */
int64_t tron_pooled12_bytes(int64_t batch);
int tron_conv3x3_ws_fwd_pool12(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16,
                               void *pooled, int64_t batch, void *stream);
/* The learner's forward of the same layer pair (DDQN.py:127): conv6's pre-activation is kept as the PX16 image pre_px16
 * (as tron_conv3x3_ws_train_fwd keeps it), the pooled output leaves as f32 planes pooled_f32 [batch][64][6][6] with the bits
 * tron_pool12_px16 gives on conv6's PX16 output; that output itself — read by nothing but the pooling — is not stored. */
int tron_conv3x3_ws_train_fwd_pool12(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16,
                                     void *pre_px16, float *pooled_f32, int64_t batch, void *stream);
int tron_dqn_head_fwd_pooled(const void *pooled, int64_t batch, int32_t side, const float *conv7_w,
                             const float *conv7_b, const float *fc1_w, const float *fc1_b, const float *fc2_w,
                             const float *fc2_b, const float *actor1_w, const float *actor1_b, const float *actor2_w,
                             const float *actor2_b, void *workspace, float *q_out, int8_t *greedy_out, void *stream);
/* The same two layers on the training path (their products are library GEMMs on conv7's dense form):
 * tron_pool12: AvgPool2d(3, stride 2, padding 1) (Net/DQNNet.py:20,52) of `planes` 12x12 f32 planes -> 6x6 (backward 0),
 * or its gradient 6x6 -> 12x12 (backward 1).  tron_conv7_dense: fold 0: conv7's weight f32[cout][cin][7][7] -> the
 * matrix f32[cout*9][cin*36] that maps a flattened 6x6 input to the flattened 3x3 output of the 7x7 / stride 2 / pad 3
 * convolution (DQNNet.py:22,53); fold 1: that matrix's gradient -> the weight's gradient.                          */
int tron_pool12(const float *x, float *y, int64_t planes, int32_t backward, void *stream);
/* The forward pooling alone at both supported board sizes, side 12 or 26 -> side/2 (for a caller that keeps conv7 on a
 * library: Net.infer in exact-f32 mode), also 34 -> 17 (32x32 boards); other sides TRON_ERR_UNSUPPORTED.              */
int tron_pool_s2(const float *x, float *y, int64_t planes, int32_t side, void *stream);
/* ... and its gradient: grad_x f32[planes][side][side] from grad_y f32[planes][side/2][side/2] (what autograd needs of
 * DQNNet.py:52 / ACNet.py's pooling on the training path).  side 12, 26 or 34 (tron_pool_s2 likewise).                */
int tron_pool_s2_bwd(const float *grad_y, float *grad_x, int64_t planes, int32_t side, void *stream);
int tron_conv7_dense(const float *src, float *dst, int32_t cout, int32_t cin, int32_t fold, void *stream);
/* The same layers on the training path at 24x24 / 32x32 boards — x = pool(x); x = mish(conv7(x)); x.view(-1, 64*O*O)
 * (Net/DQNNet.py:52-55; the actor-critic nets' tails in Net/ACNet.py alike) and what loss.backward() computes for them
 * (DDQN.py:148) — on the split-f16 matrix cores end to end, replacing the library's NHWC convolution kernels and their layout
 * transposes.  side 26 (13x13 pooled planes, O = 7) or 34 (17x17, O = 9); conv7 is 64 -> 64 channels, 7x7 / stride 2 / pad 3.
 *   fwd: x f32[batch][64][side][side] -> pooled planes kept channels-last and split in `saved` (tron_pool_conv7_saved_bytes),
 *        pre f32[batch][O*O][64] = conv7 + bias (channels-last), y f32[batch][64*O*O] = mish(pre) in NCHW-flatten order.
 *   bwd: grad_y f32[batch][64*O*O] -> grad_x f32[batch][64][side][side] (four implicit GEMMs, one per parity class of the
 *        pooled pixel, then the pooling's backward), grad_weight f32[64][64][7][7] (operands read transposed out of LDS, the
 *        stride-2 im2col done by the read addresses; per-slice partial sums added in a fixed order), grad_bias f32[64];
 *        each may be NULL.  The gradient is scaled by a power of two taken from max |grad_y| on the device.
 * workspace: tron_pool_conv7_workspace(batch, side) bytes, 16-byte aligned like every pointer here; batch <= 2^20.        */
int64_t tron_pool_conv7_saved_bytes(int64_t batch, int32_t side);
int64_t tron_pool_conv7_workspace(int64_t batch, int32_t side);
int tron_pool_conv7_fwd(const float *x, int64_t batch, int32_t side, const float *weight, const float *bias, void *saved, float *pre,
                        float *y, void *workspace, void *stream);
int tron_pool_conv7_bwd(const float *grad_y, const float *pre, const void *saved, const float *weight, int64_t batch, int32_t side,
                        float *grad_x, float *grad_weight, float *grad_bias, void *workspace, void *stream);
/* conv7 alone on the same kernels, NCHW in and out — the nn.Conv2d module itself, for callers that need the layer's input and
 * output as tensors: KFACOptimizer's hooks on the actor-critic nets' conv7 (kfac.py:156-189; Net/ACNet.py:68).
 *   fwd: pooled f32[batch][64][P][P] (P = pooled_side, 13 or 17) -> y f32[batch][64][O][O] = conv7(pooled) + bias (bias may be NULL:
 *        KFACOptimizer's SplitBias moves it out of the module); `saved` as above (tron_pool_conv7_saved_bytes(batch, 2 P)).
 *   bwd: grad_y f32[batch][64][O][O] -> grad_pooled f32[batch][64][P][P], grad_weight f32[64][64][7][7], grad_bias f32[64]; each may be NULL.
 * workspace: tron_pool_conv7_workspace(batch, 2 P) bytes.                                                                  */
int tron_conv7_fwd(const float *pooled, int64_t batch, int32_t pooled_side, const float *weight, const float *bias, void *saved, float *y,
                   void *workspace, void *stream);
int tron_conv7_bwd(const float *grad_y, const void *saved, const float *weight, int64_t batch, int32_t pooled_side, float *grad_pooled,
                   float *grad_weight, float *grad_bias, void *workspace, void *stream);

/* Wait for `stream` and report what the kernels queued on it did: launch_status-style calls above only
 * see a REJECTED launch; a fault inside a kernel surfaces at the next synchronisation.  Returns TRON_OK
 * or TRON_ERR_LAUNCH (the HIP error is consumed).  The one blocking call of this ABI.               */
int tron_synchronize(void *stream);

const char *tron_strerror(int status);
/* ---- the learner's trunk on the weight-stationary design (csrc/tron_conv_ws_train.hip; DDQN.py:115-151 on DQNNet.py:33-50) ----
 * Every tensor between conv1 and conv6 is a PX16 image (tron_px16_bytes): activations a_k and pre-activations z_k at the fixed
 * 2^-6, GRADIENT images as PX16 of g * s with s a power of two kept in a device record info = f32[68]: {s, 1 / s, -, -, max |g| of each
 * channel [<= 64]}.
 *
 * tron_conv1_px16_train / tron_conv3x3_ws_train_fwd: tron_conv1_px16 / tron_conv3x3_ws_fwd (mish on), which also write the
 *   layer's pre-activation image pre_px16 (what the backward pass takes mish' of); out_f32 (optional, f32 NCHW) for the last
 *   layer, whose consumer is the head.
 * tron_conv3x3_ws_split_weights_bwd: the fragment images of the input gradient's weights — W'[ci][co][tap] = W[co][ci][8 - tap] —
 *   for n layers in one launch (workspaces[k]: tron_conv3x3_ws_workspace(couts[k], cins[k]) bytes) and wnorms[k] = the largest
 *   absolute row sum of W' (the bound the next gradient image's scale is chosen from).  cins / couts are the FORWARD layers'.
 * tron_px16_grad_from_f32: the chain's entry: grad_px16 = PX16 of (grad_out * mish'(pre)) * s, grad_out f32 [batch][channels][side^2]
 *   (the gradient at the trunk's output); scale4 = tron_absmax_pow2(grad_out, n, 14, ...)'s record (s = scale4[0]); bias_grad
 *   (may be NULL) = the column sums; grad_info = the image's record (f32[68]).  workspace: tron_px16_grad_workspace(batch, channels) bytes.
 * tron_conv3x3_ws_dgrad: out = (conv^T(grad, W) + extra) * mish'(pre_below) as a gradient image (out_px16; out_f32 optional, f32
 *   NCHW unscaled: the consumer is conv1's weight gradient), extra (may be NULL) = the gradient image arriving along a residual
 *   connection, bias_grad_below f32[cin] (may be NULL) = the column sums of out, out_info = its record.  cin / cout: the FORWARD
 *   layer's.  workspace: tron_conv3x3_ws_dgrad_workspace(cin, cout) bytes.  Deterministic (fixed-order sums).
 * tron_conv3x3_wgrad_px16: grad_weight f32[cout][cin][3][3] = sum over images and pixels of grad * shifted in (both PX16),
 *   workspace: tron_conv3x3_wgrad_px16_workspace(batch, cin, cout, side) bytes (0: shape not covered).  Deterministic.
 * Shapes: side 12 / 26, (cin, cout) in {(32,32), (32,64), (64,64)}.                                                        */
int tron_conv1_px16_train(const int8_t *codes, const float *weight, const float *bias, int32_t cin, float plane4,
                          int64_t batch, int32_t side, void *out_px16, void *pre_px16, void *stream);
int tron_conv3x3_ws_train_fwd(const void *in_px16, const void *wfrag, const float *bias, const void *res_px16,
                              void *out_px16, float *out_f32, void *pre_px16, int64_t batch, int32_t cin, int32_t cout,
                              int32_t side, void *stream);
int tron_conv3x3_ws_split_weights_bwd(const float *const *weights, const int32_t *cins, const int32_t *couts,
                                      void *const *workspaces, float *wnorms, int32_t n, void *stream);
int64_t tron_px16_grad_workspace(int64_t batch, int32_t channels);
int tron_px16_grad_from_f32(const float *grad_out, const void *pre_px16, const float *scale4, int64_t batch, int32_t channels,
                            int32_t side, void *grad_px16, float *grad_info, float *bias_grad, void *workspace, void *stream);
/* The same entry from the gradient of the POOLED planes (AvgPool2d(3, 2, 1) behind conv6, DQNNet.py:52): the pooling's backward is
 * taken on the fly, so the f32 gradient planes of the trunk's output are never written.  grad_pooled: f32 [batch][channels][(side/2)^2]
 * (channels_last = 0: the 12x12 boards' dense conv7 path) or f32 [batch][(side/2)^2][channels] (channels_last = 1:
 * tron_pool_conv7_bwd_pooled); scale4 = tron_absmax_pow2(grad_pooled, n, 15, ...)'s record.                                   */
int tron_px16_grad_from_pooled(const float *grad_pooled, int32_t channels_last, const void *pre_px16, const float *scale4,
                               int64_t batch, int32_t channels, int32_t side, void *grad_px16, float *grad_info,
                               float *bias_grad, void *workspace, void *stream);
/* The head's first two layers behind a trunk that ends in a PX16 image: tron_pool12_px16 = tron_pool12's forward (pooled f32
 * [batch][64][6][6]) from conv6's PX16 output at 12x12; tron_pool_conv7_fwd_px16 = tron_pool_conv7_fwd reading it at 26x26;
 * tron_pool_conv7_bwd_pooled = tron_pool_conv7_bwd stopping at the pooled gradient (grad_pooled f32 [batch][13*13][64]).    */
int tron_pool12_px16(const void *x_px16, float *pooled, int64_t batch, void *stream);
int tron_pool_conv7_fwd_px16(const void *x_px16, int64_t batch, int32_t side, const float *weight, const float *bias, void *saved,
                             float *pre, float *y, void *workspace, void *stream);
int tron_pool_conv7_bwd_pooled(const float *grad_y, const float *pre, const void *saved, const float *weight, int64_t batch,
                               int32_t side, float *grad_pooled, float *grad_weight, float *grad_bias, void *workspace, void *stream);
int64_t tron_conv3x3_ws_dgrad_workspace(int32_t cin, int32_t cout);
int tron_conv3x3_ws_dgrad(const void *grad_px16, const float *grad_info, const void *wfrag_rot, const float *wnorm,
                          const void *extra_px16, const float *extra_info, const void *pre_below_px16, void *out_px16,
                          float *out_f32, float *out_info, float *bias_grad_below, int64_t batch, int32_t cin, int32_t cout,
                          int32_t side, void *workspace, void *stream);
int64_t tron_conv3x3_wgrad_px16_workspace(int64_t batch, int32_t cin, int32_t cout, int32_t side);
int tron_conv3x3_wgrad_px16(const void *in_px16, const void *grad_px16, const float *grad_info, float *grad_weight,
                            int64_t batch, int32_t cin, int32_t cout, int32_t side, void *workspace, void *stream);
/* conv1's weight gradient (DDQN.py:148 on DQNNet.py:10,34) straight from the int8 observation codes and the gradient image at
 * conv1's pre-activation (32 channels, PX16 of g * s with its record grad_info): grad_weight f32[32][cin][3][3], cin = 3 or 4
 * (the fourth plane = plane4 on every cell, game.py:124-132).  The f32 planes util.pop_up would build are never written.
 * side 12 or 26; workspace: tron_conv1_wgrad_px16_workspace(batch, side) bytes (0: unsupported).                             */
int64_t tron_conv1_wgrad_px16_workspace(int64_t batch, int32_t side);
int tron_conv1_wgrad_px16(const int8_t *codes, const void *grad_px16, const float *grad_info, int64_t batch, int32_t side,
                          int32_t cin, float plane4, float *grad_weight, void *workspace, void *stream);

int tron_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* TRON_HIP_H */
