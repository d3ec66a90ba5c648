// Device-resident replay memory (include/tron_hip.h, tron_replay_*).
//
// Replaces DDQN.ReplayBuffer (DDQN.py:167-203): deque(maxlen) -> a ring in HBM,
// random.sample -> distinct uniform slots drawn in-kernel (a Philox-keyed permutation
// of the filled slots, any batch size), np.vstack + .to(device) -> one gather kernel that also expands
// the stored int8 code planes (map.py:67-84) into the f32 pop_up planes
// (util.py:11-37) the CNN reads.  Nothing crosses PCIe.
#include "tron_device.hpp"
#include "../../include/tron_hip.h"

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <new>

using namespace tron;

struct tron_replay {
    int64_t capacity, size, head;
    int32_t cells;
    uint32_t seed, stream, calls;
    int device;
    int8_t *states, *next_states, *actions, *dones;
    float *rewards;
    int64_t *indices;      // last sampled slots
    int32_t max_batch;
    void *blob;
};

namespace {

constexpr int MAX_BATCH = 1 << 20;
constexpr int64_t MAX_CAPACITY = 1ll << 32;   // the sampler's permutation works on <= 32 bits

inline hipStream_t S_(void *s) { return reinterpret_cast<hipStream_t>(s); }
inline int launch_status() { return hipGetLastError() == hipSuccess ? TRON_OK : TRON_ERR_LAUNCH; }

// rows [0,n) of src -> slots [slot0, slot0+n) (no wrap inside one launch)
__global__ void k_push_planes(const int8_t *__restrict__ src, int8_t *__restrict__ dst, size_t nbytes)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | nbytes) & 3u) == 0) {
        const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
        uint32_t *d = reinterpret_cast<uint32_t *>(dst);
        for (size_t i = tid; i < nbytes / 4; i += stride) d[i] = s[i];
    } else {
        for (size_t i = tid; i < nbytes; i += stride) dst[i] = src[i];
    }
}

__global__ void k_push_scalars(int64_t n, const int8_t *__restrict__ action, const float *__restrict__ reward,
                               const int8_t *__restrict__ done, int8_t *__restrict__ actions,
                               float *__restrict__ rewards, int8_t *__restrict__ dones)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    actions[i] = action[i];
    rewards[i] = reward[i];
    dones[i] = done[i];
}

// random.sample(memory, k) (DDQN.py:191-200): k DISTINCT uniform slots, for any k <= size.
// Draw j is pi(j) for a keyed pseudo-random permutation pi of [0, size): the first k images of a random
// permutation are a uniform sample without replacement.  pi is a balanced Feistel network on 2h bits
// (4^h >= size, h minimal, so the domain is < 4 * size) with cycle-walking: an image >= size is pushed
// through the network again until it lands inside — a permutation restricted to the orbit of a subset
// is a permutation of the subset, so the draws are distinct by construction, and the walk ends because
// it started inside.  O(k) work, one thread per draw, any number of workgroups; the round keys come
// from Philox keyed (seed, stream) at counter (call), so every sample() call has its own permutation.
constexpr int FEISTEL_ROUNDS = 8;

__device__ __forceinline__ uint32_t feistel_f(uint32_t x, uint32_t k)
{
    x = (x ^ k) * 0x9E3779B1u;
    x ^= x >> 15;
    x *= 0x85EBCA77u;
    x ^= x >> 13;
    x *= 0xC2B2AE3Du;
    return x ^ (x >> 16);
}

__global__ __launch_bounds__(256) void k_sample_indices(uint64_t size, int batch, int half_bits, uint32_t seed,
                                                        uint32_t stream, uint32_t call, int64_t *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= batch) return;
    uint32_t key[FEISTEL_ROUNDS];
#pragma unroll
    for (int b = 0; b < FEISTEL_ROUNDS / 4; ++b)
        philox4x32_10(call, (uint32_t)b, 0u, 0x5A4D504Cu /* "SMPL" */, seed, stream, key + 4 * b);
    const uint32_t mask = (1u << half_bits) - 1u;
    uint64_t x = (uint64_t)j;
    do {
        uint32_t l = (uint32_t)(x >> half_bits) & mask, r = (uint32_t)x & mask;
#pragma unroll
        for (int t = 0; t < FEISTEL_ROUNDS; ++t) {
            const uint32_t n = l ^ (feistel_f(r, key[t]) & mask);
            l = r;
            r = n;
        }
        x = ((uint64_t)l << half_bits) | r;
    } while (x >= size);
    out[j] = (int64_t)x;
}

// gather + expand: codes int8[cells] -> planes f32[channels][cells]
__global__ __launch_bounds__(256) void k_sample_gather(const int64_t *__restrict__ idx, int batch, int cells,
                                                       int channels, float plane4,
                                                       const int8_t *__restrict__ states,
                                                       const int8_t *__restrict__ next_states,
                                                       const int8_t *__restrict__ actions,
                                                       const float *__restrict__ rewards,
                                                       const int8_t *__restrict__ dones, float *__restrict__ o_s,
                                                       int64_t *__restrict__ o_a, float *__restrict__ o_r,
                                                       float *__restrict__ o_s2, float *__restrict__ o_d)
{
    const int b = blockIdx.x >> 1;          // batch row
    const int which = blockIdx.x & 1;       // 0: state, 1: next_state
    const int64_t slot = idx[b];
    const int8_t *src = (which ? next_states : states) + (size_t)slot * cells;
    float *dst = (which ? o_s2 : o_s) + (size_t)b * channels * cells;
    if (!which && threadIdx.x == 0) {
        o_a[b] = (int64_t)actions[slot];
        o_r[b] = rewards[slot];
        o_d[b] = (float)dones[slot];
    }
    if ((cells & 3) == 0) {
        const int D = cells >> 2;
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
        float4 *d4 = reinterpret_cast<float4 *>(dst);
        for (int i = threadIdx.x; i < D; i += blockDim.x) {
            const uint32_t w = s32[i];
            float wl[4], my[4], en[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int v = (int)(int8_t)(w >> (8 * k));
                wl[k] = (v == -1) ? 1.0f : 0.0f;                                  // util.py:18-19
                my[k] = (v == -2) ? 1.0f : (v == 10) ? 10.0f : 0.0f;              // util.py:20-21,26-27
                en[k] = (v == -3) ? 1.0f : (v == -10) ? 10.0f : 0.0f;             // util.py:22-25
            }
            d4[i] = make_float4(wl[0], wl[1], wl[2], wl[3]);
            d4[D + i] = make_float4(my[0], my[1], my[2], my[3]);
            d4[2 * D + i] = make_float4(en[0], en[1], en[2], en[3]);
            if (channels == 4) d4[3 * D + i] = make_float4(plane4, plane4, plane4, plane4);
        }
    } else {
        for (int i = threadIdx.x; i < cells; i += blockDim.x) {
            const int v = src[i];
            dst[i] = (v == -1) ? 1.0f : 0.0f;
            dst[cells + i] = (v == -2) ? 1.0f : (v == 10) ? 10.0f : 0.0f;
            dst[2 * cells + i] = (v == -3) ? 1.0f : (v == -10) ? 10.0f : 0.0f;
            if (channels == 4) dst[3 * cells + i] = plane4;
        }
    }
}

// the same gather without the expansion: the rows stay int8 observation codes (what conv1's CODES staging and the
// weight-stationary chain read) — 1 byte per cell out instead of 12 / 16
__global__ __launch_bounds__(64) void k_sample_gather_codes(const int64_t *__restrict__ idx, int cells,
                                                            const int8_t *__restrict__ states,
                                                            const int8_t *__restrict__ next_states,
                                                            const int8_t *__restrict__ actions,
                                                            const float *__restrict__ rewards,
                                                            const int8_t *__restrict__ dones, int8_t *__restrict__ o_s,
                                                            int64_t *__restrict__ o_a, float *__restrict__ o_r,
                                                            int8_t *__restrict__ o_s2, float *__restrict__ o_d)
{
    const int b = blockIdx.x >> 1, which = blockIdx.x & 1;
    const int64_t slot = idx[b];
    const int8_t *src = (which ? next_states : states) + (size_t)slot * cells;
    int8_t *dst = (which ? o_s2 : o_s) + (size_t)b * cells;
    if (!which && threadIdx.x == 0) {
        o_a[b] = (int64_t)actions[slot];
        o_r[b] = rewards[slot];
        o_d[b] = (float)dones[slot];
    }
    if ((cells & 3) == 0) {                                              // (ring rows and output rows are 4-byte aligned then)
        const uint32_t *s32 = reinterpret_cast<const uint32_t *>(src);
        uint32_t *d32 = reinterpret_cast<uint32_t *>(dst);
        for (int i = threadIdx.x; i < (cells >> 2); i += blockDim.x) d32[i] = s32[i];
    } else {
        for (int i = threadIdx.x; i < cells; i += blockDim.x) dst[i] = src[i];
    }
}

__global__ void k_copy_i64(const int64_t *src, int64_t *dst, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

inline bool bad(tron_replay_handle r)
{
    if (!r) return true;
    int dev = -1;
    return hipGetDevice(&dev) != hipSuccess || dev != r->device;
}

}  // namespace

extern "C" {

int tron_replay_create(int64_t capacity, int32_t cells, uint32_t seed, uint32_t rng_stream, tron_replay_handle *out)
{
    if (!out) return TRON_ERR_BAD_ARG;
    *out = nullptr;
    if (capacity < 1 || cells < 1 || cells > 98 * 98) return TRON_ERR_BAD_ARG;
    if (capacity > MAX_CAPACITY) return TRON_ERR_UNSUPPORTED;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return TRON_ERR_NO_DEVICE; }
    tron_replay *r = new (std::nothrow) tron_replay();
    if (!r) return TRON_ERR_ALLOC;
    r->capacity = capacity; r->size = 0; r->head = 0; r->cells = cells;
    r->seed = seed; r->stream = rng_stream; r->calls = 0; r->device = dev;
    r->max_batch = (int32_t)(capacity < MAX_BATCH ? capacity : MAX_BATCH);
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t cap = (size_t)capacity;
    const size_t o_s = 0, o_s2 = align(o_s + cap * cells + 16), o_a = align(o_s2 + cap * cells + 16),
                 o_d = align(o_a + cap), o_r = align(o_d + cap), o_i = align(o_r + 4 * cap),
                 total = align(o_i + 8 * (size_t)r->max_batch);
    char *blob = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&blob), total) != hipSuccess) {
        (void)hipGetLastError();
        delete r;
        return TRON_ERR_ALLOC;
    }
    r->blob = blob;
    r->states = reinterpret_cast<int8_t *>(blob + o_s);
    r->next_states = reinterpret_cast<int8_t *>(blob + o_s2);
    r->actions = reinterpret_cast<int8_t *>(blob + o_a);
    r->dones = reinterpret_cast<int8_t *>(blob + o_d);
    r->rewards = reinterpret_cast<float *>(blob + o_r);
    r->indices = reinterpret_cast<int64_t *>(blob + o_i);
    *out = r;
    return TRON_OK;
}

int tron_replay_destroy(tron_replay_handle r)
{
    if (!r) return TRON_ERR_BAD_ARG;
    (void)hipFree(r->blob);
    delete r;
    return TRON_OK;
}

int tron_replay_size(tron_replay_handle r, int64_t *size, int64_t *capacity)
{
    if (!r) return TRON_ERR_BAD_ARG;
    if (size) *size = r->size;
    if (capacity) *capacity = r->capacity;
    return TRON_OK;
}

int tron_replay_push(tron_replay_handle r, int64_t n, const int8_t *state, const int8_t *action, const float *reward,
                     const int8_t *next_state, const int8_t *done, void *stream)
{
    if (bad(r)) return r ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (n < 0 || n > r->capacity || (n > 0 && (!action || !reward || !next_state || !done)))   // state NULL: tron_replay_push_states wrote it
        return TRON_ERR_BAD_ARG;
    int64_t row = 0;
    while (row < n) {                                   // at most two segments (ring wrap)
        const int64_t seg = (n - row < r->capacity - r->head) ? n - row : r->capacity - r->head;
        const size_t nbytes = (size_t)seg * r->cells;
        size_t blocks = (nbytes / 4 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (blocks < 1) blocks = 1;
        if (state)
            hipLaunchKernelGGL(k_push_planes, dim3((unsigned)blocks), dim3(256), 0, S_(stream), state + (size_t)row * r->cells,
                               r->states + (size_t)r->head * r->cells, nbytes);
        hipLaunchKernelGGL(k_push_planes, dim3((unsigned)blocks), dim3(256), 0, S_(stream),
                           next_state + (size_t)row * r->cells, r->next_states + (size_t)r->head * r->cells, nbytes);
        hipLaunchKernelGGL(k_push_scalars, dim3((unsigned)((seg + 255) / 256)), dim3(256), 0, S_(stream), seg,
                           action + row, reward + row, done + row, r->actions + r->head, r->rewards + r->head,
                           r->dones + r->head);
        if (launch_status() != TRON_OK) return TRON_ERR_LAUNCH;
        row += seg;
        r->head = (r->head + seg) % r->capacity;
        r->size = (r->size + seg < r->capacity) ? r->size + seg : r->capacity;
    }
    return TRON_OK;
}

int tron_replay_push_states(tron_replay_handle r, int64_t n, const int8_t *state, void *stream)
{
    if (bad(r)) return r ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (n < 0 || n > r->capacity || (n > 0 && !state)) return TRON_ERR_BAD_ARG;
    int64_t row = 0, head = r->head;                     // where the NEXT tron_replay_push will put its rows; not advanced here
    while (row < n) {
        const int64_t seg = (n - row < r->capacity - head) ? n - row : r->capacity - head;
        const size_t nbytes = (size_t)seg * r->cells;
        size_t blocks = (nbytes / 4 + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(k_push_planes, dim3((unsigned)blocks), dim3(256), 0, S_(stream), state + (size_t)row * r->cells,
                           r->states + (size_t)head * r->cells, nbytes);
        if (launch_status() != TRON_OK) return TRON_ERR_LAUNCH;
        row += seg;
        head = (head + seg) % r->capacity;
    }
    return TRON_OK;
}

int tron_replay_sample(tron_replay_handle r, int32_t batch, int32_t channels, float plane4, float *states,
                       int64_t *actions, float *rewards, float *next_states, float *dones, void *stream)
{
    if (bad(r)) return r ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (batch < 1 || batch > r->max_batch || (channels != 3 && channels != 4)) return TRON_ERR_BAD_ARG;
    if (!states || !actions || !rewards || !next_states || !dones) return TRON_ERR_BAD_ARG;
    if (r->size < batch) return TRON_ERR_BAD_ARG;       // random.sample raises ValueError likewise
    const uint32_t call = r->calls++;
    int half_bits = 1;                                   // smallest h with 4^h >= size
    while (half_bits < 16 && (1ull << (2 * half_bits)) < (uint64_t)r->size) ++half_bits;
    hipLaunchKernelGGL(k_sample_indices, dim3((batch + 255) / 256), dim3(256), 0, S_(stream), (uint64_t)r->size, batch,
                       half_bits, r->seed, r->stream, call, r->indices);
    hipLaunchKernelGGL(k_sample_gather, dim3(2 * batch), dim3(256), 0, S_(stream), r->indices, batch, r->cells,
                       channels, plane4, r->states, r->next_states, r->actions, r->rewards, r->dones, states, actions,
                       rewards, next_states, dones);
    return launch_status();
}

int tron_replay_sample_codes(tron_replay_handle r, int32_t batch, int8_t *states, int64_t *actions, float *rewards,
                             int8_t *next_states, float *dones, void *stream)
{
    if (bad(r)) return r ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (batch < 1 || batch > r->max_batch) return TRON_ERR_BAD_ARG;
    if (!states || !actions || !rewards || !next_states || !dones) return TRON_ERR_BAD_ARG;
    if ((reinterpret_cast<uintptr_t>(states) | reinterpret_cast<uintptr_t>(next_states)) & 3u) return TRON_ERR_BAD_ARG;
    if (r->size < batch) return TRON_ERR_BAD_ARG;       // random.sample raises ValueError likewise
    const uint32_t call = r->calls++;                    // (the same draw sequence as tron_replay_sample: one permutation per call)
    int half_bits = 1;
    while (half_bits < 16 && (1ull << (2 * half_bits)) < (uint64_t)r->size) ++half_bits;
    hipLaunchKernelGGL(k_sample_indices, dim3((batch + 255) / 256), dim3(256), 0, S_(stream), (uint64_t)r->size, batch,
                       half_bits, r->seed, r->stream, call, r->indices);
    hipLaunchKernelGGL(k_sample_gather_codes, dim3(2 * batch), dim3(64), 0, S_(stream), r->indices, r->cells, r->states,
                       r->next_states, r->actions, r->rewards, r->dones, states, actions, rewards, next_states, dones);
    return launch_status();
}

int tron_replay_get_cursor(tron_replay_handle r, int64_t *head, int64_t *size, uint32_t *sample_calls)
{
    if (!r) return TRON_ERR_BAD_ARG;
    if (head) *head = r->head;
    if (size) *size = r->size;
    if (sample_calls) *sample_calls = r->calls;
    return TRON_OK;
}

int tron_replay_set_cursor(tron_replay_handle r, int64_t head, int64_t size, uint32_t sample_calls)
{
    if (!r || head < 0 || head >= r->capacity || size < 0 || size > r->capacity) return TRON_ERR_BAD_ARG;
    if (size < r->capacity && head != size) return TRON_ERR_BAD_ARG;     // a ring that has not wrapped yet fills from slot 0
    r->head = head;
    r->size = size;
    r->calls = sample_calls;
    return TRON_OK;
}

static int replay_copy(tron_replay_handle r, int64_t first, int64_t n, void *const user[5], bool to_ring, void *stream)
{
    if (bad(r)) return r ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (first < 0 || n < 0 || first + n > r->capacity) return TRON_ERR_BAD_ARG;
    if (n == 0) return TRON_OK;
    char *ring[5] = {reinterpret_cast<char *>(r->states) + (size_t)first * r->cells,
                     reinterpret_cast<char *>(r->next_states) + (size_t)first * r->cells,
                     reinterpret_cast<char *>(r->actions) + first, reinterpret_cast<char *>(r->rewards) + 4 * first,
                     reinterpret_cast<char *>(r->dones) + first};
    const size_t bytes[5] = {(size_t)n * r->cells, (size_t)n * r->cells, (size_t)n, 4 * (size_t)n, (size_t)n};
    for (int k = 0; k < 5; ++k) {
        if (!user[k]) continue;
        void *dst = to_ring ? (void *)ring[k] : user[k];
        const void *src = to_ring ? (const void *)user[k] : (const void *)ring[k];
        if (hipMemcpyAsync(dst, src, bytes[k], hipMemcpyDeviceToDevice, S_(stream)) != hipSuccess) {
            (void)hipGetLastError();
            return TRON_ERR_LAUNCH;
        }
    }
    return TRON_OK;
}

int tron_replay_export(tron_replay_handle r, int64_t first, int64_t n, int8_t *states, int8_t *next_states, int8_t *actions,
                       float *rewards, int8_t *dones, void *stream)
{
    void *const user[5] = {states, next_states, actions, rewards, dones};
    return replay_copy(r, first, n, user, false, stream);
}

int tron_replay_import(tron_replay_handle r, int64_t first, int64_t n, const int8_t *states, const int8_t *next_states,
                       const int8_t *actions, const float *rewards, const int8_t *dones, void *stream)
{
    void *const user[5] = {const_cast<int8_t *>(states), const_cast<int8_t *>(next_states), const_cast<int8_t *>(actions),
                           const_cast<float *>(rewards), const_cast<int8_t *>(dones)};
    return replay_copy(r, first, n, user, true, stream);
}

int tron_replay_indices(tron_replay_handle r, int32_t batch, int64_t *indices_out, void *stream)
{
    if (bad(r)) return r ? TRON_ERR_NO_DEVICE : TRON_ERR_BAD_ARG;
    if (batch < 1 || batch > r->max_batch || !indices_out) return TRON_ERR_BAD_ARG;
    hipLaunchKernelGGL(k_copy_i64, dim3((batch + 255) / 256), dim3(256), 0, S_(stream), r->indices, indices_out, batch);
    return launch_status();
}

}  // extern "C"
