// tron_minimax.hpp — internal: where the minimax kernel reads its boards from.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct MinimaxSrc {
    const int8_t *codes;    // observation codes of the player to move, board i at codes + i*stride; or nullptr
    size_t stride;
    const int8_t *grid;     // else [n][S*S] tiles, coded for `player` on the way in
    int player;             // 1 or 2 (selects the code table for `grid`, and the Philox counter)
    const uint4 *st4;       // env state words: tick for the Philox draw, done flag; or nullptr
    uint32_t seed, stream;
    const uint32_t *rnd;    // explicit draws (one per board) when st4 is null; nullptr = 0
};

int launch_minimax(const MinimaxSrc &src, int n, int S, int mode, int8_t *out_actions, int32_t *out_values,
                   int8_t *out_expanded, hipStream_t stream);
