// Internal: the two implementations behind tron_conv3x3_fwd (include/tron_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// tron_conv_f16.hip: split-f16 matrix-core path; TRON_ERR_UNSUPPORTED when it has no instantiation for the shape.
// `workspace` (>= tron_conv3x3_f16x3_workspace(cin, cout) bytes, 16-byte aligned) receives the split weights.
// dgrad: the data-gradient call — `weight` is the forward layer's [cin][cout][3][3], read with the channel axes swapped
// and the taps reversed.  grad_absmax (may be NULL): `in` is a gradient; scale it by the power of two its per-block maxima
// give instead of the activations' fixed 2^-6.
// presplit: `workspace` already holds this layer's split weights (tron_conv3x3_split_weights).
// in_fmt: TRON_CONV_IN_*; out_split (may be NULL): also emit the output as the split-f16 image the next layer stages.
int tron_conv3x3_f16x3(const void *in, int in_fmt, const float *weight, const float *bias, const float *residual,
                       float *out, float *pre_out, int64_t batch, int cin, int cout, int side, float plane4,
                       int apply_mish, void *workspace, void *out_split, int dgrad, const float *grad_absmax,
                       int n_absmax, int presplit, hipStream_t st);
int64_t tron_conv3x3_f16x3_workspace(int cin, int cout);
