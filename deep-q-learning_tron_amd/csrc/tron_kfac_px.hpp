// tron_kfac_px.hpp — internal interface of csrc/tron_kfac_px.hip (called by tron_kfac_patch_gram in csrc/tron_kfac.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

bool tron_kfac_px_supported(int64_t batch, int C, int H, int W, int kh, int kw, int pad, int stride);
int64_t tron_kfac_px_workspace(int64_t batch, int C, int S);            // bytes (0: shape not covered)
int tron_kfac_px_gram(const float *x, int64_t batch, int C, int S, float scale, float *gram, void *workspace, hipStream_t st);
