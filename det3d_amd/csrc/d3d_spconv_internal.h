// Internal (C++ linkage) entry points shared between the sparse-conv translation units of libd3d_hip.so.
#pragma once
#include "d3d_common.h"

// round-1 weight-gradient kernels (one block per kernel offset, f32 atomics into dw; generic / f32 /
// small-channel fallbacks): used for dtypes and channel counts the tile kernel does not cover
int d3d_wgrad_legacy(const void *feat, const void *dout, const int32_t *nbr, int cap, int n_out, int kvol,
                     int cin, int cout, int dtype, float *dw, hipStream_t stream);

// d3d_set_tuning(1, blocks): run-time override of the weight-gradient kernel's target block count (-1 = default)
// d3d_set_tuning(2, 0|1): XCD-aware block -> tile order of the gather kernels (default off: measured neutral)
