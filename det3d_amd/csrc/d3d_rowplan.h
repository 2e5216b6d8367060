// Row plan of a strided convolution's data gradient (d3d_dgrad_plan_build -> d3d_sparse_conv_dgrad_planned).
//
// The data gradient of a strided layer is the gather kernel over the INPUT-stationary table: input site i receives
// through kernel offset k from the output cell o = (i + p - k) / s, which only exists where the division is exact.
// With stride 2 that is one offset in two per axis: an input row has 27 / 8 candidate offsets on average and which
// ones is decided by the parities r = (i + p) mod s of its coordinates alone.  In cell order the parities alternate
// from row to row, so every 16-row tile sees every offset and the kernel walks all 27 stages to fill an eighth of
// the MFMA rows (round 2: `<64,32>` 84 us at 11 % of HBM, the worst hand-written kernel by fraction x time).
//
// The plan processes the rows in parity-class order instead: a stable counting sort of the rows by class (at most 8
// classes), the class boundaries, and per class the list of offsets that can be non-empty.  A block whose rows share
// a class walks that list (1 to 8 stages instead of 27), a block across a boundary walks all of them.  Each row
// still adds its non-empty offsets in ascending k, so the results are bit-identical to the unplanned launch.
//
// int32 words:
//   [0]                    number of classes C (product of the strides, <= 8)
//   [1 .. 1 + C]           first plan row of each class, [1 + C] = number of rows
//   [16 + 32 c ...]        c = 0 .. C: count, then the offsets of class c (c = C: every offset, for mixed blocks);
//                          words 28..31 of a list: the first 24 offsets again, 5 bits each in two 64-bit words
//                          (what the kernel reads; a list that is not "every offset" has at most 18 entries)
//   [304 ...]              cap_in words: plan row -> input row (stable within a class)
//   [304 + cap_in ...]     8 words per 1024 rows: scratch of the sort
#pragma once
#define D3D_PLAN_START 1
#define D3D_PLAN_LISTS 16
#define D3D_PLAN_LIST_WORDS 32
#define D3D_PLAN_LIST_PACKED 28
#define D3D_PLAN_MAX_KVOL 27
#define D3D_PLAN_MAX_CLASSES 8
#define D3D_PLAN_PERM (D3D_PLAN_LISTS + (D3D_PLAN_MAX_CLASSES + 1) * D3D_PLAN_LIST_WORDS)
#define D3D_PLAN_SORT_ROWS 1024
