/*
 * d3d.h -- C ABI of libd3d_hip.so: the MI355X (gfx950) hot path of Det3D,
 *          voxelize -> VFE -> sparse 3D conv -> dense -> rotated NMS.
 *
 * Conventions (all entry points):
 *   - plain C, no torch / pybind types; every pointer marked "dev" is a device
 *     (HBM) pointer owned by the caller, the library never allocates or frees
 *     caller-visible memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every
 *     call only enqueues work on that stream and returns, no hidden syncs;
 *   - returns 0 on success, a negative D3D_E* code otherwise; the message for
 *     the calling thread's last failure is d3d_last_error();
 *   - "count" outputs (number of voxels, number of active output sites) are
 *     written to DEVICE memory so a whole geometry chain can be enqueued
 *     without a host round trip; buffers are sized by the caller to the
 *     capacities documented per call;
 *   - workspace is caller-provided scratch; *_workspace_bytes() gives the size.
 *
 * Each entry point names the reference interface it replaces (path:line under
 * the Det3D tree) -- the reference has no FFI of its own for most of them: the
 * seams are Python functions and the un-vendored
 * spconv extension.
 */
#ifndef D3D_H
#define D3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define D3D_ABI_VERSION 12 /* 12: d3d_sparse_stack_forward / _backward; 11: d3d_ctx_current, pair-compacted sparse convolution (d3d_pair_table_*, d3d_sparse_conv_fwd_pairs); 10: caller-owned contexts (d3d_ctx_*), d3d_set_tuning knob 11; 9: d3d_conv_rulebook_chain / d3d_subm_rulebook_ranked, training = 2 of d3d_bn_apply, d3d_set_tuning knob 10; 8: d3d_conv2d_nhwc_dgrad_s2; 7: residual forms of the BatchNorm sweeps, D3dConvEpilogue.addend; 4: deterministic BatchNorm reductions (fold workspaces instead of zeroed atomic accumulators), dense NHWC conv; 5: planned strided data gradient, d3d_adam_step; 6: deterministic PillarFeatureNet sums (workspace arguments) */

#define D3D_OK 0
#define D3D_EINVAL -1    /* bad argument / unsupported shape           */
#define D3D_ECAPACITY -2 /* workspace or key space too small            */
#define D3D_EHIP -3      /* HIP runtime error (see d3d_last_error())    */

/* feature element types of the sparse-conv kernels */
#define D3D_F32 0
#define D3D_BF16 1

typedef void *d3d_stream_t;

int d3d_abi_version(void);
const char *d3d_last_error(void);

/* ------------------------------------------------------------------------
 * Voxelizer.  Replaces det3d/ops/point_cloud/point_cloud_ops.py:112-184
 * (points_to_voxel + _points_to_voxel_reverse_kernel :7-55), its C++ twin
 * point_cloud_ops.h:12-71, VoxelGenerator.generate
 * (det3d/core/input/voxel_generator.py:19-27) and, with `mean_feats`, the
 * mean VFE det3d/models/readers/voxel_encoder.py:206-211.  One call handles a
 * whole batch laid out like collate_kitti does
 * (det3d/torchie/parallel/collate.py:100-137): frames concatenated along dim
 * 0, frame b = rows [frame_offsets[b], frame_offsets[b+1]).
 *
 *  points        dev  [n_points, point_stride] f32 (x,y,z,extra...)
 *  frame_offsets dev  [batch+1] i32
 *  voxel_size[3], range[6] (xyzxyz), grid[3] (x,y,z) -- host; grid is
 *                 round((hi-lo)/vs) computed in f32 exactly as the reference.
 *  max_points T, max_voxels M (per frame, with the reference's early `break`)
 * outputs (capacity rows = batch*M):
 *  coors      dev [batch*M, 4] i32 (b, z, y, x), frames packed back to back
 *  num_points dev [batch*M]    i32
 *  voxels     dev [batch*M, T, point_stride] f32 zero padded, or NULL
 *  mean_feats dev [batch*M, n_mean_ch] f32 = sum_t voxels[:, t, :n_mean_ch] /
 *             num_points, or NULL
 *  num_voxels dev [batch+1] i32: per-frame voxel count, then the total
 */
size_t d3d_voxelize_workspace_bytes(int64_t n_points, int batch, int max_points, int max_voxels);
int d3d_voxelize(const float *points, int64_t n_points, int point_stride,
                 const int32_t *frame_offsets, int batch, const float *voxel_size,
                 const float *range, const int32_t *grid, int max_points, int max_voxels,
                 int32_t *coors, int32_t *num_points, float *voxels, float *mean_feats,
                 int n_mean_ch, int32_t *num_voxels, void *workspace, size_t workspace_bytes,
                 d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Rulebooks.  Replace spconv's get_indice_pairs (called from
 * SubMConv3d/SparseConv3d.forward; call sites det3d/models/backbones/scn.py:
 * 22-43,106-157,323-355; spconv itself is not vendored -- SURVEY.md App. A).
 * Layout is output-stationary: nbr[k*cap + o] = input row that reaches output
 * row o through kernel offset k (k = (kz*kH + ky)*kW + kx), or -1.
 *
 *  coords dev [n,4] i32 (b,z,y,x); n is read from DEVICE memory (*n_dev) so
 *  the chain voxelize -> rulebooks needs no host sync; `cap` bounds n.
 *  pair_count dev [kvol] i32: pairs per offset (centre included), or NULL to skip the
 *  counting (it costs same-address atomics; only reporting needs it).
 */
size_t d3d_subm_rulebook_workspace_bytes(int cap);
int d3d_subm_rulebook(const int32_t *coords, const int32_t *n_dev, int cap, int batch,
                      const int32_t *spatial_shape /* host [3] z,y,x */,
                      const int32_t *ksize /* host [3] */, int32_t *nbr /* dev [kvol*cap] */,
                      int32_t *pair_count, void *workspace, size_t workspace_bytes,
                      d3d_stream_t stream);

/* strided (regular) sparse conv: out = floor((in + 2p - k)/s) + 1 per dim;
 * output sites = cells with >= 1 active input in their receptive field,
 * emitted sorted ascending by ((b*D + z)*H + y)*W + x (canonical order of
 * SURVEY.md App. A.3).
 *  coords_out dev [cap_out,4]; n_out_dev dev [1]
 *  nbr_out dev [kvol*cap_out]: output row o, offset k -> input row / -1
 *  nbr_in  dev [kvol*cap_in] : input row i, offset k -> output row / -1
 */
size_t d3d_conv_rulebook_workspace_bytes(int batch, const int32_t *out_shape);
int d3d_conv_rulebook(const int32_t *coords_in, const int32_t *n_in_dev, int cap_in, int batch,
                      const int32_t *in_shape, const int32_t *ksize, const int32_t *stride,
                      const int32_t *padding, const int32_t *out_shape, int32_t *coords_out,
                      int32_t *n_out_dev, int cap_out, int32_t *nbr_out, int32_t *nbr_in,
                      int32_t *pair_count, void *workspace, size_t workspace_bytes,
                      d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Sparse convolution arithmetic.  Replaces spconv's indice_conv /
 * indice_conv_backward (gather -> GEMM -> scatter-add per offset, SURVEY.md
 * App. A.2).  Output-stationary: out[o,:] = bias + sum_k in[nbr[k][o],:] @ W[k].
 *
 *  feat   dev [n_in, cin]  dtype (D3D_F32 | D3D_BF16)
 *  wpack  dev [kvol, cout, cin] dtype  (W[k] transposed: cin contiguous)
 *  bias   dev [cout] f32 or NULL
 *  nbr    dev [kvol*cap] (see above), n_out rows are produced
 *  out    dev [n_out, cout] dtype
 * The same entry point computes the data gradient: pass grad_out as `feat`,
 * nbr_in (strided) or nbr with flipped k (submanifold) and W[k] untransposed.
 */
/* spconv's parameter layout [kvol][cin][cout] f32 (det3d checkpoints) -> the two
 * packed forms the kernels read, in the feature dtype, one launch:
 *   wpack_fwd   dev [kvol][cout][cin]  (or NULL)           for d3d_sparse_conv_fwd
 *   wpack_dgrad dev [kvol][cin][cout]  (or NULL), k -> kvol-1-k when flip_dgrad (submanifold layers) */
int d3d_pack_conv_weights(const float *weight, int kvol, int cin, int cout, int dtype, int flip_dgrad,
                          void *wpack_fwd, void *wpack_dgrad, d3d_stream_t stream);

int d3d_sparse_conv_fwd(const void *feat, const void *wpack, const float *bias,
                        const int32_t *nbr, int cap, int n_out, int kvol, int cin, int cout,
                        int dtype, void *out, d3d_stream_t stream);

/* The same launch with a per-channel reduction folded into its epilogue -- each replaces one full sweep over
 * the activations that the reference pays as a separate BatchNorm kernel (scn.py:106-157 applies
 * nn.BatchNorm1d + nn.ReLU modules to `.features` after every conv):
 *   forward: stats != NULL       -> sum / sum-of-squares of the produced rows (as stored, i.e. rounded to
 *            dtype) are written to stats[0 .. 2*cout) f32 = what d3d_bn_stats computes;
 *   dgrad  : bn_partial != NULL  -> the rows produced are dL/dy of the BatchNorm(+ReLU) whose output was this
 *            conv's input; with that layer's pre-BN activations bn_x [n_out, cout] (dtype), saved mean /
 *            invstd and gamma / beta (NULL = 1 / 0), sum dz*xhat | sum dz (dz = dy masked by the ReLU) are
 *            written to bn_partial[0 .. 2*cout) f32 = what d3d_bn_bwd_reduce computes.
 * The sums are DETERMINISTIC (ABI 4): blocks hand per-block partial vectors to a fixed-order fold, no float
 * atomics; two identical launches give bit-equal sums.  The buffer behind stats / bn_partial therefore also
 * holds the per-block slabs: `ws_floats` is its capacity in floats, d3d_conv_epilogue_ws_floats(n_out, cout)
 * the amount needed; it need not be zeroed.  Only the channel pairs d3d_sparse_conv_has_epilogue() accepts
 * (the MFMA tilings) can carry a reduction; others return D3D_EINVAL instead of silently skipping it. */
typedef struct D3dConvEpilogue {
  float *stats;
  const void *bn_x;
  const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;
  int bn_relu;
  float *bn_partial;
  int ws_floats; /* capacity of the stats / bn_partial buffer in floats */
  /* ABI 7: a [n_out, cout] matrix of the feature dtype added to the rows before they are stored (and before any of the
   * reductions above sees them) -- the accumulating data gradient of a residual block: gin = dgrad(gout) + addend,
   * where addend is the gradient that reached the block's input through the identity branch (scn.py:84-87) */
  const void *addend;
} D3dConvEpilogue;
int d3d_sparse_conv_has_epilogue(int cin, int cout, int dtype);
size_t d3d_conv_epilogue_ws_floats(int n_out, int cout);
/* tuning hook for the tools/tune_*.py, tools/bench_bn.py and tools/bench_conv2d.py sweeps (a process-wide setting, not
 * part of the product path): knob 0 = tiling variant of the sparse forward kernel (0 = default; 50-55 f32 64->64, 60-63
 * f32 128-channel layers), 1 = sparse weight gradient: value % 100000 = target block count (0 = default: one block per
 * CU, two rounds for the three-offset f32 layers), value / 100000 = kernel variant (0 default, 1 = round-3 kernel,
 * 2 = dout fragments in registers on 16 waves), 2 = XCD-aware tile order, 3 = BatchNorm reduction variant for inputs of
 * >= 2^23 elements, 4 = ignored (round 2: leave the BatchNorm global atomics out), 5 = rows in flight in bn_bwd_dx
 * (1, 2, 4), 6 = grid cap of the BatchNorm streaming sweeps (0 = default), 7 = dense conv tiling variant (-1 default),
 * 8 = dense weight-gradient kernel / waves, 9 = cut the last partial round of the dense 3x3 kernel into 32-channel
 * blocks (1 default), 10 = TEST knob: f32 sparse features on exact-f32 VALU kernels (one fmaf per product, deterministic
 * sums) instead of the split-arithmetic MFMA kernels -- d3d_sparse_conv_has_epilogue / _dgrad_plan_supported then answer
 * 0 for f32 and the packers write plain f32 copies; for "split vs exact" comparisons (det3d_amd/precision.py),
 * 11 = f32 sparse forward with the weights resident in LDS (spconv_fwd_resident_kernel): -1 = the measured default (the
 * 16 / 32-channel layers), 0 = off, 1..3 = tiling set (tools/tune_resident.py).
 * The knobs belong to the calling thread's context (below): d3d_set_tuning changes the bound context, or the process
 * default context when the thread has bound none. */
int d3d_set_tuning(int knob, int value);

/* Caller-owned library state (ABI 10).  SURVEY 8(b) asks for a re-entrant C ABI without process-global state; what the
 * library keeps between calls is (a) the tuning knobs above and (b) the zeroed ticket words of the deterministic
 * cross-block folds (csrc/d3d_fold.h), one ring of 2^16 words per device, allocated on first use.  Both live in a
 * context.  d3d_ctx_create makes one (knobs start from the D3D_* environment defaults), d3d_ctx_bind attaches it to the
 * CALLING HOST THREAD -- every entry point that thread calls afterwards reads its knobs and draws its ticket words
 * there -- and d3d_ctx_bind(NULL) returns the thread to the process default context, which is what threads that never
 * bind anything share (the behaviour of ABI <= 9).  One context per stream-owning thread makes two concurrent callers
 * independent.  d3d_ctx_destroy frees the rings; the caller guarantees that none of the context's launches is still in
 * flight and that no other thread has it bound.  d3d_ctx_get_tuning reads a knob of the calling thread's context.
 * d3d_ctx_current (ABI 11) returns the handle the calling thread has bound (NULL: the process default) -- a host framework
 * whose backward pass runs on another thread than its forward (PyTorch's autograd engine) reads it in the forward and
 * re-binds it around the backward, so that knobs which change operand formats (10) apply to both directions.
 * (The reference keeps the corresponding state -- spconv's algorithm choice, cuDNN handles -- in module globals:
 * det3d/ops/syncbn/syncbn.py:1-40, spconv's ConvAlgo.) */
typedef struct D3dContext *d3d_ctx_t;
int d3d_ctx_create(d3d_ctx_t *ctx);
int d3d_ctx_bind(d3d_ctx_t ctx);
int d3d_ctx_destroy(d3d_ctx_t ctx);
int d3d_ctx_get_tuning(int knob, int *value);
int d3d_ctx_current(d3d_ctx_t *ctx);
int d3d_sparse_conv_fwd_ex(const void *feat, const void *wpack, const float *bias,
                           const int32_t *nbr, int cap, int n_out, int kvol, int cin, int cout,
                           int dtype, void *out, const D3dConvEpilogue *epilogue, d3d_stream_t stream);

/* Pair-compacted forward / data gradient (ABI 11; det3d_amd/csrc/d3d_spconv_pairs.h).
 * spconv (the reference's dependency; det3d/models/backbones/scn.py:106-157 runs it through SubMConv3d / SparseConv3d)
 * multiplies exactly the pairs of its rulebook: per kernel offset a gather of n_k input rows, a GEMM, a scatter-add.
 * d3d_sparse_conv_fwd_ex instead walks all kvol offsets of every 16-row output tile and lets absent neighbours read
 * zeros -- on LiDAR frames half of its gathers and MFMAs.  These entry points restore the pair lists without giving up
 * the single launch and the deterministic sums: d3d_pair_table_build compacts a neighbour table (output-stationary
 * `nbr_out` for the forward and submanifold data gradients, input-stationary `nbr_in` for strided data gradients) per
 * slab of 64 consecutive rows and kernel offset; d3d_sparse_conv_fwd_pairs runs MFMA tiles over 16 existing pairs and
 * accumulates a slab's rows in LDS, one wave per slab, offsets in ascending order.  Same arguments and epilogues as
 * d3d_sparse_conv_fwd_ex with `table` in place of (nbr, cap); results agree with it to f32 rounding (an offset's
 * products are summed before they are added to the row), not bit for bit.
 *  table   dev, d3d_pair_table_bytes(n_rows, kvol) bytes: [kvol][ceil64(n_rows)] u32 entries (input row | (row % 64) << 26,
 *          the pairs of a slab packed to the front of its 64-entry segment in ascending row order), then [slabs][32] u8
 *          pair counts and [slabs][32] u8 counts of the pairs of a slab's rows 0..31 (where a segment splits into the
 *          lists of its two 32-row halves: the 16-wave tilings give a wave half a slab);
 *          built once per rulebook (one launch, no host sync), shared by every layer, direction and step using it
 *  d3d_sparse_conv_pairs_supported: f32 features (split arithmetic), channel pairs 32/64 -> 32/64; kvol <= 32 */
size_t d3d_pair_table_bytes(int n_rows, int kvol);
int d3d_pair_table_build(const int32_t *nbr, int cap, int n_rows, int kvol, void *table, size_t table_bytes,
                         d3d_stream_t stream);
int d3d_sparse_conv_pairs_supported(int cin, int cout, int dtype);
int d3d_sparse_conv_fwd_pairs(const void *feat, const void *wpack, const float *bias, const void *table, int n_out,
                              int kvol, int cin, int cout, int dtype, void *out, const D3dConvEpilogue *epilogue,
                              d3d_stream_t stream);

/* Data gradient of a STRIDED layer in parity-class row order (round 3; det3d_amd/csrc/d3d_rowplan.h).
 * The reference's backward of spconv.SparseConv3d (scn.py:113, :127, :141, :150: stride 2, and the last (3,1,1) /
 * (2,1,1) layer) is indice_conv_backward over the same pairs; here it is the gather kernel over the input-stationary
 * table nbr_in, in which an input row only has the offsets k = (i + p) mod s per axis.  d3d_dgrad_plan_build sorts the
 * input rows by that class (stable; <= 8 classes) and lists each class's offsets, d3d_sparse_conv_dgrad_planned walks
 * 1..8 offsets per block instead of 27.  Same sums in the same order per row: bit-identical to
 * d3d_sparse_conv_fwd_ex(dout, wpack, NULL, nbr_in, ...).
 *  plan dev int32 [d3d_dgrad_plan_ints(cap_in)], written by the build (three launches on `stream`, no host sync);
 *  coords_in dev [n,4] int32 (batch, z, y, x), n_in_dev dev int32 [1];
 *  d3d_dgrad_plan_supported: 2 <= stride product <= 8 and kernel volume <= 27;
 *  d3d_sparse_conv_dgrad_plan_supported: channel pairs (gradient channels cin -> input channels cout) with a
 *  planned tiling; the epilogue argument is as in d3d_sparse_conv_fwd_ex. */
size_t d3d_dgrad_plan_ints(int cap_in);
int d3d_dgrad_plan_supported(const int32_t *ksize, const int32_t *stride);
int d3d_dgrad_plan_build(const int32_t *coords_in, const int32_t *n_in_dev, int cap_in, const int32_t *ksize,
                         const int32_t *stride, const int32_t *padding, int32_t *plan, size_t plan_ints,
                         d3d_stream_t stream);
/* The rulebooks of a whole sparse backbone as ONE chain of fat launches (ABI 9; csrc/d3d_rbchain.hip).  Same tables, bit
 * for bit, as d3d_conv_rulebook + d3d_dgrad_plan_build per layer and d3d_subm_rulebook per level -- what
 * spconv.SparseSequential-style stacks (det3d/models/backbones/scn.py:106-157, :323-355) ask get_indice_pairs for layer
 * by layer -- in 1 + 4 launches per strided layer (10 per layer there) and 1 launch per submanifold level (3 there):
 *   d3d_conv_rulebook_chain   layers[l] describes strided layer l (its input grid = layer l-1's output grid; layer 0 reads
 *       coords0 / *n0_dev / cap0): geometry as in d3d_conv_rulebook, caller-owned outputs coords_out / n_out_dev / nbr_out /
 *       nbr_in of capacity cap_out rows (cap_in of layer l = cap_out of layer l-1), optional pair_count, optional row plan
 *       of the data gradient (plan_ints >= d3d_dgrad_plan_ints(cap_in); NULL = none).  On return rank_bitmap / rank_prefix
 *       point INTO `workspace`: the output grid's occupancy bitmap and per-word popcount prefix -- the rank structure
 *       of the level the layer produced: row(cell) = rank_prefix[cell / 32] + popcount(rank_bitmap[cell / 32] & bits below cell).
 *       They stay valid as long as the caller keeps `workspace` (device-side counts throughout: no host sync).
 *   d3d_subm_rulebook_ranked  the submanifold table of a level that came out of a chain layer, by rank lookups into that
 *       layer's structure (rows of such a level are in ascending cell order, so rank == row): no table build, no prefill. */
typedef struct D3dChainLayer {
  int32_t in_shape[3], ksize[3], stride[3], padding[3], out_shape[3]; /* host, z y x */
  int32_t cap_out;
  int32_t *coords_out;       /* dev [cap_out, 4] */
  int32_t *n_out_dev;        /* dev [1] */
  int32_t *nbr_out;          /* dev [kvol * cap_out] */
  int32_t *nbr_in;           /* dev [kvol * cap_in]  */
  int32_t *pair_count;       /* dev [kvol] or NULL */
  int32_t *plan;             /* dev [plan_ints] or NULL */
  uint64_t plan_ints;
  const uint32_t *rank_bitmap; /* out: dev pointers into the workspace */
  const int32_t *rank_prefix;
} D3dChainLayer;
size_t d3d_conv_rulebook_chain_workspace_bytes(int n_layers, int batch, const D3dChainLayer *layers);
int d3d_conv_rulebook_chain(int n_layers, D3dChainLayer *layers, int batch, const int32_t *coords0, const int32_t *n0_dev,
                            int cap0, void *workspace, size_t workspace_bytes, d3d_stream_t stream);
int d3d_subm_rulebook_ranked(const int32_t *coords, const int32_t *n_dev, int cap, int batch, const int32_t *spatial_shape,
                             const int32_t *ksize, const uint32_t *rank_bitmap, const int32_t *rank_prefix, int32_t *nbr,
                             int32_t *pair_count, d3d_stream_t stream);
int d3d_sparse_conv_dgrad_plan_supported(int cin, int cout, int dtype);
int d3d_sparse_conv_dgrad_planned(const void *dout, const void *wpack, const int32_t *nbr_in, int cap_in, int n_in,
                                  int kvol, int cin, int cout, int dtype, void *din,
                                  const D3dConvEpilogue *epilogue, const int32_t *plan, d3d_stream_t stream);

/* weight gradient: dw[k][ci][co] = sum_o feat[nbr[k][o]][ci] * dout[o][co]
 *  dw dev [kvol, cin, cout] f32, overwritten.
 *  workspace: device scratch of d3d_sparse_conv_wgrad_workspace_bytes(...) bytes holding the per-split
 *  partial sums of the two-pass (atomic-free, bit-reproducible) bf16 path; f32 and channel counts outside
 *  {16,32,64,128} take the one-pass kernels and ignore it. */
size_t d3d_sparse_conv_wgrad_workspace_bytes(int n_out, int kvol, int cin, int cout, int dtype);
int d3d_sparse_conv_wgrad(const void *feat, const void *dout, const int32_t *nbr, int cap,
                          int n_out, int kvol, int cin, int cout, int dtype, float *dw,
                          void *workspace, size_t workspace_bytes, d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Optimizer step of the training hot path (round 3): global-norm gradient clipping + true weight decay + Adam over
 * all parameter tensors in two launches.  Replaces, for f32 parameters on the device, OptimWrapper.step of
 * det3d/solver/fastai_optim.py (p.data.mul_(1 - wd * lr), then torch.optim.Adam.step with the one-cycle momentum as
 * beta1) and clip_grad_norm_ of det3d/torchie/trainer/hooks/optimizer.py:
 *    g  = grad * grad_scale * min(1, max_norm / (||grad * grad_scale||_2 + 1e-6))        (max_norm <= 0: no clipping)
 *    m  = m + (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2
 *    p  = p * decay - lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 *  tensors dev int64 [n_tensors, 4]: address of the parameter, of exp_avg, of exp_avg_sq (f32, numel elements each),
 *          numel -- static, uploaded once by the caller;
 *  chunks dev int32 [n_chunks, 2]: (tensor, first element) of every chunk of chunk_elems (>= 256) elements, tensor by
 *          tensor; tensor_chunk_start HOST int32 [n_tensors + 1]: first chunk of every tensor;
 *  grads_host HOST uint64 [n_tensors]: device addresses of this step's gradient tensors (they travel as kernel
 *          arguments, D3D_ADAM_MAX_TENSORS per launch);
 *  partial dev f32 [n_chunks] scratch; grad_norm dev f32 [1]: the norm before clipping.
 * Deterministic (fixed-order sums), no host synchronisation. */
#define D3D_ADAM_MAX_TENSORS 448
typedef struct D3dAdamHyper {
  float lr, beta1, beta2, eps;
  float decay;      /* the parameter is multiplied by this before the update: 1 - wd * lr (true weight decay) */
  float max_norm;   /* global-norm clipping threshold, <= 0 disables */
  float grad_scale; /* gradients are multiplied by this first (1 / world size for summed gradients) */
  int step;         /* 1-based step count of the bias corrections */
} D3dAdamHyper;
int d3d_adam_step(const int64_t *tensors, int n_tensors, const int32_t *chunks, int n_chunks, int chunk_elems,
                  const int32_t *tensor_chunk_start, const uint64_t *grads_host, const D3dAdamHyper *hyper,
                  float *partial, float *grad_norm, d3d_stream_t stream);

/* Every packed conv operand rewritten in ONE launch behind the optimizer step (ABI 7; csrc/d3d_optim.hip).  The conv
 * layers read their weights in packed forms -- d3d_pack_conv_weights (sparse) / d3d_conv2d_pack_weights (dense) -- and
 * an optimizer step invalidates all of them: d3d_pack_all takes one descriptor per weight tensor (HOST array) and does
 * what n calls of those two functions would, same bytes.
 *  kind 0 (sparse): weight dev f32 [a = kvol][b = cin][c = cout]; fwd / dgrad as d3d_pack_conv_weights' wpack_fwd /
 *                   wpack_dgrad (either may be NULL), flip = its flip_dgrad, dtype = the feature dtype;
 *  kind 1 (dense):  weight dev f32 [a = cout][b = cin][c = k*k]; fwd, fwd_lo, dgrad, dgrad_lo as d3d_conv2d_pack_weights'
 *                   fwd_hi / fwd_lo / dgrad_hi / dgrad_lo (NULL = not wanted). */
#define D3D_PACK_MAX 40
typedef struct D3dPackDesc {
  const float *weight;
  void *fwd, *fwd_lo, *dgrad, *dgrad_lo;
  int kind, a, b, c;
  int flip, dtype;
} D3dPackDesc;
int d3d_pack_all(const D3dPackDesc *descs, int n, d3d_stream_t stream);

/* A run of (sparse conv -> BatchNorm1d -> ReLU) layers as ONE call per direction (ABI 12; csrc/d3d_stack.hip).  Replaces
 * the launch sequence of spconv.SparseSequential over det3d/models/backbones/scn.py:106-157 (SubMConv3d / SparseConv3d +
 * build_norm_layer + nn.ReLU triples; training-mode BatchNorm, no convolution bias) that det3d_amd/sparse_stack.py issues
 * call by call: the entry points named below, in the same order, on the same streams -- bit-identical results, one
 * foreign call instead of three to five per layer and direction.  `layers` is a HOST array, first layer first; every
 * buffer is the caller's (sizes as for the per-layer entry points).
 *  forward, per layer:  d3d_sparse_conv_fwd_ex(x, wpack, nbr_out .. -> out, epilogue.stats = stats) then
 *                       d3d_bn_apply(out -> y, training = 1, stats_ws = stats, save_mean / save_invstd written).
 *  backward, last layer first; dy of the last layer = dy_last, of layer i = layers[i + 1].gin:
 *    d3d_bn_bwd_reduce(out, dy -> partial) unless layers[i].partial_by_epilogue and layer i + 1 has a data gradient;
 *    d3d_bn_bwd_dx(out, dy, partial -> gout, gbn[0 .. cout) = dgamma, gbn[cout .. 2 cout) = dbeta);
 *    if gin: the data gradient gout -> gin (d3d_sparse_conv_dgrad_planned when the layer is strided, carries a plan and
 *      the channel pair supports it, else d3d_sparse_conv_fwd_ex over nbr_out (submanifold) / nbr_in with wd), with the
 *      BatchNorm-backward sums of layer i - 1 reduced in its epilogue when layers[i - 1].partial_by_epilogue;
 *    d3d_sparse_conv_wgrad(x, gout -> gw) on `side` behind an event recorded on `main` after d3d_bn_bwd_dx
 *      (side == NULL: on `main`);
 *    at the end `main` waits for `side`. */
typedef struct D3dStackLayer {
  const int32_t *nbr_out; /* output-stationary table, kvol * cap_out */
  const int32_t *nbr_in;  /* input-stationary twin of a strided layer, kvol * cap_in (NULL for submanifold) */
  const int32_t *dgrad_plan; /* row plan of the strided data gradient or NULL */
  int cap_out, cap_in, n_in, n_out, kvol, subm;
  int cin, cout; /* channels as the kernels see them (a stem's input channels already padded) */
  int relu;
  int partial_by_epilogue; /* the data gradient of layer i + 1 reduces this layer's BatchNorm-backward sums */
  float momentum, eps;
  const void *x; /* [n_in, cin] input rows */
  void *out;     /* [n_out, cout] conv output (kept for the backward) */
  void *y;       /* [n_out, cout] after BatchNorm (+ ReLU) */
  const void *wpack; /* d3d_pack_conv_weights' wpack_fwd */
  const void *wd;    /* its wpack_dgrad; NULL when gin is NULL */
  float *stats; /* d3d_conv_epilogue_ws_floats(n_out, cout) floats */
  int stats_floats, partial_floats;
  const float *gamma, *beta;
  float *running_mean, *running_var, *save_mean, *save_invstd;
  void *gout;     /* backward: [n_out, cout] gradient at the conv output */
  void *gin;      /* backward: [n_in, cin] gradient at the layer input, or NULL (first layer without input gradient) */
  float *partial; /* backward: fold workspace of the BatchNorm-backward sums (d3d_bn_reduce_ws_floats / epilogue size) */
  float *gbn;     /* backward: [2, cout] dgamma | dbeta */
  float *gw;      /* backward: [kvol, cin, cout] weight gradient */
  void *wgrad_ws; /* backward: d3d_sparse_conv_wgrad_workspace_bytes(n_out, kvol, cin, cout, dtype) bytes */
  size_t wgrad_ws_bytes;
} D3dStackLayer;
int d3d_sparse_stack_forward(const D3dStackLayer *layers, int n, int dtype, d3d_stream_t stream);
int d3d_sparse_stack_backward(const D3dStackLayer *layers, int n, int dtype, const void *dy_last, d3d_stream_t main,
                              d3d_stream_t side);

/* ------------------------------------------------------------------------
 * BatchNorm1d (+ReLU) over the active sites: replaces the nn.BatchNorm1d / nn.ReLU module
 * pairs that spconv.SparseSequential applies to `.features` after every sparse conv
 * (det3d/models/backbones/scn.py:106-157; residual blocks :76-87), F.batch_norm semantics
 * (biased variance to normalise, unbiased into running_var, eps, momentum).
 *  x, y, dy, dx dev [n,c] dtype; gamma/beta/running_* dev [c] f32 (gamma/beta may be NULL)
 *  training=1: batch statistics (stats_ws dev f32 scratch of d3d_bn_reduce_ws_floats(n, c, dtype) floats:
 *              [2c] sums | per-block slabs of the deterministic fold; need not be zeroed),
 *              running stats updated in place, save_mean/save_invstd dev [c] written for the backward
 *  training=0: running statistics, stats_ws unused
 *  training=2 (d3d_bn_apply and its _residual / _split forms): stats_ws holds the FINISHED per-channel mean | variance
 *              ([2c] f32; e.g. moments all-reduced over ranks by DistributedSyncBN, det3d/ops/syncbn/syncbn.py:39-52):
 *              the sweep normalises with them, writes save_mean / save_invstd and leaves the running statistics to the caller
 *  relu=1 fuses max(0, .) into the apply pass (and its mask into the backward).
 *  bwd overwrites dgamma/dbeta dev [c] f32 and dx; partial_ws: scratch of d3d_bn_reduce_ws_floats floats.
 *  Works on any row-major [n,c] matrix, e.g. the [B*H*W, C] view of a channels-last feature map
 *  (the RPN's BatchNorm2d + ReLU pairs, det3d/models/necks/rpn.py:124-142).
 */
#define D3D_BN_COPIES 1 /* the reductions fold themselves: consumers read ONE [2c] vector */
size_t d3d_bn_reduce_ws_floats(int n, int c, int dtype);
int d3d_bn_relu_fwd(const void *x, int n, int c, int dtype, const float *gamma, const float *beta,
                    float *running_mean, float *running_var, float momentum, float eps,
                    int training, int relu, void *y, float *save_mean, float *save_invstd,
                    float *stats_ws, d3d_stream_t stream);
int d3d_bn_relu_bwd(const void *x, const void *dy, int n, int c, int dtype, const float *gamma,
                    const float *beta, const float *save_mean, const float *save_invstd, int relu,
                    void *dx, float *dgamma, float *dbeta, float *partial_ws, d3d_stream_t stream);
/* The halves of the two calls above as separate entry points (d3d_bn_relu_fwd = d3d_bn_stats + d3d_bn_apply,
 * d3d_bn_relu_bwd = d3d_bn_bwd_reduce + d3d_bn_bwd_dx; same buffers).  The reductions can then come from a
 * convolution epilogue (d3d_sparse_conv_fwd_ex) instead of their own sweep.  Both reductions leave their [2c]
 * result at the start of their workspace; the apply / dx halves read `copies` consecutive [2c] vectors and add
 * them in order (copies = 1 for the producers of this library; > 1 lets a caller hand in several partial
 * results, e.g. per-rank sums). */
int d3d_bn_stats(const void *x, int n, int c, int dtype, float *stats_ws, d3d_stream_t stream);
int d3d_bn_apply(const void *x, int n, int c, int dtype, const float *gamma, const float *beta,
                 float *running_mean, float *running_var, float momentum, float eps, int training, int relu,
                 const float *stats_ws, int copies, void *y, float *save_mean, float *save_invstd,
                 d3d_stream_t stream);
int d3d_bn_bwd_reduce(const void *x, const void *dy, int n, int c, int dtype, const float *gamma,
                      const float *beta, const float *save_mean, const float *save_invstd, int relu,
                      float *partial_ws, d3d_stream_t stream);
int d3d_bn_bwd_dx(const void *x, const void *dy, int n, int c, int dtype, const float *gamma, const float *beta,
                  const float *save_mean, const float *save_invstd, int relu, const float *partial_ws, int copies,
                  void *dx, float *dgamma, float *dbeta, d3d_stream_t stream);
/* Residual forms (ABI 7; SparseBasicBlock, det3d/models/backbones/scn.py:46-89: out = relu(bn2(conv2(.)) + identity)):
 *  y = relu?(bn(x) + residual) in one sweep; in the backward the ReLU mask is [bn(x) + residual > 0], and
 *  d3d_bn_bwd_dx_residual also writes dresidual = dy * mask (the gradient of the identity branch).  residual /
 *  dresidual: dev [n, c] of `dtype`; everything else as in d3d_bn_apply / d3d_bn_bwd_reduce / d3d_bn_bwd_dx. */
int d3d_bn_apply_residual(const void *x, const void *residual, int n, int c, int dtype, const float *gamma,
                          const float *beta, float *running_mean, float *running_var, float momentum, float eps,
                          int training, int relu, const float *stats_ws, int copies, void *y, float *save_mean,
                          float *save_invstd, d3d_stream_t stream);
int d3d_bn_bwd_reduce_residual(const void *x, const void *residual, const void *dy, int n, int c, int dtype,
                               const float *gamma, const float *beta, const float *save_mean,
                               const float *save_invstd, int relu, float *partial_ws, d3d_stream_t stream);
int d3d_bn_bwd_dx_residual(const void *x, const void *residual, const void *dy, int n, int c, int dtype,
                           const float *gamma, const float *beta, const float *save_mean, const float *save_invstd,
                           int relu, const float *partial_ws, int copies, void *dx, void *dresidual, float *dgamma,
                           float *dbeta, d3d_stream_t stream);
/* The same two sweeps on f32 rows with the result in the split format of the dense conv kernels (two bf16 planes
 * hi = bf16(v), lo = bf16(v - hi); d3d_conv2d_nhwc_fwd, D3D_CONV_SPLIT): the BatchNorm2d + ReLU between two RPN
 * convolutions writes the next convolution's operand directly (rpn.py:124-142). */
int d3d_bn_apply_split(const float *x, int n, int c, const float *gamma, const float *beta, float *running_mean,
                       float *running_var, float momentum, float eps, int training, int relu, const float *stats_ws,
                       int copies, void *y_hi, void *y_lo, float *save_mean, float *save_invstd, d3d_stream_t stream);
int d3d_bn_bwd_dx_split(const float *x, const float *dy, int n, int c, const float *gamma, const float *beta,
                        const float *save_mean, const float *save_invstd, int relu, const float *partial_ws,
                        int copies, void *dx_hi, void *dx_lo, float *dgamma, float *dbeta, d3d_stream_t stream);
/* Per-channel affine sweeps for the reference's cross-rank BatchNorm (det3d/ops/syncbn/syncbn.py:15-118,
 * src/syncbn_cuda_kernel.cu:166-198 transform_input, :595-640 input_backward): the all-reduced statistics
 * arrive as [c] f32 vectors, so the passes take ready coefficients:
 *   d3d_channel_affine      y  = relu ? max(0, x*scale + shift) : x*scale + shift
 *   d3d_channel_affine_bwd  dx = coef_dz*dz + coef_x*x + coef_1,  dz = dy masked by [x*scale + shift > 0] if relu */
int d3d_channel_affine(const void *x, int n, int c, int dtype, const float *scale, const float *shift, int relu,
                       void *y, d3d_stream_t stream);
int d3d_channel_affine_bwd(const void *x, const void *dy, int n, int c, int dtype, const float *scale,
                           const float *shift, int relu, const float *coef_dz, const float *coef_x,
                           const float *coef_1, void *dx, d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Dense 2D convolution on channels-last maps: the RPN's conv layers (det3d/models/necks/rpn.py:70-159, nn.Conv2d
 * 3x3 stride 1/2 and the 1x1 / 2x2 deblock convs, bias-free; the reference runs them on cuDNN).  Implicit GEMM on
 * MFMA, both operands through LDS-DMA, f32 accumulation.
 *   x      dev [batch, h, w, cin]    bf16 (D3D_CONV_BF16) or two bf16 planes hi / lo (D3D_CONV_SPLIT: v = hi + lo,
 *          hi = bf16(v), lo = bf16(v - hi) -- 16 mantissa bits; x_lo NULL in bf16 mode)
 *   w      dev [ksz*ksz, cout, cin]  packed weights, W[tap][co][ci] = weight[co][ci][ky][kx] of nn.Conv2d, same planes
 *   y      dev [batch, ho, wo, cout] bf16 (BF16 mode) or f32 (SPLIT mode); ho = (h + 2 pad - ksz) / stride + 1
 *   stats_ws (optional): the per-channel sum | sum of squares of the produced rows (as stored) are left in
 *          stats_ws[0 .. 2*cout) -- what d3d_bn_stats computes on y -- deterministically (fixed-order fold);
 *          capacity d3d_conv2d_stats_ws_floats(batch, ho, wo, cout) floats, need not be zeroed.
 *   cin in {64, 128, 256}, cout % 128 == 0, ksz 1..3, stride 1..2 (d3d_conv2d_supported).
 * SPLIT mode multiplies hi*hi + hi*lo + lo*hi on the bf16 MFMA: relative error ~2^-16 of the products (bf16:
 * 2^-8) at a fifth of the exact-f32 MFMA's time -- the precision mode that meets the 1e-3 box tolerance.
 * The data gradient of a stride-1 layer is the same call on dL/dy with the taps mirrored and the weights packed
 * as [tap'][cin][cout] (pad' = ksz - 1 - pad).
 * d3d_conv2d_nhwc_wgrad: dw [cout, cin, ksz, ksz] f32 (nn.Conv2d's layout) = sum over output pixels of
 * g (x) x(tap), g = dL/dy [batch, ho, wo, cout] and x in the same element format; two-pass, bit-reproducible;
 * cin, cout in {128, 256}.
 * d3d_conv2d_pack_weights: nn.Conv2d weight [cout, cin, ksz, ksz] f32 -> the forward operand [tap][cout][cin] and
 * the data-gradient operand [mirrored tap][cin][cout], each as a bf16 hi plane and (optional, split mode) lo plane.
 * d3d_split_planes: f32 [n] -> the hi / lo planes of the split format. */
#define D3D_CONV_BF16 0
#define D3D_CONV_SPLIT 1
/* With `bn` set, d3d_conv2d_nhwc_fwd is a DATA-GRADIENT launch whose produced rows are dL/dz of the BatchNorm(+ReLU)
 * that made the convolution's input (rpn.py:124-142): instead of the statistics of the produced rows, the epilogue
 * leaves sum dz * xhat | sum dz in stats_ws[0 .. 2*cout) -- what d3d_bn_bwd_reduce computes -- with x = that layer's
 * pre-BN activations [batch, ho, wo, cout] in this launch's OUTPUT element type, its saved mean / invstd and its
 * gamma / beta (NULL = 1 / 0), dz = dy masked by the ReLU when relu != 0.
 * Cost: the 3x3 / stride-1 D3D_CONV_SPLIT layers (conv2d_halo_kernel) stage the block's tile of x through LDS in whole pixel
 * rows and replace the separate d3d_bn_bwd_reduce sweep at a gain (det3d_amd/dense_stack.py does this for every such layer but
 * the last of a run); the other layer shapes read x at pixel stride and are faster with the sweep. */
typedef struct D3dDenseBnBackward {
  const void *x;
  const float *mean, *invstd, *gamma, *beta;
  int relu;
} D3dDenseBnBackward;
int d3d_conv2d_supported(int cin, int cout, int ksz, int stride);
/* ABI 7: the 3x3 / stride-1 layers in D3D_CONV_SPLIT mode also exist with 64 output channels (the first block of the
 * PointPillars necks, det3d/models/necks/rpn.py:80-110 with ds_num_filters[0] = 64): forward and data gradient through
 * d3d_conv2d_nhwc_fwd; their weight gradient is not covered by d3d_conv2d_nhwc_wgrad (multiples of 128 only). */
int d3d_conv2d_supported_split3x3(int cin, int cout);
size_t d3d_conv2d_stats_ws_floats(int batch, int ho, int wo, int cout);
int d3d_conv2d_nhwc_fwd(const void *x_hi, const void *x_lo, const void *w_hi, const void *w_lo, void *y, int batch,
                        int h, int w, int cin, int cout, int ksz, int stride, int pad, int mode, float *stats_ws,
                        size_t stats_ws_floats, const D3dDenseBnBackward *bn, d3d_stream_t stream);
/* ABI 8: the data gradient of a STRIDE-2 3x3 layer (the first convolution of an RPN block, det3d/models/necks/rpn.py:64-68
 * ZeroPad2d(1) + Conv2d(3, stride 2); the reference gets it from cuDNN through autograd).  g = dL/dy [batch, ho, wo, cout]
 * in the operand format of `mode`, w_hi / w_lo = the layer's dgrad planes of d3d_conv2d_pack_weights, dx [batch, h, w, cin]
 * f32 (split) or bf16.  cout 128 | 256, cin a multiple of 128, ksz 3, pad 0..2.  (Stride-1 layers: d3d_conv2d_nhwc_fwd on
 * g with the dgrad planes and pad' = ksz - 1 - pad.) */
int d3d_conv2d_nhwc_dgrad_s2(const void *g_hi, const void *g_lo, const void *w_hi, const void *w_lo, void *dx, int batch,
                             int h, int w, int cin, int cout, int ksz, int pad, int mode, d3d_stream_t stream);
int d3d_conv2d_pack_weights(const float *weight, int cout, int cin, int ksz, void *fwd_hi, void *fwd_lo,
                            void *dgrad_hi, void *dgrad_lo, d3d_stream_t stream);
int d3d_split_planes(const float *x, int64_t n, void *hi, void *lo, d3d_stream_t stream);
size_t d3d_conv2d_wgrad_workspace_bytes(int batch, int ho, int wo, int cin, int cout, int ksz);
int d3d_conv2d_nhwc_wgrad(const void *g_hi, const void *g_lo, const void *x_hi, const void *x_lo, float *dw,
                          int batch, int h, int w, int cin, int cout, int ksz, int stride, int pad, int mode,
                          void *workspace, size_t workspace_bytes, d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * The prediction convolutions of one MultiGroupHead task group: conv_box / conv_cls / conv_dir, 1x1 Conv2d with bias on
 * the same channels-last map (det3d/models/bbox_heads/mg_head.py:193-232 `Head`, :553-560), as ONE launch per
 * direction; exact f32 FMA arithmetic for f32 or bf16 maps; the weight / bias gradients are summed in a fixed order.
 *   x        dev [m, cin] rows of the map (m = batch*H*W), dtype D3D_F32 | D3D_BF16; cin = 128
 *   weights  HOST array of `count` (1..3) dev pointers, weight k = [n[k], cin] f32 (nn.Conv2d's [n, cin, 1, 1])
 *   biases   HOST array of dev pointers [n[k]] f32 (entries or the array may be NULL)
 *   y / g    HOST array of dev pointers [m, n[k]] in x's dtype: predictions (fwd) or their gradients
 *   dx       dev [m, cin]: the SUM of the data gradients of the `count` convolutions
 *   wgrad    ws[0 .. 32*cin) = dW in concatenated channel order ([sum n][cin], zero rows up to 32),
 *            ws[32*cin .. 32*cin + 32) = db; ws needs d3d_head_conv_wgrad_ws_floats(m) floats (fold workspace)
 *   sum n <= 32 (d3d_head_conv_supported). */
int d3d_head_conv_supported(int cin, int ctot);
int d3d_head_conv_fwd(const void *x, int64_t m, int cin, int dtype, const float *const *weights,
                      const float *const *biases, void *const *y, const int *n, int count, d3d_stream_t stream);
int d3d_head_conv_dgrad(void *const *g, const int *n, int count, int64_t m, int cin, int dtype,
                        const float *const *weights, void *dx, d3d_stream_t stream);
size_t d3d_head_conv_wgrad_ws_floats(int64_t m);
int d3d_head_conv_wgrad(const void *x, void *const *g, const int *n, int count, int64_t m, int cin, int dtype,
                        float *ws, size_t ws_floats, d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * PillarFeatureNet with one (last) PFNLayer: det3d/models/readers/pillar_encoder.py:115-155
 * (decorate + mask) and :42-57 (Linear(c+5 -> f, no bias), BatchNorm1d over all m*t rows, ReLU,
 * max over t), fused; nothing of size [m, t, *] is materialised.
 *  voxels dev [m,t,c] f32 zero padded, num_points dev [m] i32, coors dev [m,4] i32 (b,z,y,x)
 *  weight dev [f, c+5] f32 (nn.Linear layout), gamma/beta/running_* dev [f] f32, f <= 64, c in 3..5
 *  out dev [m,f] f32; argmax dev [m,f] i32 (row index of the max, >= num_points for a padded row)
 *  training=1: batch statistics via stats_ws, running stats updated, save_* written
 *  bwd overwrites dweight [f,c+5], dgamma [f], dbeta [f]  (points carry no gradient)
 *  stats_ws / workspace: dev f32 scratch of d3d_pfn_ws_floats(m, c, f) floats, need not be zeroed.  Every wave of a
 *  reducing launch leaves one slab of partial sums there and a second launch adds the slabs in a fixed order: the
 *  sums are bit-reproducible (ABI 6; before: float atomics).
 */
size_t d3d_pfn_ws_floats(int m, int c, int f);
int d3d_pfn_fwd(const float *voxels, const int32_t *num_points, const int32_t *coors, int m, int t,
                int c, float vx, float vy, float x_offset, float y_offset, const float *weight, int f,
                const float *gamma, const float *beta, float *running_mean, float *running_var,
                float momentum, float eps, int training, float *out, int32_t *argmax,
                float *save_mean, float *save_invstd, float *stats_ws, size_t stats_ws_floats,
                d3d_stream_t stream);
int d3d_pfn_bwd(const float *voxels, const int32_t *num_points, const int32_t *coors, int m, int t,
                int c, float vx, float vy, float x_offset, float y_offset, const float *weight, int f,
                const float *gamma, const float *save_mean, const float *save_invstd,
                const float *out, const float *dout, const int32_t *argmax, float *dweight,
                float *dgamma, float *dbeta, float *workspace, size_t workspace_floats, d3d_stream_t stream);

/* SparseConvTensor.dense() (spconv; call sites scn.py:192-195,363-367):
 * scatter feat[n,c] to dense [batch, c, D, H, W] (zero elsewhere; the caller
 * views it as [batch, c*D, H, W]).  `dense` is fully written.  The backward
 * is d3d_dense_gather.
 * channels_last = 1 lays `dense` out as [batch, H, W, c*D] (channel = ch*D + z), i.e. the
 * memory of a torch channels-last [batch, c*D, H, W] tensor. */
int d3d_dense_scatter(const void *feat, const int32_t *coords, int n, int c, int batch,
                      const int32_t *spatial_shape, int dtype, int channels_last, void *dense,
                      d3d_stream_t stream);
int d3d_dense_gather(const void *dense, const int32_t *coords, int n, int c, int batch,
                     const int32_t *spatial_shape, int dtype, int channels_last, void *feat,
                     d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Anchor target assignment (the first "next" row): create_target_np
 * (det3d/core/anchor/target_ops.py:28-222) with the nearest-IoU similarity
 * (region_similarity.py:73-93, box_np_ops.py:343-355,955-994) and second_box_encode, for one
 * task with 7-dim boxes, no positive/negative subsampling (sample_positive_fraction = -1).
 *  anchors dev [A,7] f32; gt dev [batch,gmax,7] f32 padded; gt_count dev [batch] i32;
 *  gt_classes dev [batch,gmax] i32 (1-based) or NULL (all class 1)
 *  labels dev [batch,A] i32 (class / 0 background / -1 ignored), reg_targets dev [batch,A,7] f32,
 *  reg_weights dev [batch,A] f32
 */
size_t d3d_assign_targets_workspace_bytes(int num_anchors, int batch, int gmax);
int d3d_assign_targets(const float *anchors, int num_anchors, const float *gt,
                       const int32_t *gt_classes, const int32_t *gt_count, int batch, int gmax,
                       float matched_threshold, float unmatched_threshold, int32_t *labels,
                       float *reg_targets, float *reg_weights, void *workspace,
                       size_t workspace_bytes, d3d_stream_t stream);

/* One task group of a multi-group head, as TargetAssigner.assign_v2 does it
 * (det3d/core/anchor/target_assigner.py:68-142): every class of the task has its own anchors and
 * thresholds and is matched only against the gt boxes of that class.
 *  anchors dev [A, box_dim] f32 in the concatenated layout of generate_anchors (:144-166): anchor a
 *    belongs to class (a / anchors_per_class) % num_classes;  box_dim 7 or 9 (nuScenes: + vx, vy)
 *  matched_thresholds / unmatched_thresholds HOST [num_classes] f32 (num_classes <= 8)
 *  encode_angle_vector: box coder option (ground_box3d_coder); code = box_dim - 1 + (vector ? 2 : 1)
 *  gt dev [batch,gmax,box_dim] f32 padded (NaN entries are read as 0 like np.nan_to_num, :96),
 *  gt_classes dev [batch,gmax] i32, 1-based WITHIN the task; gt_count dev [batch] i32
 *  labels dev [batch,A] i32, reg_targets dev [batch,A,code] f32, reg_weights dev [batch,A] f32
 *  workspace: d3d_assign_targets_workspace_bytes(A, batch, gmax) */
int d3d_assign_targets_task(const float *anchors, int num_anchors, int box_dim, int num_classes,
                            int anchors_per_class, const float *matched_thresholds,
                            const float *unmatched_thresholds, int encode_angle_vector, const float *gt,
                            const int32_t *gt_classes, const int32_t *gt_count, int batch, int gmax,
                            int32_t *labels, float *reg_targets, float *reg_weights, void *workspace,
                            size_t workspace_bytes, d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Loss of one MultiGroupHead task group, forward and gradient w.r.t. the predictions in two launches.
 * Replaces MultiGroupHead.loss (det3d/models/bbox_heads/mg_head.py:580-695: prepare_loss_weights :147-188
 * with NormByNumPositives, create_loss :535-578, add_sin_difference :29-34, get_direction_target :54-63,
 * _get_pos_neg_loss :36-52) with SigmoidFocalLoss, WeightedSmoothL1Loss (codewise; code_weights ignored as in
 * losses.py:173) and WeightedSoftmaxClassificationLoss (losses.py:146-446), encode_background_as_zeros.
 *  box_preds [B,A,code], cls_preds [B,A,nc], dir_preds [B,A,2] or NULL: dev, dtype (D3D_F32 | D3D_BF16)
 *  labels dev [B,A] i32 (class / 0 / -1), reg_targets dev [B,A,code] f32
 *  anchors dev [B,A,box_dim] f32 (anchors_per_batch = 1) or [A,box_dim] (0); only read with dir_preds
 *  grad_box / grad_cls / grad_dir: dev, same shape and dtype as the predictions: d(loss)/d(prediction)
 *  scalars dev [D3D_LOSS_SCALARS] f32: [0] loss = loc + cls + dir_weight * dir   [1] loc_loss_reduced
 *    [2] cls_loss_reduced  [3] dir_loss (unweighted)  [4] cls_pos_loss  [5] cls_neg_loss
 *    [6] num_pos of frame 0  [7] num_neg of frame 0  [8 .. 8+code) loc_loss_elem
 *  code <= 10, nc <= 8.  workspace: d3d_head_loss_workspace_bytes(batch). */
#define D3D_LOSS_SCALARS 18
size_t d3d_head_loss_workspace_bytes(int batch);
int d3d_head_loss(const void *box_preds, const void *cls_preds, const void *dir_preds, int dtype,
                  const int32_t *labels, const float *reg_targets, const float *anchors,
                  int anchors_per_batch, int batch, int num_anchors, int code_size, int num_class,
                  int box_dim, int sin_difference, float dir_offset, float pos_cls_weight,
                  float neg_cls_weight, float focal_alpha, float focal_gamma, float smooth_l1_sigma,
                  float loc_weight, float cls_weight, float dir_weight, void *grad_box, void *grad_cls,
                  void *grad_dir, float *scalars, void *workspace, size_t workspace_bytes,
                  d3d_stream_t stream);

/* ------------------------------------------------------------------------
 * Rotated NMS.  Replaces rotate_nms_cc (det3d/ops/nms/nms_cpu.py:34-45) +
 * rotate_non_max_suppression_cpu (det3d/ops/nms/nms_cpu.h:73-169), as driven
 * by box_torch_ops.rotate_nms (det3d/core/bbox/box_torch_ops.py:528-549).
 *  boxes dev [n,5] f32 (x, y, w, l, r) ALREADY sorted by descending score
 *  keep  dev [max_keep] i32 indices into boxes, ascending = score order
 *  n_keep dev [1] i32
 * Semantics: skip pair if standup-AABB IoU <= 0; suppress if polygon IoU >=
 * thresh (the reference CPU rule, not the `>` of its unused GPU kernels).
 */
size_t d3d_rotate_nms_workspace_bytes(int n);
int d3d_rotate_nms(const float *boxes, int n, float thresh, int max_keep, int32_t *keep,
                   int32_t *n_keep, void *workspace, size_t workspace_bytes,
                   d3d_stream_t stream);

/* All frames x one task group of MultiGroupHead.predict's front half on the device (mg_head.py:697-803, 805-1008;
 * second_box_decode box_torch_ops.py:80-148): sigmoid + max over the group's classes, `>= score_threshold`,
 * top-`pre_max_size` by score (descending; equal scores by ascending anchor index), decode of the survivors,
 * class / direction argmax -- two launches instead of ~40 per frame.
 *  box_preds dev [batch, A, code]  cls_preds dev [batch, A, nc]  dir_preds dev [batch, A, 2] or NULL  (dtype)
 *  anchors dev [A, nd] (anchors_per_batch = 0) or [batch, A, nd] f32; nd = 7 | 9, code = nd (+1 with the angle vector)
 *  out_boxes dev [batch, K, nd] f32, out_scores [batch, K] f32, out_labels / out_dir_labels [batch, K] i32
 *  (out_dir_labels may be NULL), out_count dev [batch] i32; K = pre_max_size <= 1024; rows >= count are undefined */
size_t d3d_decode_filter_topk_workspace_bytes(int batch, int anchors);
int d3d_decode_filter_topk(const void *box_preds, const void *cls_preds, const void *dir_preds, int dtype,
                           const float *anchors, int anchors_per_batch, int batch, int num_anchors, int code_size,
                           int box_ndim, int num_classes, int encode_angle_vector, float score_threshold,
                           int pre_max_size, float *out_boxes, float *out_scores, int32_t *out_labels,
                           int32_t *out_dir_labels, int32_t *out_count, void *workspace, size_t workspace_bytes,
                           d3d_stream_t stream);

/* d3d_rotate_nms over `lists` independent, score-sorted box lists in one grid (frames x task groups):
 *  boxes dev [lists, k, box_ndim] f32 with the BEV box in columns (0, 1, 3, 4, box_ndim-1) (mg_head.py:1008),
 *  counts dev [lists] i32 valid rows per list; keep dev [lists, max_keep] i32 (-1 padded), n_keep dev [lists]. */
size_t d3d_rotate_nms_batched_workspace_bytes(int lists, int k);
int d3d_rotate_nms_batched(const float *boxes, int box_ndim, const int32_t *counts, int lists, int k, float thresh,
                           int max_keep, int32_t *keep, int32_t *n_keep, void *workspace, size_t workspace_bytes,
                           d3d_stream_t stream);

/* pairwise rotated BEV IoU, iou[m,n]; boxes as above.  Covers the north-star
 * "rotated-IoU / iou3d" item (det3d/ops/iou3d/src/iou3d_kernel.cu:214-221,
 * dead code in the reference). */
int d3d_rotate_iou(const float *boxes_a, int m, const float *boxes_b, int n, float *iou,
                   d3d_stream_t stream);

/* axis-aligned NMS with the reference's +1 pixel convention and `>` rule
 * (det3d/ops/nms/nms_gpu.py:22-33,67-126; nms_kernel.cu:38-157).
 *  boxes dev [n,4] f32 (x1,y1,x2,y2) sorted by descending score. */
int d3d_nms(const float *boxes, int n, float thresh, int max_keep, int32_t *keep,
            int32_t *n_keep, void *workspace, size_t workspace_bytes, d3d_stream_t stream);

/* ---- iou3d: BEV boxes dev [*,5] f32 (x1, y1, x2, y2, ry)  (det3d/ops/iou3d/iou3d_utils.py) ------------
 * d3d_iou3d_pairwise replaces iou3d_cuda.boxes_overlap_bev_gpu / boxes_iou_bev_gpu
 * (det3d/ops/iou3d/src/iou3d.cpp:30-70 -> iou3d_kernel.cu:108-248): out[m,n] = intersection area
 * (D3D_IOU3D_OVERLAP), rotated IoU s/(sa+sb-s, floored at 1e-8) (D3D_IOU3D_IOU_BEV) or the axis-aligned
 * IoU that ignores ry (D3D_IOU3D_IOU_NORMAL, iou3d_kernel.cu:295-304). */
#define D3D_IOU3D_OVERLAP 0
#define D3D_IOU3D_IOU_BEV 1
#define D3D_IOU3D_IOU_NORMAL 2
int d3d_iou3d_pairwise(const float *boxes_a, int m, const float *boxes_b, int n, int mode, float *out,
                       d3d_stream_t stream);

/* replaces iou3d_cuda.nms_gpu (normal = 0) / nms_normal_gpu (normal = 1)  (iou3d.cpp:73-160, kernels
 * iou3d_kernel.cu:250-352).  boxes sorted by descending score; box j > i is suppressed when
 * IoU(i, j) > thresh (strict).  thresh must be >= 0.  keep dev [n] int32 positions, *n_keep dev int32.
 * The reference copies the bitmask to the host and loops there; here everything stays on the device. */
size_t d3d_iou3d_nms_workspace_bytes(int n);
int d3d_iou3d_nms(const float *boxes, int n, float thresh, int normal, int32_t *keep, int32_t *n_keep,
                  void *workspace, size_t workspace_bytes, d3d_stream_t stream);

/* replaces rotate_iou_gpu_eval (det3d/datasets/utils/kitti_object_eval_python/rotate_iou.py:305-341;
 * numba-CUDA in the reference): boxes dev [n,5], query_boxes dev [k,5] as (cx, cy, dx, dy, r), r clockwise
 * positive; out[n,k] = area(box ^ query) divided by: criterion -1 the union, 0 area(query),
 * 1 area(box), 2 nothing (raw intersection area)  (devRotateIoUEval :246-258). */
int d3d_rotate_iou_eval(const float *boxes, int n, const float *query_boxes, int k, int criterion,
                        float *out, d3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* D3D_H */
