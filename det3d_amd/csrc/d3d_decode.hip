// d3d_decode_filter_topk: the front half of MultiGroupHead.predict / get_task_detections on the device
// (det3d/models/bbox_heads/mg_head.py:697-803, 805-1008; det3d/core/bbox/box_torch_ops.py:80-148), one task
// group, all frames of the batch:
//
//   total_scores = sigmoid(cls_preds); top_scores, top_labels = max over the task's classes (mg_head.py:984-993)
//   keep top_scores >= score_threshold (:995-1000)
//   topk(scores, min(n, nms_pre_max_size)), sorted descending (box_torch_ops.py:534)
//   boxes = second_box_decode(box_preds, anchors) of the survivors only (decode is element-wise, so decoding
//           after the selection gives the same boxes as the reference's decode-everything-first, :738-741)
//   dir_labels = argmax(dir_cls_preds) (:969)
//
// The reference (and round 1 of this package) spends ~40 tiny launches and several host syncs per frame and
// task on this; here it is two launches per task group for the whole batch:
//   k1  decode_score_kernel : one thread per anchor -> 64-bit key (score bits << 32 | ~anchor) or 0 if below
//       the threshold; dense [B, A] array, no atomics
//   k2  decode_select_kernel: one 1024-thread block per frame -> 4 x 8-bit radix select of the K-th largest
//       score, deterministic tie handling (equal scores: lowest anchor index first), bitonic sort of the <= K
//       selected keys in LDS, then the box decode of the survivors
// Score ties: torch.topk leaves their order unspecified and the reference's later argsort is not stable
// (nms_cpu.py:36), so "lowest anchor first" is a legal refinement.
#include "d3d_common.h"
#include "d3d_fill.h"

#define DEC_THREADS 1024
#define DEC_KMAX 1024

__device__ __forceinline__ float dec_load(const void *p, int64_t i, int bf16) {
  if (bf16) return __uint_as_float((uint32_t)((const uint16_t *)p)[i] << 16);
  return ((const float *)p)[i];
}

__global__ void __launch_bounds__(256)
    decode_score_kernel(const void *__restrict__ cls, int bf16, int B, int A, int nc, float thr,
                        unsigned long long *__restrict__ keys) {
  const int64_t total = (int64_t)B * A;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int a = (int)(t % A);
    float best = -INFINITY;
    for (int c = 0; c < nc; ++c) best = fmaxf(best, dec_load(cls, t * nc + c, bf16));
    // torch.sigmoid in float: 1 / (1 + exp(-x)); the maximum of the sigmoids is the sigmoid of the maximum
    const float s = 1.f / (1.f + expf(-best));
    unsigned long long key = 0ull;
    if (s >= thr && !(s != s)) key = ((unsigned long long)__float_as_uint(s) << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)a);
    keys[t] = key;
  }
}

struct DecodeArgs {
  const void *box, *cls, *dir;
  const float *anchors;
  int bf16, anchors_per_batch, B, A, code, nd, nc, vec, K;
};

__global__ void __launch_bounds__(DEC_THREADS)
    decode_select_kernel(DecodeArgs g, const unsigned long long *__restrict__ keys_all, float *__restrict__ out_boxes,
                         float *__restrict__ out_scores, int32_t *__restrict__ out_labels,
                         int32_t *__restrict__ out_dir, int32_t *__restrict__ out_count) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long sel[DEC_KMAX];
  __shared__ unsigned int scan[DEC_THREADS];
  __shared__ unsigned int s_prefix, s_need, s_cnt;
  const int b = blockIdx.x, tid = threadIdx.x;
  const unsigned long long *keys = keys_all + (int64_t)b * g.A;
  const int K = g.K;

  // ---- how many candidates at all?
  unsigned int mine = 0;
  for (int a = tid; a < g.A; a += DEC_THREADS) mine += keys[a] != 0ull;
  scan[tid] = mine;
  __syncthreads();
  for (int off = DEC_THREADS / 2; off > 0; off >>= 1) {
    if (tid < off) scan[tid] += scan[tid + off];
    __syncthreads();
  }
  const unsigned int ncand = scan[0];
  __syncthreads();
  const unsigned int want = ncand < (unsigned)K ? ncand : (unsigned)K;

  // ---- radix select on the 32 score bits: after the loop `prefix` is the score of the want-th largest
  // candidate and `need` how many candidates with exactly that score are still to be taken
  if (tid == 0) {
    s_prefix = 0u;
    s_need = want;
  }
  __syncthreads();
  if (ncand > (unsigned)K) {
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      if (tid < 256) hist[tid] = 0u;
      __syncthreads();
      const unsigned int prefix = s_prefix;
      const unsigned int himask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int a = tid; a < g.A; a += DEC_THREADS) {
        const unsigned long long k = keys[a];
        if (k == 0ull) continue;
        const unsigned int sb = (unsigned int)(k >> 32);
        if ((sb & himask) == (prefix & himask)) atomicAdd(&hist[(sb >> shift) & 255u], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        unsigned int need = s_need, bin = 255;
        for (;; --bin) {  // from the largest digit down: the bin holding the need-th largest
          if (hist[bin] >= need || bin == 0) break;
          need -= hist[bin];
        }
        s_prefix = prefix | (bin << shift);
        s_need = need;
      }
      __syncthreads();
    }
  }
  const unsigned int tscore = s_prefix, need_ties = s_need;
  const bool take_all = ncand <= (unsigned)K;

  // ---- compaction.  Threads own CONTIGUOUS anchor ranges so that ties are ranked by anchor index.
  const int per = (g.A + DEC_THREADS - 1) / DEC_THREADS;
  const int a0 = tid * per, a1 = min(a0 + per, g.A);
  unsigned int n_gt = 0, n_eq = 0;
  for (int a = a0; a < a1; ++a) {
    const unsigned long long k = keys[a];
    if (k == 0ull) continue;
    const unsigned int sb = (unsigned int)(k >> 32);
    if (take_all || sb > tscore) ++n_gt;
    else if (sb == tscore) ++n_eq;
  }
  // exclusive scans of both counts over the block (Hillis-Steele, 10 steps each)
  unsigned int off_gt, off_eq;
  {
    scan[tid] = n_gt;
    __syncthreads();
    for (int d = 1; d < DEC_THREADS; d <<= 1) {
      unsigned int v = tid >= d ? scan[tid - d] : 0u;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    off_gt = scan[tid] - n_gt;
    const unsigned int total_gt = scan[DEC_THREADS - 1];
    __syncthreads();
    scan[tid] = n_eq;
    __syncthreads();
    for (int d = 1; d < DEC_THREADS; d <<= 1) {
      unsigned int v = tid >= d ? scan[tid - d] : 0u;
      __syncthreads();
      scan[tid] += v;
      __syncthreads();
    }
    off_eq = scan[tid] - n_eq;  // rank of this thread's first tie among all ties, in anchor order
    __syncthreads();
    if (tid == 0) s_cnt = total_gt + (take_all ? 0u : need_ties);
    scan[0] = total_gt;
  }
  __syncthreads();
  const unsigned int total_gt = scan[0];
  const unsigned int count = s_cnt < (unsigned)K ? s_cnt : (unsigned)K;
  {
    unsigned int wg = off_gt, rank_eq = off_eq;
    for (int a = a0; a < a1; ++a) {
      const unsigned long long k = keys[a];
      if (k == 0ull) continue;
      const unsigned int sb = (unsigned int)(k >> 32);
      if (take_all || sb > tscore) {
        if (wg < (unsigned)DEC_KMAX) sel[wg] = k;
        ++wg;
      } else if (sb == tscore) {
        // ties behind the strictly greater ones; the need_ties lowest anchors win
        if (rank_eq < need_ties && total_gt + rank_eq < (unsigned)DEC_KMAX) sel[total_gt + rank_eq] = k;
        ++rank_eq;
      }
    }
  }
  for (int i = tid; i < DEC_KMAX; i += DEC_THREADS)
    if ((unsigned)i >= count) sel[i] = 0ull;
  __syncthreads();

  // ---- bitonic sort, descending: larger score first, equal scores by lower anchor (keys hold ~anchor)
  for (int size = 2; size <= DEC_KMAX; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int i = tid;
      const int j = i ^ stride;
      if (j > i && i < DEC_KMAX) {
        const unsigned long long x = sel[i], y = sel[j];
        const bool desc = (i & size) == 0;
        if (desc ? (x < y) : (x > y)) {
          sel[i] = y;
          sel[j] = x;
        }
      }
      __syncthreads();
    }
  }

  // ---- decode the survivors (second_box_decode, box_torch_ops.py:80-148)
  if (tid == 0) out_count[b] = (int32_t)count;
  if ((unsigned)tid < count) {
    const unsigned long long k = sel[tid];
    const int a = (int)(0xFFFFFFFFu - (uint32_t)(k & 0xFFFFFFFFull));
    const int64_t row = (int64_t)b * g.A + a;
    const float *an = g.anchors + ((int64_t)(g.anchors_per_batch ? b : 0) * g.A + a) * g.nd;
    float t[10];
    for (int c = 0; c < g.code; ++c) t[c] = dec_load(g.box, row * g.code + c, g.bf16);
    const float xa = an[0], ya = an[1], za = an[2], wa = an[3], la = an[4], ha = an[5], ra = an[g.nd - 1];
    const float diag = sqrtf(la * la + wa * wa);
    float *o = out_boxes + ((int64_t)b * K + tid) * g.nd;
    o[0] = t[0] * diag + xa;
    o[1] = t[1] * diag + ya;
    o[2] = t[2] * ha + za;
    o[3] = expf(t[3]) * wa;
    o[4] = expf(t[4]) * la;
    o[5] = expf(t[5]) * ha;
    int r = 6;
    if (g.nd == 9) {
      o[6] = t[6] + an[6];
      o[7] = t[7] + an[7];
      r = 8;
    }
    if (g.vec) o[g.nd - 1] = atan2f(t[r + 1] + sinf(ra), t[r] + cosf(ra));
    else o[g.nd - 1] = t[r] + ra;
    out_scores[(int64_t)b * K + tid] = __uint_as_float((uint32_t)(k >> 32));
    // label = argmax over classes (first maximum, like torch.max), direction = argmax of the 2 dir logits
    int lab = 0;
    float best = dec_load(g.cls, row * g.nc, g.bf16);
    for (int c = 1; c < g.nc; ++c) {
      const float v = dec_load(g.cls, row * g.nc + c, g.bf16);
      if (v > best) {
        best = v;
        lab = c;
      }
    }
    out_labels[(int64_t)b * K + tid] = lab;
    if (out_dir) {
      int d = 0;
      if (g.dir) d = dec_load(g.dir, row * 2 + 1, g.bf16) > dec_load(g.dir, row * 2, g.bf16) ? 1 : 0;
      out_dir[(int64_t)b * K + tid] = d;
    }
  }
}

extern "C" size_t d3d_decode_filter_topk_workspace_bytes(int batch, int anchors) {
  return d3d_align_up((size_t)batch * anchors * sizeof(unsigned long long)) + 256;
}

extern "C" int d3d_decode_filter_topk(const void *box_preds, const void *cls_preds, const void *dir_preds,
                                      int dtype, const float *anchors, int anchors_per_batch, int batch,
                                      int num_anchors, int code_size, int box_ndim, int num_classes,
                                      int encode_angle_vector, float score_threshold, int pre_max_size,
                                      float *out_boxes, float *out_scores, int32_t *out_labels,
                                      int32_t *out_dir_labels, int32_t *out_count, void *workspace,
                                      size_t workspace_bytes, d3d_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  D3D_REQUIRE(dtype == D3D_F32 || dtype == D3D_BF16, "d3d_decode_filter_topk: dtype %d", dtype);
  D3D_REQUIRE(batch >= 0 && num_anchors >= 0 && num_classes >= 1, "d3d_decode_filter_topk: sizes");
  D3D_REQUIRE(box_ndim == 7 || box_ndim == 9, "d3d_decode_filter_topk: box_ndim %d (7 or 9)", box_ndim);
  D3D_REQUIRE(code_size == box_ndim + (encode_angle_vector ? 1 : 0) && code_size <= 10,
              "d3d_decode_filter_topk: code_size %d does not match box_ndim %d / angle vector %d", code_size,
              box_ndim, encode_angle_vector);
  D3D_REQUIRE(pre_max_size >= 1 && pre_max_size <= DEC_KMAX, "d3d_decode_filter_topk: pre_max_size %d (1..%d)",
              pre_max_size, DEC_KMAX);
  D3D_REQUIRE(out_count, "d3d_decode_filter_topk: null out_count");
  if (batch == 0) return D3D_OK;
  if (num_anchors == 0) {
    D3D_HIP_TRY(d3d_fill_async(stream, out_count, 0, (size_t)batch * 4));
    return D3D_OK;
  }
  D3D_REQUIRE(box_preds && cls_preds && anchors && out_boxes && out_scores && out_labels && workspace,
              "d3d_decode_filter_topk: null pointer");
  if (workspace_bytes < d3d_decode_filter_topk_workspace_bytes(batch, num_anchors))
    return d3d_set_error(D3D_ECAPACITY, "d3d_decode_filter_topk: workspace too small");
  unsigned long long *keys = (unsigned long long *)workspace;
  const int64_t total = (int64_t)batch * num_anchors;
  int nb = (int)((total + 255) / 256);
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(decode_score_kernel, dim3(nb), dim3(256), 0, stream, cls_preds, dtype == D3D_BF16, batch,
                     num_anchors, num_classes, score_threshold, keys);
  D3D_LAUNCH_TRY("decode_score_kernel");
  DecodeArgs g;
  g.box = box_preds; g.cls = cls_preds; g.dir = dir_preds; g.anchors = anchors;
  g.bf16 = dtype == D3D_BF16; g.anchors_per_batch = anchors_per_batch; g.B = batch; g.A = num_anchors;
  g.code = code_size; g.nd = box_ndim; g.nc = num_classes; g.vec = encode_angle_vector; g.K = pre_max_size;
  hipLaunchKernelGGL(decode_select_kernel, dim3(batch), dim3(DEC_THREADS), 0, stream, g, keys, out_boxes,
                     out_scores, out_labels, out_dir_labels, out_count);
  D3D_LAUNCH_TRY("decode_select_kernel");
  return D3D_OK;
}
