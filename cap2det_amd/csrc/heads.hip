// MIDN / OICR / label kernels (latency-bound, tiny): column softmax over proposals with
// wave64 shuffle reductions, OICR pseudo-label + cross-entropy, caption -> label MLP.
//
// Reference call sites replaced:
//   c2d_midn_fwd/bwd            models/cap2det_model.py:53-109  (+ core/utils.py:101-113,172-184)
//   c2d_sigmoid_ce_fwd_bwd      models/cap2det_model.py:293-297
//   c2d_oicr_select             models/utils.py:39-62           (+ core/utils.py:187-199)
//   c2d_oicr_loss_fwd_bwd       models/utils.py:64-103, models/cap2det_model.py:314-328
//   c2d_labels_from_ids         models/label_extractor.py:15-39,183-207
//   c2d_text_classifier_fwd     models/label_extractor.py:353-421,442-472
//   c2d_word_vector_match_fwd   models/label_extractor.py:232-328
#include "c2d_common.h"

namespace {

constexpr float kBig = 1e10f;     // core/utils.py:10
constexpr float kSmall = 1e-10f;  // core/utils.py:11

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = c2d_wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = c2d_wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s = fmaxf(s, red[i]);
  return s;
}
__device__ __forceinline__ float block_min(float v, float* red) {
  v = c2d_wave_min(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  float s = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) s = fminf(s, red[i]);
  return s;
}

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

// grid (C, B).  logits: [B*N][ld], r|c logits at column off_r / off_c.
constexpr int kMidnCache = 8;        // proposals per thread held in registers (N <= 2048)
__global__ __launch_bounds__(256) void midn_fwd_kernel(
    const float* __restrict__ logits, int ld, int off_r, int off_c,
    const int32_t* __restrict__ num_proposals, float* __restrict__ proba,
    float* __restrict__ class_logits, float* __restrict__ scores, int N, int C) {
  __shared__ float red[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const int nb = num_proposals[b];
  const float* L = logits + (size_t)b * N * ld;
  if (N <= kMidnCache * 256) {
    // The class's two logit columns live in registers (proposal r = tid + 256 j): every strided
    // column load of the block is in flight at once instead of four dependent sweeps.  Same
    // values, same per-thread summation order, same block reductions as the sweeps below.
    float u[kMidnCache], lc[kMidnCache];
#pragma unroll
    for (int j = 0; j < kMidnCache; ++j) {
      const int r = threadIdx.x + 256 * j;
      const float m = r < nb ? 1.0f : 0.0f;
      const bool in = r < N;
      const float lr = in ? L[(size_t)r * ld + off_r + c] : 0.f;
      lc[j] = in ? L[(size_t)r * ld + off_c + c] : 0.f;
      u[j] = m * lr - kBig * (1.0f - m);
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < kMidnCache; ++j)
      if (threadIdx.x + 256 * j < N) mx = fmaxf(mx, u[j]);
    mx = block_max(mx, red);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < kMidnCache; ++j)
      if (threadIdx.x + 256 * j < N) se += expf(u[j] - mx);
    se = block_sum(se, red);
    float cl = 0.f;
#pragma unroll
    for (int j = 0; j < kMidnCache; ++j) {
      const int r = threadIdx.x + 256 * j;
      if (r < N) {
        const float m = r < nb ? 1.0f : 0.0f;
        u[j] = m * (expf(u[j] - mx) / se);             // (u now holds the probability)
        cl += m * (lc[j] * u[j]);
      }
    }
    cl = block_sum(cl, red);
    if (threadIdx.x == 0) class_logits[b * C + c] = cl;
    const float sg = sigmoidf(cl);
#pragma unroll
    for (int j = 0; j < kMidnCache; ++j) {
      const int r = threadIdx.x + 256 * j;
      if (r < N) {
        const size_t i = ((size_t)b * N + r) * C + c;
        proba[i] = u[j];
        scores[i] = sg * u[j];
      }
    }
    return;
  }
  // u_r = mask*Lr - 1e10*(1-mask)   (models/cap2det_model.py:92-93, core/utils.py:183-184)
  float mx = -INFINITY;
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    const float m = r < nb ? 1.0f : 0.0f;
    const float u = m * L[(size_t)r * ld + off_r + c] - kBig * (1.0f - m);
    mx = fmaxf(mx, u);
  }
  mx = block_max(mx, red);
  float se = 0.f;
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    const float m = r < nb ? 1.0f : 0.0f;
    const float u = m * L[(size_t)r * ld + off_r + c] - kBig * (1.0f - m);
    se += expf(u - mx);
  }
  se = block_sum(se, red);
  float cl = 0.f;
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    const float m = r < nb ? 1.0f : 0.0f;
    const float u = m * L[(size_t)r * ld + off_r + c] - kBig * (1.0f - m);
    const float p = m * (expf(u - mx) / se);
    proba[((size_t)b * N + r) * C + c] = p;
    cl += m * (L[(size_t)r * ld + off_c + c] * p);
  }
  cl = block_sum(cl, red);
  if (threadIdx.x == 0) class_logits[b * C + c] = cl;
  const float sg = sigmoidf(cl);
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    const size_t i = ((size_t)b * N + r) * C + c;
    scores[i] = sg * proba[i];
  }
}

// dLc = g*P ; dLr = g*P*(mask*Lc - cl)      (P already carries the mask)
__global__ __launch_bounds__(256) void midn_bwd_kernel(
    const float* __restrict__ dclass_logits, const float* __restrict__ logits, int ld, int off_r,
    int off_c, const int32_t* __restrict__ num_proposals, const float* __restrict__ proba,
    const float* __restrict__ class_logits, float* __restrict__ dlogits, int lddl, int N,
    int C) {
  const int c = blockIdx.x, b = blockIdx.y;
  const int nb = num_proposals[b];
  const float g = dclass_logits[b * C + c];
  const float cl = class_logits[b * C + c];
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    const size_t row = (size_t)b * N + r;
    const float m = r < nb ? 1.0f : 0.0f;
    const float p = proba[row * C + c];
    const float lc = logits[row * ld + off_c + c];
    dlogits[row * lddl + off_c + c] = g * p;
    dlogits[row * lddl + off_r + c] = g * p * (m * lc - cl);
  }
}

// loss[0] += weight * mean(bce);  dlogits = weight * (sigmoid(x) - z) / n
__global__ __launch_bounds__(256) void sigmoid_ce_kernel(const float* __restrict__ logits,
                                                         const float* __restrict__ labels,
                                                         int n, float weight,
                                                         float* __restrict__ loss,
                                                         float* __restrict__ dlogits) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float x = logits[i], z = labels[i];
    s += fmaxf(x, 0.f) - x * z + log1pf(expf(-fabsf(x)));
    if (dlogits) dlogits[i] = weight * (sigmoidf(x) - z) / (float)n;
  }
  s = block_sum(s, red);
  if (threadIdx.x == 0 && loss) atomicAdd(loss, weight * s / (float)n);
}

// grid (C, B): picks the most confident proposal of class c (masked_argmax, first index on
// ties, axis minimum taken over padded rows too) and stores its index and box.
__device__ __forceinline__ void oicr_select_body(
    const float* __restrict__ s0, int ld, int off, const int32_t* __restrict__ num_proposals,
    const float* __restrict__ boxes, int32_t* __restrict__ idx_out, float* __restrict__ box_out,
    int N, int C) {
  __shared__ float red[4];
  __shared__ float sval[4];
  __shared__ int sidx[4];
  const int c = blockIdx.x, b = blockIdx.y;
  const int nb = num_proposals[b];
  const float* S = s0 + (size_t)b * N * ld + off + c;
  float mn = INFINITY;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (N <= kMidnCache * 256) {
    // the class column in registers (proposal r = tid + 256 j): one strided sweep instead of two
    float sv[kMidnCache];
#pragma unroll
    for (int j = 0; j < kMidnCache; ++j) {
      const int r = threadIdx.x + 256 * j;
      sv[j] = r < N ? S[(size_t)r * ld] : INFINITY;
      mn = fminf(mn, sv[j]);
    }
    mn = block_min(mn, red);
#pragma unroll
    for (int j = 0; j < kMidnCache; ++j) {
      const int r = threadIdx.x + 256 * j;
      if (r < N) {
        const float v = (sv[j] - mn) * (r < nb ? 1.0f : 0.0f);
        if (v > best) { best = v; bi = r; }   // r ascending per thread: keeps the first
      }
    }
  } else {
    for (int r = threadIdx.x; r < N; r += blockDim.x) mn = fminf(mn, S[(size_t)r * ld]);
    mn = block_min(mn, red);
    for (int r = threadIdx.x; r < N; r += blockDim.x) {
      const float v = (S[(size_t)r * ld] - mn) * (r < nb ? 1.0f : 0.0f);
      if (v > best) { best = v; bi = r; }   // r ascending per thread: keeps the first
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if ((threadIdx.x & 63) == 0) { sval[threadIdx.x >> 6] = best; sidx[threadIdx.x >> 6] = bi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i)
      if (sval[i] > best || (sval[i] == best && sidx[i] < bi)) { best = sval[i]; bi = sidx[i]; }
    idx_out[b * C + c] = bi;
    const float* bx = boxes + ((size_t)b * N + bi) * 4;
    float* o = box_out + ((size_t)b * C + c) * 4;
    o[0] = bx[0]; o[1] = bx[1]; o[2] = bx[2]; o[3] = bx[3];
  }
}

__global__ __launch_bounds__(256) void oicr_select_kernel(
    const float* __restrict__ s0, int ld, int off, const int32_t* __restrict__ num_proposals,
    const float* __restrict__ boxes, int32_t* __restrict__ idx_out, float* __restrict__ box_out,
    int N, int C) {
  oicr_select_body(s0, ld, off, num_proposals, boxes, idx_out, box_out, N, C);
}

// All K refinement stages in one launch (grid (C, B, K)): stage 0 searches s0, stage k > 0 the
// class columns of softmax(scores of stage k - 1) in Q (rows of C + 1, background first), which
// oicr_softmax_kernel has written; indices / boxes of stage k at idx_out + k B C / box_out + 4 k B C.
__global__ __launch_bounds__(256) void oicr_select_multi_kernel(
    const float* __restrict__ s0, int ld, int off, const float* __restrict__ Q,
    const int32_t* __restrict__ num_proposals, const float* __restrict__ boxes,
    int32_t* __restrict__ idx_out, float* __restrict__ box_out, int B, int N, int C) {
  const int k = blockIdx.z;
  const size_t bc = (size_t)B * C;
  if (k > 0) {
    s0 = Q + (size_t)(k - 1) * B * N * (C + 1);
    ld = C + 1;
    off = 1;
  }
  oicr_select_body(s0, ld, off, num_proposals, boxes, idx_out + k * bc, box_out + 4 * k * bc, N, C);
}

__device__ __forceinline__ float box_area(float y0, float x0, float y1, float x1) {
  return fmaxf(x1 - x0, 0.0f) * fmaxf(y1 - y0, 0.0f);
}

// One wave per proposal row; lanes stride over the C+1 columns.
__device__ __forceinline__ void oicr_loss_body(
    const float* __restrict__ S, int ld, int off, const float* __restrict__ top_boxes,
    const float* __restrict__ boxes, const float* __restrict__ labels,
    const int32_t* __restrict__ num_proposals, float iou_thr, float weight, int B, int N, int C,
    float* __restrict__ loss, float* __restrict__ dS, int lddS, int doff,
    float* __restrict__ Q) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  float contrib = 0.f;
  if (row < (long long)B * N) {
    const int b = (int)(row / N), r = (int)(row % N);
    const int nb = num_proposals[b];
    const float m = r < nb ? 1.0f : 0.0f;
    const float* bx = boxes + row * 4;
    const float y0 = bx[0], x0 = bx[1], y1 = bx[2], x1 = bx[3];
    const float a1 = box_area(y0, x0, y1, x1);
    const float* s = S + row * ld + off;
    const int C1 = C + 1;
    // pass 1: targets (classes live at columns 1..C) and the row max
    float tsum = 0.f, mx = -INFINITY;
    for (int j = lane; j < C1; j += 64) {
      mx = fmaxf(mx, s[j]);
      if (j >= 1) {
        const int c = j - 1;
        const float* tb = top_boxes + ((size_t)b * C + c) * 4;
        const float iy0 = fmaxf(y0, tb[0]), ix0 = fmaxf(x0, tb[1]);
        const float iy1 = fminf(y1, tb[2]), ix1 = fminf(x1, tb[3]);
        const float inter = box_area(iy0, ix0, iy1, ix1);
        const float uni = a1 + box_area(tb[0], tb[1], tb[2], tb[3]) - inter;
        const float iou = inter / uni;  // 0/0 -> NaN -> compares false (models/utils.py:73)
        const float t = (labels[b * C + c] > 0.f && iou >= iou_thr) ? 1.0f : 0.0f;
        tsum += t;
      }
    }
    tsum = c2d_wave_sum(tsum);
    mx = c2d_wave_max(mx);
    float se = 0.f;
    for (int j = lane; j < C1; j += 64) se += expf(s[j] - mx);
    se = c2d_wave_sum(se);
    const float lse = logf(se);
    const float bkg = tsum > 0.f ? 0.0f : 1.0f;
    const float tot = bkg + tsum;
    const float denom = fmaxf(kSmall, (float)nb);
    float ce = 0.f;
    for (int j = lane; j < C1; j += 64) {
      float t;
      if (j == 0) {
        t = bkg;
      } else {
        const int c = j - 1;
        const float* tb = top_boxes + ((size_t)b * C + c) * 4;
        const float iy0 = fmaxf(y0, tb[0]), ix0 = fmaxf(x0, tb[1]);
        const float iy1 = fminf(y1, tb[2]), ix1 = fminf(x1, tb[3]);
        const float inter = box_area(iy0, ix0, iy1, ix1);
        const float uni = a1 + box_area(tb[0], tb[1], tb[2], tb[3]) - inter;
        const float iou = inter / uni;
        t = (labels[b * C + c] > 0.f && iou >= iou_thr) ? 1.0f : 0.0f;
      }
      const float lab = t / tot;
      const float z = s[j] - mx;
      const float q = expf(z) / se;
      ce -= lab * (z - lse);
      if (Q) Q[row * C1 + j] = q;
      if (dS) dS[row * lddS + doff + j] = weight * (q - lab) * m / denom / (float)B;
    }
    ce = c2d_wave_sum(ce);
    contrib = weight * ce * m / denom / (float)B;
  }
  if (lane != 0) contrib = 0.f;
  const float tot = block_sum(contrib, red);
  if (threadIdx.x == 0 && loss && tot != 0.f) atomicAdd(loss, tot);
}

__global__ __launch_bounds__(256) void oicr_loss_kernel(
    const float* __restrict__ S, int ld, int off, const float* __restrict__ top_boxes,
    const float* __restrict__ boxes, const float* __restrict__ labels,
    const int32_t* __restrict__ num_proposals, float iou_thr, float weight, int B, int N, int C,
    float* __restrict__ loss, float* __restrict__ dS, int lddS, int doff,
    float* __restrict__ Q) {
  oicr_loss_body(S, ld, off, top_boxes, boxes, labels, num_proposals, iou_thr, weight, B, N, C,
                 loss, dS, lddS, doff, Q);
}

// Stage k = blockIdx.y of K: score / gradient columns off + k (C + 1), its own top boxes, loss
// scalar and softmax plane.
__global__ __launch_bounds__(256) void oicr_loss_multi_kernel(
    const float* __restrict__ S, int ld, int off, const float* __restrict__ top_boxes,
    const float* __restrict__ boxes, const float* __restrict__ labels,
    const int32_t* __restrict__ num_proposals, float iou_thr, float weight, int B, int N, int C,
    float* __restrict__ loss, float* __restrict__ dS, int lddS, int doff,
    float* __restrict__ Q) {
  const int k = blockIdx.y;
  oicr_loss_body(S, ld, off + k * (C + 1), top_boxes + (size_t)4 * k * B * C, boxes, labels,
                 num_proposals, iou_thr, weight, B, N, C, loss + k, dS, lddS, doff + k * (C + 1),
                 Q + (size_t)k * B * N * (C + 1));
}

// softmax of the score rows of stages 0 .. K - 2 (blockIdx.y), exactly as oicr_loss_body computes
// and stores it (same lanes, same reductions: the same bits) — what the selection of the NEXT stage
// searches, ahead of the loss kernels that used to produce it one stage at a time.
__global__ __launch_bounds__(256) void oicr_softmax_kernel(const float* __restrict__ S, int ld,
                                                           int off, int B, int N, int C,
                                                           float* __restrict__ Q) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * N) return;
  const int k = blockIdx.y;
  const int C1 = C + 1;
  const float* s = S + row * ld + off + k * C1;
  float mx = -INFINITY;
  for (int j = lane; j < C1; j += 64) mx = fmaxf(mx, s[j]);
  mx = c2d_wave_max(mx);
  float se = 0.f;
  for (int j = lane; j < C1; j += 64) se += expf(s[j] - mx);
  se = c2d_wave_sum(se);
  float* q = Q + ((size_t)k * B * N + row) * C1;
  for (int j = lane; j < C1; j += 64) q[j] = expf(s[j] - mx) / se;
}

// labels[b][c] = 1 if any ids[b][t] == c (ids >= C are out-of-vocabulary)
__global__ __launch_bounds__(256) void labels_from_ids_kernel(const int32_t* __restrict__ ids,
                                                              int T, int C,
                                                              float* __restrict__ labels) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < C; c += blockDim.x) labels[b * C + c] = 0.f;
  __syncthreads();
  for (int t = threadIdx.x; t < T; t += blockDim.x) {
    const int id = ids[b * T + t];
    if (id >= 0 && id < C) labels[b * C + id] = 1.0f;
  }
}

// One 1024-thread block per caption.  hidden = relu(masked_maximum_t(E[id_t] . W1 + b1));
// logits = hidden . W2 + b2.  masked_maximum is the reference's literal formula
// max_t((v_t - min_T v) * m_t) + min_T v (core/utils.py:75-79), so the pre-activations are
// evaluated twice (sweep 0: min over ALL tokens, sweep 1: the masked maximum) instead of being
// kept: T x H floats do not fit beside the staged embedding rows, and the second sweep costs
// ~10 us.  Work split: thread -> (hidden unit h, token slice); the embedding rows of a chunk of
// TXT_TC tokens are staged in LDS and read as broadcasts, a W1 element is loaded once per chunk
// and reused for the slice's tokens of that chunk (the previous form walked tokens serially with
// a block barrier per token: 8 ms per caption at T = 60 — longer than the whole detector step).
constexpr int TXT_TC = 16;      // tokens per staged chunk
constexpr int TXT_MAXJ = 16;    // tokens of a chunk one thread may own (= TXT_TC when nslice = 1)
__global__ __launch_bounds__(1024) void text_classifier_kernel(
    const int32_t* __restrict__ ids, int T, const float* __restrict__ emb, int vocab, int E,
    const float* __restrict__ w1, const float* __restrict__ b1, int H,
    const float* __restrict__ w2, const float* __restrict__ b2, int C,
    float* __restrict__ logits) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Epad = (E + 3) & ~3;
  float* hid = smem;                      // [H]   final hidden layer
  float* part = hid + H;                  // [nslice][H] per-slice partial min / max
  const int b = blockIdx.x;
  const int hspan = min(((H + 63) / 64) * 64, (int)blockDim.x);   // threads per token slice
  const int nslice = max(1, min((int)blockDim.x / hspan, TXT_TC));
  float* erow = part + nslice * H;        // [TXT_TC][Epad] staged embedding rows
  int* eid = reinterpret_cast<int*>(erow + TXT_TC * Epad);    // [TXT_TC] their (clamped) ids
  const int slice = threadIdx.x / hspan;
  const int hl = threadIdx.x - slice * hspan;
  const bool worker = slice < nslice;
  for (int h0 = 0; h0 < H; h0 += hspan) {
    const int h = h0 + hl;
    const bool on = worker && h < H;
    const float bias = on ? b1[h] : 0.f;
    float mn = INFINITY;
    for (int sweep = 0; sweep < 2; ++sweep) {
      float best = -INFINITY, smin = INFINITY;
      for (int t0 = 0; t0 < T; t0 += TXT_TC) {
        const int nt = min(TXT_TC, T - t0);
        __syncthreads();                  // the previous chunk's rows are no longer read
        for (int i = threadIdx.x; i < nt * Epad; i += blockDim.x) {
          const int t = i / Epad, e = i - t * Epad;
          int id = ids[b * T + t0 + t];
          if (id < 0 || id > vocab) id = vocab;
          erow[i] = e < E ? emb[(size_t)id * E + e] : 0.f;
          if (e == 0) eid[t] = id;
        }
        __syncthreads();
        if (on) {
          float acc[TXT_MAXJ];
#pragma unroll
          for (int j = 0; j < TXT_MAXJ; ++j) acc[j] = 0.f;
          for (int e = 0; e < E; ++e) {
            const float w = w1[(size_t)e * H + h];
#pragma unroll
            for (int j = 0; j < TXT_MAXJ; ++j) {
              const int t = slice + j * nslice;
              if (t < nt) acc[j] += erow[t * Epad + e] * w;
            }
          }
#pragma unroll
          for (int j = 0; j < TXT_MAXJ; ++j) {
            const int t = slice + j * nslice;
            if (t < nt) {
              const float v = acc[j] + bias;
              if (sweep == 0) smin = fminf(smin, v);
              else best = fmaxf(best, (v - mn) * (eid[t] != vocab ? 1.0f : 0.0f));
            }
          }
        }
      }
      // combine the token slices of hidden unit h (min / max: order-independent, exact)
      if (on) part[slice * H + h] = sweep == 0 ? smin : best;
      __syncthreads();
      if (on) {
        float r = part[h];
        for (int q = 1; q < nslice; ++q)
          r = sweep == 0 ? fminf(r, part[q * H + h]) : fmaxf(r, part[q * H + h]);
        if (sweep == 0) mn = r;
        else if (slice == 0) hid[h] = fmaxf(r + mn, 0.0f);
      }
      __syncthreads();
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = 0.f;
    for (int h = 0; h < H; ++h) v += hid[h] * w2[(size_t)h * C + c];
    logits[b * C + c] = v + b2[c];
  }
}

// Workspace form (the one the detector's label extractor uses): the hidden layer is independent
// per hidden unit, so the grid is (hidden units / 64, captions) instead of one workgroup per
// caption, and a workgroup keeps ITS 64 columns of W1 in LDS ([E][64] fp32, 75 KiB at E = 300)
// for both sweeps and every token chunk: 20 000 LDS-fed FMAs per lane and no global load inside
// the loops (the one-workgroup form re-reads W1 from L2 once per chunk and sweep: ~1 ms per
// caption).  hidden[b][h] goes to the caller's workspace; text_logits_kernel finishes.
constexpr int TXT_HB = 64;      // hidden units per workgroup
__global__ __launch_bounds__(256) void text_hidden_kernel(
    const int32_t* __restrict__ ids, int T, const float* __restrict__ emb, int vocab, int E,
    const float* __restrict__ w1, const float* __restrict__ b1, int H,
    float* __restrict__ hidden) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Epad = (E + 3) & ~3;
  float* wl = smem;                               // [E][TXT_HB]
  float* erow = wl + (size_t)E * TXT_HB;          // [TXT_TC][Epad]
  float* part = erow + TXT_TC * Epad;             // [4][TXT_HB]
  int* eid = reinterpret_cast<int*>(part + 4 * TXT_HB);
  const int b = blockIdx.y, h0 = blockIdx.x * TXT_HB;
  const int hl = threadIdx.x & (TXT_HB - 1), slice = threadIdx.x >> 6;   // 4 token slices
  const int h = h0 + hl;
  const bool on = h < H;
  for (int i = threadIdx.x; i < E * TXT_HB; i += 256) {
    const int e = i >> 6, c = i & (TXT_HB - 1);
    wl[i] = h0 + c < H ? w1[(size_t)e * H + h0 + c] : 0.f;
  }
  const float bias = on ? b1[h] : 0.f;
  float mn = INFINITY;
  for (int sweep = 0; sweep < 2; ++sweep) {
    float best = -INFINITY, smin = INFINITY;
    for (int t0 = 0; t0 < T; t0 += TXT_TC) {
      const int nt = min(TXT_TC, T - t0);
      __syncthreads();
      for (int i = threadIdx.x; i < nt * Epad; i += 256) {
        const int t = i / Epad, e = i - t * Epad;
        int id = ids[b * T + t0 + t];
        if (id < 0 || id > vocab) id = vocab;
        erow[i] = e < E ? emb[(size_t)id * E + e] : 0.f;
        if (e == 0) eid[t] = id;
      }
      __syncthreads();
      float acc[TXT_TC / 4];
#pragma unroll
      for (int j = 0; j < TXT_TC / 4; ++j) acc[j] = 0.f;
      for (int e = 0; e < E; ++e) {
        const float w = wl[e * TXT_HB + hl];
#pragma unroll
        for (int j = 0; j < TXT_TC / 4; ++j) {
          const int t = slice + 4 * j;            // (rows past nt hold stale data: never used below)
          acc[j] += erow[t * Epad + e] * w;
        }
      }
#pragma unroll
      for (int j = 0; j < TXT_TC / 4; ++j) {
        const int t = slice + 4 * j;
        if (t < nt) {
          const float v = acc[j] + bias;
          if (sweep == 0) smin = fminf(smin, v);
          else best = fmaxf(best, (v - mn) * (eid[t] != vocab ? 1.0f : 0.0f));
        }
      }
    }
    part[slice * TXT_HB + hl] = sweep == 0 ? smin : best;
    __syncthreads();
    float r = part[hl];
    for (int q = 1; q < 4; ++q)
      r = sweep == 0 ? fminf(r, part[q * TXT_HB + hl]) : fmaxf(r, part[q * TXT_HB + hl]);
    if (sweep == 0) mn = r;
    else if (slice == 0 && on) hidden[(size_t)b * H + h] = fmaxf(r + mn, 0.0f);
    __syncthreads();
  }
}

// logits[b][c] = hidden[b] . W2[:, c] + b2[c]; one wave per class (lanes stride the hidden units,
// wave-shuffle sum), then the label merge of text_labels_merge_kernel.
__global__ __launch_bounds__(256) void text_logits_kernel(
    const float* __restrict__ hidden, int H, const float* __restrict__ w2,
    const float* __restrict__ b2, int C, const float* __restrict__ exact, float thr,
    float* __restrict__ logits, float* __restrict__ labels) {
  extern __shared__ __attribute__((aligned(16))) float hid[];   // [H]
  __shared__ int any_exact;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) any_exact = 0;
  for (int h = threadIdx.x; h < H; h += blockDim.x) hid[h] = hidden[(size_t)b * H + h];
  __syncthreads();
  if (labels)
    for (int c = threadIdx.x; c < C; c += blockDim.x)
      if (exact[b * C + c] > 0.f) any_exact = 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = wave; c < C; c += 4) {
    float v = 0.f;
    for (int h = lane; h < H; h += 64) v += hid[h] * w2[(size_t)h * C + c];
    v = c2d_wave_sum(v);
    if (lane == 0) logits[b * C + c] = v + b2[c];
  }
  __syncthreads();
  if (labels)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float ml = sigmoidf(logits[b * C + c]) > thr ? 1.0f : 0.0f;
      labels[b * C + c] = any_exact ? exact[b * C + c] : ml;
    }
}

// ---------------------------------------------------------------------------------------------
// Text-classifier TRAINING (models/text_model.py:31-129 through
// models/label_extractor.py:353-421 with is_training=True).  The two FC layers run on the MFMA
// GEMM kernels (c2d_conv_fwd / dgrad / wgrad with 1x1 geometry); these kernels do the rest.
// ---------------------------------------------------------------------------------------------

// x[b*T + t][0..ld) = embedding[id] zero-padded from E to ld columns (ld % 16 == 0 for the GEMM).
__global__ __launch_bounds__(256) void embedding_gather_kernel(const int32_t* __restrict__ ids,
                                                               long long rows,
                                                               const float* __restrict__ emb,
                                                               int vocab, int E, int ld,
                                                               float* __restrict__ x) {
  const long long total = rows * ld;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / ld;
    const int e = (int)(i - r * ld);
    int id = ids[r];
    if (id < 0 || id > vocab) id = vocab;
    x[i] = e < E ? emb[(size_t)id * E + e] : 0.0f;
  }
}

// hidden[b][h] = dropout(relu(masked_maximum_t(pre[b][t][h]))) with masked_maximum =
// max_t((pre - min_t pre) * mask_t) + min_t pre, mask_t = (id_t != OOV) (core/utils.py:63-79),
// dropout = x * keepmask / keep_prob (slim.dropout; keepmask NULL = no dropout).
__global__ __launch_bounds__(256) void text_pool_fwd_kernel(
    const float* __restrict__ pre, const int32_t* __restrict__ ids, int T, int H, int vocab,
    const uint8_t* __restrict__ keepmask, float inv_keep, float* __restrict__ hidden) {
  const int b = blockIdx.y;
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const float* p = pre + (size_t)b * T * H + h;
  float mn = INFINITY;
  for (int t = 0; t < T; ++t) mn = fminf(mn, p[(size_t)t * H]);
  float best = -INFINITY;
  for (int t = 0; t < T; ++t) {
    const int id = ids[b * T + t];
    const float m = (id >= 0 && id < vocab) ? 1.0f : 0.0f;
    best = fmaxf(best, (p[(size_t)t * H] - mn) * m);
  }
  float y = fmaxf(best + mn, 0.0f);
  if (keepmask) y = keepmask[(size_t)b * H + h] ? y * inv_keep : 0.0f;
  hidden[(size_t)b * H + h] = y;
}

// Gradient of the above w.r.t. pre, with TensorFlow's tie rules (reduce_max / reduce_min split
// the gradient equally among tied extrema): with z_t = (x_t - m) mu_t, y = max_t z_t + m,
//   dy/dx_s = mu_s [z_s == max z] / n_max + ([x_s == m] / n_min) (1 - sum_t mu_t [z_t == max z] / n_max)
__global__ __launch_bounds__(256) void text_pool_bwd_kernel(
    const float* __restrict__ dhidden, const float* __restrict__ pre,
    const int32_t* __restrict__ ids, int T, int H, int vocab, const uint8_t* __restrict__ keepmask,
    float inv_keep, float* __restrict__ dpre) {
  const int b = blockIdx.y;
  const int h = blockIdx.x * blockDim.x + threadIdx.x;
  if (h >= H) return;
  const float* p = pre + (size_t)b * T * H + h;
  float* d = dpre + (size_t)b * T * H + h;
  float mn = INFINITY;
  for (int t = 0; t < T; ++t) mn = fminf(mn, p[(size_t)t * H]);
  float zmax = -INFINITY;
  int nmin = 0;
  for (int t = 0; t < T; ++t) {
    const int id = ids[b * T + t];
    const float m = (id >= 0 && id < vocab) ? 1.0f : 0.0f;
    const float x = p[(size_t)t * H];
    zmax = fmaxf(zmax, (x - mn) * m);
    nmin += x == mn ? 1 : 0;
  }
  int nmax = 0;
  float smu = 0.f;
  for (int t = 0; t < T; ++t) {
    const int id = ids[b * T + t];
    const float m = (id >= 0 && id < vocab) ? 1.0f : 0.0f;
    if ((p[(size_t)t * H] - mn) * m == zmax) { ++nmax; smu += m; }
  }
  float g = dhidden[(size_t)b * H + h];
  if (keepmask) g = keepmask[(size_t)b * H + h] ? g * inv_keep : 0.0f;
  if (!(zmax + mn > 0.0f)) g = 0.0f;                       // relu'(y): y > 0
  const float via_min = g * (1.0f - smu / (float)nmax) / (float)nmin;
  for (int t = 0; t < T; ++t) {
    const int id = ids[b * T + t];
    const float m = (id >= 0 && id < vocab) ? 1.0f : 0.0f;
    const float x = p[(size_t)t * H];
    float v = 0.f;
    if ((x - mn) * m == zmax) v += g * m / (float)nmax;
    if (x == mn) v += via_min;
    d[(size_t)t * H] = v;
  }
}

// labels = any(exact > 0) ? exact : (sigmoid(logits) > thr)
__global__ __launch_bounds__(256) void text_labels_merge_kernel(const float* __restrict__ logits,
                                                                const float* __restrict__ exact,
                                                                float thr, int C,
                                                                float* __restrict__ labels) {
  __shared__ int any_exact;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) any_exact = 0;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    if (exact[b * C + c] > 0.f) any_exact = 1;
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float ml = sigmoidf(logits[b * C + c]) > thr ? 1.0f : 0.0f;
    labels[b * C + c] = any_exact ? exact[b * C + c] : ml;
  }
}


// WordVectorMatchExtractor (models/label_extractor.py:232-328): cosine similarity between every
// caption token and every class name in GloVe space, masked max-pool over the non-OOV tokens
// (core/utils.py:63-79 formula), one-hot of the best class; overridden by the exact match.
// One block per caption; sim[T][C] lives in LDS.
__global__ __launch_bounds__(256) void word_vector_match_kernel(
    const int32_t* __restrict__ ids, int T, const float* __restrict__ emb, int vocab, int E,
    const int32_t* __restrict__ class_ids, int C, const float* __restrict__ exact,
    float* __restrict__ labels) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sim = smem;                 // [T][C]
  float* tinv = smem + T * C;        // [T] 1/||token||
  float* cinv = tinv + T;            // [C] 1/||class||
  float* pooled = cinv + C;          // [C]
  __shared__ int any_token, any_exact, best_c;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) { any_token = 0; any_exact = 0; }
  __syncthreads();
  // tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12))
  for (int i = threadIdx.x; i < T + C; i += blockDim.x) {
    int id = i < T ? ids[b * T + i] : class_ids[i - T];
    if (id < 0 || id > vocab) id = vocab;
    const float* v = emb + (size_t)id * E;
    float ss = 0.f;
    for (int e = 0; e < E; ++e) ss += v[e] * v[e];
    const float inv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
    if (i < T) { tinv[i] = inv; if (id != vocab) any_token = 1; }
    else cinv[i - T] = inv;
  }
  for (int c = threadIdx.x; c < C; c += blockDim.x)
    if (exact[b * C + c] > 0.f) any_exact = 1;
  __syncthreads();
  for (int i = threadIdx.x; i < T * C; i += blockDim.x) {
    const int t = i / C, c = i - t * C;
    int id = ids[b * T + t];
    if (id < 0 || id > vocab) id = vocab;
    int cid = class_ids[c];
    if (cid < 0 || cid > vocab) cid = vocab;
    const float* tv = emb + (size_t)id * E;
    const float* cv = emb + (size_t)cid * E;
    float d = 0.f;
    for (int e = 0; e < E; ++e) d += (tv[e] * tinv[t]) * (cv[e] * cinv[c]);
    sim[i] = d;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float mn = INFINITY;
    for (int t = 0; t < T; ++t) mn = fminf(mn, sim[t * C + c]);
    float best = -INFINITY;
    for (int t = 0; t < T; ++t) {
      int id = ids[b * T + t];
      const float m = (id >= 0 && id < vocab) ? 1.0f : 0.0f;
      best = fmaxf(best, (sim[t * C + c] - mn) * m);
    }
    pooled[c] = best + mn;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int bc = 0;
    float bv = pooled[0];
    for (int c = 1; c < C; ++c)
      if (pooled[c] > bv) { bv = pooled[c]; bc = c; }   // tf.argmax: first maximum
    best_c = bc;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float ms = (any_token && c == best_c) ? 1.0f : 0.0f;
    labels[b * C + c] = any_exact ? exact[b * C + c] : ms;
  }
}

}  // namespace

extern "C" int c2d_midn_fwd(const float* logits, int ld, int off_r, int off_c,
                            const int32_t* num_proposals, float* proba, float* class_logits,
                            float* scores, int batch, int n, int num_classes, void* stream) {
  C2D_CHECK_ARG(logits && num_proposals && proba && class_logits && scores);
  C2D_CHECK_ARG(batch > 0 && n > 0 && num_classes > 0);
  hipLaunchKernelGGL(midn_fwd_kernel, dim3(num_classes, batch), dim3(256), 0,
                     (hipStream_t)stream, logits, ld, off_r, off_c, num_proposals, proba,
                     class_logits, scores, n, num_classes);
  return c2d_launch_status();
}

extern "C" int c2d_midn_bwd(const float* dclass_logits, const float* logits, int ld, int off_r,
                            int off_c, const int32_t* num_proposals, const float* proba,
                            const float* class_logits, float* dlogits, int lddl, int batch,
                            int n, int num_classes, void* stream) {
  C2D_CHECK_ARG(dclass_logits && logits && num_proposals && proba && class_logits && dlogits);
  C2D_CHECK_ARG(batch > 0 && n > 0 && num_classes > 0);
  hipLaunchKernelGGL(midn_bwd_kernel, dim3(num_classes, batch), dim3(256), 0,
                     (hipStream_t)stream, dclass_logits, logits, ld, off_r, off_c,
                     num_proposals, proba, class_logits, dlogits, lddl, n, num_classes);
  return c2d_launch_status();
}

extern "C" int c2d_sigmoid_ce_fwd_bwd(const float* logits, const float* labels, int n,
                                      float weight, float* loss, float* dlogits, void* stream) {
  C2D_CHECK_ARG(logits && labels && n > 0);
  hipLaunchKernelGGL(sigmoid_ce_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits,
                     labels, n, weight, loss, dlogits);
  return c2d_launch_status();
}

extern "C" int c2d_oicr_select(const float* s0, int ld, int off, const int32_t* num_proposals,
                               const float* boxes, int32_t* idx, float* top_boxes, int batch,
                               int n, int num_classes, void* stream) {
  C2D_CHECK_ARG(s0 && num_proposals && boxes && idx && top_boxes);
  C2D_CHECK_ARG(batch > 0 && n > 0 && num_classes > 0);
  hipLaunchKernelGGL(oicr_select_kernel, dim3(num_classes, batch), dim3(256), 0,
                     (hipStream_t)stream, s0, ld, off, num_proposals, boxes, idx, top_boxes, n,
                     num_classes);
  return c2d_launch_status();
}

extern "C" int c2d_oicr_loss_fwd_bwd(const float* scores, int ld, int off,
                                     const float* top_boxes, const float* boxes,
                                     const float* labels, const int32_t* num_proposals,
                                     float iou_threshold, float weight, int batch, int n,
                                     int num_classes, float* loss, float* dscores, int lddl,
                                     int doff, float* softmax_out, void* stream) {
  C2D_CHECK_ARG(scores && top_boxes && boxes && labels && num_proposals);
  C2D_CHECK_ARG(batch > 0 && n > 0 && num_classes > 0);
  const long long rows = (long long)batch * n;
  hipLaunchKernelGGL(oicr_loss_kernel, dim3(c2d_ceil_div(rows, 4)), dim3(256), 0,
                     (hipStream_t)stream, scores, ld, off, top_boxes, boxes, labels,
                     num_proposals, iou_threshold, weight, batch, n, num_classes, loss, dscores,
                     lddl, doff, softmax_out);
  return c2d_launch_status();
}

extern "C" int c2d_oicr_refine_fwd_bwd(const float* scores, int ld, int off, int stages,
                                       const float* s0, int s0_ld, int s0_off,
                                       const float* boxes, const float* labels,
                                       const int32_t* num_proposals, float iou_threshold,
                                       float weight, int batch, int n, int num_classes,
                                       float* loss, float* dscores, int lddl, int doff,
                                       float* softmax_out, int32_t* idx, float* top_boxes,
                                       void* stream) {
  C2D_CHECK_ARG(scores && s0 && boxes && labels && num_proposals && softmax_out && idx && top_boxes);
  C2D_CHECK_ARG(batch > 0 && n > 0 && num_classes > 0 && stages > 0);
  const long long rows = (long long)batch * n;
  hipStream_t st = (hipStream_t)stream;
  if (stages > 1)
    hipLaunchKernelGGL(oicr_softmax_kernel, dim3(c2d_ceil_div(rows, 4), stages - 1), dim3(256), 0, st,
                       scores, ld, off, batch, n, num_classes, softmax_out);
  hipLaunchKernelGGL(oicr_select_multi_kernel, dim3(num_classes, batch, stages), dim3(256), 0, st,
                     s0, s0_ld, s0_off, softmax_out, num_proposals, boxes, idx, top_boxes, batch, n,
                     num_classes);
  hipLaunchKernelGGL(oicr_loss_multi_kernel, dim3(c2d_ceil_div(rows, 4), stages), dim3(256), 0, st,
                     scores, ld, off, top_boxes, boxes, labels, num_proposals, iou_threshold, weight,
                     batch, n, num_classes, loss, dscores, lddl, doff, softmax_out);
  return c2d_launch_status();
}

extern "C" int c2d_labels_from_ids(const int32_t* ids, int batch, int num_tokens,
                                   int num_classes, float* labels, void* stream) {
  C2D_CHECK_ARG(labels && batch > 0 && num_classes > 0 && num_tokens >= 0);
  C2D_CHECK_ARG(ids || num_tokens == 0);
  hipLaunchKernelGGL(labels_from_ids_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream,
                     ids, num_tokens, num_classes, labels);
  return c2d_launch_status();
}

extern "C" int c2d_text_classifier_fwd(const int32_t* ids, int batch, int num_tokens,
                                       const float* embedding, int vocab_size, int emb_dims,
                                       const float* w1, const float* b1, int hidden_units,
                                       const float* w2, const float* b2, int num_classes,
                                       const float* exact_labels, float label_threshold,
                                       float* logits, float* labels, void* stream) {
  C2D_CHECK_ARG(ids && embedding && w1 && b1 && w2 && b2 && logits);
  C2D_CHECK_ARG(batch > 0 && num_tokens > 0 && vocab_size > 0 && emb_dims > 0);
  C2D_CHECK_ARG(hidden_units > 0 && num_classes > 0);
  C2D_CHECK_ARG(!labels || exact_labels);
  const int hspan = hidden_units > 1024 ? 1024 : ((hidden_units + 63) / 64) * 64;
  int nslice = 1024 / hspan;
  if (nslice < 1) nslice = 1;
  if (nslice > TXT_TC) nslice = TXT_TC;
  const size_t smem = ((size_t)hidden_units * (1 + nslice) + (size_t)TXT_TC * ((emb_dims + 3) & ~3)) *
                          sizeof(float) + TXT_TC * sizeof(int);
  if (smem > 64 * 1024) return C2D_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(text_classifier_kernel, dim3(batch), dim3(1024), smem, (hipStream_t)stream,
                     ids, num_tokens, embedding, vocab_size, emb_dims, w1, b1, hidden_units, w2,
                     b2, num_classes, logits);
  if (labels)
    hipLaunchKernelGGL(text_labels_merge_kernel, dim3(batch), dim3(256), 0, (hipStream_t)stream,
                       logits, exact_labels, label_threshold, num_classes, labels);
  return c2d_launch_status();
}

extern "C" long long c2d_text_classifier_workspace_bytes(int batch, int hidden_units) {
  if (batch <= 0 || hidden_units <= 0) return -1;
  return (long long)batch * hidden_units * (long long)sizeof(float);
}

extern "C" int c2d_text_classifier_fwd_ws(const int32_t* ids, int batch, int num_tokens,
                                          const float* embedding, int vocab_size, int emb_dims,
                                          const float* w1, const float* b1, int hidden_units,
                                          const float* w2, const float* b2, int num_classes,
                                          const float* exact_labels, float label_threshold,
                                          float* logits, float* labels, void* workspace,
                                          long long workspace_bytes, void* stream) {
  C2D_CHECK_ARG(ids && embedding && w1 && b1 && w2 && b2 && logits && workspace);
  C2D_CHECK_ARG(batch > 0 && num_tokens > 0 && vocab_size > 0 && emb_dims > 0);
  C2D_CHECK_ARG(hidden_units > 0 && num_classes > 0);
  C2D_CHECK_ARG(!labels || exact_labels);
  if (workspace_bytes < c2d_text_classifier_workspace_bytes(batch, hidden_units))
    return C2D_ERR_WORKSPACE;
  const size_t smem = ((size_t)emb_dims * TXT_HB + (size_t)TXT_TC * ((emb_dims + 3) & ~3) +
                       4 * TXT_HB) * sizeof(float) + TXT_TC * sizeof(int);
  if (smem > 150 * 1024 || (size_t)hidden_units * sizeof(float) > 60 * 1024)
    return C2D_ERR_UNSUPPORTED;
  static const hipError_t attr = hipFuncSetAttribute(
      (const void*)text_hidden_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  (void)attr;
  float* hidden = (float*)workspace;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(text_hidden_kernel, dim3(c2d_ceil_div(hidden_units, TXT_HB), batch),
                     dim3(256), smem, st, ids, num_tokens, embedding, vocab_size, emb_dims, w1,
                     b1, hidden_units, hidden);
  hipLaunchKernelGGL(text_logits_kernel, dim3(batch), dim3(256),
                     (size_t)hidden_units * sizeof(float), st, hidden, hidden_units, w2, b2,
                     num_classes, exact_labels, label_threshold, logits, labels);
  return c2d_launch_status();
}

extern "C" int c2d_word_vector_match_fwd(const int32_t* ids, int batch, int num_tokens,
                                         const float* embedding, int vocab_size, int emb_dims,
                                         const int32_t* class_ids, int num_classes,
                                         const float* exact_labels, float* labels,
                                         void* stream) {
  C2D_CHECK_ARG(ids && embedding && class_ids && exact_labels && labels);
  C2D_CHECK_ARG(batch > 0 && num_tokens > 0 && vocab_size > 0 && emb_dims > 0 && num_classes > 0);
  const size_t smem =
      ((size_t)num_tokens * num_classes + num_tokens + 2 * (size_t)num_classes) * sizeof(float);
  if (smem > 64 * 1024) return C2D_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(word_vector_match_kernel, dim3(batch), dim3(256), smem, (hipStream_t)stream,
                     ids, num_tokens, embedding, vocab_size, emb_dims, class_ids, num_classes,
                     exact_labels, labels);
  return c2d_launch_status();
}

extern "C" int c2d_embedding_gather(const int32_t* ids, long long rows, const float* embedding,
                                    int vocab_size, int emb_dims, int ld, float* x, void* stream) {
  C2D_CHECK_ARG(ids && embedding && x && rows > 0 && vocab_size > 0 && emb_dims > 0 &&
                ld >= emb_dims);
  long long blocks = (rows * ld + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(embedding_gather_kernel, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream,
                     ids, rows, embedding, vocab_size, emb_dims, ld, x);
  return c2d_launch_status();
}

extern "C" int c2d_text_pool_fwd(const float* pre, const int32_t* ids, int batch, int num_tokens,
                                 int hidden_units, int vocab_size, const uint8_t* keep_mask,
                                 float keep_prob, float* hidden, void* stream) {
  C2D_CHECK_ARG(pre && ids && hidden && batch > 0 && num_tokens > 0 && hidden_units > 0);
  C2D_CHECK_ARG(vocab_size > 0 && keep_prob > 0.0f);
  hipLaunchKernelGGL(text_pool_fwd_kernel, dim3(c2d_ceil_div(hidden_units, 256), batch), dim3(256),
                     0, (hipStream_t)stream, pre, ids, num_tokens, hidden_units, vocab_size,
                     keep_mask, 1.0f / keep_prob, hidden);
  return c2d_launch_status();
}

extern "C" int c2d_text_pool_bwd(const float* dhidden, const float* pre, const int32_t* ids,
                                 int batch, int num_tokens, int hidden_units, int vocab_size,
                                 const uint8_t* keep_mask, float keep_prob, float* dpre,
                                 void* stream) {
  C2D_CHECK_ARG(dhidden && pre && ids && dpre && batch > 0 && num_tokens > 0 && hidden_units > 0);
  C2D_CHECK_ARG(vocab_size > 0 && keep_prob > 0.0f);
  hipLaunchKernelGGL(text_pool_bwd_kernel, dim3(c2d_ceil_div(hidden_units, 256), batch), dim3(256),
                     0, (hipStream_t)stream, dhidden, pre, ids, num_tokens, hidden_units,
                     vocab_size, keep_mask, 1.0f / keep_prob, dpre);
  return c2d_launch_status();
}
