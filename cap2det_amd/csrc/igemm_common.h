// Shared declarations of the implicit-GEMM convolution kernels (conv_gemm.hip: fp32 kernels, the
// dispatch and the C-ABI entry points; igemm_bf16.hip: the bf16 ring kernel).  gfx950 only.
#pragma once
#include "c2d_common.h"

namespace c2d_ig {

typedef float f32x16 __attribute__((ext_vector_type(16)));
// Native vector type for register staging: HIP's float4 struct is copied with memcpy, which
// keeps staged arrays in scratch memory (private segment) instead of VGPRs.
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;                // floats of K per LDS slab (one 128-B line per row)

struct ConvGeom {
  int ih, iw;      // conv input spatial size
  int oh, ow;      // conv output spatial size
  int kh, kw;      // taps
  int stride;      // 1 or 2
  int pad_t, pad_l;
  int mode;        // 0: rows index conv OUTPUT pixels (forward / wgrad); 1: rows index conv
                   // INPUT pixels and src() yields OUTPUT pixels (dgrad)
  unsigned long long magic_hw, magic_w;  // ceil(2^40 / (h*w)), ceil(2^40 / w) of the ROW space
  int rh, rw;      // row-space extent: (oh, ow) in mode 0; in mode 1 the sub-grid of input
                   // pixels (y0 + sub*yy, x0 + sub*xx) handled by this launch
  int sub, y0, x0; // mode 1 only: stride-2 dgrad is split into the 4 parity classes of the
                   // input pixel, each of which sees only the taps of matching parity
  int ky0, kx0, kstep, nky, nkx;  // tap subset: ky = ky0 + kstep*t, t < nky (same for kx)
  int nimg;        // images (ROIs) in the batch
  // Pixel-major launches whose blocks are one pixel each (pm = 7, 128-row blocks): the blocks of
  // an interior pixel visit nine taps, those of a corner four.  lpt_ngx > 0: block ids are dealt
  // HEAVY PIXELS FIRST inside every XCD (longest-processing-time-first: the dispatcher hands blocks
  // to free slots in id order, so the long blocks start first and the short ones fill the tail);
  // lpt_ngx = image groups per XCD, px_order = the pixels by falling tap count (pm_tile_order).
  int lpt_ngx;
  unsigned char px_order[64];
  int pm;          // PIXEL-MAJOR row order (small maps, see decompose<true>): 0 = off, else log2 of
                   // the image-group size (5: groups of 32 images, 7: groups of 128)
};

// Row m of the iteration space -> (image, y, x).  Exact for m * d < 2^40 (always here).
struct RowPos {
  int img, y, x;
  bool valid;
};

// PM (pixel-major, used for the 3x3 convolutions over the tiny per-ROI maps): rows are ordered
// (group of G images, pixel, image in group), m = ((grp * rh*rw) + pixel) * G + r, G = 2^g.pm, so
// every aligned 32-row MFMA tile holds ONE pixel position of 32 images.  Whether a tap falls into
// the SAME padding is then uniform over the tile and its MFMAs are skipped instead of multiplying
// zeros (4x4 map: 100 of 144 (pixel, tap) pairs are real; 7x7: 361 of 441).  Round 4: G = 128 = the
// block's rows wherever the image count allows, so ALL of a block's tiles are one pixel: a tap is
// then real or padding for the whole block, padding taps are never visited and no slab is staged
// for a tap only some of the tiles use (with G = 32 a block of four pixels visited 7.5 taps on
// average on a 4x4 map for the 6.25 its MFMAs needed: the matrix pipe of the pixel-major launches
// was busy 0.61 of the time against 0.64-0.79 for the row-major ones, profiles/r04 per-kernel).
template <bool PM = false>
__device__ __forceinline__ RowPos decompose(int m, int M, const ConvGeom& g) {
  RowPos p;
  if (PM) {
    const unsigned sh = (unsigned)g.pm;
    const unsigned t = (unsigned)m >> sh, r = (unsigned)m & ((1u << sh) - 1u);
    const unsigned grp = (unsigned)(((unsigned long long)t * g.magic_hw) >> 40);
    const unsigned px = t - grp * (unsigned)(g.rh * g.rw);
    p.img = (int)((grp << sh) + r);
    p.valid = m < M && p.img < g.nimg;
    p.y = (int)(((unsigned long long)px * g.magic_w) >> 40);
    p.x = (int)px - p.y * g.rw;
    return p;
  }
  p.valid = m < M;
  const unsigned mm = p.valid ? (unsigned)m : 0u;
  p.img = (int)(((unsigned long long)mm * g.magic_hw) >> 40);
  const unsigned r = mm - (unsigned)p.img * (unsigned)(g.rh * g.rw);
  p.y = (int)(((unsigned long long)r * g.magic_w) >> 40);
  p.x = (int)r - p.y * g.rw;
  return p;
}

// Does the aligned 32-row MFMA tile that starts at row m hold any real row?  (PM: the images of
// the last group beyond nimg are padding — with groups of 128 that can be whole tiles.)
template <bool PM>
__device__ __forceinline__ bool tile_has_rows(int m, int M, const ConvGeom& g) {
  if (m >= M) return false;
  if (!PM) return true;
  const unsigned sh = (unsigned)g.pm;
  const unsigned t = (unsigned)m >> sh;
  const unsigned grp = (unsigned)(((unsigned long long)t * g.magic_hw) >> 40);
  return (int)((grp << sh) + ((unsigned)m & ((1u << sh) - 1u))) < g.nimg;
}

// Source row (in the A operand's row space) for iteration row `p` and tap (ky,kx); -1 if none.
// Branch-free (bitwise predicates) so that the loads that follow can be issued back to back.
template <int MODE>
__device__ __forceinline__ int src_row(const ConvGeom& g, const RowPos& p, int ky, int kx) {
  if (MODE == 0) {
    const int iy = p.y * g.stride - g.pad_t + ky;
    const int ix = p.x * g.stride - g.pad_l + kx;
    const int ok = (int)p.valid & (int)(iy >= 0) & (int)(iy < g.ih) & (int)(ix >= 0) &
                   (int)(ix < g.iw);
    const int row = (p.img * g.ih + iy) * g.iw + ix;
    return ok ? row : -1;
  } else {
    const int ty = p.y * g.sub + g.y0 + g.pad_t - ky;
    const int tx = p.x * g.sub + g.x0 + g.pad_l - kx;
    const int sh = g.stride - 1;  // stride is 1 or 2
    const int oy = ty >> sh, ox = tx >> sh;
    const int ok = (int)p.valid & (int)(ty >= 0) & (int)(tx >= 0) &
                   (int)(((ty | tx) & sh) == 0) & (int)(oy < g.oh) & (int)(ox < g.ow);
    const int row = (p.img * g.oh + oy) * g.ow + ox;
    return ok ? row : -1;
  }
}

// Does tap (ky,kx) of a row at pixel (y,x) of the row space read a real pixel?  (the image-
// independent part of src_row)
template <int MODE>
__host__ __device__ __forceinline__ bool tap_ok(const ConvGeom& g, int y, int x, int ky, int kx) {
  if (MODE == 0) {
    const int iy = y * g.stride - g.pad_t + ky;
    const int ix = x * g.stride - g.pad_l + kx;
    return iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw;
  } else {
    const int ty = y * g.sub + g.y0 + g.pad_t - ky;
    const int tx = x * g.sub + g.x0 + g.pad_l - kx;
    const int sh = g.stride - 1;
    return ty >= 0 && tx >= 0 && ((ty | tx) & sh) == 0 && (ty >> sh) < g.oh && (tx >> sh) < g.ow;
  }
}

// Raw buffer descriptor over [p, p + bytes) built from wave-uniform inputs (readfirstlane makes
// that provable to the compiler: no waterfall loop around the loads, cdna_hip_programming.md T20).
constexpr unsigned OOB_OFFSET = 0x7FFFFF00u;   // >= any buffer size here: the load returns zeros
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* p, long long bytes) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  const unsigned nb = __builtin_amdgcn_readfirstlane(
      (unsigned)(bytes < (long long)OOB_OFFSET ? bytes : (long long)OOB_OFFSET));
  return __builtin_amdgcn_make_buffer_rsrc(
      (void*)(((unsigned long long)hi << 32) | lo), (short)0, (int)nb, 0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc_b(const void* p, long long bytes) {
  return make_rsrc((const float*)p, bytes);
}
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0);
  return __builtin_bit_cast(f32x4, v);
}

// four consecutive operand elements (fp32: one 16-B load; bf16: one 8-B load, widened) as fp32
template <int ES>
__device__ __forceinline__ f32x4 buf_load_elems4(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  if constexpr (ES == 4) {
    return buf_load4(rs, voff, soff);
  } else {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)voff, soff, 0);
    f32x4 o;   // bf16 -> fp32 = the 16 bits moved to the top half
    o.x = __uint_as_float(v.x << 16); o.y = __uint_as_float(v.x & 0xffff0000u);
    o.z = __uint_as_float(v.y << 16); o.w = __uint_as_float(v.y & 0xffff0000u);
    return o;
  }
}

__device__ __forceinline__ f32x4 mask4(f32x4 v, bool keep) {
  v.x = keep ? v.x : 0.0f; v.y = keep ? v.y : 0.0f;
  v.z = keep ? v.z : 0.0f; v.w = keep ? v.w : 0.0f;
  return v;
}

struct IgemmArgs {
  const float* A; int lda; int a_off;
  long long a_rows;         // rows of the A operand's buffer (for the buffer-load range check)
  const float* Bt;          // [taps][N][K]
  // f32x9 launches (igemm_x9.hip): the weights as three bf16 planes — plane p of the element at Bt[i]
  // is ((const bf16*)Bp)[i] bp_stride BYTES further per plane; segBp[s] likewise for segB[s]
  const void* Bp; long long bp_stride; const void* segBp[4];
  float* C; int ldc; int c_off;
  const float* scale;       // [N] or null (=1)
  const float* shift;       // [N] or null (=0)
  int relu;
  int accumulate;           // C += result
  int M, N, K;
  int m_tiles, n_tiles;
  // Multi-segment 1x1 mode (nseg > 1): the reduction runs over the concatenation of `nseg`
  // (A_s [rows][lda_s] (+off_s), Bt_s [N][K_s]) pairs — one GEMM for the input gradient of an
  // Inception block whose branches all start with a 1x1 convolution of the same input.
  int nseg;
  const float* segA[4]; const float* segB[4];
  int seg_lda[4], seg_off[4], segK[4];   // (every segment's A has a_rows rows)
  int total_slabs;
  int es;                   // operand / output element size: 4 (fp32) or 2 (bf16)
  // Fused BatchNorm/ReLU backward of the layer that PRODUCED this convolution's input (input-
  // gradient launches only, fy != null): the epilogue turns dx into dc = dx * (y > 0) * fscale
  // and block (m-tile) mt stores the column sums of dz = dx * (y > 0) and dz * (y - beta) / gamma
  // over its rows at fpart[(fpart_row0 + mt) * 2 * N ..] (layout of c2d_bn_relu_bwd_partial).
  // The columns may belong to up to four producers (the branches feeding a concat buffer):
  // producer p owns columns [fseg_end[p-1], fseg_end[p]), its vectors are indexed from its first
  // column; fident[p]: a pooling branch (no BN/ReLU: the gradient passes unchanged, sums zero).
  const void* fy; int fldy, fyoff;
  int fnprod; int fseg_end[4]; int fident[4];
  const float* fscale[4]; const float* fbeta[4]; const float* fgamma[4];
  float* fpart; int fpart_row0;
  // Several 1x1 convolutions of the SAME input as one GEMM (forward launches only, mo_n > 0):
  // output columns [mo_end[s-1], mo_end[s]) belong to convolution s — its weight rows start
  // mo_boff[s] elements behind Bt (all of them inside mo_bbytes bytes), its BN scale / shift
  // vectors, destination (pointer, row stride, column offset) and ReLU flag are its own.
  int mo_n; int mo_end[4]; int mo_relu[4];
  long long mo_boff[4]; long long mo_bbytes;
  float* mo_C[4]; int mo_ldc[4], mo_coff[4];
  const float* mo_scale[4]; const float* mo_shift[4];
  int dbg;                  // ablation bits (C2D_TUNE=igemm_dbg; ring kernel): 2 no weight DMA, 4 no MFMA, 8 no epilogue, 16 no B fragment reads, 64 no DMA after the prologue
  ConvGeom g;
  unsigned long long* trace;   // diagnostic build (C2D_TRACE) only: 8 x u64 per block (tools/trace_igemm.py)
};


// XCD-aware tile order (cdna_hip_programming.md T1, bijective form): consecutive logical tiles
// run on one XCD, and the n-tiles of one m-tile are consecutive, so the A rows they share are
// served by that XCD's L2.
__device__ __forceinline__ int xcd_remap(int id, int total) {
  const int q = total >> 3, r = total & 7;
  const int xcd = id & 7, local = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

// Block id -> (m-tile, n-tile).  Default: xcd_remap (tile order, n-tiles of an m-tile adjacent).
// Heavy-first pixel-major launches (ConvGeom::lpt_ngx): XCD x owns the image groups
// [x * ngx, (x + 1) * ngx); inside it the blocks run pixel-class-major (all groups' blocks of the
// heaviest pixel first), the n-tiles of an m-tile still adjacent.
__device__ __forceinline__ void block_tile(const ConvGeom& g, int n_tiles, int id, int total, int* mt,
                                           int* nt) {
  if (g.lpt_ngx > 0) {
    const int xcd = id & 7, local = id >> 3;
    const int t = local / n_tiles;
    *nt = local - t * n_tiles;
    const int j = t / g.lpt_ngx, gl = t - j * g.lpt_ngx;
    *mt = (xcd * g.lpt_ngx + gl) * (g.rh * g.rw) + (int)g.px_order[j];
    return;
  }
  const int lb = xcd_remap(id, total);
  *mt = lb / n_tiles;
  *nt = lb - *mt * n_tiles;
}

// Stream-K plan (host-computed, passed by value).  The iteration space of a launch is the
// concatenation, in tile order (m-tile major, n-tile minor), of every tile's slab iterations
// (cost = visited taps x K slabs; PM tiles on the map border visit fewer taps).  `grid`
// persistent workgroups each take `share` consecutive iterations, so all of them finish together
// whatever the tile count is: no partial last round of tiles, no idle CUs at the end (the
// per-block timeline of the one-tile-per-block form showed 25-35 % of a launch spent in such a
// tail).  A tile cut by a share boundary is finished by whichever of its pieces arrives last
// (cdna_hip_programming.md §5, in-launch split-K recipe): every piece stores its fp32
// accumulators to its own slab, releases, and draws a ticket on the tile's counter; the piece
// that draws the last ticket acquires, sums all slabs IN PIECE ORDER (bitwise reproducible) and
// runs the epilogue.  Nobody ever waits, so no residency or dispatch-order assumption is made.
constexpr int SK_MAX_PERIOD = 64;
// ---- several 1x1 convolutions of one input as one GEMM (IgemmArgs::mo_n) ---------------------
// element offset (relative to Bt) of the weight row of output column n
__device__ __forceinline__ int mo_weight_row(const IgemmArgs& a, int n) {
  int s = 0, lo = 0;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (q + 1 < a.mo_n && n >= a.mo_end[q]) { s = q + 1; lo = a.mo_end[q]; }
  return (int)a.mo_boff[s] + (n - lo) * a.K;      // (host: every weight row within 2^31 elements of Bt)
}
struct MoOut { float* C; int ldc, coff, lo, relu; const float* scale; const float* shift; };
__device__ __forceinline__ MoOut mo_output(const IgemmArgs& a, int ncol) {
  int s = 0, lo = 0;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (q + 1 < a.mo_n && ncol >= a.mo_end[q]) { s = q + 1; lo = a.mo_end[q]; }
  return MoOut{a.mo_C[s], a.mo_ldc[s], a.mo_coff[s], lo, a.mo_relu[s], a.mo_scale[s], a.mo_shift[s]};
}

// ---- fused BN/ReLU backward in the input-gradient epilogue (IgemmArgs::fy) -------------------
template <int ES>
__device__ __forceinline__ f32x4 load_act4(const void* base, size_t idx) {
  if constexpr (ES == 4) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + idx);
  } else {
    const bf16x4 o = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(base) + idx);
    return f32x4{(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
  }
}
// the producer parameters of the four columns starting at ncol (all inside one producer)
__device__ __forceinline__ bool fused_bn_params(const IgemmArgs& a, int ncol, f32x4& sc, f32x4& be,
                                                f32x4& ig) {
  int p = 0, lo = 0;
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (q + 1 < a.fnprod && ncol >= a.fseg_end[q]) { p = q + 1; lo = a.fseg_end[q]; }
  if (a.fident[p]) return true;
  sc = *reinterpret_cast<const f32x4*>(a.fscale[p] + (ncol - lo));
  if (a.fgamma[p]) {
    be = *reinterpret_cast<const f32x4*>(a.fbeta[p] + (ncol - lo));
    const f32x4 ga = *reinterpret_cast<const f32x4*>(a.fgamma[p] + (ncol - lo));
    ig = f32x4{ga.x != 0.f ? 1.f / ga.x : 0.f, ga.y != 0.f ? 1.f / ga.y : 0.f,
               ga.z != 0.f ? 1.f / ga.z : 0.f, ga.w != 0.f ? 1.f / ga.w : 0.f};
  }
  return false;
}
// one epilogue item: v = four input-gradient values of a row, yv = the producer's outputs there
__device__ __forceinline__ f32x4 fused_bn_item(f32x4 v, f32x4 yv, f32x4 sc, f32x4 be, f32x4 ig,
                                               f32x4& sb, f32x4& sg) {
  f32x4 dz;
  dz.x = yv.x > 0.f ? v.x : 0.f; dz.y = yv.y > 0.f ? v.y : 0.f;
  dz.z = yv.z > 0.f ? v.z : 0.f; dz.w = yv.w > 0.f ? v.w : 0.f;
  sb += dz;
  sg.x += dz.x * (yv.x - be.x) * ig.x; sg.y += dz.y * (yv.y - be.y) * ig.y;
  sg.z += dz.z * (yv.z - be.z) * ig.z; sg.w += dz.w * (yv.w - be.w) * ig.w;
  return dz * sc;
}
// Column sums of a block: lane sums -> LDS -> fixed-order sums over the lanes / waves that share
// a column -> the block's row of the partials (no atomics: bitwise reproducible).
template <int WM, int WN, int SCOLS, int RPP>
__device__ __forceinline__ void fused_bn_finish(float* red, const float* dummy, float* fpart,
                                                int fpart_row0, int N, const f32x4& sb,
                                                const f32x4& sg, int tid, int wave, int ec4, int er,
                                                bool lane_on, int n0, int mt) {
  constexpr int BN = WN * SCOLS, NTHREADS = WM * WN * 64;
  (void)dummy;
  __syncthreads();            // every wave is done with its epilogue staging slice
  if (lane_on) {
    float* p = red + ((wave * RPP + er) * 2) * SCOLS + ec4 * 4;
    *reinterpret_cast<f32x4*>(p) = sb;
    *reinterpret_cast<f32x4*>(p + SCOLS) = sg;
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * BN; idx += NTHREADS) {
    const int k = idx / BN, c = idx - k * BN;
    const int wn_c = c / SCOLS, cl = c - wn_c * SCOLS;
    float t = 0.f;
    for (int wm = 0; wm < WM; ++wm)
      for (int e = 0; e < RPP; ++e) t += red[(((wm * WN + wn_c) * RPP + e) * 2 + k) * SCOLS + cl];
    if (n0 + c < N) fpart[((size_t)(fpart_row0 + mt) * 2 + k) * N + n0 + c] = t;
  }
}

// ---- filter gradients (per-tap TN GEMM over rows) -------------------------------------------
struct WgradArgs {
  const float* A; int lda; int a_off;   // activations x (rows of the conv input)
  long long a_rows;                     // rows of x
  const float* G; int ldg; int g_off;   // dC rows (conv output rows)
  float* dW;                            // [taps][I][J], pre-zeroed or accumulated into
  int M;                                // conv output rows (reduction length)
  int I, J;                             // cin, cout
  int rows_per_split;                   // multiple of WBK
  int tiles_x, tiles_y, nsplits;        // 1-D grid = tiles_x (tap, i-tile) * tiles_y (j-tile) * nsplits
  long long part_stride;                // > 0: split z stores (plain) into dW + z * part_stride
                                        // floats (its own slab) instead of adding atomically
  ConvGeom g;                           // mode 0
};

// Block -> (x, y, split) with all tiles of one row split consecutive on ONE XCD (xcd_remap): the
// x / dC rows of a split are then fetched into that XCD's L2 once and shared by its tiles.
struct WgradBlock { int x, y, z; };
__device__ __forceinline__ WgradBlock wgrad_block(const WgradArgs& a) {
  const int txy = a.tiles_x * a.tiles_y;
  const int logical = xcd_remap(blockIdx.x, txy * a.nsplits);
  WgradBlock b;
  b.z = logical / txy;
  const int t = logical - b.z * txy;
  b.y = t / a.tiles_x;
  b.x = t - b.y * a.tiles_x;
  return b;
}


// Several 1x1 / stride-1 filter gradients of ONE input in one launch (conv_gemm.hip:
// wgrad_tn_group_kernel, igemm_bf16.hip: wgrad1x1_bf16_ring_group_kernel).
constexpr int WGRAD_GROUP_MAX = 4;
struct WgradGroupArgs {
  WgradArgs a[WGRAD_GROUP_MAX];
  int first_tile[WGRAD_GROUP_MAX + 1];   // first tile of problem p inside a split; [num] = tiles per split
  int num, nsplits;
};
__device__ __forceinline__ WgradBlock wgrad_group_block(const WgradGroupArgs& g, int* p_out) {
  const int txy = g.first_tile[g.num];
  const int logical = xcd_remap(blockIdx.x, txy * g.nsplits);
  WgradBlock b;
  b.z = logical / txy;
  int t = logical - b.z * txy;
  int p = 0;
  for (int i = 1; i < g.num; ++i)
    if (t >= g.first_tile[i]) p = i;
  t -= g.first_tile[p];
  b.y = t / g.a[p].tiles_x;
  b.x = t - b.y * g.a[p].tiles_x;
  *p_out = p;
  return b;
}

// ---- cross-file plumbing -----------------------------------------------------------------------
// Dispatch record of the calling thread's last convolution entry point (conv_gemm.hip).
void dispatch_note_ext(const char* fmt, int a = 0, int b = 0, int c = 0, int d = 0, int e = 0,
                       int f = 0, int g = 0, int h = 0);
// bf16 ring kernel (igemm_bf16.hip): launches the instance for block tile (WM x MT x 32) x
// (WN x NT x 32) in `a.g.mode`, row order `pm`; fills m_tiles / n_tiles / total_slabs itself.
// *m_tiles_out = row blocks of the launch.  query: record only, no launch.
// Returns C2D_ERR_UNSUPPORTED when no instance exists for the tile (the caller keeps its own).
int launch_igemm_bf16_ring(const IgemmArgs& a, int wm, int wn, int mt, int nt, bool pm,
                           hipStream_t s, int* m_tiles_out, bool query);
// fp32 operands as nine bf16 partial products (igemm_x9.hip): as launch_igemm_bf16_ring for the 128 x
// (64 NT) tiles (wm = wn = mt = 2, nt = 1..4) and 64 x 64; a.Bp / a.bp_stride / a.segBp filled by the
// caller from x9_planes_of (null: the weight operand lies in no bound arena).
int launch_igemm_x9_ring(const IgemmArgs& a, int wm, int wn, int mt, int nt, bool pm, hipStream_t s,
                         int* m_tiles_out, bool query);
const void* x9_planes_of(const void* w, long long bytes, long long* stride_bytes);
#ifdef C2D_RING_TRACE
unsigned long long* ring_trace_buffer();     // diagnostic build: the buffer c2d_debug_set_ring_trace set
#endif
// Between ring_group_begin() and ring_group_end(s) the calling thread's launch_igemm_bf16_ring calls
// that pick a groupable instance (row-major bf16 input gradients on the 128x256 / 128x64 / 64x64
// tiles) are held back; ring_group_end launches them — ONE launch when they all picked the same
// instance (igemm_ring_group_kernel), one by one otherwise.  At most four independent problems.
void ring_group_begin();
int ring_group_end(hipStream_t s);
// The same ring on fp32 operands (v_mfma_f32_32x32x2_f32); C2D_ERR_UNSUPPORTED unless enabled / an
// instance exists for the tile.
int launch_igemm_f32_ring(const IgemmArgs& a, int wm, int wn, int mt, int nt, bool pm,
                          hipStream_t s, int* m_tiles_out, bool query);
// bf16 filter gradient of a 1x1 / stride-1 convolution (igemm_bf16.hip: wgrad1x1_bf16_ring_kernel):
// fills the tiling fields of `a` itself.  *splits_out = row splits (slabs of a.part_stride floats
// when a.part_stride > 0, else atomics into a.dW).  splits_only: compute the split count only.
int launch_wgrad1x1_bf16_ring(WgradArgs a, hipStream_t s, int* splits_out, bool splits_only);
// `num` (<= WGRAD_GROUP_MAX) 1x1 / stride-1 filter gradients of ONE input as one launch; fills the
// tiling fields of a[p] itself (atomics into a[p].dW).
int launch_wgrad1x1_bf16_ring_group(WgradArgs* a, int num, hipStream_t s);
// f32x9 (igemm_x9.hip): true when the process has bound weight planes and the switch is on; the
// filter gradient of a 1x1 / stride-1 convolution on fp32 operands as nine bf16 partial products
// (fills the tiling fields of `a` itself; atomics into a.dW; C2D_ERR_UNSUPPORTED: not this shape).
bool x9_active();
int launch_wgrad1x1_x9(WgradArgs a, hipStream_t s);
// the same for a 3x3 / SAME convolution over n maps of wcx x wcx input pixels: stride 1 (4 or 7) or 2 (7)
int launch_wgrad3x3_x9(WgradArgs a, int n, int wcx, int stride, hipStream_t s);

}  // namespace c2d_ig
