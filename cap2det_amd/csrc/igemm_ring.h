// The DMA-ring implicit-GEMM kernel body shared by igemm_bf16.hip (bf16 operands, and fp32 operands on the
// fp32 matrix pipe) and igemm_x9.hip (fp32 operands as nine bf16 partial products).  gfx950 only.
//
// BASELINE configs[2] / [4]: slim.conv2d of models/utils.py:165-167 and its input gradient).
//
// Same iteration space, tap tables, pixel-major tile skipping, multi-segment / multi-output modes
// and epilogue as round 2's igemm_bf16_kernel (which this kernel replaces), which staged ONE slab ahead: a
// workgroup then waits out the whole global -> LDS latency of a slab (~2,600 cycles under load)
// for 512-1,400 cycles of MFMAs per slab, and issues all of a slab's DMA pieces in front of its
// MFMAs (6-8 wave-instructions of ~100 issue cycles each, both waves of a SIMD at the same time
// behind the barrier).  Here
//   * operand stages go through a ring of D buffers: D - 2 stages stay in flight across the
//     barrier (counted s_waitcnt vmcnt, raw s_barrier — never __syncthreads() with a DMA pending,
//     cdna_hip_programming.md §5 "Pipelining across barriers");
//   * a stage is BKT = 64 or 32 elements of K deep (128- or 64-byte rows): the 32-deep form makes
//     a ring of three small enough for TWO workgroups per CU on the 128x256 tile.
// What the sweeps showed (launch_tile below): deep rings lose, co-resident workgroups win.
// LDS image of a stage: unpadded rows; 16-byte chunk c of row r sits at position c ^ swz(r) with
// swz(r) = (r >> 1) & 7 for 128-byte rows and (r >> 2) & 3 for 64-byte rows — applied to the DMA's
// per-lane GLOBAL address and to the fragment reads (an LDS-DMA image is lane-linear) — which puts
// the 16 lanes of every ds_read_b128 group on 16 distinct 16-byte slots (MI355X_MICROARCH.md §LDS).
#pragma once
#include "igemm_common.h"
#include <stdlib.h>
#include <type_traits>
#include <utility>

namespace c2d_ig {
namespace {

typedef __attribute__((address_space(3))) void lds_void_t;

struct SlabCursor {
  int tap, kc, sgi, Kc;
  unsigned long long taps_left;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int... Is, class F>
__device__ __forceinline__ void static_for_ring(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// Position swizzle of a staged row with CPR 16-byte chunks: chunk c of row r sits at c ^ ring_swz(r),
// which puts the 16 lanes of every ds_read_b128 group (16 consecutive rows, one chunk) on 16
// distinct 16-byte slots of a 256-byte bank period (128-, 64- and 32-byte rows).
template <int CPR>
__device__ __forceinline__ int ring_swz(int r) {
  static_assert(CPR == 8 || CPR == 4 || CPR == 2, "row bytes");
  return CPR == 8 ? (r >> 1) & 7 : CPR == 4 ? (r >> 2) & 3 : (r >> 3) & 1;
}

// Truncation split of eight fp32 values into three bf16 planes, x = hi + mid + lo EXACTLY (8 + 8 + 8
// significant bits; |x| below 2^-110 loses what bf16's subnormal range cannot hold, non-finite x
// gives a NaN mid plane): hi = the top half of the word, mid = the top half of x - hi (exact in
// fp32), lo = x - hi - mid (at most 8 significant bits left: its low half is zero).  44 single-issue
// vector instructions per fragment, placed in the gaps between the MFMAs they feed
// (MI355X_MICROARCH.md "vector-instruction ISSUE cost": five 4-cycle fillers hide per 32-cycle gap;
// packed f32 adds do not — the subtractions are kept scalar, see the Makefile's -fno-slp-vectorize).
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3_frag(const f32x4 c0, const f32x4 c1, bf16x8& hi, bf16x8& mid,
                                            bf16x8& lo) {
  float x[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
  float r1[8], r2[8];
  u32x4_t h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) r1[e] = x[e] - __uint_as_float(__float_as_uint(x[e]) & 0xffff0000u);
#pragma unroll
  for (int e = 0; e < 8; ++e) r2[e] = r1[e] - __uint_as_float(__float_as_uint(r1[e]) & 0xffff0000u);
#pragma unroll
  for (int q = 0; q < 4; ++q) {      // (low half = element 2q, high half = element 2q + 1)
    h[q] = __builtin_amdgcn_perm(__float_as_uint(x[2 * q + 1]), __float_as_uint(x[2 * q]), 0x07060302u);
    m[q] = __builtin_amdgcn_perm(__float_as_uint(r1[2 * q + 1]), __float_as_uint(r1[2 * q]), 0x07060302u);
    l[q] = __builtin_amdgcn_perm(__float_as_uint(r2[2 * q + 1]), __float_as_uint(r2[2 * q]), 0x07060302u);
  }
  hi = __builtin_bit_cast(bf16x8, h);
  mid = __builtin_bit_cast(bf16x8, m);
  lo = __builtin_bit_cast(bf16x8, l);
}

// largest divisor of n that is at most 8
constexpr int ring_pass_group(int n) {
  int g = 1;
  for (int d = 2; d <= 8; ++d)
    if (n % d == 0) g = d;
  return g;
}

constexpr int ring_blocks_per_cu(int lds_bytes) {
  return (160 * 1024) / lds_bytes >= 5 ? 5 : (160 * 1024) / lds_bytes;
}

// ES = operand / result element size: 2 = bf16 on v_mfma_f32_32x32x16_bf16; 4 = fp32 on
// v_mfma_f32_32x32x2_f32 (exact fp32; round 3: the same ring with 16- or 32-float stages, i.e. the
// same 64- / 128-byte rows — the register-staged igemm_nt_kernel of conv_gemm.hip spends 11 % of
// its launches on operand staging and needs a second barrier per slab).
// FUSED: the instance whose epilogue carries the producer layer's BN/ReLU backward (IgemmArgs::fy;
// input-gradient launches of bf16 networks).  Its own instantiation: the epilogue's column sums and
// per-column facts must not cost the plain kernels a register.
// (the body: `bid` of `nbid` blocks of the problem `a` — the launch's own block index, or the index
//  inside one problem of a grouped launch, igemm_ring_group_kernel below)
// BP = 3 (igemm_x9.hip, ES = 4): the weights come as THREE bf16 planes (hi / mid / lo of the
// truncation split x = hi + mid + lo, c2d_split3_bf16; plane p of a weight element sits a.bp_stride
// bytes behind plane p - 1), the fp32 activation rows are staged as they are and split by the lane
// that reads a fragment, and every k16 step issues the NINE partial products on
// v_mfma_f32_32x32x16_bf16 (288 matrix-pipe cycles against 512 for eight v_mfma_f32_32x32x2_f32).
template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED = false,
          int BP = 1>
__device__ __forceinline__ void igemm_ring_body(const IgemmArgs& a, const int bid, const int nbid) {
  constexpr bool X9 = BP == 3;
  static_assert(BP == 1 || (BP == 3 && ES == 4), "weight planes: fp32 activations only");
  constexpr int ESB = X9 ? 2 : ES;              // element size of the staged weights
  constexpr int RB = BKT * ES;                  // bytes per staged activation row (128 or 64)
  constexpr int CPR = RB / 16;                  // 16-byte chunks per row (8 or 4)
  constexpr int RBB = BKT * ESB;                // bytes per staged weight row (of one plane)
  constexpr int CPRB = RBB / 16;
  constexpr int KS = (ES == 2 || X9) ? BKT / 16 : CPR / 2;   // bf16 / x9: MFMA k-steps per stage; fp32:
                                                // 16-byte chunks per lane and stage (4 MFMAs each)
  constexpr int EPC = 16 / ES;                  // elements per 16-byte chunk
  constexpr int EPCB = 16 / ESB;
  constexpr int BM = WM * MT * 32;
  constexpr int BN = WN * NT * 32;
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int ROWS_PER_PASS = NTHREADS / CPR; // a wave-instruction stages 1 KiB = 1024 / RB rows
  constexpr int ROWS_PER_PASS_B = NTHREADS / CPRB;
  constexpr int A_LOADS = BM / ROWS_PER_PASS;
  // weight rows: whole passes, plus a last partial pass that only the first waves take part in
  // (32-deep stages of 8-wave blocks stage 128 rows per pass; BN = 192 / 320 leave half a pass)
  constexpr int B_LOADS1 = (BN + ROWS_PER_PASS_B - 1) / ROWS_PER_PASS_B;   // per plane
  constexpr int B_LOADS = BP * B_LOADS1;
  constexpr bool B_TAIL = BN % ROWS_PER_PASS_B != 0;
  constexpr int PER = A_LOADS + B_LOADS;        // DMA wave-instructions per stage and wave (PER - BP
                                                // for the waves outside a partial last pass)
  constexpr int A_BYTES = BM * RB, BP_BYTES = BN * RBB, B_BYTES = BP * BP_BYTES;
  constexpr int RING_BYTES = D * (A_BYTES + B_BYTES);
  // (the epilogue stages 16-row half strips of every wave through the same memory)
  constexpr int EPI_BYTES = WM * WN * 16 * (NT * 32 + 4) * 4 + (BM + 3 * BN) * 4;   // (+ its row / column tables)
  constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static_assert(BM % ROWS_PER_PASS == 0 && BN % (1024 / RBB) == 0, "tile vs block size");
  static_assert(D >= 2 && D <= 6 && (D - 2) * PER <= 63, "ring depth vs the 6-bit vmcnt");
  static_assert(LDS_BYTES <= 160 * 1024, "ring exceeds LDS");
  static_assert((RB == 128 || RB == 64) && (RBB == 128 || RBB == 64 || RBB == 32), "stage depth");
  __shared__ __attribute__((aligned(1024))) char smem[LDS_BYTES];
  char* const smemB = smem + D * A_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
#ifdef C2D_RING_TRACE
  // diagnostic build only (tools/trace_ring.py): per-block phase stamps of wave 0
  const unsigned long long tr0 = __builtin_amdgcn_s_memtime();
  const unsigned long long tr_rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long tr_wait = 0, tr_issue = 0;
#endif
  // DMA: this lane fetches the chunk that belongs at LDS position tid % CPR of its row
  const int lrow = tid / CPR;                                       // row inside a pass
  const int lswz = ring_swz<CPR>(lrow);                             // (pass rows are multiples of 16)
  const int kchunk = (tid % CPR) ^ lswz;
  const int q4 = kchunk * EPC;                                      // element offset inside the stage
  // (the same for the weight rows: other row bytes when they are bf16 planes)
  const int lrowB = tid / CPRB;
  const int q4B = ((tid % CPRB) ^ ring_swz<CPRB>(lrowB)) * EPCB;
  // does this wave hold a piece of the (partial) last pass of the weight rows?
  const bool b_last = !B_TAIL || wave * (1024 / RBB) + (B_LOADS1 - 1) * ROWS_PER_PASS_B < BN;

  int mt, nt;
  block_tile(a.g, a.n_tiles, bid, nbid, &mt, &nt);
  const int m0 = mt * BM, n0 = nt * BN;
  const int ntaps = a.g.nky * a.g.nkx;
  const int kslabs = (a.K + BKT - 1) / BKT;
  const size_t tap_stride = (size_t)a.N * a.K;

  RowPos apos[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) apos[i] = decompose<PM>(m0 + lrow + i * ROWS_PER_PASS, a.M, a.g);
  int brow_off[B_LOADS1];
#pragma unroll
  for (int i = 0; i < B_LOADS1; ++i) brow_off[i] = min(n0 + lrowB + i * ROWS_PER_PASS_B, a.N - 1);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  unsigned row_bits = 0;
#pragma unroll
  for (int i = 0; i < MT; ++i)
    if (tile_has_rows<PM>(m0 + (wm * MT + i) * 32, a.M, a.g)) row_bits |= 1u << i;
  row_bits = __builtin_amdgcn_readfirstlane(row_bits);

  // Tap table, one tap per lane (every wave holds all of it): what changes from tap to tap —
  // the row delta of the activation rows, the offset of the tap's weight plane and (PM) which
  // 32-row tiles of the block are real for the tap — is computed once.
  int tab_delta = 0, tab_toff = 0;
  unsigned tab_tv = 0;
  if (lane < ntaps) {
    const int ty_ = lane / a.g.nkx;
    const int ky = a.g.ky0 + a.g.kstep * ty_, kx = a.g.kx0 + a.g.kstep * (lane - ty_ * a.g.nkx);
    tab_toff = (ky * a.g.kw + kx) * (int)tap_stride;
    if (MODE == 0) {
      tab_delta = (ky - a.g.pad_t) * a.g.iw + (kx - a.g.pad_l);
    } else {
      const int sh = a.g.stride - 1;     // (stride-2 launches hold the taps of ONE parity class)
      tab_delta = ((a.g.y0 + a.g.pad_t - ky) >> sh) * a.g.ow + ((a.g.x0 + a.g.pad_l - kx) >> sh);
    }
    tab_tv = 0xffu;
    if (PM) {
      tab_tv = 0;
      const int hw = a.g.rh * a.g.rw;
#pragma unroll
      for (int tb = 0; tb < BM / 32; ++tb) {
        const unsigned t = (unsigned)(m0 + tb * 32) >> a.g.pm;
        const unsigned grp = (unsigned)(((unsigned long long)t * a.g.magic_hw) >> 40);
        const unsigned px = t - grp * (unsigned)hw;
        const int y = (int)(((unsigned long long)px * a.g.magic_w) >> 40);
        const int x = (int)px - y * a.g.rw;
        tab_tv |= (tap_ok<MODE>(a.g, y, x, ky, kx) ? 1u : 0u) << tb;
      }
    }
  }
  const unsigned long long tapmask = __ballot(lane < ntaps && tab_tv != 0);
  int cnt = __builtin_popcountll(tapmask) * kslabs;
  if (a.nseg > 1) cnt = a.total_slabs;

  int row_base[A_LOADS];
  unsigned long long amask[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    row_base[i] = MODE == 0 ? (apos[i].img * a.g.ih + apos[i].y * a.g.stride) * a.g.iw + apos[i].x * a.g.stride
                            : (apos[i].img * a.g.oh + apos[i].y) * a.g.ow + apos[i].x;
    amask[i] = 0;
    for (int ty_ = 0, tp = 0; ty_ < a.g.nky; ++ty_)
      for (int tx_ = 0; tx_ < a.g.nkx; ++tx_, ++tp)
        if (src_row<MODE>(a.g, apos[i], a.g.ky0 + a.g.kstep * ty_, a.g.kx0 + a.g.kstep * tx_) >= 0)
          amask[i] |= 1ull << tp;
  }

  // One cursor over the stage sequence (K stages of a tap, real taps, segments).
  SlabCursor cur = {0, 0, 0, a.K, tapmask};
  int lda = a.lda;
  __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * ES,
                                           (a.a_rows * a.lda - a.a_off) * ES);
  // (x9: the descriptor spans the three planes of the weights, plane p is reached through the scalar offset)
  const int bp_stride = X9 ? (int)a.bp_stride : 0;
  __amdgpu_buffer_rsrc_t rsB = make_rsrc_b(
      X9 ? a.Bp : (const void*)a.Bt,
      (a.mo_n ? a.mo_bbytes / ES * ESB
              : (a.nseg > 1 ? (long long)a.N * a.K : (long long)a.g.kh * a.g.kw * a.N * a.K) * ESB) +
          (long long)(BP - 1) * bp_stride);
  int brow_base[B_LOADS1];             // element offset of the staged weight row inside a tap's plane
#pragma unroll
  for (int i = 0; i < B_LOADS1; ++i)
    brow_base[i] = a.mo_n ? mo_weight_row(a, brow_off[i]) : brow_off[i] * a.K;
  unsigned tv_load = ~0u;
  unsigned tvq_lo = ~0u, tvq_hi = ~0u;   // tap validity bits (8 per ring slot) of the stages in the ring
  unsigned aoff[A_LOADS], boff[B_LOADS1];

#define K_RETAP()                                                                            \
  {                                                                                            \
    const int delta = __builtin_amdgcn_readlane(tab_delta, cur.tap);                           \
    const int toff = __builtin_amdgcn_readlane(tab_toff, cur.tap);                             \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                        \
        aoff[i] = ((amask[i] >> cur.tap) & 1ull)                                               \
                      ? (unsigned)((row_base[i] + delta) * lda + q4) * (unsigned)ES : OOB_OFFSET; \
    _Pragma("unroll") for (int i = 0; i < B_LOADS1; ++i)                                       \
        boff[i] = (unsigned)((a.nseg > 1 ? brow_off[i] * cur.Kc : brow_base[i]) + toff + q4B) * (unsigned)ESB; \
    if (PM)                                                                                    \
      tv_load = ((unsigned)__builtin_amdgcn_readlane((int)tab_tv, cur.tap) >> (wm * MT)) &     \
                ((1u << MT) - 1u);                                                             \
  }
  // one DMA piece (8 or 16 rows x 128 / 64 B per wave-instruction); lanes past a K tail fetch zeros
#define K_PIECE_A(SLOT, I)                                                                   \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                    \
      rsA, (lds_void_t*)(smem + (SLOT) * A_BYTES + wave * 1024 + (I) * ROWS_PER_PASS * RB), 16, \
      (int)(cur.kc + q4 < cur.Kc ? aoff[I] : OOB_OFFSET), cur.kc * ES, 0, 0);
#define K_PIECE_B1(SLOT, I, P)                                                               \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                    \
      rsB, (lds_void_t*)(smemB + (SLOT) * B_BYTES + (P) * BP_BYTES + wave * 1024 +             \
                         (I) * ROWS_PER_PASS_B * RBB), 16,                                     \
      (int)(cur.kc + q4B < cur.Kc ? boff[I] : OOB_OFFSET), cur.kc * ESB + (P) * bp_stride, 0, 0);
#define K_PIECE_B(SLOT, I)                                                                   \
  { _Pragma("unroll") for (int pl = 0; pl < BP; ++pl) { K_PIECE_B1(SLOT, I, pl) } }
#define K_NOTE_TV(SLOT)                                                                      \
  {                                                                                            \
    if ((SLOT) < 4) tvq_lo = (tvq_lo & ~(0xffu << (8 * ((SLOT) & 3)))) | ((tv_load & 0xffu) << (8 * ((SLOT) & 3))); \
    else tvq_hi = (tvq_hi & ~(0xffu << (8 * ((SLOT) & 3)))) | ((tv_load & 0xffu) << (8 * ((SLOT) & 3))); \
  }
#define K_ISSUE_ALL(SLOT)                                                                    \
  {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_LOADS1; ++i)                                       \
      if (i + 1 < B_LOADS1 || b_last) { K_PIECE_B(SLOT, i) }                                 \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) { K_PIECE_A(SLOT, i) }               \
    K_NOTE_TV(SLOT)                                                                          \
  }
  // advance the cursor by one stage (next K stage, next real tap, or next segment)
#define K_ADVANCE()                                                                          \
  {                                                                                            \
    cur.kc += BKT;                                                                             \
    if (cur.kc >= cur.Kc) {                                                                    \
      cur.kc = 0;                                                                              \
      if (a.nseg > 1) {                                                                        \
        ++cur.sgi;                                                                             \
        lda = a.seg_lda[cur.sgi]; cur.Kc = a.segK[cur.sgi];                                    \
        rsA = make_rsrc_b((const char*)a.segA[cur.sgi] + (size_t)a.seg_off[cur.sgi] * ES,     \
                          (a.a_rows * lda - a.seg_off[cur.sgi]) * ES);                         \
        rsB = make_rsrc_b(X9 ? a.segBp[cur.sgi] : (const void*)a.segB[cur.sgi],               \
                          (long long)a.N * cur.Kc * ESB + (long long)(BP - 1) * bp_stride);    \
      } else {                                                                                 \
        cur.taps_left &= cur.taps_left - 1ull;                                                 \
        cur.tap = cur.taps_left ? __builtin_ctzll(cur.taps_left) : 0;                          \
      }                                                                                        \
      K_RETAP();                                                                             \
    }                                                                                          \
  }
  // prologue: stages 0 .. D - 2
  if (cnt > 0) {
    cur.tap = cur.taps_left ? __builtin_ctzll(cur.taps_left) : 0;
    K_RETAP();
    K_ISSUE_ALL(0);
#pragma unroll
    for (int d = 1; d < D - 1; ++d)
      if (d < cnt) {
        K_ADVANCE();
        K_ISSUE_ALL(d);
      }
  }
  // fragment addresses inside a stage buffer: row r, chunk c -> r * RB + ((c ^ swz(r)) << 4)
  int arow_b[MT], brow_b[NT], asw[MT], bsw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int r = (wm * MT + i) * 32 + li;
    arow_b[i] = r * RB; asw[i] = ring_swz<CPR>(r);
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int r = (wn * NT + j) * 32 + li;
    brow_b[j] = r * RBB; bsw[j] = ring_swz<CPRB>(r);
  }

  // One stage of the K loop.  `slot` / `slot_in` (ring slot of stage `it` / of the stage issued
  // in this iteration) are plain ints in the general loop and compile-time constants in the
  // steady-state loop below, which also knows that a further stage is issued and how many are in
  // flight: round 3's stamps (tools/trace_ring.py) showed the ≈75 scalar / address
  // instructions of a general stage — ring-slot arithmetic, the `ahead` / `more` cases, tuning
  // bits — to cost as much as its MFMAs once sixteen waves share the CU's scalar issue.
  typedef typename std::conditional<ES == 2, bf16x8, f32x4>::type frag_t;
  auto stage = [&](auto slot, auto slot_in, auto steady_c, int it) __attribute__((always_inline)) {
    constexpr bool STEADY = decltype(steady_c)::value;
#ifdef C2D_RING_TRACE
    const unsigned long long tw0 = __builtin_amdgcn_s_memtime();
#endif
    // Stage `it` has landed: a wave's DMA pieces retire in order, so "at most the pieces of the
    // `ahead` newest stages outstanding" says this wave's pieces of stage `it` are done;
    // everybody's: the barrier.  The barrier also says every wave is done reading the buffer of
    // stage it - 1, which the pieces issued below overwrite.
    if constexpr (STEADY) {
      if (b_last) wait_vmcnt<(D - 2) * PER>();
      else wait_vmcnt<(D - 2) * (PER - BP)>();
    } else {
      const int ahead = min(cnt, it + D - 1) - it - 1;      // stages issued beyond `it` (block-uniform)
      if (b_last) {
        if (D > 5 && ahead >= 4) wait_vmcnt<(D > 5 ? 4 : 0) * PER>();
        else if (D > 4 && ahead >= 3) wait_vmcnt<(D > 4 ? 3 : 0) * PER>();
        else if (D > 3 && ahead >= 2) wait_vmcnt<(D > 3 ? 2 : 0) * PER>();
        else if (D > 2 && ahead >= 1) wait_vmcnt<(D > 2 ? 1 : 0) * PER>();
        else wait_vmcnt<0>();
      } else {
        if (D > 5 && ahead >= 4) wait_vmcnt<(D > 5 ? 4 : 0) * (PER - BP)>();
        else if (D > 4 && ahead >= 3) wait_vmcnt<(D > 4 ? 3 : 0) * (PER - BP)>();
        else if (D > 3 && ahead >= 2) wait_vmcnt<(D > 3 ? 2 : 0) * (PER - BP)>();
        else if (D > 2 && ahead >= 1) wait_vmcnt<(D > 2 ? 1 : 0) * (PER - BP)>();
        else wait_vmcnt<0>();
      }
    }
    __builtin_amdgcn_s_barrier();
#ifdef C2D_RING_TRACE
    const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
    tr_wait += tw1 - tw0;
#endif
    const bool more = STEADY || (it + D - 1 < cnt && !(a.dbg & 64));
    if (more) K_ADVANCE();
    // (row-major launches compute every 32-row tile: rows beyond M are zeros and are not stored)
    unsigned onbits = ~0u;
    if (PM || !STEADY) {
      const unsigned tvq = slot < 4 ? tvq_lo >> (8 * (slot & 3)) : tvq_hi >> (8 * (slot & 3));
      onbits = (!STEADY && (a.dbg & 4)) ? 0u : __builtin_amdgcn_readfirstlane(row_bits & tvq);
    }
    const char* const bufa = smem + slot * A_BYTES;
    const char* const bufb = smemB + slot * B_BYTES;
    // B fragments of the whole stage, then the DMA pieces of the stage D - 1 ahead (behind the
    // fragment reads, so that their issue overlaps the LDS latency), then, per 32-row tile of this
    // wave (pixel-major: one scalar branch each — a tile whose rows are SAME padding for this tap
    // costs nothing) its A fragments and KS x NT MFMAs.
    // (fp32: lane half lh takes the chunks lh * KS .. of a row — a permutation of the k order common
    //  to both operands — and feeds four v_mfma_f32_32x32x2_f32 from every 16-byte chunk)
    // (one-row-tile waves of many column tiles: the fragments of ONE k-step at a time — those of a
    //  whole 64-deep stage, 12..24 B fragments, do not fit beside the accumulators at four waves
    //  per SIMD)
    constexpr bool PER_STEP = MT == 1 && NT * KS > 8;
    if constexpr (X9) {
      // One k16 step at a time: the three planes of NT weight fragments, then per 32-row tile its raw
      // fp32 fragment (two 16-byte chunks), the split, and 9 x NT MFMAs — the partial products that
      // involve a low plane first.  No per-tile branch (x9 pixel-major launches are one pixel per
      // block: a visited tap is real for every tile; tiles without rows multiply zeros): the stage
      // is ONE scheduling region, and the DMA pieces of the stage D - 1 ahead leave one by one
      // BETWEEN the MFMA groups instead of in front of them (a piece costs 60 - 180 issue cycles,
      // MI355X_MICROARCH.md: five to eight of them in front of the MFMAs idled the matrix pipe for a
      // third of a stage at one wave per SIMD).
      constexpr int GROUPS = KS * MT * NT;
      (void)onbits;
      if (more) K_NOTE_TV(slot_in)
      int piece_q = 0;
#pragma unroll
      for (int st = 0; st < KS; ++st) {
        bf16x8 bh[NT], bm[NT], bl[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const char* bp = bufb + brow_b[j] + (((2 * st + lh) ^ bsw[j]) << 4);
          bh[j] = *reinterpret_cast<const bf16x8*>(bp);
          bm[j] = *reinterpret_cast<const bf16x8*>(bp + BP_BYTES);
          bl[j] = *reinterpret_cast<const bf16x8*>(bp + 2 * BP_BYTES);
        }
#ifdef C2D_RING_TRACE
        if (st == 0) tr_issue += __builtin_amdgcn_s_memtime() - tw1;
#endif
        if (st == 0) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int c0 = 4 * st + 2 * lh;
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(bufa + arow_b[i] + ((c0 ^ asw[i]) << 4));
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(bufa + arow_b[i] + (((c0 + 1) ^ asw[i]) << 4));
          bf16x8 ah, am, al;
          if (STEADY || !(a.dbg & 32)) {
            split3_frag(r0, r1, ah, am, al);
          } else {                                  // (ablation: no split arithmetic)
            ah = __builtin_bit_cast(bf16x8, r0); am = __builtin_bit_cast(bf16x8, r1); al = ah;
          }
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if (STEADY || !(a.dbg & 4)) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bm[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            } else {                                // (ablation: no MFMAs; the fragments stay live)
              acc[i][j][0] += (float)al[0] + (float)am[0] + (float)ah[0] + (float)bl[j][0] + (float)bm[j][0] + (float)bh[j][0];
            }
            // this group's share of the DMA pieces (weights first: they have the partial last pass)
            const int g = (st * MT + i) * NT + j;
            const int q_end = ((g + 1) * PER + GROUPS - 1) / GROUPS;
            if (more) {
#pragma unroll
              for (int q = 0; q < PER; ++q) {
                if (q >= piece_q && q < q_end) {
                  if (q < B_LOADS) {
                    if (q / BP + 1 < B_LOADS1 || b_last) { K_PIECE_B1(slot_in, q / BP, q % BP) }
                  } else {
                    K_PIECE_A(slot_in, q - B_LOADS)
                  }
                }
              }
            }
            piece_q = q_end;
          }
        }
      }
    } else if constexpr (PER_STEP) {
      if (more) {
#pragma unroll
        for (int i = 0; i < B_LOADS1; ++i)
          if (i + 1 < B_LOADS1 || b_last) { K_PIECE_B(slot_in, i) }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) { K_PIECE_A(slot_in, i) }
        K_NOTE_TV(slot_in)
      }
#ifdef C2D_RING_TRACE
      tr_issue += __builtin_amdgcn_s_memtime() - tw1;
#endif
      __builtin_amdgcn_s_setprio(1);
      if ((STEADY && !PM) || (onbits & 1u)) {
#pragma unroll
        for (int st = 0; st < KS; ++st) {
          const int ch = ES == 2 ? 2 * st + lh : lh * KS + st;
          const frag_t afc = *reinterpret_cast<const frag_t*>(bufa + arow_b[0] + ((ch ^ asw[0]) << 4));
          frag_t bfc[NT];
#pragma unroll
          for (int j = 0; j < NT; ++j)
            bfc[j] = *reinterpret_cast<const frag_t*>(bufb + brow_b[j] + ((ch ^ bsw[j]) << 4));
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            if constexpr (ES == 2) {
              acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afc, bfc[j], acc[0][j], 0, 0, 0);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afc[e], bfc[j][e], acc[0][j], 0, 0, 0);
            }
          }
        }
      }
    } else {
      frag_t bf[NT][KS];
      if (!STEADY && (a.dbg & 16)) {          // (ablation: no B fragment reads)
  #pragma unroll
        for (int st = 0; st < KS; ++st)
  #pragma unroll
          for (int j = 0; j < NT; ++j)
            bf[j][st] = __builtin_bit_cast(frag_t, f32x4{(float)it, (float)st, (float)j, 1.0f});
      } else {
  #pragma unroll
      for (int st = 0; st < KS; ++st)
  #pragma unroll
        for (int j = 0; j < NT; ++j)
          bf[j][st] = *reinterpret_cast<const frag_t*>(
              bufb + brow_b[j] + (((ES == 2 ? 2 * st + lh : lh * KS + st) ^ bsw[j]) << 4));
      }
      if (more) {
        if (STEADY || !(a.dbg & 2)) {
  #pragma unroll
          for (int i = 0; i < B_LOADS1; ++i)
            if (i + 1 < B_LOADS1 || b_last) { K_PIECE_B(slot_in, i) }
        }
  #pragma unroll
        for (int i = 0; i < A_LOADS; ++i) { K_PIECE_A(slot_in, i) }
        K_NOTE_TV(slot_in)
      }
  #ifdef C2D_RING_TRACE
      tr_issue += __builtin_amdgcn_s_memtime() - tw1;
  #endif
      __builtin_amdgcn_s_setprio(1);
  #pragma unroll
      for (int i = 0; i < MT; ++i) {
        if ((STEADY && !PM) || ((onbits >> i) & 1u)) {
          frag_t af[KS];
  #pragma unroll
          for (int st = 0; st < KS; ++st)
            af[st] = *reinterpret_cast<const frag_t*>(
                bufa + arow_b[i] + (((ES == 2 ? 2 * st + lh : lh * KS + st) ^ asw[i]) << 4));
  #pragma unroll
          for (int st = 0; st < KS; ++st)
  #pragma unroll
            for (int j = 0; j < NT; ++j) {
              if constexpr (ES == 2) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st], bf[j][st], acc[i][j], 0, 0, 0);
              } else {
  #pragma unroll
                for (int e = 0; e < 4; ++e)
                  acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[st][e], bf[j][st][e], acc[i][j], 0, 0, 0);
              }
            }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };

#ifdef C2D_RING_TRACE
  const unsigned long long tr1 = __builtin_amdgcn_s_memtime();
#endif
  int it = 0;
  // steady state, D stages per trip (the stage of a trip's position d sits in ring slot d): every
  // stage of the trip issues a further one and has D - 2 newer ones in flight
  if (a.dbg == 0)
    for (; it + 2 * D - 2 < cnt; it += D)
      static_for_ring(std::make_integer_sequence<int, D>{}, [&](auto d) __attribute__((always_inline)) {
        stage(d, std::integral_constant<int, (decltype(d)::value + D - 1) % D>{}, std::true_type{},
              it + decltype(d)::value);
      });
  // the last stages (and launches run with tuning bits): the general form
  int slot = 0;                          // (`it` is a multiple of D here) ring slot of stage `it`
  int slot_in = D - 1;                   // ring slot the stage issued in iteration `it` goes to
  for (; it < cnt; ++it) {
    stage(slot, slot_in, std::false_type{}, it);
    if (++slot == D) slot = 0;
    if (++slot_in == D) slot_in = 0;
  }
#undef K_RETAP
#undef K_PIECE_A
#undef K_PIECE_B
#undef K_PIECE_B1
#undef K_NOTE_TV
#undef K_ISSUE_ALL
#undef K_ADVANCE
#ifdef C2D_RING_TRACE
  const unsigned long long tr2 = __builtin_amdgcn_s_memtime();
#endif
  __syncthreads();     // every wave is done with the stage buffers: the epilogue reuses them
#ifdef C2D_RING_TRACE
  const unsigned long long te1 = __builtin_amdgcn_s_memtime();
  unsigned long long te2 = te1, te3 = te1;
#endif
  if (a.dbg & 8) return;

  // Epilogue: every wave transposes its 32-row strips through a private LDS slice so that a lane
  // stores 16 B (8 bf16 / 4 fp32) of ONE output row, neighbouring lanes neighbouring columns.
  constexpr int SCOLS = NT * 32;
  constexpr int SSTR = SCOLS + 4;
  if constexpr (FUSED) {
    // Fused BN/ReLU backward of the producer layer (see IgemmArgs::fy; input-gradient launches):
    // dc = dx * [y > 0] * scale, and per column the sums of dz = dx * [y > 0] and dz * (y - beta) /
    // gamma over the block's rows.  Round 5: the same structure as the plain epilogue below — the
    // output row of every tile row and the producer's per-column facts once per block in LDS tables,
    // 16 bytes per lane and store — with ONE difference: a lane keeps a FIXED 16-byte column chunk
    // (lane % CPRW) for the whole block, because it sums per column; a pass covers 64 / CPRW rows
    // (tiles whose chunk count does not divide 64 leave the last lanes idle: 128x192 tiles use 60
    // of 64).  Round 2's form (four columns per lane, a row decomposition with two divisions and the
    // producer routing from kernarg in every pass, inside the plain kernel: its registers were the
    // plain kernel's) made the fused launches of the bf16 step cost more than the separate
    // bn_relu_bwd kernels they replaced (2.98 against 2.95 ms per step; this form 2.915 against
    // 2.925, same box, alternating runs).  Fetching the producer's outputs of the whole block tile
    // up front (one load latency per block instead of one per strip) was tried on top and lost to
    // its 32 spilled registers (2.96 ms).
    constexpr int CW = 16 / ES;                    // columns per lane and store
    constexpr int CPRW = SCOLS / CW;               // chunks per strip row
    constexpr int RPPF = 64 / CPRW;                // rows per pass
    constexpr int TAB_BYTES = (BM + 3 * BN) * 4;
    constexpr int HALVES = WM * WN * 32 * SSTR * 4 + TAB_BYTES <= LDS_BYTES ? 1 : 2;
    constexpr int HROWS = 32 / HALVES;
    constexpr int STAGE_BYTES = WM * WN * HROWS * SSTR * 4;
    constexpr int NPASSF = (HROWS + RPPF - 1) / RPPF;
    static_assert(STAGE_BYTES + TAB_BYTES <= LDS_BYTES, "epilogue staging exceeds LDS");
    static_assert(NTHREADS * 2 * CW * 4 <= LDS_BYTES, "column-sum exchange exceeds LDS");
    static_assert(CPRW <= 64 && RPPF >= 1, "strip row wider than a wave");
    float* stage = reinterpret_cast<float*>(smem) + wave * (HROWS * SSTR);
    int* const drow_tab = reinterpret_cast<int*>(smem + STAGE_BYTES);   // output row of tile row r, or -1
    float* const tab_sc = reinterpret_cast<float*>(drow_tab + BM);      // producer's BN scale (NaN: the
    float* const tab_be = tab_sc + BN;                                  // column passes unchanged), beta
    float* const tab_ig = tab_be + BN;                                  // and 1 / gamma (0: no gamma)
    for (int r = tid; r < BM; r += NTHREADS) {
      const int m = m0 + r;
      bool row_ok = m < a.M;
      int drow = m;
      if (PM || a.g.sub > 1) {
        const RowPos p = decompose<PM>(m, a.M, a.g);
        row_ok = p.valid;
        drow = (p.img * a.g.ih + p.y * a.g.sub + a.g.y0) * a.g.iw + p.x * a.g.sub + a.g.x0;
      }
      drow_tab[r] = row_ok ? drow : -1;
    }
    for (int c = tid; c < BN; c += NTHREADS) {
      const int ncol = n0 + c;
      float sc = __builtin_nanf(""), be = 0.f, ig = 0.f;
      if (ncol < a.N) {
        int p = 0, lo = 0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (q + 1 < a.fnprod && ncol >= a.fseg_end[q]) { p = q + 1; lo = a.fseg_end[q]; }
        if (!a.fident[p]) {
          sc = a.fscale[p][ncol - lo];
          if (a.fgamma[p]) {
            be = a.fbeta[p][ncol - lo];
            const float ga = a.fgamma[p][ncol - lo];
            ig = ga != 0.f ? 1.f / ga : 0.f;
          }
        }
      }
      tab_sc[c] = sc; tab_be[c] = be; tab_ig[c] = ig;
    }
    __syncthreads();
    const int chunk = lane % CPRW, rsub = lane / CPRW;
    const bool lane_on = rsub < RPPF;
    const int lcol = wn * SCOLS + chunk * CW;
    const int ncol = n0 + lcol;
    const bool col_ok = lane_on && ncol < a.N;
    float csc[CW], cbe[CW], cig[CW], sb[CW], sg[CW];
#pragma unroll
    for (int e = 0; e < CW; ++e) {
      csc[e] = tab_sc[lcol + e]; cbe[e] = tab_be[lcol + e]; cig[e] = tab_ig[lcol + e];
      sb[e] = 0.f; sg[e] = 0.f;
    }
    __syncthreads();       // (every lane has its column facts: the exchange below reuses the memory)
    constexpr int GP = NPASSF < 4 ? NPASSF : 4;     // passes whose loads are in flight together
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int h = 0; h < HALVES; ++h) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int rr = 0; rr < 16 / HALVES; ++rr) {
            const int r = h * (16 / HALVES) + rr;
            stage[((rr & 3) + 8 * (rr >> 2) + 4 * lh) * SSTR + j * 32 + li] = acc[i][j][r];
          }
        __builtin_amdgcn_s_waitcnt(0xc07f);      // the strip is in LDS
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pg = 0; pg < NPASSF; pg += GP) {
          f32x4 yraw[GP], oldv[GP];
          int drow[GP];
#pragma unroll
          for (int q = 0; q < GP; ++q) {
            const int row = (pg + q) * RPPF + rsub;
            drow[q] = (pg + q < NPASSF && col_ok && row < HROWS)
                          ? drow_tab[(wm * MT + i) * 32 + h * HROWS + row] : -1;
            if (drow[q] >= 0) {
              const char* ysrc = reinterpret_cast<const char*>(a.fy) +
                                 ((size_t)drow[q] * a.fldy + a.fyoff + ncol) * ES;
              yraw[q] = *reinterpret_cast<const f32x4*>(ysrc);
              if (a.accumulate)
                oldv[q] = *reinterpret_cast<const f32x4*>(
                    reinterpret_cast<const char*>(a.C) + ((size_t)drow[q] * a.ldc + a.c_off + ncol) * ES);
            }
          }
#pragma unroll
          for (int q = 0; q < GP; ++q) {
            if (drow[q] < 0) continue;
            const int row = (pg + q) * RPPF + rsub;
            float v[CW], yv[CW];
#pragma unroll
            for (int e = 0; e < CW; e += 4)
              *reinterpret_cast<f32x4*>(&v[e]) =
                  *reinterpret_cast<const f32x4*>(&stage[row * SSTR + chunk * CW + e]);
            if constexpr (ES == 2) {
              const bf16x8 yb = __builtin_bit_cast(bf16x8, yraw[q]);
#pragma unroll
              for (int e = 0; e < 8; ++e) yv[e] = (float)yb[e];
              if (a.accumulate) {
                const bf16x8 o = __builtin_bit_cast(bf16x8, oldv[q]);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += (float)o[e];
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) yv[e] = yraw[q][e];
              if (a.accumulate) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += oldv[q][e];
              }
            }
#pragma unroll
            for (int e = 0; e < CW; ++e) {
              if (csc[e] == csc[e]) {             // (NaN: a pooling branch's column, plain gradient)
                const float dz = yv[e] > 0.f ? v[e] : 0.f;
                sb[e] += dz;
                sg[e] += dz * (yv[e] - cbe[e]) * cig[e];
                v[e] = dz * csc[e];
              }
            }
            char* dst = reinterpret_cast<char*>(a.C) + ((size_t)drow[q] * a.ldc + a.c_off + ncol) * ES;
            if constexpr (ES == 2) {
              bf16x8 o;
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
              *reinterpret_cast<bf16x8*>(dst) = o;
            } else {
              *reinterpret_cast<f32x4*>(dst) = f32x4{v[0], v[1], v[2], v[3]};
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    // column sums of the block: lane sums -> LDS -> fixed-order sums over the lanes (rows of a pass)
    // and waves (row tiles) that share a column -> the block's row of the partials (no atomics)
    __syncthreads();
    float* const red = reinterpret_cast<float*>(smem);
    if (lane_on) {
#pragma unroll
      for (int e = 0; e < CW; ++e) {
        red[(wave * 64 + lane) * 2 * CW + e] = sb[e];
        red[(wave * 64 + lane) * 2 * CW + CW + e] = sg[e];
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * BN; idx += NTHREADS) {
      const int k = idx / BN, c = idx - k * BN;
      const int wn_c = c / SCOLS, cl = c - wn_c * SCOLS;
      const int ch = cl / CW, e = cl - ch * CW;
      float t = 0.f;
      for (int wm_ = 0; wm_ < WM; ++wm_)
        for (int r = 0; r < RPPF; ++r)
          t += red[((wm_ * WN + wn_c) * 64 + r * CPRW + ch) * 2 * CW + k * CW + e];
      if (n0 + c < a.N) a.fpart[((size_t)(a.fpart_row0 + mt) * 2 + k) * a.N + n0 + c] = t;
    }
  } else {
    // Round 3 (tools/trace_ring.py): the passes above cost 400 - 600 cycles each — per pass a
    // row decomposition (two divisions for pixel-major rows), the routing of a multi-output
    // launch from memory, 4 elements per lane — and were a sixth to a third of a launch.  Here
    // the per-row and per-column facts are computed once per block into LDS tables and a pass
    // handles 16 B per lane of a flat (row, 16-byte chunk) index, so no lane idles either.
    constexpr int CW = 16 / ES;                    // columns per lane and store
    constexpr int CPRW = SCOLS / CW;               // chunks per strip row
    constexpr int TAB_BYTES = (BM + 3 * BN) * 4;
    constexpr int HALVES = WM * WN * 32 * SSTR * 4 + TAB_BYTES <= LDS_BYTES ? 1 : 2;
    constexpr int HROWS = 32 / HALVES;
    constexpr int STAGE_BYTES = WM * WN * HROWS * SSTR * 4;
    static_assert(STAGE_BYTES + TAB_BYTES <= LDS_BYTES, "epilogue staging exceeds LDS");
    constexpr int NPASS = HROWS * CPRW / 64;
    static_assert(HROWS * CPRW % 64 == 0, "epilogue chunks vs wave size");
    constexpr int GP = ring_pass_group(NPASS);           // passes whose old values are held at once
    static_assert(NPASS % GP == 0 && GP <= 8, "epilogue pass groups");
    float* stage = reinterpret_cast<float*>(smem) + wave * (HROWS * SSTR);
    int* const drow_tab = reinterpret_cast<int*>(smem + STAGE_BYTES);   // output row of tile row r, or -1
    float* const tab_sc = reinterpret_cast<float*>(drow_tab + BM);      // BN scale / shift of column c,
    float* const tab_sh = tab_sc + BN;                                  // and the lower bound of its
    float* const tab_lb = tab_sh + BN;                                  // activation (0 = ReLU, -inf)
    const bool has_bn = a.scale || a.shift || a.relu || a.mo_n;         // (block-uniform)
    for (int r = tid; r < BM; r += NTHREADS) {
      const int m = m0 + r;
      bool row_ok = m < a.M;
      int drow = m;
      if (PM || (MODE == 1 && a.g.sub > 1)) {
        const RowPos p = decompose<PM>(m, a.M, a.g);
        row_ok = p.valid;
        drow = MODE == 0 ? (p.img * a.g.rh + p.y) * a.g.rw + p.x
                         : (p.img * a.g.ih + p.y * a.g.sub + a.g.y0) * a.g.iw + p.x * a.g.sub + a.g.x0;
      }
      drow_tab[r] = row_ok ? drow : -1;
    }
    if (has_bn)
      for (int c = tid; c < BN; c += NTHREADS) {
        const int ncol = n0 + c;
        float sc = 1.f, sh = 0.f;
        int relu = a.relu;
        if (ncol < a.N) {
          if (MODE == 0 && a.mo_n) {
            const MoOut o = mo_output(a, ncol);
            sc = o.scale[ncol - o.lo]; sh = o.shift[ncol - o.lo]; relu = o.relu;
          } else {
            if (a.scale) sc = a.scale[ncol];
            if (a.shift) sh = a.shift[ncol];
          }
        }
        tab_sc[c] = sc; tab_sh[c] = sh; tab_lb[c] = relu ? 0.f : -__builtin_inff();
      }
    // (the tables' global loads are consumed by their LDS writes: no load is pending when the
    //  first store is issued — on gfx9 stores count in vmcnt too, and a wait for a load placed
    //  after a store would also wait for that store's acknowledgement)
    __syncthreads();
#ifdef C2D_RING_TRACE
    te2 = __builtin_amdgcn_s_memtime();
#endif
    // where chunk `c` of strip (i, h) goes: nullptr = nowhere (row beyond M / padding, column beyond N)
    const int mo_n = MODE == 0 ? a.mo_n : 0;
    auto locate = [&](int i, int h, int c, int& row, int& lcol) -> char* {
      row = c / CPRW;
      lcol = wn * SCOLS + (c - row * CPRW) * CW;
      const int ncol = n0 + lcol;
      const int drow = drow_tab[(wm * MT + i) * 32 + h * HROWS + row];
      if (drow < 0 || ncol >= a.N) return nullptr;
      float* oC = a.C; int oldc = a.ldc, ocoff = a.c_off + ncol;
      if (mo_n) {      // several convolutions in one GEMM: the chunk belongs to one of them
        oC = a.mo_C[0]; oldc = a.mo_ldc[0]; ocoff = a.mo_coff[0] + ncol;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const bool t = q + 1 < mo_n && ncol >= a.mo_end[q];
          oC = t ? a.mo_C[q + 1] : oC; oldc = t ? a.mo_ldc[q + 1] : oldc;
          ocoff = t ? a.mo_coff[q + 1] + ncol - a.mo_end[q] : ocoff;
        }
      }
      return reinterpret_cast<char*>(oC) + ((size_t)drow * oldc + ocoff) * ES;
    };
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int h = 0; h < HALVES; ++h) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int rr = 0; rr < 16 / HALVES; ++rr) {
            const int r = h * (16 / HALVES) + rr;
            // accumulator register r of lane (li, lh) is row (r & 3) + 8 * (r >> 2) + 4 * lh of the
            // strip, i.e. row (rr & 3) + 8 * (rr >> 2) + 4 * lh of half h
            stage[((rr & 3) + 8 * (rr >> 2) + 4 * lh) * SSTR + j * 32 + li] = acc[i][j][r];
          }
#pragma unroll
        for (int pg = 0; pg < NPASS / GP; ++pg) {
          f32x4 oldv[GP];
          if (a.accumulate) {     // the old values of GP passes in flight together, ONE wait
#pragma unroll
            for (int q = 0; q < GP; ++q) {
              int row, lcol;
              const char* src = locate(i, h, (pg * GP + q) * 64 + lane, row, lcol);
              if (src) oldv[q] = *reinterpret_cast<const f32x4*>(src);
            }
            __builtin_amdgcn_s_waitcnt(0x0f70);
          }
          if (pg == 0) {
            __builtin_amdgcn_s_waitcnt(0xc07f);      // the strip is in LDS
            __builtin_amdgcn_wave_barrier();
          }
#pragma unroll
          for (int q = 0; q < GP; ++q) {
            int row, lcol;
            char* dst = locate(i, h, (pg * GP + q) * 64 + lane, row, lcol);
            float v[CW];
#pragma unroll
            for (int e = 0; e < CW; e += 4)
              *reinterpret_cast<f32x4*>(&v[e]) = *reinterpret_cast<const f32x4*>(&stage[row * SSTR + (lcol - wn * SCOLS) + e]);
            if (has_bn) {
#pragma unroll
              for (int e = 0; e < CW; e += 4) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(&tab_sc[lcol + e]);
                const f32x4 sh = *reinterpret_cast<const f32x4*>(&tab_sh[lcol + e]);
                const f32x4 lb = *reinterpret_cast<const f32x4*>(&tab_lb[lcol + e]);
#pragma unroll
                for (int u = 0; u < 4; ++u) v[e + u] = fmaxf(v[e + u] * sc[u] + sh[u], lb[u]);
              }
            }
            if (dst) {
              if constexpr (ES == 2) {
                if (a.accumulate) {
                  const bf16x8 o = __builtin_bit_cast(bf16x8, oldv[q]);
#pragma unroll
                  for (int e = 0; e < 8; ++e) v[e] += (float)o[e];
                }
                bf16x8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
                *reinterpret_cast<bf16x8*>(dst) = o;
              } else {
                f32x4 o = {v[0], v[1], v[2], v[3]};
                if (a.accumulate) o += oldv[q];
                *reinterpret_cast<f32x4*>(dst) = o;
              }
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef C2D_RING_TRACE
        if (i == 0 && h == 0) te3 = __builtin_amdgcn_s_memtime();
#endif
      }
    }
  }
#ifdef C2D_RING_TRACE
  if (a.trace && tid == 0) {
    const unsigned long long tr3 = __builtin_amdgcn_s_memtime();
    unsigned long long* t = a.trace + (size_t)blockIdx.x * 16;
    t[0] = tr0; t[1] = tr1 - tr0; t[2] = tr2 - tr1; t[3] = tr3 - tr2; t[4] = tr_wait; t[5] = tr_issue;
    t[6] = (unsigned long long)cnt; t[7] = __builtin_amdgcn_s_memrealtime();
    t[8] = te1 - tr2; t[9] = te2 - te1; t[10] = te3 - te2; t[11] = tr3 - te3;
    t[12] = tr_rt0;      // (in-kernel clock of the block = (t[1] + t[2] + t[3]) / (t[7] - t[12]) x 100 MHz)
  }
#endif
}

// (x9: the accumulators, three weight planes per column tile and the split need the registers of
//  at most two waves per SIMD)
#define K_RING_LDS (D * (WM * MT * ES + WN * NT * (BP == 3 ? 6 : ES)) * 32 * BKT)
#define K_RING_BOUNDS                                                                          \
  __launch_bounds__(WM * WN * 64,                                                                \
                    BP == 3 ? (ring_blocks_per_cu(K_RING_LDS) * (WM * WN) / 4 >= 2 ? 2 : 1)    \
                    : ring_blocks_per_cu(K_RING_LDS) * (WM * WN) / 4 > 0                       \
                        ? ring_blocks_per_cu(K_RING_LDS) * (WM * WN) / 4 : 1)

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED = false,
          int BP = 1>
__global__ K_RING_BOUNDS void igemm_ring_kernel(IgemmArgs a) {
  igemm_ring_body<MODE, WM, WN, MT, NT, PM, BKT, D, ES, FUSED, BP>(a, (int)blockIdx.x, (int)gridDim.x);
}

// Up to four INDEPENDENT problems of one instance in one launch: the parity classes of a stride-2
// input gradient (conv_gemm.hip conv_dgrad_impl: four launches of 141-250 workgroups — less than one
// per CU each, 14-27 us apiece, most of it ramp and epilogue — whose workgroups now share the CUs).
// Block b belongs to problem p with first[p] <= b < first[p + 1] and is its block b - first[p].
struct IgemmRingGroup {
  IgemmArgs a[4];
  int first[5];
  int num;
};
template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED = false,
          int BP = 1>
__global__ K_RING_BOUNDS void igemm_ring_group_kernel(IgemmRingGroup g) {
  int p = 0;
  for (int i = 1; i < g.num; ++i)
    if ((int)blockIdx.x >= g.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  igemm_ring_body<MODE, WM, WN, MT, NT, PM, BKT, D, ES, FUSED, BP>(g.a[p], (int)blockIdx.x - g.first[p],
                                                                    g.first[p + 1] - g.first[p]);
}
#undef K_RING_BOUNDS
#undef K_RING_LDS

}  // namespace
}  // namespace c2d_ig
