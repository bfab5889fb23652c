// bf16 implicit-GEMM convolution with a DEEP LDS-DMA RING (second stage in the bf16 storage mode,
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
template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED = false>
__device__ __forceinline__ void igemm_ring_body(const IgemmArgs& a, const int bid, const int nbid) {
  constexpr int RB = BKT * ES;                  // bytes per staged row (128 or 64)
  constexpr int CPR = RB / 16;                  // 16-byte chunks per row (8 or 4)
  constexpr int KS = ES == 2 ? BKT / 16 : CPR / 2;   // bf16: MFMA k-steps per stage; fp32: 16-byte
                                                     // chunks per lane and stage (4 MFMAs each)
  constexpr int EPC = 16 / ES;                  // elements per 16-byte chunk
  constexpr int BM = WM * MT * 32;
  constexpr int BN = WN * NT * 32;
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int ROWS_PER_PASS = NTHREADS / CPR; // a wave-instruction stages 1 KiB = 1024 / RB rows
  constexpr int A_LOADS = BM / ROWS_PER_PASS;
  // weight rows: whole passes, plus a last partial pass that only the first waves take part in
  // (32-deep stages of 8-wave blocks stage 128 rows per pass; BN = 192 / 320 leave half a pass)
  constexpr int B_LOADS = (BN + ROWS_PER_PASS - 1) / ROWS_PER_PASS;
  constexpr bool B_TAIL = BN % ROWS_PER_PASS != 0;
  constexpr int PER = A_LOADS + B_LOADS;        // DMA wave-instructions per stage and wave (PER - 1
                                                // for the waves outside a partial last pass)
  constexpr int A_BYTES = BM * RB, B_BYTES = BN * RB;
  constexpr int RING_BYTES = D * (A_BYTES + B_BYTES);
  // (the epilogue stages 16-row half strips of every wave through the same memory)
  constexpr int EPI_BYTES = WM * WN * 16 * (NT * 32 + 4) * 4 + (BM + 3 * BN) * 4;   // (+ its row / column tables)
  constexpr int LDS_BYTES = RING_BYTES > EPI_BYTES ? RING_BYTES : EPI_BYTES;
  static_assert(BM % ROWS_PER_PASS == 0 && BN % (1024 / RB) == 0, "tile vs block size");
  static_assert(D >= 2 && D <= 6 && (D - 2) * PER <= 63, "ring depth vs the 6-bit vmcnt");
  static_assert(LDS_BYTES <= 160 * 1024, "ring exceeds LDS");
  static_assert(RB == 128 || RB == 64, "stage depth");
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
  unsigned long long tr_wait = 0, tr_issue = 0;
#endif
  // DMA: this lane fetches the chunk that belongs at LDS position tid % CPR of its row
  const int lrow = tid / CPR;                                       // row inside a pass
  const int lswz = CPR == 8 ? (lrow >> 1) & 7 : (lrow >> 2) & 3;    // (pass rows are multiples of 16)
  const int kchunk = (tid % CPR) ^ lswz;
  const int q4 = kchunk * EPC;                                      // element offset inside the stage
  // does this wave hold a piece of the (partial) last pass of the weight rows?
  const bool b_last = !B_TAIL || wave * (1024 / RB) + (B_LOADS - 1) * ROWS_PER_PASS < BN;

  int mt, nt;
  block_tile(a.g, a.n_tiles, bid, nbid, &mt, &nt);
  const int m0 = mt * BM, n0 = nt * BN;
  const int ntaps = a.g.nky * a.g.nkx;
  const int kslabs = (a.K + BKT - 1) / BKT;
  const size_t tap_stride = (size_t)a.N * a.K;

  RowPos apos[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) apos[i] = decompose<PM>(m0 + lrow + i * ROWS_PER_PASS, a.M, a.g);
  int brow_off[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) brow_off[i] = min(n0 + lrow + i * ROWS_PER_PASS, a.N - 1);

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
  __amdgpu_buffer_rsrc_t rsB = make_rsrc_b(
      a.Bt, a.mo_n ? a.mo_bbytes
                   : (a.nseg > 1 ? (long long)a.N * a.K : (long long)a.g.kh * a.g.kw * a.N * a.K) * ES);
  int brow_base[B_LOADS];              // element offset of the staged weight row inside a tap's plane
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i)
    brow_base[i] = a.mo_n ? mo_weight_row(a, brow_off[i]) : brow_off[i] * a.K;
  unsigned tv_load = ~0u;
  unsigned tvq_lo = ~0u, tvq_hi = ~0u;   // tap validity bits (8 per ring slot) of the stages in the ring
  unsigned aoff[A_LOADS], boff[B_LOADS];

#define C2D_RETAP()                                                                            \
  {                                                                                            \
    const int delta = __builtin_amdgcn_readlane(tab_delta, cur.tap);                           \
    const int toff = __builtin_amdgcn_readlane(tab_toff, cur.tap);                             \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                        \
        aoff[i] = ((amask[i] >> cur.tap) & 1ull)                                               \
                      ? (unsigned)((row_base[i] + delta) * lda + q4) * (unsigned)ES : OOB_OFFSET; \
    _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                        \
        boff[i] = (unsigned)((a.nseg > 1 ? brow_off[i] * cur.Kc : brow_base[i]) + toff + q4) * (unsigned)ES; \
    if (PM)                                                                                    \
      tv_load = ((unsigned)__builtin_amdgcn_readlane((int)tab_tv, cur.tap) >> (wm * MT)) &     \
                ((1u << MT) - 1u);                                                             \
  }
  // one DMA piece (8 or 16 rows x 128 / 64 B per wave-instruction); lanes past a K tail fetch zeros
#define C2D_PIECE_A(SLOT, I)                                                                   \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                    \
      rsA, (lds_void_t*)(smem + (SLOT) * A_BYTES + wave * 1024 + (I) * ROWS_PER_PASS * RB), 16, \
      (int)(cur.kc + q4 < cur.Kc ? aoff[I] : OOB_OFFSET), cur.kc * ES, 0, 0);
#define C2D_PIECE_B(SLOT, I)                                                                   \
  __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                    \
      rsB, (lds_void_t*)(smemB + (SLOT) * B_BYTES + wave * 1024 + (I) * ROWS_PER_PASS * RB), 16, \
      (int)(cur.kc + q4 < cur.Kc ? boff[I] : OOB_OFFSET), cur.kc * ES, 0, 0);
#define C2D_NOTE_TV(SLOT)                                                                      \
  {                                                                                            \
    if ((SLOT) < 4) tvq_lo = (tvq_lo & ~(0xffu << (8 * ((SLOT) & 3)))) | ((tv_load & 0xffu) << (8 * ((SLOT) & 3))); \
    else tvq_hi = (tvq_hi & ~(0xffu << (8 * ((SLOT) & 3)))) | ((tv_load & 0xffu) << (8 * ((SLOT) & 3))); \
  }
#define C2D_ISSUE_ALL(SLOT)                                                                    \
  {                                                                                            \
    _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                        \
      if (i + 1 < B_LOADS || b_last) { C2D_PIECE_B(SLOT, i) }                                  \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) { C2D_PIECE_A(SLOT, i) }               \
    C2D_NOTE_TV(SLOT)                                                                          \
  }
  // advance the cursor by one stage (next K stage, next real tap, or next segment)
#define C2D_ADVANCE()                                                                          \
  {                                                                                            \
    cur.kc += BKT;                                                                             \
    if (cur.kc >= cur.Kc) {                                                                    \
      cur.kc = 0;                                                                              \
      if (a.nseg > 1) {                                                                        \
        ++cur.sgi;                                                                             \
        lda = a.seg_lda[cur.sgi]; cur.Kc = a.segK[cur.sgi];                                    \
        rsA = make_rsrc_b((const char*)a.segA[cur.sgi] + (size_t)a.seg_off[cur.sgi] * ES,     \
                          (a.a_rows * lda - a.seg_off[cur.sgi]) * ES);                         \
        rsB = make_rsrc_b(a.segB[cur.sgi], (long long)a.N * cur.Kc * ES);                      \
      } else {                                                                                 \
        cur.taps_left &= cur.taps_left - 1ull;                                                 \
        cur.tap = cur.taps_left ? __builtin_ctzll(cur.taps_left) : 0;                          \
      }                                                                                        \
      C2D_RETAP();                                                                             \
    }                                                                                          \
  }
  // prologue: stages 0 .. D - 2
  if (cnt > 0) {
    cur.tap = cur.taps_left ? __builtin_ctzll(cur.taps_left) : 0;
    C2D_RETAP();
    C2D_ISSUE_ALL(0);
#pragma unroll
    for (int d = 1; d < D - 1; ++d)
      if (d < cnt) {
        C2D_ADVANCE();
        C2D_ISSUE_ALL(d);
      }
  }
  // fragment addresses inside a stage buffer: row r, chunk c -> r * RB + ((c ^ swz(r)) << 4)
  int arow_b[MT], brow_b[NT], asw[MT], bsw[NT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int r = (wm * MT + i) * 32 + li;
    arow_b[i] = r * RB; asw[i] = CPR == 8 ? (r >> 1) & 7 : (r >> 2) & 3;
  }
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int r = (wn * NT + j) * 32 + li;
    brow_b[j] = r * RB; bsw[j] = CPR == 8 ? (r >> 1) & 7 : (r >> 2) & 3;
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
      else wait_vmcnt<(D - 2) * (PER - 1)>();
    } else {
      const int ahead = min(cnt, it + D - 1) - it - 1;      // stages issued beyond `it` (block-uniform)
      if (b_last) {
        if (D > 5 && ahead >= 4) wait_vmcnt<(D > 5 ? 4 : 0) * PER>();
        else if (D > 4 && ahead >= 3) wait_vmcnt<(D > 4 ? 3 : 0) * PER>();
        else if (D > 3 && ahead >= 2) wait_vmcnt<(D > 3 ? 2 : 0) * PER>();
        else if (D > 2 && ahead >= 1) wait_vmcnt<(D > 2 ? 1 : 0) * PER>();
        else wait_vmcnt<0>();
      } else {
        if (D > 5 && ahead >= 4) wait_vmcnt<(D > 5 ? 4 : 0) * (PER - 1)>();
        else if (D > 4 && ahead >= 3) wait_vmcnt<(D > 4 ? 3 : 0) * (PER - 1)>();
        else if (D > 3 && ahead >= 2) wait_vmcnt<(D > 3 ? 2 : 0) * (PER - 1)>();
        else if (D > 2 && ahead >= 1) wait_vmcnt<(D > 2 ? 1 : 0) * (PER - 1)>();
        else wait_vmcnt<0>();
      }
    }
    __builtin_amdgcn_s_barrier();
#ifdef C2D_RING_TRACE
    const unsigned long long tw1 = __builtin_amdgcn_s_memtime();
    tr_wait += tw1 - tw0;
#endif
    const bool more = STEADY || (it + D - 1 < cnt && !(a.dbg & 64));
    if (more) C2D_ADVANCE();
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
    if constexpr (PER_STEP) {
      if (more) {
#pragma unroll
        for (int i = 0; i < B_LOADS; ++i)
          if (i + 1 < B_LOADS || b_last) { C2D_PIECE_B(slot_in, i) }
#pragma unroll
        for (int i = 0; i < A_LOADS; ++i) { C2D_PIECE_A(slot_in, i) }
        C2D_NOTE_TV(slot_in)
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
          for (int i = 0; i < B_LOADS; ++i)
            if (i + 1 < B_LOADS || b_last) { C2D_PIECE_B(slot_in, i) }
        }
  #pragma unroll
        for (int i = 0; i < A_LOADS; ++i) { C2D_PIECE_A(slot_in, i) }
        C2D_NOTE_TV(slot_in)
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
#undef C2D_RETAP
#undef C2D_PIECE_A
#undef C2D_PIECE_B
#undef C2D_NOTE_TV
#undef C2D_ISSUE_ALL
#undef C2D_ADVANCE
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
    constexpr int GP = NPASS <= 8 ? NPASS : NPASS / 2;   // passes whose old values are held at once
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
  }
#endif
}

#define C2D_RING_BOUNDS                                                                          \
  __launch_bounds__(WM * WN * 64,                                                                \
                    ring_blocks_per_cu(D * (WM * MT + WN * NT) * 32 * BKT * ES) * (WM * WN) / 4 > 0 \
                        ? ring_blocks_per_cu(D * (WM * MT + WN * NT) * 32 * BKT * ES) * (WM * WN) / 4 : 1)

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED = false>
__global__ C2D_RING_BOUNDS void igemm_ring_kernel(IgemmArgs a) {
  igemm_ring_body<MODE, WM, WN, MT, NT, PM, BKT, D, ES, FUSED>(a, (int)blockIdx.x, (int)gridDim.x);
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
template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED = false>
__global__ C2D_RING_BOUNDS void igemm_ring_group_kernel(IgemmRingGroup g) {
  int p = 0;
  for (int i = 1; i < g.num; ++i)
    if ((int)blockIdx.x >= g.first[i]) p = i;
  p = __builtin_amdgcn_readfirstlane(p);
  igemm_ring_body<MODE, WM, WN, MT, NT, PM, BKT, D, ES, FUSED>(g.a[p], (int)blockIdx.x - g.first[p],
                                                                g.first[p + 1] - g.first[p]);
}
#undef C2D_RING_BOUNDS

// ---- host side ---------------------------------------------------------------------------------
// Tuning hooks (read once, only with C2D_TUNE=1): C2D_RING_BK=32|64, C2D_RING_D=2|3 (more in a
// -DC2D_RING_SWEEP build) force the stage depth / ring depth where an instance exists;
// C2D_RING=0 makes every bf16 convolution fail with C2D_ERR_UNSUPPORTED (there is no second bf16
// GEMM: a check that nothing falls back silently).
#ifdef C2D_RING_TRACE
unsigned long long* g_ring_trace = nullptr;
#endif
struct RingTune { int bk, d, off; };
const RingTune& ring_tune() {
  static const RingTune t = [] {
    RingTune r = {0, 0, 0};
    if (getenv("C2D_TUNE")) {
      if (const char* e = getenv("C2D_RING_BK")) r.bk = atoi(e);
      if (const char* e = getenv("C2D_RING_D")) r.d = atoi(e);
      if (const char* e = getenv("C2D_RING")) r.off = e[0] == '0';
    }
    return r;
  }();
  return t;
}

// Collector of ring_group_begin() .. ring_group_end(): launches of group-eligible instances are held
// back (every problem with the launchers of ITS instance) and leave as one grouped launch when they
// all chose the same instance, one by one otherwise.
struct RingCollect {
  bool active;
  int num;
  IgemmArgs a[4];
  int (*single[4])(const IgemmArgs&, hipStream_t);
  int (*group[4])(const IgemmRingGroup&, hipStream_t);
};
thread_local RingCollect g_ring_collect = {false, 0, {}, {}, {}};

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED>
int ring_launch_single(const IgemmArgs& a, hipStream_t s) {
  // (spelled as rocprofv3 prints the instance: the FUSED flag included)
  dispatch_note_ext(PM ? (FUSED ? "igemm_ring_kernel<%d, %d, %d, %d, %d, true, %d, %d, %d, true>"
                                : "igemm_ring_kernel<%d, %d, %d, %d, %d, true, %d, %d, %d, false>")
                       : (FUSED ? "igemm_ring_kernel<%d, %d, %d, %d, %d, false, %d, %d, %d, true>"
                                : "igemm_ring_kernel<%d, %d, %d, %d, %d, false, %d, %d, %d, false>"),
                    MODE, WM, WN, MT, NT, BKT, D, ES);
  hipLaunchKernelGGL((igemm_ring_kernel<MODE, WM, WN, MT, NT, PM, BKT, D, ES, FUSED>),
                     dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, s, a);
  return c2d_launch_status();
}

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED>
int ring_launch_group(const IgemmRingGroup& g, hipStream_t s) {
  dispatch_note_ext(FUSED ? "igemm_ring_group_kernel<%d, %d, %d, %d, %d, false, %d, %d, %d, true>"
                          : "igemm_ring_group_kernel<%d, %d, %d, %d, %d, false, %d, %d, %d, false>",
                    MODE, WM, WN, MT, NT, BKT, D, ES);
  hipLaunchKernelGGL((igemm_ring_group_kernel<MODE, WM, WN, MT, NT, PM, BKT, D, ES, FUSED>),
                     dim3(g.first[g.num]), dim3(WM * WN * 64), 0, s, g);
  return c2d_launch_status();
}

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES = 2>
int launch_one(IgemmArgs a, hipStream_t s) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  a.m_tiles = c2d_ceil_div(a.M, BM);
  a.n_tiles = c2d_ceil_div(a.N, BN);
  if (a.nseg > 1) {
    a.total_slabs = 0;
    for (int i = 0; i < a.nseg; ++i) a.total_slabs += c2d_ceil_div(a.segK[i], BKT);
  }
  static const int dbg_env = (getenv("C2D_TUNE") && getenv("C2D_IGEMM_DBG")) ? atoi(getenv("C2D_IGEMM_DBG")) : 0;
  a.dbg = dbg_env;
#ifdef C2D_RING_TRACE
  a.trace = g_ring_trace;
#endif
  // the row-major input-gradient instances of the stride-2 layers' tiles can leave in a grouped launch
  constexpr bool GROUPABLE = MODE == 1 && ES == 2 && !PM && WM == 2 && MT <= 2 &&
                             ((WN == 4 && MT == 2 && NT == 2) || (WN == 2 && NT == 1));
  if constexpr (MODE == 1 && ES == 2) {
    if (a.fy != nullptr) {      // the producer's BN/ReLU backward rides in the epilogue
      if constexpr (GROUPABLE) {
        RingCollect& c = g_ring_collect;
        if (c.active && c.num < 4) {
          c.a[c.num] = a;
          c.single[c.num] = &ring_launch_single<MODE, WM, WN, MT, NT, PM, BKT, D, ES, true>;
          c.group[c.num] = &ring_launch_group<MODE, WM, WN, MT, NT, PM, BKT, D, ES, true>;
          ++c.num;
          return C2D_OK;
        }
      }
      return ring_launch_single<MODE, WM, WN, MT, NT, PM, BKT, D, ES, true>(a, s);
    }
  } else if (a.fy != nullptr) {
    return C2D_ERR_UNSUPPORTED;   // (fp32 networks fuse in igemm_nt_kernel, conv_gemm.hip)
  }
  if constexpr (GROUPABLE) {
    RingCollect& c = g_ring_collect;
    if (c.active && c.num < 4) {
      c.a[c.num] = a;
      c.single[c.num] = &ring_launch_single<MODE, WM, WN, MT, NT, PM, BKT, D, ES, false>;
      c.group[c.num] = &ring_launch_group<MODE, WM, WN, MT, NT, PM, BKT, D, ES, false>;
      ++c.num;
      return C2D_OK;
    }
  }
  return ring_launch_single<MODE, WM, WN, MT, NT, PM, BKT, D, ES, false>(a, s);
}

// (stage depth, ring depth) of a block tile, measured per GEMM call of the step (tools/
// sweep_step_gemms.sh + tools/sweep_cold.py, N = 2000 ROIs; with the operands warm from the previous
// repetition and, as inside a step, cold: C2D_BENCH_COLD=1).  What decides is how many workgroups
// share a CU, not how deep one workgroup prefetches: these GEMMs are short (K = 128 .. 2304: 2 - 36
// stages), so a workgroup spends as long in its ramp (first stages in flight) and its epilogue as
// in its K loop, and only ANOTHER workgroup's K loop fills the matrix pipe meanwhile.  Rings of
// four or more stages (one workgroup per CU) measured 10 - 45 % slower on every call.  So:
//   * 128x256 (8 waves): 32-deep stages in a ring of three (72 KiB: TWO workgroups per CU) —
//     first for launches of more than 1.5 rounds of the chip (block-entry input gradients 109 ->
//     84 us, 192->256 3x3 on 7x7 maps 126 -> 108 us); since the steady-state loop (half the
//     instructions per stage) also for the one-round launches of 4x4 maps (224->224 3x3 forward /
//     input gradient 47 -> 41 / 46 -> 39 us cold, 38.6 -> 34.5 / 37.4 -> 33.2 warm);
//   * 128x192 (8 waves, full width): the same ring for one-round launches (61 KiB: two per CU;
//     160->224 / 192->224 input gradients 41 -> 37.5 us cold); 80 KiB of 64-deep stages otherwise;
//   * 128x64 (4 waves): many-tile launches likewise (36 KiB: four per CU; 576-wide entry gradient
//     142 -> 102 us);
//   * everything else: 64-deep stages, two buffers.
#ifdef C2D_RING_SWEEP
#define C2D_RING_COMBOS(X) X(64, 2) X(32, 3) X(64, 3) X(32, 4)
#else
#define C2D_RING_COMBOS(X) X(64, 2) X(32, 3)
#endif
template <int MODE, int WM, int WN, int MT, int NT, bool PM>
int launch_tile(const IgemmArgs& a, hipStream_t s) {
  constexpr int ROWS = (WM * MT + WN * NT) * 32;
  constexpr int LDS = 160 * 1024;
  const RingTune& t = ring_tune();
  const long long blocks = (long long)c2d_ceil_div(a.M, WM * MT * 32) * c2d_ceil_div(a.N, WN * NT * 32);
  int bk = 64, d = 2;
  if (WM == 2 && WN == 4) { bk = 32; d = 3; }
  if (WM == 4 && WN == 2 && NT == 3 && blocks <= 384) { bk = 32; d = 3; }
  if (WM == 2 && WN == 2 && MT == 2 && NT == 1 && blocks > 768) { bk = 32; d = 3; }
  if (t.bk) bk = t.bk;
  if (t.d) d = t.d;
#define C2D_RING_CASE(BKT_, D_)                                                                \
  if constexpr (D_ * ROWS * BKT_ * 2 <= LDS && (D_ - 2) * (ROWS * BKT_ * 2 / (WM * WN * 1024) + 1) <= 63) \
    if (bk == BKT_ && d == D_) return launch_one<MODE, WM, WN, MT, NT, PM, BKT_, D_>(a, s);
  C2D_RING_COMBOS(C2D_RING_CASE)
#undef C2D_RING_CASE
  // (a forced combination that does not exist for this tile: the default)
  return launch_one<MODE, WM, WN, MT, NT, PM, 64, 2>(a, s);
}

template <int WM, int WN, int MT, int NT>
int launch_shape(const IgemmArgs& a, bool pm, hipStream_t s) {
  if (a.g.mode == 0)
    return pm ? launch_tile<0, WM, WN, MT, NT, true>(a, s) : launch_tile<0, WM, WN, MT, NT, false>(a, s);
  return pm ? launch_tile<1, WM, WN, MT, NT, true>(a, s) : launch_tile<1, WM, WN, MT, NT, false>(a, s);
}

// fp32 operands on the same ring.  Swept on the step's 26 forward / input-gradient calls (tools/
// bench_step_gemms.py fp32; 16- / 32-float stages, rings of 2-4): the register-staged
// igemm_nt_kernel of conv_gemm.hip stays ahead by 3-6 % overall (3.66 + 3.57 ms against 3.77 + 3.79
// with the best ring) — at 64 cycles per fp32 MFMA the staging instructions of a workgroup sit
// under the MFMAs of the three others on its CU — with ONE exception the default takes: the 3x3
// forward convolutions with 257..384 output columns on the per-ROI maps (192->320 of Mixed_5b/5c:
// five 64-column tiles per row block), 284 -> 234 us per call with 16-float stages in two buffers
// (24 KiB: five workgroups per CU).  C2D_TUNE=1 C2D_RING_FP32=1|0 forces the ring on (all tiles,
// sweep build) / off; C2D_RINGF_BK = 16 | 32, C2D_RINGF_D = ring depth (sweep build).
struct RingF32Tune { int on, bk, d; };
const RingF32Tune& ringf_tune() {
  static const RingF32Tune t = [] {
    RingF32Tune r = {-1, 0, 0};
    if (getenv("C2D_TUNE")) {
      if (const char* e = getenv("C2D_RING_FP32")) r.on = atoi(e);
      if (const char* e = getenv("C2D_RINGF_BK")) r.bk = atoi(e);
      if (const char* e = getenv("C2D_RINGF_D")) r.d = atoi(e);
    }
    return r;
  }();
  return t;
}

#ifdef C2D_RING_SWEEP
template <int MODE, int WM, int WN, int MT, int NT, bool PM>
int launch_tile_f32(const IgemmArgs& a, hipStream_t s) {
  const RingF32Tune& t = ringf_tune();
  const int bk = t.bk == 32 ? 32 : 16;
  const int d = t.d >= 2 && t.d <= 4 ? t.d : 2;
  if (bk == 16 && d == 3) return launch_one<MODE, WM, WN, MT, NT, PM, 16, 3, 4>(a, s);
  if (bk == 16 && d == 4) return launch_one<MODE, WM, WN, MT, NT, PM, 16, 4, 4>(a, s);
  if (bk == 16 && d == 2) return launch_one<MODE, WM, WN, MT, NT, PM, 16, 2, 4>(a, s);
  if (bk == 32 && d == 3) return launch_one<MODE, WM, WN, MT, NT, PM, 32, 3, 4>(a, s);
  return launch_one<MODE, WM, WN, MT, NT, PM, 32, 2, 4>(a, s);
}

template <int WM, int WN, int MT, int NT>
int launch_shape_f32(const IgemmArgs& a, bool pm, hipStream_t s) {
  if (a.g.mode == 0)
    return pm ? launch_tile_f32<0, WM, WN, MT, NT, true>(a, s) : launch_tile_f32<0, WM, WN, MT, NT, false>(a, s);
  return pm ? launch_tile_f32<1, WM, WN, MT, NT, true>(a, s) : launch_tile_f32<1, WM, WN, MT, NT, false>(a, s);
}
#endif

}  // namespace

int launch_igemm_f32_ring(const IgemmArgs& a, int wm, int wn, int mt, int nt, bool pm,
                          hipStream_t s, int* m_tiles_out, bool query) {
  const int on = ringf_tune().on;
  if (on == 0) return C2D_ERR_UNSUPPORTED;
  const int key = ((wm * 10 + wn) * 10 + mt) * 10 + nt;
  const bool pick = key == 2221 && pm && a.g.mode == 0 && a.N > 256 && a.N <= 384 && !a.fy && !a.mo_n &&
                    a.nseg == 1;
#ifdef C2D_RING_SWEEP
  if (on == 1) {
    if (key != 2221 && key != 2222 && key != 2211) return C2D_ERR_UNSUPPORTED;
    if (m_tiles_out) *m_tiles_out = c2d_ceil_div(a.M, wm * mt * 32);
    if (query) return C2D_OK;
    switch (key) {
      case 2221: return launch_shape_f32<2, 2, 2, 1>(a, pm, s);      // 128 x 64
      case 2222: return launch_shape_f32<2, 2, 2, 2>(a, pm, s);      // 128 x 128
      default: return launch_shape_f32<2, 2, 1, 1>(a, pm, s);        // 64 x 64
    }
  }
#endif
  if (!pick) return C2D_ERR_UNSUPPORTED;
  if (m_tiles_out) *m_tiles_out = c2d_ceil_div(a.M, wm * mt * 32);
  if (query) return C2D_OK;
  return launch_one<0, 2, 2, 2, 1, true, 16, 2, 4>(a, s);
}

namespace {
}  // namespace

void ring_group_begin() {
  g_ring_collect.active = true;
  g_ring_collect.num = 0;
}

int ring_group_end(hipStream_t s) {
  RingCollect& c = g_ring_collect;
  c.active = false;
  const int num = c.num;
  c.num = 0;
  if (num == 0) return C2D_OK;
  bool same = num > 1;
  for (int i = 1; i < num; ++i) same = same && c.group[i] == c.group[0];
  if (!same) {
    for (int i = 0; i < num; ++i) {
      const int rc = c.single[i](c.a[i], s);
      if (rc) return rc;
    }
    return C2D_OK;
  }
  IgemmRingGroup g;
  g.num = num;
  int blocks = 0;
  for (int i = 0; i < 4; ++i) {
    g.first[i] = blocks;
    if (i < num) {
      g.a[i] = c.a[i];
      blocks += c.a[i].m_tiles * c.a[i].n_tiles;
    } else {
      g.a[i] = c.a[0];
    }
  }
  g.first[4] = blocks;
  for (int i = num; i < 4; ++i) g.first[i] = blocks;
  return c.group[0](g, s);
}

int launch_igemm_bf16_ring(const IgemmArgs& a, int wm, int wn, int mt, int nt, bool pm,
                           hipStream_t s, int* m_tiles_out, bool query) {
  if (ring_tune().off) return C2D_ERR_UNSUPPORTED;
  const int bm = wm * mt * 32;
  int rc = C2D_ERR_UNSUPPORTED;
  const int key = ((wm * 10 + wn) * 10 + mt) * 10 + nt;
  switch (key) {
    case 2221: case 2422: case 4212: case 4213: case 4214: case 4215: case 4216: case 2222: case 2211:
      break;
    default:
      return C2D_ERR_UNSUPPORTED;
  }
  if (m_tiles_out) *m_tiles_out = c2d_ceil_div(a.M, bm);
  if (query) return C2D_OK;
  switch (key) {
    case 2221: rc = launch_shape<2, 2, 2, 1>(a, pm, s); break;      // 128 x 64, 4 waves
    case 2222: rc = launch_shape<2, 2, 2, 2>(a, pm, s); break;      // 128 x 128, 4 waves
    case 2211: rc = launch_shape<2, 2, 1, 1>(a, pm, s); break;      // 64 x 64, 4 waves
    case 2422: rc = launch_shape<2, 4, 2, 2>(a, pm, s); break;      // 128 x 256, 8 waves (2 x 4)
    case 4212: rc = launch_shape<4, 2, 1, 2>(a, pm, s); break;      // 128 x full width, 8 waves (4 x 2)
    case 4213: rc = launch_shape<4, 2, 1, 3>(a, pm, s); break;
    case 4214: rc = launch_shape<4, 2, 1, 4>(a, pm, s); break;
    case 4215: rc = launch_shape<4, 2, 1, 5>(a, pm, s); break;
    case 4216: rc = launch_shape<4, 2, 1, 6>(a, pm, s); break;
  }
  return rc;
}


// ---------------------------------------------------------------------------------------------
// Filter gradient of a 1x1 / stride-1 convolution, bf16 operands (Conv2DBackpropFilter of the
// block-entry and branch-closing convolutions of Mixed_5a-c, train/trainer.py:141-146):
//   dW[i][j] += sum_m x[m][i] * dC[m][j]        (K = rows, split over workgroups)
// Round 2's kernel (wgrad_tn_bf16_kernel) staged 32-row slabs through registers with two barriers
// per slab and ran at ~270 TFLOP/s on these calls.  Here the same 128 x (64 | 128) output tile gets
//   * its operand stages global -> LDS by DMA (no staging registers, no ds_write), a ring of D
//     stages of BKT rows, ONE barrier per stage (as igemm_ring_kernel);
//   * 64-row stages in two buffers (64 KiB: two workgroups per CU);
//   * the row-major tiles read transposed with ds_read_b64_tr_b16 (cdna_hip_programming.md T10)
//     from a lane-linear DMA image: the 64-byte granule g of row r sits at position g ^ (r & 3)
//     (256-byte rows) / g ^ ((r >> 1) & 1) (128-byte rows), applied to the DMA's source address
//     and to the reads, so that the four rows a 32-lane half reads hit four different bank groups.
// ---------------------------------------------------------------------------------------------
namespace {
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag2(const char* lo_p, const char* hi_p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo_p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi_p);
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NTJ, int BKT, int D>
__device__ __forceinline__ void wgrad1x1_bf16_ring_body(const WgradArgs& a, const WgradBlock blk) {
  constexpr int BJ = NTJ * 64;                         // output columns per block (64 or 128)
  constexpr int A_RB = 256, G_RB = BJ * 2;             // bytes per staged row
  constexpr int A_BYTES = BKT * A_RB, G_BYTES = BKT * G_RB;
  constexpr int A_LOADS = A_BYTES / 4096;              // a pass = 256 lanes x 16 B = 4 KiB
  constexpr int G_LOADS = G_BYTES / 4096;
  constexpr int PER = A_LOADS + G_LOADS;
  constexpr int KS = BKT / 16;
  static_assert(A_BYTES % 4096 == 0 && G_BYTES % 4096 == 0 && D >= 2 && D <= 4, "stage geometry");
  __shared__ __attribute__((aligned(1024))) char smem[D * (A_BYTES + G_BYTES)];
  char* const smemG = smem + D * A_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int itiles = (a.I + 127) / 128;
  const int i0 = (blk.x % itiles) * 128;
  const int j0 = blk.y * BJ;
  const int mbeg = blk.z * a.rows_per_split;
  const int cnt = (min(a.rows_per_split, a.M - mbeg) + BKT - 1) / BKT;

  // DMA lanes: A pass = 16 rows x 16 chunks; G pass = 16 rows x 16 chunks (BJ = 128) or 32 rows x
  // 8 chunks (BJ = 64).  The lane at LDS position p of row r fetches source chunk p ^ swz(r).
  const int arow = tid >> 4, apos = tid & 15;
  const int achunk = apos ^ ((arow & 3) << 2);
  constexpr int GCPR = G_RB / 16;                      // chunks per dC row (16 or 8)
  const int grow = tid / GCPR, gpos = tid % GCPR;
  const int gchunk = GCPR == 16 ? gpos ^ ((grow & 3) << 2) : gpos ^ (((grow >> 1) & 1) << 2);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * 2,
                                                 (a.a_rows * a.lda - a.a_off) * 2);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * 2,
                                                 ((long long)a.M * a.ldg - a.g_off) * 2);
  // (columns beyond I / J are clamped: their products land in dW rows / columns never stored)
  const unsigned aoff = (unsigned)(arow * a.lda + min(i0 + achunk * 8, a.I - 8)) * 2u;
  const unsigned goff = (unsigned)(grow * a.ldg + min(j0 + gchunk * 8, a.J - 8)) * 2u;
  constexpr int A_PASS_ROWS = 16, G_PASS_ROWS = 256 / GCPR;

  f32x16 acc[2][NTJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  unsigned tile_bits = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
      if ((i0 + wm * 64 + i * 32 < a.I) && (j0 + (wn * NTJ + j) * 32 < a.J))
        tile_bits |= 1u << (i * NTJ + j);
  tile_bits = __builtin_amdgcn_readfirstlane(tile_bits);

  // stage `st` (rows mbeg + st * BKT ...) into ring slot `slot`; rows >= M lie outside the
  // descriptors and come back as zeros (splits end on stage boundaries: no row is counted twice)
#define C2D_W_ISSUE(SLOT, ST)                                                                  \
  {                                                                                            \
    const int m0 = mbeg + (ST) * BKT;                                                          \
    _Pragma("unroll") for (int p = 0; p < A_LOADS; ++p)                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                \
          rsA, (lds_void_t*)(smem + (SLOT) * A_BYTES + p * 4096 + wave * 1024), 16, (int)aoff, \
          (m0 + p * A_PASS_ROWS) * a.lda * 2, 0, 0);                                           \
    _Pragma("unroll") for (int p = 0; p < G_LOADS; ++p)                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                \
          rsG, (lds_void_t*)(smemG + (SLOT) * G_BYTES + p * 4096 + wave * 1024), 16, (int)goff, \
          (m0 + p * G_PASS_ROWS) * a.ldg * 2, 0, 0);                                           \
  }
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (d < cnt) C2D_W_ISSUE(d, d)

  // transposed-read addresses: k rows 16 s + 8 lh + tq (+4), 64-byte granule of the tile's columns
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  int a_lo[2], g_lo[NTJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    a_lo[i] = (8 * lh + tq) * A_RB + (((wm * 2 + i) ^ tq) << 6) + 32 * tg + 8 * tp;
#pragma unroll
  for (int j = 0; j < NTJ; ++j)
    g_lo[j] = (8 * lh + tq) * G_RB +
              (((wn * NTJ + j) ^ (GCPR == 16 ? tq : (tq >> 1))) << 6) + 32 * tg + 8 * tp;

  int slot = 0, slot_in = D - 1;
  for (int it = 0; it < cnt; ++it) {
    const int ahead = min(cnt, it + D - 1) - it - 1;
    if (D > 3 && ahead >= 2) wait_vmcnt<(D > 3 ? 2 : 0) * PER>();
    else if (D > 2 && ahead >= 1) wait_vmcnt<(D > 2 ? 1 : 0) * PER>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (it + D - 1 < cnt) C2D_W_ISSUE(slot_in, it + D - 1)
    const char* const As = smem + slot * A_BYTES;
    const char* const Gs = smemG + slot * G_BYTES;
    bf16x8 af[2][KS], bf[NTJ][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i][s] = tr_frag2(As + s * 16 * A_RB + a_lo[i], As + (s * 16 + 4) * A_RB + a_lo[i]);
#pragma unroll
      for (int j = 0; j < NTJ; ++j)
        bf[j][s] = tr_frag2(Gs + s * 16 * G_RB + g_lo[j], Gs + (s * 16 + 4) * G_RB + g_lo[j]);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTJ; ++j)
        if ((tile_bits >> (i * NTJ + j)) & 1u) {
#pragma unroll
          for (int s = 0; s < KS; ++s)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    if (++slot == D) slot = 0;
    if (++slot_in == D) slot_in = 0;
  }
#undef C2D_W_ISSUE

  // split-K result: fp32 atomics into dW, or (part_stride > 0) plain stores into this split's slab
  float* dw = a.dW + (size_t)blk.z * a.part_stride;
  const bool part = a.part_stride > 0;
#pragma unroll
  for (int j = 0; j < NTJ; ++j) {
    const int jj = j0 + (wn * NTJ + j) * 32 + li;
    if (jj >= a.J) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ii >= a.I) continue;
        if (part) dw[(size_t)ii * a.J + jj] = acc[i][j][r];
        else atomicAdd(dw + (size_t)ii * a.J + jj, acc[i][j][r]);
      }
  }
}

#define C2D_WRING_BOUNDS(NTJ, BKT, D)                                                            \
  __launch_bounds__(256, (160 * 1024) / (D * BKT * (128 + NTJ * 64) * 2) >= 3                      \
                             ? 3 : ((160 * 1024) / (D * BKT * (128 + NTJ * 64) * 2) >= 2 ? 2 : 1))
template <int NTJ, int BKT, int D>
__global__ C2D_WRING_BOUNDS(NTJ, BKT, D) void wgrad1x1_bf16_ring_kernel(WgradArgs a) {
  wgrad1x1_bf16_ring_body<NTJ, BKT, D>(a, wgrad_block(a));
}
// several filter gradients of one input in one launch (see WgradGroupArgs)
template <int NTJ, int BKT, int D>
__global__ C2D_WRING_BOUNDS(NTJ, BKT, D) void wgrad1x1_bf16_ring_group_kernel(WgradGroupArgs g) {
  int p;
  const WgradBlock blk = wgrad_group_block(g, &p);
  wgrad1x1_bf16_ring_body<NTJ, BKT, D>(g.a[p], blk);
}
#undef C2D_WRING_BOUNDS

struct WgradTune { int bk, d, slots, off; };
const WgradTune& wgrad_tune() {
  static const WgradTune t = [] {
    WgradTune r = {0, 0, 0, 0};
    if (getenv("C2D_TUNE")) {
      if (const char* e = getenv("C2D_WRING_BK")) r.bk = atoi(e);
      if (const char* e = getenv("C2D_WRING_D")) r.d = atoi(e);
      if (const char* e = getenv("C2D_WRING_SLOTS")) r.slots = atoi(e);
      if (const char* e = getenv("C2D_WRING")) r.off = e[0] == '0';
    }
    return r;
  }();
  return t;
}
}  // namespace

int launch_wgrad1x1_bf16_ring(WgradArgs a, hipStream_t s, int* splits_out, bool splits_only) {
  const WgradTune& t = wgrad_tune();
  if (t.off) return C2D_ERR_UNSUPPORTED;
  if (a.I % 8 != 0 || a.J % 8 != 0 || a.I < 8 || a.J < 8 || a.lda % 8 != 0 || a.ldg % 8 != 0 ||
      a.a_off % 8 != 0 || a.g_off % 8 != 0)
    return C2D_ERR_UNSUPPORTED;
  // Measured per call of the step (tools/sweep_wgrad.sh, N = 2000): these launches are bound by
  // their split-K atomics as much as by staging — every workgroup adds a whole 128 x 128 fp32 tile
  // (16.8 MB per 256 workgroups at the chip-wide 1.3 TB/s of float atomics, all at the end of the
  // launch) — so the DMA ring only pays where the tile count keeps the splits low: output widths
  // of 129..256 columns (576->192: 68 -> 49 us, 1024->192: 40 -> 33, 1024->160: 39 -> 30);
  // 128-wide layers stay on the two-K-group kernel of conv_gemm.hip (30 us); the 352-wide ones
  // measured equal with warm operands (49 us either way) and 78 -> 65 us with cold ones
  // (C2D_BENCH_COLD=1: what the step sees), so they take the ring as well.
  if (!t.bk && !t.d && !t.slots && !(a.J > 128 && a.J <= 384)) return C2D_ERR_UNSUPPORTED;
  const bool narrow = a.J % 128 != 0 && a.J % 128 <= 64;    // 128 x 64 block tiles
  const int bj = narrow ? 64 : 128;
  const int bk = t.bk == 32 ? 32 : 64;
  const int d = t.d >= 2 && t.d <= 4 ? t.d : 2;
  a.tiles_x = c2d_ceil_div(a.I, 128);
  a.tiles_y = c2d_ceil_div(a.J, bj);
  const int tiles = a.tiles_x * a.tiles_y;
  // two workgroups per CU in one round; at least 4 stages per workgroup
  int splits = (t.slots > 0 ? t.slots : 512) / tiles;
  const int max_splits = c2d_ceil_div(a.M, 4 * bk);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.rows_per_split = c2d_ceil_div(c2d_ceil_div(a.M, splits), bk) * bk;
  a.nsplits = c2d_ceil_div(a.M, a.rows_per_split);
  if (splits_out) *splits_out = a.nsplits;
  if (splits_only) return C2D_OK;
  const dim3 grid(tiles * a.nsplits), block(256);
  dispatch_note_ext("wgrad1x1_bf16_ring_kernel<%d, %d, %d>", narrow ? 1 : 2, bk, d);
#define C2D_WR(NTJ_, BK_, D_)                                                                   \
  if ((narrow ? 1 : 2) == NTJ_ && bk == BK_ && d == D_) {                                      \
    hipLaunchKernelGGL((wgrad1x1_bf16_ring_kernel<NTJ_, BK_, D_>), grid, block, 0, s, a);      \
    return c2d_launch_status();                                                                \
  }
  C2D_WR(1, 64, 2) C2D_WR(2, 64, 2) C2D_WR(1, 32, 3) C2D_WR(2, 32, 3)
#undef C2D_WR
  return C2D_ERR_UNSUPPORTED;
}


// Grouped form (c2d_conv1x1_wgrad_multi_bf16): `num` 1x1 / stride-1 filter gradients that share
// the activations x; 128-column tiles for every output, ONE split count from the tiles of all.
int launch_wgrad1x1_bf16_ring_group(WgradArgs* a, int num, hipStream_t s) {
  if (num < 1 || num > WGRAD_GROUP_MAX) return C2D_ERR_UNSUPPORTED;
  WgradGroupArgs g;
  int tiles = 0;
  for (int p = 0; p < num; ++p) {
    if (a[p].I % 8 != 0 || a[p].J % 8 != 0 || a[p].I < 8 || a[p].J < 8 || a[p].lda % 8 != 0 ||
        a[p].ldg % 8 != 0 || a[p].a_off % 8 != 0 || a[p].g_off % 8 != 0 || a[p].M != a[0].M)
      return C2D_ERR_UNSUPPORTED;
    a[p].tiles_x = c2d_ceil_div(a[p].I, 128);
    a[p].tiles_y = c2d_ceil_div(a[p].J, 128);
    g.first_tile[p] = tiles;
    tiles += a[p].tiles_x * a[p].tiles_y;
  }
  for (int p = num; p <= WGRAD_GROUP_MAX; ++p) g.first_tile[p] = tiles;
  const WgradTune& t = wgrad_tune();
  int splits = (t.slots > 0 ? t.slots : 512) / tiles;       // two workgroups per CU in one round
  const int max_splits = c2d_ceil_div(a[0].M, 4 * 64);      // at least 4 stages per workgroup
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int rps = c2d_ceil_div(c2d_ceil_div(a[0].M, splits), 64) * 64;
  g.nsplits = c2d_ceil_div(a[0].M, rps);
  g.num = num;
  for (int p = 0; p < num; ++p) {
    a[p].rows_per_split = rps; a[p].nsplits = g.nsplits; a[p].part_stride = 0;
    a[p].a_rows = a[p].M;
    g.a[p] = a[p];
  }
  dispatch_note_ext("wgrad1x1_bf16_ring_group_kernel<2, 64, 2>");
  hipLaunchKernelGGL((wgrad1x1_bf16_ring_group_kernel<2, 64, 2>), dim3(tiles * g.nsplits), dim3(256), 0, s, g);
  return c2d_launch_status();
}

}  // namespace c2d_ig

#ifdef C2D_RING_TRACE
// Diagnostic build only (make EXTRA_CXXFLAGS=-DC2D_RING_TRACE; tools/trace_ring.py): 16 x u64 per
// block — start, prologue, K loop, epilogue, time at wait + barrier, time issuing DMA (cycles of
// wave 0), stages, s_memrealtime at the end.
extern "C" int c2d_debug_set_ring_trace(void* buf) {
  c2d_ig::g_ring_trace = (unsigned long long*)buf;
  return C2D_OK;
}
#endif
