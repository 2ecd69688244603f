// Strip form of the stride-1 3x3 gather-GEMM for the high-resolution, 32-channel layers (the last Conv2DTranspose
// of the decoder, model.py:120-135, forward and data gradient, and the output head conv, model.py:137).
//
// In gconv2.hip every K chunk re-gathers its A tile from L2: a 3x3 layer reads its input nine times.  For the
// 64 x 64 x 32 layers that is 1.2 GB of L2 -> LDS traffic per launch against 125 us of MFMA work and 9 K chunks per
// tile - gather and prologue bound (77-90 TFLOP/s; the head, N = 16, 50).  Here a workgroup walks strips of R output
// rows of one stamp (R * W <= 256 pixels): the (R+2) x (W+2) x 32 input patch arrives ONCE by LDS-DMA
// (global_load_lds_dwordx4, zero page outside the image) into a double-buffered LDS ring while the previous strip is
// being multiplied, all nine taps read their A fragments from it at shifted addresses, and the layer's whole weight
// tensor (9 x 32 x Cout) stays in LDS for the workgroup's lifetime.  There is no K loop to pipeline: 18 fragment
// steps (tap x 16-channel half) of straight-line code per strip.
//
// LDS layouts: patch [pixel][8 quads of 4 channels], weights [tap][n][8 quads of 4 k]; the quad index is XOR-ed with
// ((pixel or n) >> 1) & 7 so that the ds_read_b128 fragment reads of 16 consecutive pixels / columns are conflict
// free (the DMA applies the swizzle on the global side: the lane that owns LDS slot s of pixel L fetches channel
// quad s ^ swz(L)).  Eight waves, each 32 pixels x Cout; epilogue through a per-wave LDS tile (bias / PReLU on
// float4 rows) as in gconv2.hip.
#include "common.h"
#include <stdlib.h>

namespace dv {

typedef const __attribute__((address_space(1))) void* gs_gptr_t;
typedef __attribute__((address_space(3))) void* gs_lptr_t;

namespace {
constexpr int GS_WAVES = 8;
constexpr int GS_THREADS = 64 * GS_WAVES;
constexpr int GS_MAXG = 8;       // 1-KiB DMA pieces per wave and strip
constexpr int GS_LDC = 36;
__device__ __forceinline__ int gs_dh(unsigned long long code, int t) { return (int)((code >> (4 * t)) & 3) - 1; }
__device__ __forceinline__ int gs_dw(unsigned long long code, int t) { return (int)((code >> (4 * t + 2)) & 3) - 1; }
__device__ __forceinline__ int gs_wt(unsigned long long code, int t) { return (int)((code >> (4 * t)) & 15); }
}  // namespace

// CIN: input channels (32, or 16 for the head's data gradient); COUT: columns held in LDS / accumulators (16 or 32,
// >= p.Cout); NMAJOR: weights stored W[wt][n][k] (else W[wt][k][n])
template <int CIN, int COUT, bool NMAJOR>
__global__ __launch_bounds__(GS_THREADS, 1) void gconv_strip_kernel(const GStripParams p) {
  constexpr int CQ = CIN / 4;                     // channel quads per pixel (8 or 4)
  constexpr int QG = CIN / 16;                    // 16-channel fragment groups per tap
  constexpr int NSTEPS = 9 * QG;
  // quad swizzle: rows (pixels, weight columns) are CQ*16 bytes apart; 16 consecutive rows must cover 256 bytes
  constexpr int SWS = CQ == 8 ? 1 : 2;
#define GS_SW(idx) (((idx) >> SWS) & (CQ - 1))
  constexpr int TN = COUT / 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                                          // [2][patch_floats]
  float* wts = smem + 2 * p.patch_floats;                       // [9][COUT][32]
  float* stage = wts + 9 * COUT * CIN;                          // [8 waves][16][GS_LDC]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations (M0) and piece guards stay on the scalar unit
  const int l15 = lane & 15, lg = lane >> 4;
  const int W = p.Wd, PW = W + 2;
  const int strip_px = p.R * W;

  // ---- weights -> LDS (once), laid out [tap][n][k] with the quad swizzle -----------------------------------
  for (int e = tid; e < 9 * COUT * CQ; e += GS_THREADS) {
    const int t = e / (COUT * CQ), rem = e - t * (COUT * CQ);
    const int wt = gs_wt(p.wtcode, t);
    if (NMAJOR) {
      const int n = rem / CQ, kq = rem - n * CQ;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n < p.Cout) v = *reinterpret_cast<const f32x4*>(p.W + ((size_t)(wt * p.Cout + n) * CIN + kq * 4));
      *reinterpret_cast<f32x4*>(wts + ((t * COUT + n) * CQ + (kq ^ GS_SW(n))) * 4) = v;
    } else {
      // k-major source: a float4 along n is scattered to four rows of the [n][k] tile
      const int k = rem / (COUT / 4), n4 = rem - k * (COUT / 4);
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n4 * 4 < p.Cout) v = *reinterpret_cast<const f32x4*>(p.W + ((size_t)(wt * CIN + k) * p.Cout + n4 * 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n4 * 4 + j;
        wts[((t * COUT + n) * CQ + ((k >> 2) ^ GS_SW(n))) * 4 + (k & 3)] = v[j];
      }
    }
  }

  // ---- strip-invariant part of the patch DMA: slot e = 64 * (wave + 8 k) + lane -> (patch pixel L, quad slot s) ----
  const int ptot = (p.R + 2) * PW * CQ;            // 16-byte slots of one patch
  const int ngp = (ptot + 63) / 64;
  int prel[GS_MAXG], prow[GS_MAXG];
#pragma unroll
  for (int k = 0; k < GS_MAXG; ++k) {
    const int e = 64 * (wave + GS_WAVES * k) + lane;
    const int L = e / CQ, s = e - L * CQ;
    const int pr = L / PW, pc = L - pr * PW;
    const int gc = pc - 1;
    const bool ok = e < ptot && (unsigned)gc < (unsigned)W;
    prel[k] = ((pr - 1) * W + gc) * CIN + ((s ^ GS_SW(L)) << 2);
    prow[k] = ok ? pr - 1 : -100000;
  }
  auto issue_dma = [&](int sidx, int buf) {
    const int n = sidx / p.strips_per_stamp;
    const int i0 = (sidx - n * p.strips_per_stamp) * p.R;
    const int base = (n * p.H + i0) * W * CIN;      // first pixel of the strip's first output row
    float* dst = patch + buf * p.patch_floats;
#pragma unroll
    for (int k = 0; k < GS_MAXG; ++k) {
      const int g = wave + GS_WAVES * k;
      if (g < ngp) {                                  // wave-uniform
        const bool ok = (unsigned)(i0 + prow[k]) < (unsigned)p.H;
        const float* src = ok ? p.X + (unsigned)(base + prel[k]) : p.zero;
        __builtin_amdgcn_global_load_lds((gs_gptr_t)src, (gs_lptr_t)(dst + g * 256), 16, 0, 0);
      }
    }
  };

  // ---- per-lane fragment addressing (strip-invariant) -------------------------------------------------------
  int L0[2];                                        // patch pixel (tap 0,0 -> +PW+1) of this lane's row in M block mb
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    int px = wave * 32 + mb * 16 + l15;
    if (px >= strip_px) px = 0;                     // computed, never stored
    const int pr = px / W, pc = px - pr * W;
    L0[mb] = (pr + 1) * PW + pc + 1;
  }
  int nbase[TN], nsw[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = tn * 16 + l15;
    nbase[tn] = n * CIN;
    nsw[tn] = GS_SW(n);
  }

  const int s_begin = blockIdx.x * p.strips_per_wg;
  const int s_end = min(p.nstrips, s_begin + p.strips_per_wg);
  if (s_begin < s_end) issue_dma(s_begin, 0);
  int buf = 0;
  float* stg = stage + wave * (16 * GS_LDC);
  constexpr int F4 = COUT / 4;                      // float4 per output row
  constexpr int ROWS_IT = 64 / F4;
  // rows this lane finishes in the epilogue: (half mb, iteration it) -> row mb*16 + lane/F4 + it*ROWS_IT of the wave
  constexpr int NIT = 16 / ROWS_IT;
  const int ef4 = lane % F4, ecol = ef4 * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.epi >= 1 && ecol < p.Cout) bias4 = *reinterpret_cast<const f32x4*>(p.bias + ecol);
  f32x4 alr[2][NIT];                                // PReLU slopes of the pending epilogue, loaded a strip ahead of use
  auto prefetch_alpha = [&](int i0) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int px = wave * 32 + mb * 16 + lane / F4 + it * ROWS_IT;
        const int pr = px / W, pc = px - pr * W;
        const bool ok = px < strip_px && i0 + pr < p.H && ecol < p.Cout;
        alr[mb][it] = *reinterpret_cast<const f32x4*>(p.alpha + (ok ? (unsigned)(((i0 + pr) * W + pc) * p.Cout + ecol) : 0u));
      }
  };
  auto epi_half = [&](const f32x4 (&ac)[2][TN], int mb, int n, int i0) {   // 16 rows through the wave's staging tile
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) stg[(lg * 4 + r) * GS_LDC + tn * 16 + l15] = ac[mb][tn][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): wave-private region
    __builtin_amdgcn_wave_barrier();
    const int col = ecol;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int rr = lane / F4 + it * ROWS_IT;
      const int px = wave * 32 + mb * 16 + rr;
      const int pr = px / W, pc = px - pr * W;
      if (px >= strip_px || i0 + pr >= p.H || col >= p.Cout) continue;
      const unsigned aoff = (unsigned)(((i0 + pr) * W + pc) * p.Cout + col);
      const unsigned ooff = (unsigned)(n * p.H * W * p.Cout) + aoff;
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + rr * GS_LDC + col);
      v += bias4;
      if (p.U) *reinterpret_cast<f32x4*>(p.U + ooff) = v;
      if (p.epi == 2) {
        const f32x4 al = alr[mb][it];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? v[k] : al[k] * v[k];
        *reinterpret_cast<f32x4*>(p.A + ooff) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  f32x4 pacc[2][TN];
  int pn = 0, pi0 = 0;
  bool have_prev = false;
  for (int sidx = s_begin; sidx < s_end; ++sidx) {
    // this wave's DMA pieces of the strip (and, the first time, the weight staging) are retired EXPLICITLY before the
    // barrier: hipcc emits such a wait in front of s_barrier today, but nothing obliges it to (the same omission in
    // wgrad_strip8_kernel read a strip before its DMA had landed, about one run in twenty).  The barrier also fences
    // the previous strip's patch reads, so the other buffer may be refilled right behind it.
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0)
    __syncthreads();
    if (sidx + 1 < s_end) issue_dma(sidx + 1, buf ^ 1);
    if (have_prev && p.epi == 2) prefetch_alpha(pi0);
    const float* P = patch + buf * p.patch_floats;

    f32x4 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // NSTEPS fragment steps (tap, 16-channel group), fragments of step s+1 read while the MFMAs of step s run
    auto load_frags = [&](int step, f32x4 (&af)[2], f32x4 (&bf)[TN]) {
      const int t = step / QG, qg = step - t * QG;
      const int shift = gs_dh(p.tapcode, t) * PW + gs_dw(p.tapcode, t);
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int L = L0[mb] + shift;
        af[mb] = *reinterpret_cast<const f32x4*>(P + L * CIN + (((qg * 4 + lg) ^ GS_SW(L)) << 2));
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        bf[tn] = *reinterpret_cast<const f32x4*>(wts + t * COUT * CIN + nbase[tn] + (((qg * 4 + lg) ^ nsw[tn]) << 2));
    };
    auto mfma_step = [&](const f32x4 (&af)[2], const f32x4 (&bf)[TN]) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[mb][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb][jj], bf[tn][jj], acc[mb][tn], 0, 0, 0);
    };
    // The previous strip's epilogue (LDS staging, bias / PReLU, stores) is issued in two halves between the fragment
    // steps of this strip: its instructions fill the issue slack behind the MFMAs instead of leaving the matrix pipe
    // idle while all eight waves store at the same time.
    f32x4 a0[2], b0[TN], a1[2], b1[TN];
    constexpr int H1 = (NSTEPS / 4) & ~1, H2 = H1 + ((NSTEPS / 3) & ~1);
    load_frags(0, a0, b0);
#pragma unroll
    for (int step = 0; step < NSTEPS; step += 2) {
      if (step + 1 < NSTEPS) load_frags(step + 1, a1, b1);
      mfma_step(a0, b0);
      if (step + 2 < NSTEPS) load_frags(step + 2, a0, b0);
      if (step + 1 < NSTEPS) mfma_step(a1, b1);
      if (step == H1 && have_prev) epi_half(pacc, 0, pn, pi0);
      if (step == H2 && have_prev) epi_half(pacc, 1, pn, pi0);
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b2 = 0; b2 < TN; ++b2) pacc[a][b2] = acc[a][b2];
    pn = sidx / p.strips_per_stamp;
    pi0 = (sidx - pn * p.strips_per_stamp) * p.R;
    have_prev = true;
    buf ^= 1;
  }
  if (have_prev) {
    if (p.epi == 2) prefetch_alpha(pi0);
    epi_half(pacc, 0, pn, pi0);
    epi_half(pacc, 1, pn, pi0);
  }
}

#undef GS_SW

// ---------------------------------------------------------------------------------------------------------
// First-layer form: 8 input channels (6 bands + the BatchNorm "ones" channel + 1 pad, see bn_apply_kernel),
// 32 output channels, pad 1, weights k-major W[tap][8][32] (the folded W1p of fold_bn_w1_kernel).  The layer moves
// 256 MB at B = 256 (28 MB in, 2 x 114 MB out) against 35 us of MFMA work, so the point of the strip form is that
// the input is read once and the stores of one strip hide behind the next: a K step is a PAIR of taps (2 x 8
// channels = the 16 k-values of four 16x16x4 MFMAs: lanes 0-31 of a fragment read belong to the even tap, lanes
// 32-63 to the odd one), five steps per strip, the tenth tap has zero weights.
// LDS: patch [pixel][2 quads] (quad ^= (pixel >> 3) & 1), weights [step][n][4 quads] (quad ^= (n >> 2) & 3).
__global__ __launch_bounds__(GS_THREADS, 4) void gconv_strip8_kernel(const GStripParams p) {
  constexpr int CIN = 8, COUT = 32, TN = 2, NST = 5;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                                          // [2][patch_floats]
  float* wts = smem + 2 * p.patch_floats;                       // [5][32][16]
  float* stage = wts + NST * COUT * 16;                         // [8 waves][16][GS_LDC]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations (M0) and piece guards stay on the scalar unit
  const int l15 = lane & 15, lg = lane >> 4;
  const int W = p.Wd, PW = W + 2;
  const int strip_px = p.R * W;

  // weights -> LDS: element (tap t, channel c, column n) at [t >> 1][n][(t & 1) * 8 + c]
  for (int e = tid; e < NST * COUT * 16; e += GS_THREADS) {
    const int st = e / (COUT * 16), rem = e - st * (COUT * 16);
    const int n = rem >> 4, k16 = rem & 15;
    const int t = 2 * st + (k16 >> 3), c = k16 & 7;
    const float v = t < 9 ? p.W[(t * CIN + c) * COUT + n] : 0.f;
    wts[(st * COUT + n) * 16 + ((((k16 >> 2) ^ ((n >> 2) & 3)) << 2) | (k16 & 3))] = v;
  }

  const int ptot = (p.R + 2) * PW * 2;             // 16-byte slots of one patch
  const int ngp = (ptot + 63) / 64;
  int prel[2], prow[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int e = 64 * (wave + GS_WAVES * k) + lane;
    const int L = e >> 1, sl = e & 1;
    const int pr = L / PW, pc = L - pr * PW;
    const int gc = pc - 1;
    const bool ok = e < ptot && (unsigned)gc < (unsigned)W;
    prel[k] = ((pr - 1) * W + gc) * CIN + ((sl ^ ((L >> 3) & 1)) << 2);
    prow[k] = ok ? pr - 1 : -100000;
  }
  auto issue_dma = [&](int sidx, int buf) {
    const int n = sidx / p.strips_per_stamp;
    const int i0 = (sidx - n * p.strips_per_stamp) * p.R;
    const int base = (n * p.H + i0) * W * CIN;
    float* dst = patch + buf * p.patch_floats;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int g = wave + GS_WAVES * k;
      if (g < ngp) {
        const bool ok = (unsigned)(i0 + prow[k]) < (unsigned)p.H;
        const float* src = ok ? p.X + (unsigned)(base + prel[k]) : p.zero;
        __builtin_amdgcn_global_load_lds((gs_gptr_t)src, (gs_lptr_t)(dst + g * 256), 16, 0, 0);
      }
    }
  };

  // per-lane fragment addressing: patch pixel of (M block, tap pair step); the lane's tap is 2 * step + (lg >> 1)
  int Lf[2][NST];
#pragma unroll
  for (int mb = 0; mb < 2; ++mb) {
    int px = wave * 32 + mb * 16 + l15;
    if (px >= strip_px) px = 0;                     // computed, never stored
    const int pr = px / W, pc = px - pr * W;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int t = min(2 * st + (lg >> 1), 8);     // tap 9: zero weights, any valid address
      Lf[mb][st] = (pr + t / 3) * PW + pc + t % 3;
    }
  }
  int boff[TN];
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const int n = tn * 16 + l15;
    boff[tn] = n * 16 + ((lg ^ ((n >> 2) & 3)) << 2);
  }

  const int s_begin = blockIdx.x * p.strips_per_wg;
  const int s_end = min(p.nstrips, s_begin + p.strips_per_wg);
  if (s_begin < s_end) issue_dma(s_begin, 0);
  int buf = 0;
  float* stg = stage + wave * (16 * GS_LDC);
  constexpr int F4 = COUT / 4, ROWS_IT = 64 / F4, NIT = 16 / ROWS_IT;
  const int ecol = (lane % F4) * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.epi >= 1) bias4 = *reinterpret_cast<const f32x4*>(p.bias + ecol);
  f32x4 alr[2][NIT];
  auto prefetch_alpha = [&](int i0) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int px = wave * 32 + mb * 16 + lane / F4 + it * ROWS_IT;
        const int pr = px / W, pc = px - pr * W;
        const bool ok = px < strip_px && i0 + pr < p.H;
        alr[mb][it] = *reinterpret_cast<const f32x4*>(p.alpha + (ok ? (unsigned)(((i0 + pr) * W + pc) * COUT + ecol) : 0u));
      }
  };
  auto epi_half = [&](const f32x4 (&ac)[2][TN], int mb, int n, int i0) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) stg[(lg * 4 + r) * GS_LDC + tn * 16 + l15] = ac[mb][tn][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): wave-private region
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int rr = lane / F4 + it * ROWS_IT;
      const int px = wave * 32 + mb * 16 + rr;
      const int pr = px / W, pc = px - pr * W;
      if (px >= strip_px || i0 + pr >= p.H) continue;
      const unsigned ooff = (unsigned)(n * p.H * W * COUT) + (unsigned)(((i0 + pr) * W + pc) * COUT + ecol);
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + rr * GS_LDC + ecol);
      v += bias4;
      if (p.U) *reinterpret_cast<f32x4*>(p.U + ooff) = v;
      if (p.epi == 2) {
        const f32x4 al = alr[mb][it];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? v[k] : al[k] * v[k];
        *reinterpret_cast<f32x4*>(p.A + ooff) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  };
  f32x4 pacc[2][TN];
  int pn = 0, pi0 = 0;
  bool have_prev = false;
  for (int sidx = s_begin; sidx < s_end; ++sidx) {
    __builtin_amdgcn_s_waitcnt(0x0F70);             // vmcnt(0): this wave's DMA pieces (see wgrad_strip8_kernel)
    __syncthreads();
    if (sidx + 1 < s_end) issue_dma(sidx + 1, buf ^ 1);
    if (have_prev && p.epi == 2) prefetch_alpha(pi0);
    const float* P = patch + buf * p.patch_floats;
    f32x4 acc[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragments of step st + 1 are read while the MFMAs of step st run (two register sets)
    auto load_frags = [&](int st, f32x4 (&af)[2], f32x4 (&bf)[TN]) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const int L = Lf[mb][st];
        af[mb] = *reinterpret_cast<const f32x4*>(P + L * CIN + (((lg & 1) ^ ((L >> 3) & 1)) << 2));
      }
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bf[tn] = *reinterpret_cast<const f32x4*>(wts + st * COUT * 16 + boff[tn]);
    };
    auto mfma_step = [&](const f32x4 (&af)[2], const f32x4 (&bf)[TN]) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[mb][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb][jj], bf[tn][jj], acc[mb][tn], 0, 0, 0);
    };
    f32x4 a0[2], b0[TN], a1[2], b1[TN];
    load_frags(0, a0, b0);
    load_frags(1, a1, b1);
    mfma_step(a0, b0);                               // step 0
    load_frags(2, a0, b0);
    mfma_step(a1, b1);                               // step 1
    if (have_prev) epi_half(pacc, 0, pn, pi0);       // the previous strip's epilogue between the MFMA groups
    load_frags(3, a1, b1);
    mfma_step(a0, b0);                               // step 2
    load_frags(4, a0, b0);
    mfma_step(a1, b1);                               // step 3
    if (have_prev) epi_half(pacc, 1, pn, pi0);
    mfma_step(a0, b0);                               // step 4
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b2 = 0; b2 < TN; ++b2) pacc[a][b2] = acc[a][b2];
    pn = sidx / p.strips_per_stamp;
    pi0 = (sidx - pn * p.strips_per_stamp) * p.R;
    have_prev = true;
    buf ^= 1;
  }
  if (have_prev) {
    if (p.epi == 2) prefetch_alpha(pi0);
    epi_half(pacc, 0, pn, pi0);
    epi_half(pacc, 1, pn, pi0);
  }
}

// Returns 1 when the layer is not the first-layer shape (the caller then uses gconv2).
int launch_gconv_strip8(GStripParams p, hipStream_t s) {
  if (p.Cin != 8 || p.Cout != 32 || !p.zero || p.Wd < 8 || p.Wd > 64 || p.H < 1 || p.epi < 0 || p.epi > 2) return 1;
  if (p.epi == 2 && (!p.alpha || !p.A)) return 1;
  if ((long)p.NB * p.H * p.Wd * 32 >= (1L << 30)) return 1;
  int R = 256 / p.Wd;
  if (R > p.H) R = p.H;
  const int PW = p.Wd + 2;
  const int slots = (R + 2) * PW * 2;
  if ((slots + 63) / 64 > 2 * GS_WAVES) return 1;
  p.R = R;
  p.patch_floats = ((slots + 63) / 64) * 256;
  p.strips_per_stamp = (p.H + R - 1) / R;
  p.nstrips = p.NB * p.strips_per_stamp;
  // bandwidth-bound: two workgroups per CU keep stores of one strip and the DMA of the next in flight
  const int target = 512;
  p.strips_per_wg = (p.nstrips + target - 1) / target;
  const int grid = (p.nstrips + p.strips_per_wg - 1) / p.strips_per_wg;
  const size_t smem = ((size_t)2 * p.patch_floats + (size_t)5 * 32 * 16 + (size_t)GS_WAVES * 16 * GS_LDC) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gconv_strip8_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(gconv_strip8_kernel, dim3(grid), dim3(GS_THREADS), smem, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

template <int CIN, int COUT, bool NMAJOR>
static int launch_gs(const GStripParams& p, int grid, size_t smem, hipStream_t s) {
  static bool attr_set = false;
  auto kern = gconv_strip_kernel<CIN, COUT, NMAJOR>;
  if (!attr_set) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(GS_THREADS), smem, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

// Returns 1 when the layer is not one this kernel takes (the caller then uses gconv2).
int launch_gconv_strip(GStripParams p, bool nmajor, hipStream_t s) {
  if ((p.Cin != 32 && p.Cin != 16) || (p.Cout != 32 && p.Cout != 16) || !p.zero) return 1;
  if (p.Cin == 16 && p.Cout != 32) return 1;
  if (p.Wd < 8 || p.Wd > 64 || p.H < 1 || p.epi < 0 || p.epi > 2) return 1;
  if ((long)p.NB * p.H * p.Wd * 32 >= (1L << 30)) return 1;
  const int cout_pad = p.Cout <= 16 ? 16 : 32;
  int R = 256 / p.Wd;
  if (R > p.H) R = p.H;
  if (R < 1) return 1;
  const int PW = p.Wd + 2;
  const int slots = (R + 2) * PW * (p.Cin / 4);
  if ((slots + 63) / 64 > GS_MAXG * GS_WAVES) return 1;
  p.R = R;
  p.patch_floats = ((slots + 63) / 64) * 256;
  p.strips_per_stamp = (p.H + R - 1) / R;
  p.nstrips = p.NB * p.strips_per_stamp;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  p.strips_per_wg = (p.nstrips + cus - 1) / cus;
  const int grid = (p.nstrips + p.strips_per_wg - 1) / p.strips_per_wg;
  const size_t smem = ((size_t)2 * p.patch_floats + (size_t)9 * cout_pad * p.Cin + (size_t)GS_WAVES * 16 * GS_LDC) * sizeof(float);
  if (smem > 160 * 1024) return 1;
  if (p.Cin == 16) return nmajor ? launch_gs<16, 32, true>(p, grid, smem, s) : launch_gs<16, 32, false>(p, grid, smem, s);
  if (cout_pad == 16) return nmajor ? launch_gs<32, 16, true>(p, grid, smem, s) : launch_gs<32, 16, false>(p, grid, smem, s);
  return nmajor ? launch_gs<32, 32, true>(p, grid, smem, s) : launch_gs<32, 32, false>(p, grid, smem, s);
}

}  // namespace dv
