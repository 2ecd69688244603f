// Weight gradient for the high-resolution, few-channel layers (Cx in {32,64}, Cy <= 64): "strip" form.
//
// dW[(kh,kw,cx), cy] = sum over pixels of X[n, i*SX + kh - pb, j*SX + kw - pb, cx] * Y[n, i, j, cy]   (see wgrad.hip)
// For these layers the output (9*Cx x Cy) is tiny and the pixel reduction is huge (up to 1M pixels), so a
// (row tile x column tile) GEMM kernel re-reads X nine times (once per tap) for very little MFMA work per
// barrier.  Here a workgroup walks over strips of R image rows of one stamp: the X rows a strip touches
// (with their 1-pixel halo, zero-filled outside the image) and the strip's Y rows are staged in LDS ONCE, and
// all nine taps read their A fragments from the same patch at shifted addresses.  Every wave keeps its part
// of the whole dW in accumulators across all strips of the workgroup (up to 36 16x16 blocks = 144 registers)
// and writes one partial slab at the end; reduce_partials sums the slabs in a fixed order.
#include <algorithm>

#include "common.h"

namespace dv {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int CX, int CY, int SX, int WMS, int WNS, int WKS>
__global__ __launch_bounds__(256, 1) void wgrad_strip_kernel(const WStripParams p) {
  static_assert(WMS * WNS * WKS == 4, "4 waves");
  static_assert(CX == 32 * WMS, "each wave owns two 16-channel blocks of Cx for all nine taps");
  constexpr int CYB = (CY + 15) / 16;
  constexpr int NB_W = CYB / WNS;
  static_assert(NB_W * WNS == CYB && NB_W >= 1 && NB_W <= 2, "column blocks per wave");
  constexpr int CX4 = CX / 4, CY4 = CY / 4;
  constexpr int MAXGX = 12, MAXGY = 8;                // 1-KiB LDS-DMA pieces per wave and strip (X, Y)

  // LDS: two buffers of [X patch | Y rows], unpadded pixel rows so that every 1-KiB LDS-DMA piece
  // (64 lanes x 16 B, written at a wave-uniform base) is 64 consecutive 16-byte slots of the image.
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int XC = (p.Wy - 1) * SX + 3;
  const int buf_floats = p.xs_floats + p.ys_floats;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave % WKS, wn = (wave / WKS) % WNS, wm = wave / (WKS * WNS);
  const int l15 = lane & 15, lg = lane >> 4;

  f32x4 acc[9][2][NB_W];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int nb = 0; nb < NB_W; ++nb) acc[t][cb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int strip_px = p.R * p.Wy;
  const int ksteps = (strip_px + 3) / 4;
  const int kpw = (ksteps + WKS - 1) / WKS;
  const int ks0 = wk * kpw, ks1 = min(ksteps, ks0 + kpw);

  // strip-invariant part of the X gather: slot e = 64*(wave + 4k) + lane -> (patch row, column ok, offset)
  const int xtot = p.XR * XC * CX4;
  const int ngx = (xtot + 63) / 64, ngy = (strip_px * CY4 + 63) / 64;
  int xrel[MAXGX], xrow[MAXGX];
#pragma unroll
  for (int k = 0; k < MAXGX; ++k) {
    const int e = 64 * (wave + 4 * k) + lane;
    const int px = e / CX4, c4 = e - px * CX4;
    const int xr = px / XC, xc = px - xr * XC;
    const int gc = xc - p.pb;
    const bool ok = e < xtot && (unsigned)gc < (unsigned)p.Wx;
    xrel[k] = (xr * p.Wx + gc) * CX + c4 * 4;
    xrow[k] = ok ? xr : -100000;                      // invalid column / slot: never passes the row test
  }

  auto issue_dma = [&](int sidx, int buf) {
    const int n = sidx / p.strips_per_stamp;
    const int i0 = (sidx - n * p.strips_per_stamp) * p.R;
    const int gr0 = i0 * SX - p.pb;
    const int xbase = (n * p.Hx + gr0) * p.Wx * CX;    // may be "negative rows": only used when the row test passes
    float* xs = smem + buf * buf_floats;
    float* ys = xs + p.xs_floats;
#pragma unroll
    for (int k = 0; k < MAXGX; ++k) {
      const int g = wave + 4 * k;
      if (g < ngx) {                                    // wave-uniform
        const bool ok = (unsigned)(gr0 + xrow[k]) < (unsigned)p.Hx;
        const float* src = ok ? p.X + (unsigned)(xbase + xrel[k]) : p.zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + g * 256), 16, 0, 0);
      }
    }
    const unsigned ybase = (unsigned)((n * p.Hy + i0) * p.Wy) * CY;
    const int yvalid4 = min(p.R, p.Hy - i0) * p.Wy * CY4;
#pragma unroll
    for (int k = 0; k < MAXGY; ++k) {
      const int g = wave + 4 * k;
      if (g < ngy) {
        const int e = 64 * g + lane;
        const float* src = e < yvalid4 ? p.Y + ybase + (unsigned)(e * 4) : p.zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ys + g * 256), 16, 0, 0);
      }
    }
  };

  const int s_begin = blockIdx.x * p.strips_per_wg;
  const int s_end = min(p.nstrips, s_begin + p.strips_per_wg);
  if (s_begin < s_end) issue_dma(s_begin, 0);
  int buf = 0;
  unsigned long long tb = 0, td = 0, tc = 0, tmark = 0;   // dbg 4: cycles in barrier / DMA issue / compute
  for (int sidx = s_begin; sidx < s_end; ++sidx) {
    if ((p.dbg & 4)) tmark = __builtin_amdgcn_s_memtime();
    // the barrier's vmcnt(0) retires this strip's DMA pieces of every wave; it also fences the previous strip's
    // LDS reads, so the other buffer may be refilled right behind it
    __syncthreads();
    if ((p.dbg & 4)) {
      unsigned long long t = __builtin_amdgcn_s_memtime();
      tb += t - tmark;
      tmark = t;
    }
    if (sidx + 1 < s_end && !(p.dbg & 1)) issue_dma(sidx + 1, buf ^ 1);
    if ((p.dbg & 4)) {
      unsigned long long t = __builtin_amdgcn_s_memtime();
      td += t - tmark;
      tmark = t;
    }
    const float* Xs = smem + buf * buf_floats;
    const float* Ys = Xs + p.xs_floats;

    // ---- MFMA over this wave's pixels ---------------------------------------------------------
    // Fragments of k-step ks+1 are read from LDS while the 18*NB_W MFMAs of k-step ks run (two named register
    // sets, loop unrolled by two): hipcc otherwise emits read -> lgkmcnt(0) -> 4 MFMAs chains.
    int q = 4 * ks0 + lg;
    int pr = q / p.Wy, pj = q - pr * p.Wy;
    auto load_frags = [&](float (&af)[18], float (&bf)[NB_W]) {
      const bool qv = q < strip_px;
      const int abase = qv ? ((pr * SX) * XC + pj * SX) * CX + wm * 32 + l15 : wm * 32 + l15;
      const float* yb = Ys + (qv ? q : 0) * CY + wn * NB_W * 16 + l15;
#pragma unroll
      for (int nb = 0; nb < NB_W; ++nb) {
        // mask by multiplication: a select would let hipcc predicate the LDS read and split the basic block
        bf[nb] = yb[nb * 16] * ((qv && (wn * NB_W + nb) * 16 + l15 < CY) ? 1.f : 0.f);
      }
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float* xa = Xs + abase + (kh * XC + kw) * CX;
          af[(kh * 3 + kw) * 2] = xa[0];
          af[(kh * 3 + kw) * 2 + 1] = xa[16];
        }
      q += 4;
      pj += 4;
      const int wrap = pj >= p.Wy ? 1 : 0;              // Wy >= 4 (checked by the launcher)
      pj -= wrap ? p.Wy : 0;
      pr += wrap;
    };
    auto mfma_step = [&](const float (&af)[18], const float (&bf)[NB_W]) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int nb = 0; nb < NB_W; ++nb) {
          acc[t][0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2 * t], bf[nb], acc[t][0][nb], 0, 0, 0);
          acc[t][1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2 * t + 1], bf[nb], acc[t][1][nb], 0, 0, 0);
        }
    };
    // one wave per SIMD issues in order: ask the scheduler to slot each LDS read (and its address VALU) into the
    // issue slack behind a pair of MFMAs instead of grouping all reads in front of the MFMA block
    auto interleave = [&]() {
#pragma unroll
      for (int i = 0; i < 20; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, NB_W == 2 ? 2 : 1, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                  // DS read
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                  // VALU
      }
    };
    float a0[18], b0[NB_W], a1[18], b1[NB_W];
    int ks = ks0;
    if (ks < ks1) load_frags(a0, b0);
    const int ks_end = (p.dbg & 2) ? ks0 : ks1;
    if (p.dbg & 8) {                                   // ablation: MFMA issue only, no LDS reads in the loop
      load_frags(a1, b1);
      for (; ks + 2 <= ks_end; ks += 2) {
        mfma_step(a0, b0);
        mfma_step(a1, b1);
      }
    }
    for (; ks + 2 <= ks_end; ks += 2) {
      load_frags(a1, b1);
      mfma_step(a0, b0);
      interleave();
      load_frags(a0, b0);                               // may run past this wave's range: loads are clamped, unused
      mfma_step(a1, b1);
      interleave();
    }
    if (ks < ks_end) mfma_step(a0, b0);
    if ((p.dbg & 4)) {
      asm volatile("" ::"v"(acc[0][0][0]), "v"(acc[8][1][NB_W - 1]));
      unsigned long long t = __builtin_amdgcn_s_memtime();
      tc += t - tmark;
    }
    buf ^= 1;
  }
  if ((p.dbg & 4) && blockIdx.x == 7 && lane == 0) {
    float* d = p.part + p.part_capacity - 64 + wave * 4;
    d[0] = (float)tb;
    d[1] = (float)td;
    d[2] = (float)tc;
    d[3] = (float)(s_end - s_begin);
  }
  __syncthreads();                                     // all waves done with LDS before it is reused below

  // ---- one partial slab per workgroup; waves that split the pixels are summed through LDS first ---------
  float* slab = p.part + (size_t)blockIdx.x * (9 * CX) * CY;
  if (WKS == 1) {
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = t * CX + wm * 32 + cb * 16 + lg * 4 + r;
#pragma unroll
          for (int nb = 0; nb < NB_W; ++nb) {
            const int col = (wn * NB_W + nb) * 16 + l15;
            if (col < CY) slab[(size_t)row * CY + col] = acc[t][cb][nb][r];
          }
        }
  } else {
    constexpr int TC = CYB * 16;                       // staged tile: [WKS][CX][TC] floats per tap
    float* T = smem;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int nb = 0; nb < NB_W; ++nb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            T[(wk * CX + wm * 32 + cb * 16 + lg * 4 + r) * TC + (wn * NB_W + nb) * 16 + l15] = acc[t][cb][nb][r];
      __syncthreads();
      for (int e = tid; e < CX * TC; e += 256) {
        float v = T[e];
#pragma unroll
        for (int k = 1; k < WKS; ++k) v += T[k * CX * TC + e];
        const int cx = e / TC, cy = e - cx * TC;
        if (cy < CY) slab[(size_t)(t * CX + cx) * CY + cy] = v;
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// First-layer variant: Cx = 8 (6 bands + the BN "ones" channel + 1 pad), Cy = 32, stride 1.  dW rows are
// (tap, cx) = 72 -> five 16-row MFMA blocks, each spanning two taps (lanes 0-7 / 8-15); tap 9 does not exist and
// contributes zeros.  Four waves split the pixels of a strip; the partial tiles are summed through LDS.
__global__ __launch_bounds__(256, 1) void wgrad_strip8_kernel(const WStripParams p) {
  constexpr int CX = 8, CY = 32, CX4 = 2, CY4 = 8, WKS = 4, MB = 5, NBK = 2;
  constexpr int MAXGX = 8, MAXGY = 12;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int XC = p.Wy + 2;
  const int buf_floats = p.xs_floats + p.ys_floats;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave;
  const int l15 = lane & 15, lg = lane >> 4;

  f32x4 acc[MB][NBK];
#pragma unroll
  for (int a = 0; a < MB; ++a)
#pragma unroll
    for (int b = 0; b < NBK; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int strip_px = p.R * p.Wy;
  const int ksteps = (strip_px + 3) / 4;
  const int kpw = (ksteps + WKS - 1) / WKS;
  const int ks0 = wk * kpw, ks1 = min(ksteps, ks0 + kpw);

  // per-lane tap offsets of the five row blocks (floats inside the patch), -1: no such tap
  int toff[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int tap = 2 * mb + (l15 >> 3);
    toff[mb] = tap < 9 ? ((tap / 3) * XC + (tap % 3)) * CX + (l15 & 7) : -1;
  }

  const int xtot = p.XR * XC * CX4;
  const int ngx = (xtot + 63) / 64, ngy = (strip_px * CY4 + 63) / 64;
  int xrel[MAXGX], xrow[MAXGX];
#pragma unroll
  for (int k = 0; k < MAXGX; ++k) {
    const int e = 64 * (wave + 4 * k) + lane;
    const int px = e / CX4, c4 = e - px * CX4;
    const int xr = px / XC, xc = px - xr * XC;
    const int gc = xc - p.pb;
    const bool ok = e < xtot && (unsigned)gc < (unsigned)p.Wx;
    xrel[k] = (xr * p.Wx + gc) * CX + c4 * 4;
    xrow[k] = ok ? xr : -100000;
  }
  auto issue_dma = [&](int sidx, int buf) {
    const int n = sidx / p.strips_per_stamp;
    const int i0 = (sidx - n * p.strips_per_stamp) * p.R;
    const int gr0 = i0 - p.pb;
    const int xbase = (n * p.Hx + gr0) * p.Wx * CX;
    float* xs = smem + buf * buf_floats;
    float* ys = xs + p.xs_floats;
#pragma unroll
    for (int k = 0; k < MAXGX; ++k) {
      const int g = wave + 4 * k;
      if (g < ngx) {
        const bool ok = (unsigned)(gr0 + xrow[k]) < (unsigned)p.Hx;
        const float* src = ok ? p.X + (unsigned)(xbase + xrel[k]) : p.zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(xs + g * 256), 16, 0, 0);
      }
    }
    const unsigned ybase = (unsigned)((n * p.Hy + i0) * p.Wy) * CY;
    const int yvalid4 = min(p.R, p.Hy - i0) * p.Wy * CY4;
#pragma unroll
    for (int k = 0; k < MAXGY; ++k) {
      const int g = wave + 4 * k;
      if (g < ngy) {
        const int e = 64 * g + lane;
        const float* src = e < yvalid4 ? p.Y + ybase + (unsigned)(e * 4) : p.zero;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ys + g * 256), 16, 0, 0);
      }
    }
  };

  const int s_begin = blockIdx.x * p.strips_per_wg;
  const int s_end = min(p.nstrips, s_begin + p.strips_per_wg);
  if (s_begin < s_end) issue_dma(s_begin, 0);
  int buf = 0;
  for (int sidx = s_begin; sidx < s_end; ++sidx) {
    __syncthreads();
    if (sidx + 1 < s_end) issue_dma(sidx + 1, buf ^ 1);
    const float* Xs = smem + buf * buf_floats;
    const float* Ys = Xs + p.xs_floats;
    int q = 4 * ks0 + lg;
    int pr = q / p.Wy, pj = q - pr * p.Wy;
    for (int ks = ks0; ks < ks1; ++ks) {
      const bool qv = q < strip_px;
      const int base = qv ? (pr * XC + pj) * CX : 0;
      const float msk = qv ? 1.f : 0.f;
      float bf[NBK], af[MB];
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) bf[nb] = Ys[(qv ? q : 0) * CY + nb * 16 + l15] * msk;
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) af[mb] = Xs[base + max(toff[mb], 0)] * (toff[mb] >= 0 ? 1.f : 0.f);
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb], bf[nb], acc[mb][nb], 0, 0, 0);
      q += 4;
      pj += 4;
      const int wrap = pj >= p.Wy ? 1 : 0;
      pj -= wrap ? p.Wy : 0;
      pr += wrap;
    }
    buf ^= 1;
  }
  __syncthreads();
  // sum the four pixel-split copies through LDS: T[wk][80][32]
  float* T = smem;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int nb = 0; nb < NBK; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) T[(wk * 80 + mb * 16 + lg * 4 + r) * CY + nb * 16 + l15] = acc[mb][nb][r];
  __syncthreads();
  float* slab = p.part + (size_t)blockIdx.x * (9 * CX) * CY;
  for (int e = tid; e < 72 * CY; e += 256) {
    float v = T[e];
#pragma unroll
    for (int k = 1; k < WKS; ++k) v += T[k * 80 * CY + e];
    slab[e] = v;
  }
}

static int launch_strip8(WStripParams p, hipStream_t s, int* nsplit_out) {
  const int XC = p.Wy + 2;
  auto xfl = [&](int r) { return (((size_t)(r + 2) * XC * 2 + 63) / 64) * 256; };
  auto yfl = [&](int r) { return (((size_t)r * p.Wy * 8 + 63) / 64) * 256; };
  auto fits = [&](int r) {
    return 2 * (xfl(r) + yfl(r)) * sizeof(float) <= 128 * 1024 && xfl(r) / 256 <= 4 * 8 && yfl(r) / 256 <= 4 * 12;
  };
  if (p.Wy < 4 || !fits(1)) return 1;   // fall back to wgrad_kernel
  // this layer is a streaming reduction (142 MB in, 72 x 32 out): short strips and three workgroups per CU keep more
  // DMA in flight than one workgroup with long strips (82 -> 69 us, tools/layer_bench.py)
  int R = 1;
  while (R < p.Hy && R < 2 && fits(R + 1)) ++R;
  p.R = R;
  p.XR = R + 2;
  p.xs_floats = (int)xfl(R);
  p.ys_floats = (int)yfl(R);
  p.strips_per_stamp = (p.Hy + R - 1) / R;
  p.nstrips = p.NB * p.strips_per_stamp;
  int wgs = std::min(p.nstrips, 768);
  wgs = std::max(1, std::min(wgs, (int)(p.part_capacity / ((size_t)72 * 32))));
  p.strips_per_wg = (p.nstrips + wgs - 1) / wgs;
  wgs = (p.nstrips + p.strips_per_wg - 1) / p.strips_per_wg;
  const size_t smem = std::max((size_t)2 * (p.xs_floats + p.ys_floats), (size_t)4 * 80 * 32) * sizeof(float);
  static size_t attr = 0;
  if (smem > attr) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_strip8_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr = smem;
  }
  hipLaunchKernelGGL(wgrad_strip8_kernel, dim3(wgs), dim3(256), smem, s, p);
  DV_HIP(hipGetLastError());
  *nsplit_out = wgs;
  return OK;
}

template <int CX, int CY, int SX, int WMS, int WNS, int WKS>
static int launch_strip_cfg(WStripParams p, hipStream_t s, int* nsplit_out) {
  constexpr int CX4 = CX / 4, CY4 = CY / 4;
  const int XC = (p.Wy - 1) * SX + 3;
  auto xfl = [&](int r) { return (((size_t)((r - 1) * SX + 3) * XC * CX4 + 63) / 64) * 256; };   // floats, 1-KiB pieces
  auto yfl = [&](int r) { return (((size_t)r * p.Wy * CY4 + 63) / 64) * 256; };
  auto fits = [&](int r) {
    return 2 * (xfl(r) + yfl(r)) * sizeof(float) <= 128 * 1024 && xfl(r) / 256 <= 4 * 12 && yfl(r) / 256 <= 4 * 8;
  };
  int R = 1;
  if (p.Wy < 4 || !fits(1)) return 1;   // geometry does not fit the strip buffers: caller falls back to wgrad_kernel
  while (R < p.Hy && R < 8 && fits(R + 1)) ++R;
  p.R = R;
  p.XR = (R - 1) * SX + 3;
  p.xs_floats = (int)xfl(R);
  p.ys_floats = (int)yfl(R);
  p.strips_per_stamp = (p.Hy + R - 1) / R;
  p.nstrips = p.NB * p.strips_per_stamp;
  // one workgroup per CU (double-buffered strips fill most of the LDS); each keeps its accumulators over all of
  // its strips and writes a single partial slab
  int wgs = std::min(p.nstrips, 256);
  wgs = std::max(1, std::min(wgs, (int)(p.part_capacity / ((size_t)9 * CX * CY))));
  p.strips_per_wg = (p.nstrips + wgs - 1) / wgs;
  wgs = (p.nstrips + p.strips_per_wg - 1) / p.strips_per_wg;
  auto kern = wgrad_strip_kernel<CX, CY, SX, WMS, WNS, WKS>;
  const size_t smem = std::max((size_t)2 * (p.xs_floats + p.ys_floats), (size_t)WKS * CX * ((CY + 15) / 16) * 16) *
                      sizeof(float);
  static size_t attr = 0;
  if (smem > attr) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr = smem;
  }
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), smem, s, p);
  DV_HIP(hipGetLastError());
  *nsplit_out = wgs;
  return OK;
}

bool wgrad_strip_supported(int Cx, int Cy, int sx, int ntaps) {
  if (ntaps != 9 || (sx != 1 && sx != 2)) return false;
  if (Cx == 8 && Cy == 32 && sx == 1) return true;
  return (Cx == 32 && (Cy == 32 || Cy == 64 || Cy == 12 || Cy == 16)) || (Cx == 64 && Cy == 64);
}

static int g_strip_dbg = 0;
void debug_set_strip(int v) { g_strip_dbg = v; }

int launch_wgrad_strip(const WStripParams& p0, int Cx, int Cy, int sx, hipStream_t s, int* nsplit_out) {
  WStripParams p = p0;
  p.dbg = g_strip_dbg;
  if (Cx == 8 && Cy == 32 && sx == 1) {
    if (p.pb != 1 || p.Hx != p.Hy) return 1;
    return launch_strip8(p, s, nsplit_out);
  }
  if ((long)p.NB * p.Hx * p.Wx * Cx >= (1L << 30) || (long)p.NB * p.Hy * p.Wy * Cy >= (1L << 30)) {
    set_error("wgrad_strip: tensor too large for 32-bit offsets");
    return E_INVALID;
  }
#define DV_STRIP(cx, cy, wm, wn, wk)                                                          \
  if (Cx == cx && Cy == cy)                                                                   \
    return sx == 1 ? launch_strip_cfg<cx, cy, 1, wm, wn, wk>(p, s, nsplit_out)                \
                   : launch_strip_cfg<cx, cy, 2, wm, wn, wk>(p, s, nsplit_out);
  DV_STRIP(32, 32, 1, 1, 4)
  DV_STRIP(32, 64, 1, 2, 2)
  DV_STRIP(32, 12, 1, 1, 4)
  DV_STRIP(32, 16, 1, 1, 4)
  DV_STRIP(64, 64, 2, 2, 1)
#undef DV_STRIP
  set_error("wgrad_strip: unsupported channel pair (%d, %d)", Cx, Cy);
  return E_INVALID;
}

}  // namespace dv
