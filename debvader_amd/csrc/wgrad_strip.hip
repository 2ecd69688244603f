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

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS-DMA destinations (M0) and piece guards stay on the scalar unit
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
    // vmcnt(0) + barrier retire this strip's DMA pieces of every wave (spelled out: see wgrad_strip8_kernel); the
    // barrier also fences the previous strip's LDS reads, so the other buffer may be refilled right behind it
    __builtin_amdgcn_s_waitcnt(0x0F70);
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
//
// FUSED: Y is d(activation) of the first PReLU and p.U its pre-activation.  The strip's rows of both are brought in,
// d(pre-activation) = dA * (U > 0 ? 1 : alpha) is formed in LDS and never written to memory (this layer has no data
// gradient, so the weight-gradient MFMAs are its only reader), and d(alpha) / d(bias) are accumulated in registers:
// workgroup b owns row block b % strips_per_stamp of the stamps g, g + groups, ... (g = b / strips_per_stamp), so a
// thread meets the same (pixel, channel) elements in every strip.  Replaces a 342 MB read-modify-write pass.
template <bool FUSED>
__global__ __launch_bounds__(256, 1) void wgrad_strip8_kernel(const WStripParams p) {
  constexpr int CX = 8, CY = 32, CX4 = 2, CY4 = 8, WKS = 4, MB = 5, NBK = 2;
  constexpr int MAXGX = 8, MAXGY = 12, MAXE = 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int XC = p.Wy + 2;
  const int buf_floats = p.xs_floats + (FUSED ? 2 : 1) * p.ys_floats;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar loop control below
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

  // per-lane tap offsets of the five row blocks (floats inside the patch)
  int toff[MB];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int tap = 2 * mb + (l15 >> 3);
    const int t = min(tap, 8);            // tap 9 (rows 72..79 of dW) does not exist: any finite operand will do
    toff[mb] = ((t / 3) * XC + (t % 3)) * CX + (l15 & 7);
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
  // strip (stamp n, first output row i0) -> LDS buffer buf
  auto issue_dma = [&](int n, int i0, int buf) {
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
        if (FUSED) {
          const float* usrc = e < yvalid4 ? p.U + ybase + (unsigned)(e * 4) : p.zero;
          __builtin_amdgcn_global_load_lds((gptr_t)usrc, (lptr_t)(ys + p.ys_floats + g * 256), 16, 0, 0);
        }
      }
    }
  };

  // the sequence of strips of this workgroup: unfused, a contiguous range of (stamp, row block) pairs; fused, one
  // row block of every groups-th stamp
  int n_cur, i0_cur, s_cur, s_end;
  const int rb = FUSED ? (int)blockIdx.x % p.strips_per_stamp : 0;
  if (FUSED) {
    s_cur = (int)blockIdx.x / p.strips_per_stamp;      // stamp index
    s_end = p.NB;
  } else {
    s_cur = blockIdx.x * p.strips_per_wg;              // strip index
    s_end = min(p.nstrips, s_cur + p.strips_per_wg);
  }
  const int s_step = FUSED ? p.groups : 1;
  auto locate = [&](int sidx, int& n, int& i0) {
    if (FUSED) {
      n = sidx;
      i0 = rb * p.R;
    } else {
      n = sidx / p.strips_per_stamp;
      i0 = (sidx - n * p.strips_per_stamp) * p.R;
    }
  };

  // fused: this thread's PReLU slopes and gradient accumulators for its (at most MAXE) float4 elements of the block
  f32x4 al[MAXE], dal[MAXE], dbs = {0.f, 0.f, 0.f, 0.f};
  int evalid = 0;
  if (FUSED) {
    const int i0 = rb * p.R;
    evalid = min(p.R, p.Hy - i0) * p.Wy * CY4;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      const int e = tid + 256 * k;
      dal[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      al[k] = e < evalid ? *reinterpret_cast<const f32x4*>(p.alpha + (size_t)i0 * p.Wy * CY + e * 4)
                         : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  }

  if (s_cur < s_end) {
    locate(s_cur, n_cur, i0_cur);
    issue_dma(n_cur, i0_cur, 0);
  }
  int buf = 0;
  for (int sidx = s_cur; sidx < s_end; sidx += s_step) {
    // every wave's DMA pieces of this strip must have landed before any wave reads them: the wait is spelled out
    // because the compiler does not reliably place one for LDS-DMA in front of the barrier (it left it out of the
    // fused form, a race that showed up in 1 of ~20 runs)
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    __syncthreads();
    float* Xs = smem + buf * buf_floats;
    float* Ys = Xs + p.xs_floats;
    const bool more = sidx + s_step < s_end && !(p.dbg & 1);
    if (more) {                                // in flight during the transform and the MFMA loop
      int nn, ni0;
      locate(sidx + s_step, nn, ni0);
      issue_dma(nn, ni0, buf ^ 1);
    }
    if (FUSED && !(p.dbg & 16)) {
      // dA -> dU in place (rows past the image arrive as zeros and stay zeros)
      const float* Us = Ys + p.ys_floats;
#pragma unroll
      for (int k = 0; k < MAXE; ++k) {
        const int e = tid + 256 * k;
        if (e < evalid) {
          const f32x4 g = *reinterpret_cast<const f32x4*>(Ys + e * 4);
          const f32x4 uv = *reinterpret_cast<const f32x4*>(Us + e * 4);
          f32x4 d;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const bool pos = uv[c] > 0.f;
            d[c] = pos ? g[c] : g[c] * al[k][c];
            dal[k][c] += pos ? 0.f : g[c] * uv[c];
            dbs[c] += d[c];
          }
          *reinterpret_cast<f32x4*>(Ys + e * 4) = d;
        }
      }
      // LDS-only barrier: a __syncthreads() here would also wait for the DMA pieces issued just above
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    int q = 4 * ks0 + lg;
    int pr = q / p.Wy, pj = q - pr * p.Wy;
    const int ks_end = (p.dbg & 2) ? ks0 : ks1;
    // Fragment loads of k-step ks+1 are issued before the MFMAs of k-step ks (two register sets).  No masking of the
    // operands: pixels past the strip read the zero-filled tail of Ys (their X address is clamped to a valid one), and
    // the rows of the non-existent tap 9 (dW rows 72..79) are never written out.
    const int qlast = 4 * ksteps - 1;
    auto load = [&](float (&bf)[NBK], float (&af)[MB]) {
      const bool qv = q < strip_px;
      const int base = qv ? (pr * XC + pj) * CX : 0;
#pragma unroll
      for (int nb = 0; nb < NBK; ++nb) bf[nb] = Ys[min(q, qlast) * CY + nb * 16 + l15];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) af[mb] = Xs[base + toff[mb]];
      q += 4;
      pj += 4;
      const int wrap = pj >= p.Wy ? 1 : 0;
      pj -= wrap ? p.Wy : 0;
      pr += wrap;
    };
    auto mma = [&](const float (&bf)[NBK], const float (&af)[MB]) {
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBK; ++nb)
          acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[mb], bf[nb], acc[mb][nb], 0, 0, 0);
    };
    // branch-free pairs, so that the compiler can count the LDS reads in flight (a load past the last k-step
    // re-reads the strip's zero tail and is not used)
    float bf0[NBK], af0[MB], bf1[NBK], af1[MB];
    const int nks = max(ks_end - ks0, 0);
    __builtin_amdgcn_s_waitcnt(0xC07F);      // no scalar load in flight: lets the loop wait on LDS-read counts only
    load(bf0, af0);
    for (int i = 0; i < (nks >> 1); ++i) {
      load(bf1, af1);
      __builtin_amdgcn_sched_barrier(0);     // keep the reads ahead of the MFMAs they hide behind
      mma(bf0, af0);
      __builtin_amdgcn_sched_barrier(0);
      load(bf0, af0);
      __builtin_amdgcn_sched_barrier(0);
      mma(bf1, af1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (nks & 1) mma(bf0, af0);
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
  if (FUSED) {
    const int g = (int)blockIdx.x / p.strips_per_stamp;
    float* dst = p.dal_part + (size_t)g * p.alpha_elems + (size_t)rb * p.R * p.Wy * CY;
#pragma unroll
    for (int k = 0; k < MAXE; ++k) {
      const int e = tid + 256 * k;
      if (e < evalid) *reinterpret_cast<f32x4*>(dst + e * 4) = dal[k];
    }
    // d(bias): a thread's elements all belong to channel quad tid % 8
    __syncthreads();
    f32x4* shv = reinterpret_cast<f32x4*>(smem);
    shv[tid] = dbs;
    __syncthreads();
    if (tid < CY4) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
      for (int t = tid; t < 256; t += CY4) a += shv[t];
      *reinterpret_cast<f32x4*>(p.db_part + (size_t)blockIdx.x * CY + tid * 4) = a;
    }
  }
}

bool wgrad_strip8_fusable(int Hy, int Wy) {
  // the fused form keeps two output rows per strip: 2 * Wy * 8 float4 elements over 256 threads x 4
  return Wy >= 4 && (size_t)std::min(Hy, 2) * Wy * 8 <= 1024;
}

static int launch_strip8(WStripParams p, hipStream_t s, int* nsplit_out) {
  const bool fused = p.U != nullptr;
  const int XC = p.Wy + 2;
  auto xfl = [&](int r) { return (((size_t)(r + 2) * XC * 2 + 63) / 64) * 256; };
  auto yfl = [&](int r) { return (((size_t)r * p.Wy * 8 + 63) / 64) * 256; };
  auto fits = [&](int r) {
    return 2 * (xfl(r) + (fused ? 2 : 1) * yfl(r)) * sizeof(float) <= 128 * 1024 && xfl(r) / 256 <= 4 * 8 &&
           yfl(r) / 256 <= 4 * 12;
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
  int wgs;
  if (fused) {
    if ((size_t)R * p.Wy * 8 > 1024) return 1;
    size_t cap = std::min(p.part_capacity / ((size_t)72 * 32), p.db_capacity / 32) / p.strips_per_stamp;
    cap = std::min(cap, p.dal_capacity / (size_t)p.alpha_elems);
    if (cap < 1) return 1;
    // two 78 KB workgroups per CU: one full round of 512
    const int target = 512;
    p.groups = (int)std::min<size_t>(std::min(p.NB, std::max(1, target / p.strips_per_stamp)), cap);
    p.strips_per_wg = (p.NB + p.groups - 1) / p.groups;
    wgs = p.groups * p.strips_per_stamp;
  } else {
    wgs = std::min(p.nstrips, 768);
    wgs = std::max(1, std::min(wgs, (int)(p.part_capacity / ((size_t)72 * 32))));
    p.strips_per_wg = (p.nstrips + wgs - 1) / wgs;
    wgs = (p.nstrips + p.strips_per_wg - 1) / p.strips_per_wg;
  }
  const size_t smem =
      std::max((size_t)2 * (p.xs_floats + (fused ? 2 : 1) * p.ys_floats), (size_t)4 * 80 * 32) * sizeof(float);
  auto kern = fused ? wgrad_strip8_kernel<true> : wgrad_strip8_kernel<false>;
  static size_t attr[2] = {0, 0};
  if (smem > attr[fused]) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr[fused] = smem;
  }
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), smem, s, p);
  DV_HIP(hipGetLastError());
  *nsplit_out = wgs;
  if (fused) *p.groups_out = p.groups;
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
