// Gather-GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).
//
// One kernel serves Conv2D forward (reference model.py:81-91,137), Conv2DTranspose forward
// (model.py:121-134; stride 2 is launched as four output-parity classes so that no zero-inserted
// taps are multiplied), both data gradients, and the Dense layers (model.py:96-98,114,117).
// See common.h for the contraction it computes.
//
// Tiling: 256 threads = 4 waves (64 lanes each). Workgroup tile BM x BN, K consumed in chunks of
// 32 through double-buffered LDS with register staging (global -> VGPR -> LDS) so the next chunk's
// gather is in flight while the current chunk is on the MFMA pipe.  A rows are stored [m][k] with a
// 36-float stride: every lane reads its four k values with one ds_read_b128 (the MFMA k order is a
// free permutation, so lane group g of MFMA jj takes physical k = 16q + 4g + jj for A and B alike).
#include "common.h"

namespace dv {

constexpr int BK = 32;
constexpr int LDA = BK + 4;

// tap table entry (common.h TapTab): (dh + 8) | (dw + 8) << 4 | weight tap index << 8
__device__ __forceinline__ int tap_dh(unsigned e) { return (int)(e & 15u) - 8; }
__device__ __forceinline__ int tap_dw(unsigned e) { return (int)((e >> 4) & 15u) - 8; }
__device__ __forceinline__ int tap_wt(unsigned e) { return (int)(e >> 8); }

template <int BM, int BN, int WGM, int WGN, bool NMAJOR>
__global__ __launch_bounds__(256) void gconv_kernel(const GConvParams p) {
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int AROWS = BM / 32;                 // A rows gathered per thread
  constexpr int LDBK = BN + 4;                   // k-major B row stride
  constexpr int A_ELEMS = BM * LDA;
  constexpr int B_ELEMS = NMAJOR ? BN * LDA : BK * LDBK;
  constexpr int BROWS_N = (BN + 31) / 32;        // n-major: rows per thread
  constexpr int BQ = BN / 4;                     // k-major: float4 per k row
  constexpr int BKR = 256 / BQ;                  // k-major: k rows covered per pass
  constexpr int BPASS = (BK + BKR - 1) / BKR;    // k-major: passes per thread

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                              // [2][BM][LDA]
  float* Bs = smem + 2 * A_ELEMS;                // [2][...]
  // the tap of a K element varies per lane here (generic K decode), so the table sits in LDS; entries past the last
  // tap repeat tap 0 (K elements past the end are masked anyway)
  __shared__ unsigned s_tap[32];

  const int tid = threadIdx.x;
  if (tid < 32) s_tap[tid] = p.xt.t[tid < p.ntaps ? tid : 0];
  __syncthreads();
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm0 = (wave / WGN) * WM;
  const int wn0 = (wave % WGN) * WN;
  const int l15 = lane & 15;
  const int lg = lane >> 4;

  // tile coordinates: N tiles fastest so that workgroups sharing an A panel are adjacent
  const int ntn = (p.Cout + BN - 1) / BN;
  const int tile = blockIdx.x;
  const int m0 = (tile / ntn) * BM;
  const int n0 = (tile % ntn) * BN;

  // ---- per-thread gather rows -----------------------------------------------------------
  const int kq = tid & 7;
  const int r0 = tid >> 3;
  int rbase[AROWS], rih[AROWS], rjw[AROWS];
  const int HcWc = p.Hc * p.Wc;
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int m = m0 + r0 + 32 * i;
    if (m < p.M) {
      int nb = m / HcWc;
      int rem = m - nb * HcWc;
      int ii = rem / p.Wc;
      int jj = rem - ii * p.Wc;
      rbase[i] = nb * p.Hin * p.Win;
      rih[i] = ii * p.sin;
      rjw[i] = jj * p.sin;
    } else {
      rbase[i] = 0;
      rih[i] = -1000000;
      rjw[i] = 0;
    }
  }

  f32x4 areg[AROWS];
  f32x4 breg[NMAJOR ? BROWS_N : BPASS];
  unsigned amask = 0, bmask = 0;   // which staged vectors are real data (the rest are stored as zeros)

  // Loads are issued unconditionally from a clamped (always valid) address and masked when they are written
  // to LDS: a branch around a load makes hipcc wait vmcnt(0) right behind it, which would serialise the gather
  // with the MFMA work it is meant to overlap.
  auto load_global = [&](int kc) {
    const int kk = kc * BK + kq * 4;
    int tap, ci;
    if (p.cin_shift >= 0) {
      tap = kk >> p.cin_shift;
      ci = kk & (p.Cin - 1);
    } else {
      tap = kk / p.Cin;
      ci = kk - tap * p.Cin;
    }
    const bool kvalid = kk < p.K;
    const unsigned te = s_tap[kvalid ? tap : 0];
    const int dh = tap_dh(te), dw = tap_dw(te);
    amask = 0;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      int ih = rih[i] + dh, iw = rjw[i] + dw;
      bool ok = kvalid && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
      size_t off = ok ? ((size_t)(rbase[i] + ih * p.Win + iw)) * p.Cin + ci : 0;
      areg[i] = *reinterpret_cast<const f32x4*>(p.X + off);
      amask |= (ok ? 1u : 0u) << i;
    }
    bmask = 0;
    if (NMAJOR) {
      const int wt = tap_wt(te);
#pragma unroll
      for (int i = 0; i < BROWS_N; ++i) {
        int n = n0 + r0 + 32 * i;
        bool ok = kvalid && n < p.Cout && r0 + 32 * i < BN;
        size_t off = ok ? ((size_t)(wt * p.Cout + n)) * p.Cin + ci : 0;
        breg[i] = *reinterpret_cast<const f32x4*>(p.W + off);
        bmask |= (ok ? 1u : 0u) << i;
      }
    } else {
      const int nq = tid % BQ;
      const int kr0 = tid / BQ;
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        int kr = kr0 + BKR * i;
        int kb = kc * BK + kr;
        int n = n0 + nq * 4;
        bool ok = kr < BK && kb < p.K && n < p.Cout;
        int tb, cb;
        if (p.cin_shift >= 0) {
          tb = kb >> p.cin_shift;
          cb = kb & (p.Cin - 1);
        } else {
          tb = kb / p.Cin;
          cb = kb - tb * p.Cin;
        }
        int wt = tap_wt(s_tap[ok ? tb : 0]);
        size_t off = ok ? ((size_t)(wt * p.Cin + cb)) * p.Cout + n : 0;
        breg[i] = *reinterpret_cast<const f32x4*>(p.W + off);
        bmask |= (ok ? 1u : 0u) << i;
      }
    }
  };

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto store_lds = [&](int buf) {
    float* a = As + buf * A_ELEMS;
#pragma unroll
    for (int i = 0; i < AROWS; ++i)
      *reinterpret_cast<f32x4*>(a + (r0 + 32 * i) * LDA + kq * 4) = ((amask >> i) & 1u) ? areg[i] : zero4;
    float* b = Bs + buf * B_ELEMS;
    if (NMAJOR) {
#pragma unroll
      for (int i = 0; i < BROWS_N; ++i)
        if (r0 + 32 * i < BN)
          *reinterpret_cast<f32x4*>(b + (r0 + 32 * i) * LDA + kq * 4) = ((bmask >> i) & 1u) ? breg[i] : zero4;
    } else {
      const int nq = tid % BQ;
      const int kr0 = tid / BQ;
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        int kr = kr0 + BKR * i;
        if (kr < BK) *reinterpret_cast<f32x4*>(b + kr * LDBK + nq * 4) = ((bmask >> i) & 1u) ? breg[i] : zero4;
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto compute = [&](int buf) {
    const float* a = As + buf * A_ELEMS;
    const float* b = Bs + buf * B_ELEMS;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      f32x4 af[TM];
      f32x4 bf[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        af[tm] = *reinterpret_cast<const f32x4*>(a + (wm0 + tm * 16 + l15) * LDA + q * 16 + lg * 4);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        if (NMAJOR) {
          bf[tn] = *reinterpret_cast<const f32x4*>(b + (wn0 + tn * 16 + l15) * LDA + q * 16 + lg * 4);
        } else {
          const float* bp = b + (q * 16 + lg * 4) * LDBK + wn0 + tn * 16 + l15;
          bf[tn] = (f32x4){bp[0], bp[LDBK], bp[2 * LDBK], bp[3 * LDBK]};
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm][jj], bf[tn][jj], acc[tm][tn], 0, 0, 0);
    }
  };

  const int nchunks = (p.K + BK - 1) / BK;
  load_global(0);
  store_lds(0);
  __syncthreads();
  if (p.dbg == 0) {
    for (int kc = 0; kc < nchunks; ++kc) {
      const int cur = kc & 1;
      if (kc + 1 < nchunks) load_global(kc + 1);
      compute(cur);
      if (kc + 1 < nchunks) store_lds(cur ^ 1);
      __syncthreads();
    }
  } else if (p.dbg == 1) {  // ablation: MFMA + LDS reads only (timing aid, results are wrong)
    for (int kc = 0; kc < nchunks; ++kc) compute(kc & 1);
  } else if (p.dbg == 2) {  // ablation: no global loads, keep LDS stores + barrier
    for (int kc = 0; kc < nchunks; ++kc) {
      const int cur = kc & 1;
      compute(cur);
      if (kc + 1 < nchunks) store_lds(cur ^ 1);
      __syncthreads();
    }
  } else {                  // ablation: global loads + MFMA, no LDS store / barrier
    for (int kc = 0; kc < nchunks; ++kc) {
      if (kc + 1 < nchunks) load_global(kc + 1);
      compute(kc & 1);
      asm volatile("" ::"v"(areg[0]), "v"(breg[0]));
    }
  }

  // ---- epilogue: bias, per-element PReLU, scatter to the output pixel ---------------------
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm0 + tm * 16 + lg * 4 + r;
      if (m >= p.M) continue;
      int nb = m / HcWc;
      int rem = m - nb * HcWc;
      int ii = rem / p.Wc;
      int jj = rem - ii * p.Wc;
      int oh = ii * p.sout + p.ph, ow = jj * p.sout + p.pw;
      size_t apix = (size_t)(oh * p.Wout + ow) * p.Cout;
      size_t opix = ((size_t)nb * p.Hout * p.Wout) * p.Cout + apix;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn0 + tn * 16 + l15;
        if (n >= p.Cout) continue;
        float v = acc[tm][tn][r];
        if (p.epi >= 1) v += p.bias[n];
        if (p.U) p.U[opix + n] = v;
        if (p.epi == 2) {
          float al = p.alpha[apix + n];
          float av = v > 0.f ? v : al * v;
          p.A[opix + n] = av;
        }
      }
    }
  }
}

template <int BM, int BN, int WGM, int WGN, bool NMAJOR>
static int launch_cfg(const GConvParams& p, hipStream_t s) {
  constexpr int A_ELEMS = BM * LDA;
  constexpr int B_ELEMS = NMAJOR ? BN * LDA : BK * (BN + 4);
  constexpr size_t smem = (size_t)(2 * A_ELEMS + 2 * B_ELEMS) * sizeof(float);
  static bool attr_set = false;
  auto kern = gconv_kernel<BM, BN, WGM, WGN, NMAJOR>;
  if (!attr_set) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr_set = true;
  }
  int tm = (p.M + BM - 1) / BM, tn = (p.Cout + BN - 1) / BN;
  dim3 grid(tm * tn), block(256);
  hipLaunchKernelGGL(kern, grid, block, smem, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

static int g_tile_override = -1;
static int g_dbg = 0;
void debug_set_gconv_tile(int code) {
  g_tile_override = code < 0 ? -1 : code % 100;
  g_dbg = code < 0 ? 0 : code / 100;
}

// pure MFMA issue-rate probe: NACC independent accumulators per wave, no memory traffic in the loop; operands
// either trivial or pseudo-random (data-dependent power -> clock).  Writes shader cycles (s_memtime) and
// 100 MHz ticks (s_memrealtime) of block 0 so the in-kernel clock can be derived.
template <int NACC>
__global__ __launch_bounds__(256) void mfma_peak_kernel(float* out, int iters, int randomize) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a[4], b[4];
  unsigned st = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    st = st * 1664525u + 1013904223u;
    a[i] = randomize ? ((st >> 8) * (1.0f / 8388608.0f) - 1.0f) : 0.001f * threadIdx.x;
    st = st * 1664525u + 1013904223u;
    b[i] = randomize ? ((st >> 8) * (1.0f / 8388608.0f) - 1.0f) : 1.0f;
  }
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  f32x4 t = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) t += acc[i];
  out[16 + blockIdx.x * 256 + threadIdx.x] = t[0] + t[1] + t[2] + t[3];
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    out[0] = (float)(t1 - t0);
    out[1] = (float)(r1 - r0);
  }
}
int debug_mfma_peak(float* out, int blocks, int iters, int nacc, int randomize, hipStream_t s) {
  if (nacc == 36)
    hipLaunchKernelGGL(mfma_peak_kernel<36>, dim3(blocks), dim3(256), 0, s, out, iters, randomize);
  else
    hipLaunchKernelGGL(mfma_peak_kernel<16>, dim3(blocks), dim3(256), 0, s, out, iters, randomize);
  DV_HIP(hipGetLastError());
  return OK;
}

template <bool NMAJOR>
static int dispatch(const GConvParams& p, hipStream_t s) {
  const int N = p.Cout;
  switch (g_tile_override) {
    case 0: return launch_cfg<128, 128, 2, 2, NMAJOR>(p, s);
    case 1: return launch_cfg<128, 64, 2, 2, NMAJOR>(p, s);
    case 2: return launch_cfg<64, 64, 2, 2, NMAJOR>(p, s);
    case 3: return launch_cfg<128, 32, 4, 1, NMAJOR>(p, s);
    case 4: return launch_cfg<32, 64, 2, 2, NMAJOR>(p, s);
    case 5: return launch_cfg<128, 16, 4, 1, NMAJOR>(p, s);
    default: break;
  }
  // pick the tile so that the grid still fills 256 CUs when M is small (deep, low-resolution layers)
  if (N <= 16) return launch_cfg<128, 16, 4, 1, NMAJOR>(p, s);
  if (N <= 32) return launch_cfg<128, 32, 4, 1, NMAJOR>(p, s);
  if (N <= 64) {
    long tiles = ((long)(p.M + 127) / 128);
    if (tiles >= 256) return launch_cfg<128, 64, 2, 2, NMAJOR>(p, s);
    return launch_cfg<64, 64, 2, 2, NMAJOR>(p, s);
  }
  long tiles128 = ((long)(p.M + 127) / 128) * ((N + 127) / 128);
  if (tiles128 >= 256) return launch_cfg<128, 128, 2, 2, NMAJOR>(p, s);
  long tiles64 = ((long)(p.M + 63) / 64) * ((N + 63) / 64);
  if (tiles64 >= 192) return launch_cfg<64, 64, 2, 2, NMAJOR>(p, s);
  return launch_cfg<32, 64, 2, 2, NMAJOR>(p, s);
}

int launch_gconv(const GConvParams& p, hipStream_t s) {
  if (p.M <= 0) return OK;
  if ((p.Cin & 3) || (p.Cout & 3)) {
    set_error("gconv: Cin (%d) and Cout (%d) must be multiples of 4", p.Cin, p.Cout);
    return E_INVALID;
  }
  if (p.ntaps < 0 || p.ntaps > DV_MAX_TAPS || p.ntaps != p.xt.n || p.K != p.ntaps * p.Cin) {
    set_error("gconv: bad tap table (ntaps=%d K=%d Cin=%d)", p.ntaps, p.K, p.Cin);
    return E_INVALID;
  }
  if ((long)p.NB * p.Hin * p.Win * p.Cin >= (1L << 31) || (long)p.NB * p.Hout * p.Wout * p.Cout >= (1L << 31)) {
    set_error("gconv: tensor too large for 32-bit pixel indexing; lower the batch chunk");
    return E_INVALID;
  }
  if (p.epi == 2 && (!p.alpha || !p.A)) {
    set_error("gconv: PReLU epilogue needs alpha and A");
    return E_INVALID;
  }
  GConvParams q = p;
  q.dbg = g_dbg;
  return q.w_nmajor ? dispatch<true>(q, s) : dispatch<false>(q, s);
}

}  // namespace dv
