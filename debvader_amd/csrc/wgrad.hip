// Weight-gradient contraction on the fp32 matrix cores, split over pixel ranges.
//
// dW[(t, cx), cy] = sum over class-grid pixels p=(nb,i,j) of  X[nb, i*sx+dh[t], j*sx+dw[t], cx] * Y[nb, i*sy+ph, j*sy+pw, cy]
// covers d(kernel) of Conv2D (X = layer input, Y = d(pre-activation); reference model.py:81-91,137),
// of Conv2DTranspose (X = d(pre-activation) of the layer output, Y = layer input; model.py:121-134) and
// of Dense (one tap, 1x1 grid; model.py:96-98,114,117).  The reduction dimension (pixels) is the long one,
// so every workgroup owns one (row tile, column tile) of dW for a contiguous pixel range and writes
// a partial slab; reduce_partials() sums the slabs in a fixed order, which keeps results bit-reproducible.
#include <stdlib.h>

#include "common.h"

namespace dv {

constexpr int BKP = 32;  // pixels per LDS stage

// tap table entry (common.h TapTab): (dh + 8) | (dw + 8) << 4 | weight tap index << 8
__device__ __forceinline__ int wtap_dh(unsigned e) { return (int)(e & 15u) - 8; }
__device__ __forceinline__ int wtap_dw(unsigned e) { return (int)((e >> 4) & 15u) - 8; }
__device__ __forceinline__ int wtap_wt(unsigned e) { return (int)(e >> 8); }

template <int V>
struct WSet {
  static constexpr int value = V;
};

template <int BMW, int BNW, int WGM, int WGN>
__global__ __launch_bounds__(256) void wgrad_kernel(const WGradParams p) {
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int WM = BMW / WGM, WN = BNW / WGN;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int LDAW = BMW + 16, LDBW = BNW + 16;   // lane groups lg, lg+1 land 16 banks apart: conflict-free b32 reads
  constexpr int AQ = (BMW + 31) / 32, BQ = (BNW + 31) / 32;
  constexpr int A_ELEMS = BKP * LDAW, B_ELEMS = BKP * LDBW;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;
  __shared__ unsigned s_tap[32];     // a row's tap varies per lane: the table is read from LDS (prologue and epilogue only)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int l15 = lane & 15, lg = lane >> 4;
  if (tid < 32) s_tap[tid] = p.xt.t[tid < p.ntaps ? tid : 0];
  __syncthreads();

  const int rows_launch = p.ntaps * p.Cx;
  const int ntn = (p.Cy + BNW - 1) / BNW;
  const int m0 = (blockIdx.x / ntn) * BMW;
  const int n0 = (blockIdx.x % ntn) * BNW;
  const int split = blockIdx.y;
  const int pstart = split * p.pchunk;
  const int pend = min(p.P, pstart + p.pchunk);

  const int ps = tid >> 3, q = tid & 7;
  int adh[AQ], adw[AQ], acx[AQ];
  bool aok[AQ];
#pragma unroll
  for (int i = 0; i < AQ; ++i) {
    int rl = 4 * (q + 8 * i);
    int r = m0 + rl;
    aok[i] = (rl < BMW) && (r < rows_launch);
    int t = aok[i] ? r / p.Cx : 0;
    acx[i] = r - t * p.Cx;
    adh[i] = wtap_dh(s_tap[t]);
    adw[i] = wtap_dw(s_tap[t]);
  }
  const int HcWc = p.Hc * p.Wc;

  // two register sets: a chunk's loads are issued two iterations before its LDS store (see gconv2.hip)
  f32x4 areg[2][AQ], breg[2][BQ];
  unsigned amask2[2] = {0, 0}, bmask2[2] = {0, 0};
  // unconditional loads from clamped addresses, masked at the LDS store (see gconv.hip)
  auto load_global = [&](auto setc, int kc) {
    constexpr int S = decltype(setc)::value;
    unsigned amask = 0, bmask = 0;
    const int pp = pstart + kc * BKP + ps;
    const bool pv = pp < pend;
    int nb = 0, ii = 0, jj = 0;
    if (pv) {
      // exact n / d for any 32-bit n (Granlund-Montgomery: t = mulhi(m, n); q = (t + ((n - t) >> s1)) >> s2)
      const unsigned n = (unsigned)pp;
      const unsigned t1 = __umulhi(p.div_hw_m, n);
      nb = (int)((t1 + ((n - t1) >> p.div_hw_s1)) >> p.div_hw_s2);
      const unsigned rem = n - (unsigned)nb * (unsigned)HcWc;
      const unsigned t2 = __umulhi(p.div_w_m, rem);
      ii = (int)((t2 + ((rem - t2) >> p.div_w_s1)) >> p.div_w_s2);
      jj = (int)rem - ii * p.Wc;
    }
    // (element offsets fit 32 bits: launch_wgrad refuses tensors of 2^31 elements or more)
    const int xrow0 = nb * p.Hx;
#pragma unroll
    for (int i = 0; i < AQ; ++i) {
      int ih = ii * p.sx + adh[i], iw = jj * p.sx + adw[i];
      bool ok = pv && aok[i] && (unsigned)ih < (unsigned)p.Hx && (unsigned)iw < (unsigned)p.Wx;
      const unsigned off = ok ? (unsigned)(((xrow0 + ih) * p.Wx + iw) * p.Cx + acx[i]) : 0u;
      areg[S][i] = *reinterpret_cast<const f32x4*>(p.X + off);
      amask |= (ok ? 1u : 0u) << i;
    }
    const unsigned ypix = (unsigned)(((nb * p.Hy + ii * p.sy + p.ph) * p.Wy + jj * p.sy + p.pw) * p.Cy);
#pragma unroll
    for (int i = 0; i < BQ; ++i) {
      int cl = 4 * (q + 8 * i);
      int c = n0 + cl;
      bool ok = pv && cl < BNW && c < p.Cy;
      breg[S][i] = *reinterpret_cast<const f32x4*>(p.Y + (ok ? ypix + (unsigned)c : 0u));
      bmask |= (ok ? 1u : 0u) << i;
    }
    amask2[S] = amask;
    bmask2[S] = bmask;
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto store_lds = [&](auto setc, int buf) {
    constexpr int S = decltype(setc)::value;
    const unsigned amask = amask2[S], bmask = bmask2[S];
    float* a = As + buf * A_ELEMS + ps * LDAW;
    float* b = Bs + buf * B_ELEMS + ps * LDBW;
#pragma unroll
    for (int i = 0; i < AQ; ++i)
      if (4 * (q + 8 * i) < BMW) *reinterpret_cast<f32x4*>(a + 4 * (q + 8 * i)) = ((amask >> i) & 1u) ? areg[S][i] : zero4;
#pragma unroll
    for (int i = 0; i < BQ; ++i)
      if (4 * (q + 8 * i) < BNW) *reinterpret_cast<f32x4*>(b + 4 * (q + 8 * i)) = ((bmask >> i) & 1u) ? breg[S][i] : zero4;
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
    for (int s = 0; s < BKP / 4; ++s) {
      float af[TM], bf[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm) af[tm] = a[(4 * s + lg) * LDAW + wm0 + tm * 16 + l15];
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) bf[tn] = b[(4 * s + lg) * LDBW + wn0 + tn * 16 + l15];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
          acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm], bf[tn], acc[tm][tn], 0, 0, 0);
    }
  };

  const int npix = max(pend - pstart, 0);
  const int nchunks = (npix + BKP - 1) / BKP;
  if (nchunks > 0) {
    load_global(WSet<0>{}, 0);
    store_lds(WSet<0>{}, 0);
    if (nchunks > 1) load_global(WSet<1>{}, 1);
  }
  __syncthreads();
  {
    auto iter = [&](auto ld, auto st, int kc, int cur) {   // loads chunk kc+2, computes chunk kc, stores chunk kc+1
      load_global(ld, kc + 2);
      compute(cur);
      store_lds(st, cur ^ 1);
      __syncthreads();
    };
    int kc = 0;
    for (; kc + 3 < nchunks; kc += 2) {
      iter(WSet<0>{}, WSet<1>{}, kc, 0);
      iter(WSet<1>{}, WSet<0>{}, kc + 1, 1);
    }
    const int r = nchunks - kc;
    if (r == 3) {
      iter(WSet<0>{}, WSet<1>{}, kc, 0);
      compute(1);
      store_lds(WSet<0>{}, 0);
      __syncthreads();
      compute(0);
    } else if (r == 2) {
      compute(0);
      store_lds(WSet<1>{}, 1);
      __syncthreads();
      compute(1);
    } else if (r == 1) {
      compute(0);
    }
  }

  float* slab = p.part + (size_t)split * p.rows_total * p.Cy;
#pragma unroll
  for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int rl = m0 + wm0 + tm * 16 + lg * 4 + r;
      if (rl >= rows_launch) continue;
      int t = rl / p.Cx;
      int cx = rl - t * p.Cx;
      size_t grow = (size_t)(wtap_wt(s_tap[t]) * p.Cx + cx) * p.Cy;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        int c = n0 + wn0 + tn * 16 + l15;
        if (c < p.Cy) slab[grow + c] = acc[tm][tn][r];
      }
    }
  }
}

template <int BMW, int BNW, int WGM, int WGN>
static int launch_w(const WGradParams& p, hipStream_t s) {
  constexpr size_t smem = (size_t)(2 * BKP * (BMW + 16) + 2 * BKP * (BNW + 16)) * sizeof(float);
  auto kern = wgrad_kernel<BMW, BNW, WGM, WGN>;
  static bool attr_set = false;
  if (!attr_set) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr_set = true;
  }
  int rows = p.ntaps * p.Cx;
  dim3 grid(((rows + BMW - 1) / BMW) * ((p.Cy + BNW - 1) / BNW), p.nsplit), block(256);
  hipLaunchKernelGGL(kern, grid, block, smem, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

static void fast_div_consts(unsigned d, unsigned* m, unsigned* s1, unsigned* s2) {
  unsigned l = 0;
  while ((1ull << l) < d) ++l;                     // ceil(log2 d), d >= 1
  *m = (unsigned)((((1ull << 32) * ((1ull << l) - d)) / d) + 1);
  *s1 = l < 1 ? l : 1;
  *s2 = l - *s1;
}

int launch_wgrad(const WGradParams& p0, hipStream_t s) {
  if (p0.P <= 0) return OK;
  WGradParams p = p0;
  if (p.Hc < 1 || p.Wc < 1 || (long)p.NB * p.Hx * p.Wx * p.Cx >= (1L << 31) || (long)p.NB * p.Hy * p.Wy * p.Cy >= (1L << 31)) {
    set_error("wgrad: tensor too large for 32-bit element offsets; lower max_batch");
    return E_INVALID;
  }
  fast_div_consts((unsigned)(p.Hc * p.Wc), &p.div_hw_m, &p.div_hw_s1, &p.div_hw_s2);
  fast_div_consts((unsigned)p.Wc, &p.div_w_m, &p.div_w_s1, &p.div_w_s2);
  if (p.ntaps < 1 || p.ntaps > DV_MAX_TAPS || p.ntaps != p.xt.n) {
    set_error("wgrad: bad tap table (ntaps=%d)", p.ntaps);
    return E_INVALID;
  }
  if ((p.Cx & 3) || (p.Cy & 3) || (p.pchunk % BKP) || p.nsplit < 1 || (long)p.nsplit * p.pchunk < p.P) {
    set_error("wgrad: bad parameters (Cx=%d Cy=%d pchunk=%d nsplit=%d P=%d)", p.Cx, p.Cy, p.pchunk, p.nsplit, p.P);
    return E_INVALID;
  }
  const int rows = p.ntaps * p.Cx;
  if (p.Cy <= 16) return launch_w<128, 16, 4, 1>(p, s);
  if (p.Cy <= 32) return rows <= 32 ? launch_w<32, 32, 2, 2>(p, s) : launch_w<64, 32, 4, 1>(p, s);
  if (p.Cy <= 64) return rows <= 64 ? launch_w<64, 64, 2, 2>(p, s) : launch_w<128, 64, 2, 2>(p, s);
  return rows <= 64 ? launch_w<64, 128, 2, 2>(p, s) : launch_w<128, 128, 2, 2>(p, s);
}

// ---------------------------------------------------------------------------------------------
// out[(r/cpad)*creal + r%cpad][c] = sum_s part[s][r][c]; 4 consecutive columns per thread, splits summed in
// a fixed order with 4 independent accumulators so the loads of different slabs overlap.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                              int nsplit, int slab4, int ncols4, int cpad, int creal) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= slab4) return;
  const int r = e / ncols4;
  const int c4 = e - r * ncols4;
  const int ci = r % cpad;
  if (ci >= creal) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(part) + e;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    a0 += p[(size_t)(s + 0) * slab4];
    a1 += p[(size_t)(s + 1) * slab4];
    a2 += p[(size_t)(s + 2) * slab4];
    a3 += p[(size_t)(s + 3) * slab4];
  }
  for (; s < nsplit; ++s) a0 += p[(size_t)s * slab4];
  const int ro = (r / cpad) * creal + ci;
  reinterpret_cast<f32x4*>(out)[(size_t)ro * ncols4 + c4] = (a0 + a1) + (a2 + a3);
}

// small slabs with many splits: a block owns 16 float4 columns and 16 split lanes (each sums every 16th slab
// with 4 loads in flight), then the 16 partial sums are added in a fixed order through LDS.
__global__ __launch_bounds__(256) void reduce_partials_wide_kernel(const float* __restrict__ part,
                                                                   float* __restrict__ out, int nsplit, int slab4,
                                                                   int ncols4, int cpad, int creal) {
  __shared__ f32x4 sh[16][17];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + cl;
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
  if (e < slab4) {
    const f32x4* p = reinterpret_cast<const f32x4*>(part) + e;
    int s = sl;
    for (; s + 48 < nsplit; s += 64) {
      a0 += p[(size_t)(s + 0) * slab4];
      a1 += p[(size_t)(s + 16) * slab4];
      a2 += p[(size_t)(s + 32) * slab4];
      a3 += p[(size_t)(s + 48) * slab4];
    }
    for (; s < nsplit; s += 16) a0 += p[(size_t)s * slab4];
  }
  sh[sl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && e < slab4) {
    f32x4 v = sh[0][cl];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += sh[k][cl];
    const int r = e / ncols4;
    const int c4 = e - r * ncols4;
    const int ci = r % cpad;
    if (ci < creal) reinterpret_cast<f32x4*>(out)[(size_t)((r / cpad) * creal + ci) * ncols4 + c4] = v;
  }
}

// scalar variant for column counts that are not a multiple of 4 (d(alpha) slabs use ncols = 1)
__global__ __launch_bounds__(256) void reduce_partials_scalar_kernel(const float* __restrict__ part,
                                                                     float* __restrict__ out, int nsplit, int slab) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= slab) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    a0 += part[(size_t)(s + 0) * slab + e];
    a1 += part[(size_t)(s + 1) * slab + e];
    a2 += part[(size_t)(s + 2) * slab + e];
    a3 += part[(size_t)(s + 3) * slab + e];
  }
  for (; s < nsplit; ++s) a0 += part[(size_t)s * slab + e];
  out[e] = (a0 + a1) + (a2 + a3);
}

// All slab reductions of a backward pass in one launch (grid.y = entry): the same two fixed-order strategies as above,
// chosen per entry (many small slabs: 16 split lanes per column group; few large ones: one thread per float4 column).
__global__ __launch_bounds__(256) void reduce_partials_batch_kernel(const WRedBatch b) {
  __shared__ f32x4 sh[16][17];
  const WRedEntry d = b.e[blockIdx.y];
  const f32x4* part = reinterpret_cast<const f32x4*>(d.part);
  f32x4* out = reinterpret_cast<f32x4*>(d.out);
  if (d.nsplit >= 16) {
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    for (int e0 = blockIdx.x * 16; e0 < d.slab4; e0 += gridDim.x * 16) {      // (uniform trip count per block)
      const int e = e0 + cl;
      f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
      if (e < d.slab4) {
        const f32x4* p = part + e;
        int s = sl;
        for (; s + 48 < d.nsplit; s += 64) {
          a0 += p[(size_t)(s + 0) * d.slab4];
          a1 += p[(size_t)(s + 16) * d.slab4];
          a2 += p[(size_t)(s + 32) * d.slab4];
          a3 += p[(size_t)(s + 48) * d.slab4];
        }
        for (; s < d.nsplit; s += 16) a0 += p[(size_t)s * d.slab4];
      }
      __syncthreads();
      sh[sl][cl] = (a0 + a1) + (a2 + a3);
      __syncthreads();
      if (sl == 0 && e < d.slab4) {
        f32x4 v = sh[0][cl];
#pragma unroll
        for (int k = 1; k < 16; ++k) v += sh[k][cl];
        const int r = e / d.ncols4, c4 = e - r * d.ncols4, ci = r % d.cpad;
        if (ci < d.creal) out[(size_t)((r / d.cpad) * d.creal + ci) * d.ncols4 + c4] = v;
      }
    }
    return;
  }
  for (int e = blockIdx.x * 256 + threadIdx.x; e < d.slab4; e += gridDim.x * 256) {
    const int r = e / d.ncols4, c4 = e - r * d.ncols4, ci = r % d.cpad;
    if (ci >= d.creal) continue;
    const f32x4* p = part + e;
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
    int s = 0;
    for (; s + 4 <= d.nsplit; s += 4) {
      a0 += p[(size_t)(s + 0) * d.slab4];
      a1 += p[(size_t)(s + 1) * d.slab4];
      a2 += p[(size_t)(s + 2) * d.slab4];
      a3 += p[(size_t)(s + 3) * d.slab4];
    }
    for (; s < d.nsplit; ++s) a0 += p[(size_t)s * d.slab4];
    out[(size_t)((r / d.cpad) * d.creal + ci) * d.ncols4 + c4] = (a0 + a1) + (a2 + a3);
  }
}

int launch_reduce_partials_batch(const WRedBatch& b, hipStream_t s) {
  if (b.count <= 0) return OK;
  if (b.count > DV_WRED_MAX) return E_INVALID;
  for (int i = 0; i < b.count; ++i)
    if (b.e[i].slab4 <= 0 || b.e[i].ncols4 <= 0 || b.e[i].cpad < b.e[i].creal || b.e[i].nsplit < 1) return E_INVALID;
  hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3(128, (unsigned)b.count), dim3(256), 0, s, b);
  DV_HIP(hipGetLastError());
  return OK;
}

int launch_reduce_partials(const float* part, float* out, int nsplit, long slab_elems, int ncols, int cpad, int creal,
                           hipStream_t s) {
  if (slab_elems <= 0) return OK;
  if (slab_elems >= (1L << 31)) return E_INVALID;
  if ((ncols & 3) == 0 && (slab_elems & 3) == 0) {
    int slab4 = (int)(slab_elems / 4);
    if (nsplit >= 32 && slab4 <= 65536)
      hipLaunchKernelGGL(reduce_partials_wide_kernel, dim3((slab4 + 15) / 16), dim3(256), 0, s, part, out, nsplit,
                         slab4, ncols / 4, cpad, creal);
    else
      hipLaunchKernelGGL(reduce_partials_kernel, dim3((slab4 + 255) / 256), dim3(256), 0, s, part, out, nsplit,
                         slab4, ncols / 4, cpad, creal);
  } else {
    if (cpad != creal) {
      set_error("reduce_partials: row compaction needs ncols %% 4 == 0");
      return E_INVALID;
    }
    hipLaunchKernelGGL(reduce_partials_scalar_kernel, dim3((unsigned)((slab_elems + 255) / 256)), dim3(256), 0, s, part,
                       out, nsplit, (int)slab_elems);
  }
  DV_HIP(hipGetLastError());
  return OK;
}

// ---- dense kernel gradients -----------------------------------------------------------------------------------------
// G[i][j] = sum_b X[b][i] * Y[b][j]: the contraction index is the stamp and both operands are stamp-major, which is exactly
// what v_mfma_f32_16x16x4_f32 wants - its A operand is one value per lane, A[row = lane % 16][k = lane / 16], so the lanes
// of a 16-lane group read CONSECUTIVE columns of one stamp row.  A lane loads a float2 of X and a float4 of Y per four
// stamps (columns i0 + 2 (lane % 16) + {0, 1} and j0 + 4 (lane % 16) + {0 .. 3}); component q of such a load is the
// operand of a "virtual" 16-row block made of every second / fourth column, so one pair of loads feeds 2 x 4 MFMAs and the
// four accumulators that share a row block are four consecutive output columns: one 16-byte store per lane.  A wave owns
// 32 x 64 outputs and walks all stamps, four steps in flight; no LDS, no slabs, one fixed summation order.  The tiled
// wgrad_kernel took 28 us for each of the two 4096 x 560 layers of the 59-px net at 256 stamps (+ a slab-sum launch).
constexpr int DW_PF = 4;
__global__ __launch_bounds__(256) void dense_wgrad_tn_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ Y,
                                                             int ldy, int NB, int I, int J, float* __restrict__ G, int ldg) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int JT = (J + 63) >> 6, IT = (I + 31) >> 5;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= IT * JT) return;
  const int it = tile / JT, jt = tile - it * JT;
  const int lr = lane & 15, kq = lane >> 4;
  const int ic = it * 32 + 2 * lr, jc = jt * 64 + 4 * lr;        // first column of this lane's X pair / Y quad
  const bool xin = ic + 1 < I, xhalf = ic < I, yin = jc + 3 < J;  // (I even, J a multiple of 4: a quad is in or out)
  f32x4 acc[2][4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto load = [&](int m0, f32x2& x, f32x4& y) {
    const int m = m0 + kq;
    x = f32x2{0.f, 0.f};
    y = f32x4{0.f, 0.f, 0.f, 0.f};
    if (m < NB) {
      if (xin) x = *reinterpret_cast<const f32x2*>(X + (size_t)m * ldx + ic);
      else if (xhalf) x[0] = X[(size_t)m * ldx + ic];
      if (yin) y = *reinterpret_cast<const f32x4*>(Y + (size_t)m * ldy + jc);
    }
  };
  f32x2 rx[DW_PF];
  f32x4 ry[DW_PF];
#pragma unroll
  for (int q = 0; q < DW_PF; ++q) load(4 * q, rx[q], ry[q]);
  for (int m0 = 0; m0 < NB; m0 += 4 * DW_PF) {
#pragma unroll
    for (int q = 0; q < DW_PF; ++q) {
      if (m0 + 4 * q < NB) {
        const f32x2 x = rx[q];
        const f32x4 y = ry[q];
        load(m0 + 4 * (q + DW_PF), rx[q], ry[q]);                 // (past the last stamp: zeros, never used)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[a], y[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  // accumulator (a, b), register t: row 4 (lane / 16) + t of virtual block a = column i0 + 2 (4 (lane / 16) + t) + a of G's
  // rows, column lane % 16 of virtual block b = output column j0 + 4 (lane % 16) + b
  if (!yin) return;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int i = it * 32 + 2 * (4 * kq + t) + a;
      if (i < I)
        *reinterpret_cast<f32x4*>(G + (size_t)i * ldg + jc) = f32x4{acc[a][0][t], acc[a][1][t], acc[a][2][t], acc[a][3][t]};
    }
}

int launch_dense_wgrad_tn(const float* X, int ldx, const float* Y, int ldy, int NB, int I, int J, float* G, int ldg,
                          hipStream_t s) {
  if (NB < 1 || I < 1 || J < 4 || (I & 1) || (J & 3) || (ldx & 1) || (ldy & 3) || (ldg & 3)) {
    set_error("dense_wgrad_tn: widths must be even (X) / multiples of 4 (Y, G) (I %d, J %d, strides %d / %d / %d)", I, J, ldx,
              ldy, ldg);
    return E_INVALID;
  }
  const int tiles = ((I + 31) / 32) * ((J + 63) / 64);
  hipLaunchKernelGGL(dense_wgrad_tn_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, X, ldx, Y, ldy, NB, I, J, G,
                     ldg);
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
