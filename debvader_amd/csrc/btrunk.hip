// Dense trunk of the bf16 engine on the bf16 matrix cores (round 6; gfx950 / MI355X only).
//
// Between the two conv stacks of the conv-VAE sit Flatten -> PReLU -> Dense(params_size) (model.py:94-98), the sampler and
// PReLU -> Dense(560) -> PReLU -> Dense(w*w*256) -> PReLU -> Reshape (model.py:112-118).  Until round 5 the bf16 engine ran
// that trunk on the fp32 engine's kernels between two layout conversions: ~25 launches of 5 - 30 us, 0.19 ms of a 1.97 ms
// step.  Its two large products and their data gradients are M = stamps (256), N x K = 4096 x 560 GEMMs: 1.2 GFLOP and 5 MB
// of weights each - latency, not work.  This file gives them one kernel,
//
//   bgemm_kernel<AMODE, EPI>   C[m][n] = sum_k A[m][k] * B[n][k]          (v_mfma_f32_16x16x32_bf16, fp32 accumulation)
//
// whose A side reads what the neighbouring layer already has and whose epilogue writes what the next one wants, so that
// no conversion, PReLU or PReLU-backward launch is left at the two seams:
//   A modes   STAMP        bf16 stamp-inner [P][NBp][C] of the conv stacks (bf16.h), k = p * C + c (Flatten's HWC order)
//             STAMP_PRELU  the same with PReLU(alpha[k]) applied on load (the flatten PReLU of model.py:95)
//             ROWS_F32     fp32 rows [NB][lda] of the sampler side, rounded to bf16 on load
//   epilogues SLAB         fp32 partial sums [nslab][M][ldc] of a K-split product (the consumer adds them in order)
//             STAMP_BIAS_PRELU  + bias[n], PReLU(alpha[n]) -> pre-activation and activation in bf16 stamp-inner
//                               (Dense -> PReLU -> Reshape((w, w, 256)), model.py:116-118)
//             STAMP_GATE2  d(flatten PReLU output) -> d(pre-activation of the last encoder conv): both PReLU gates
//                          (model.py:92,95) and the partial column sums of d(alpha) of either and of d(bias)
// and the two dense kernel gradients one more,
//
//   bgemm_tn_kernel<XMODE, YMODE>   G[i][j] = sum_m X[m][i] * Y[m][j]     (contraction over stamps, fp32 written once)
//
// Both are register-direct: a wave owns a 32 x 32 tile (2 x 2 MFMA blocks) and loads its fragments straight from global
// memory in MFMA order - the operands of a launch are 2 - 5 MB and live in L2; what these products need is many waves in
// flight, not operand reuse (a 64 x 64 LDS tile form of the fp32 engine took 22 us per product, this takes 4 - 7).
// Products with K = 4096 split K eight ways: four waves of a workgroup take consecutive K quarters of one tile and are summed
// through LDS in wave order, two such workgroups write two slabs.  Everything is deterministic (fixed summation orders,
// no atomics).
//
// Rounding points (restated in oracle/vae_oracle_bf16.py): the two dense kernels and every A operand are rounded to bf16
// (nearest even) where they enter an MFMA; accumulators, biases, slopes, gates and every gradient sum are fp32.
#include <algorithm>

#include "common.h"
#include "bf16.h"

namespace dv {

namespace {

typedef __bf16 bt_bf16;
typedef __bf16 bt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float bt_f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bt_bf16x8 bt_zero8() {
  bt_bf16x8 z;
#pragma unroll
  for (int i = 0; i < 8; ++i) z[i] = (bt_bf16)0.f;
  return z;
}

// eight fp32 -> eight bf16 (round to nearest even)
__device__ __forceinline__ bt_bf16x8 bt_pack8(bt_f32x4 lo, bt_f32x4 hi) {
  bt_bf16x8 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[i] = (bt_bf16)lo[i];
    o[4 + i] = (bt_bf16)hi[i];
  }
  return o;
}

template <int AMODE>
__device__ __forceinline__ bt_bf16x8 bt_load_a(const BGemmParams& p, int row, int k) {
  if constexpr (AMODE == BGA_ROWS_F32) {
    if (row >= p.Mreal) return bt_zero8();
    const float* src = reinterpret_cast<const float*>(p.A) + (size_t)row * p.lda + k;
    bt_f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
    if (k + 4 <= p.Kreal) lo = *reinterpret_cast<const bt_f32x4*>(src);
    if (k + 8 <= p.Kreal) hi = *reinterpret_cast<const bt_f32x4*>(src + 4);
    return bt_pack8(lo, hi);
  } else {
    const int pix = k / p.C, c = k - pix * p.C;
    const bt_bf16* src = reinterpret_cast<const bt_bf16*>(p.A) + ((size_t)pix * p.NBp + row) * p.C + c;
    bt_bf16x8 v = *reinterpret_cast<const bt_bf16x8*>(src);
    if constexpr (AMODE == BGA_STAMP_PRELU) {
      const bt_f32x4 a0 = *reinterpret_cast<const bt_f32x4*>(p.a_alpha + k);
      const bt_f32x4 a1 = *reinterpret_cast<const bt_f32x4*>(p.a_alpha + k + 4);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float x = (float)v[i];
        const float al = i < 4 ? a0[i] : a1[i - 4];
        v[i] = (bt_bf16)(x > 0.f ? x : al * x);
      }
    }
    return v;
  }
}

struct BtFrag {
  bt_bf16x8 a[2][2];   // [row block][k half]
  bt_bf16x8 b[2][2];   // [column block][k half]
};

template <int AMODE, int EPI>
__global__ __launch_bounds__(256) void bgemm_kernel(const BGemmParams p) {
  __shared__ float red[4][1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wk = p.wg_ksplit;
  const int MT = (p.M + 31) >> 5, NT = p.N >> 5;
  const int tile = blockIdx.x * (4 / wk) + wave / wk;
  const int ks = wave % wk;
  const bool valid = tile < MT * NT;
  const int mt = valid ? tile % MT : 0, nt = valid ? tile / MT : 0;
  const int m0 = mt * 32, n0 = nt * 32;
  const int lr = lane & 15, ko = (lane >> 4) * 8;
  // K range of this wave: slice (slab, ks) of nslab * wk equal runs of 64-wide steps
  const int ksteps = p.K >> 6;
  const int nsl = p.nslab * wk;
  const int per = (ksteps + nsl - 1) / nsl;
  const int s0 = (blockIdx.y * wk + ks) * per;
  const int s1 = valid ? min(s0 + per, ksteps) : s0;
  int rowa[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    rowa[i] = m0 + 16 * i + lr;
    if constexpr (AMODE != BGA_ROWS_F32) rowa[i] = min(rowa[i], p.NBp - 1);   // (an odd last 16-row block: duplicate reads)
  }
  const bt_bf16* Bw = reinterpret_cast<const bt_bf16*>(p.B);
  bt_f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = bt_f32x4{0.f, 0.f, 0.f, 0.f};

  auto load = [&](int s, BtFrag& f) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // the two 64-byte halves of a row's 128-byte line, back to back
      const int k = s * 64 + h * 32 + ko;
#pragma unroll
      for (int i = 0; i < 2; ++i) f.a[i][h] = bt_load_a<AMODE>(p, rowa[i], k);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        f.b[j][h] = *reinterpret_cast<const bt_bf16x8*>(Bw + (size_t)(n0 + 16 * j + lr) * p.ldb + k);
    }
  };
  BtFrag cur, nxt;
  if (s0 < s1) load(s0, cur);
  for (int s = s0; s < s1; ++s) {
    if (s + 1 < s1) load(s + 1, nxt);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.a[i][h], cur.b[j][h], acc[i][j], 0, 0, 0);
    cur = nxt;
  }
  // ---- the K slices of a tile: summed through LDS in wave order (ks = 1, 2, 3 onto ks = 0) ----
  if (wk > 1) {
    if (ks > 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) red[wave][((i * 2 + j) * 4 + r) * 64 + lane] = acc[i][j][r];
    }
    __syncthreads();
    if (ks == 0) {
      for (int q = 1; q < wk; ++q)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] += red[wave + q][((i * 2 + j) * 4 + r) * 64 + lane];
    }
  }
  if (!valid || ks != 0) return;

  // ---- epilogue: lane holds rows 4 * (lane >> 4) + r, column lane & 15 of each 16 x 16 block ----
  const int rg = (lane >> 4) * 4;
  if constexpr (EPI == BGE_SLAB) {
    float* out = p.slab + (size_t)blockIdx.y * p.slab_stride;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + 16 * i + rg + r;
        if (row >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 2; ++j) out[(size_t)row * p.ldc + n0 + 16 * j + lr] = acc[i][j][r];
      }
  } else if constexpr (EPI == BGE_STAMP_BIAS_PRELU) {
    bt_bf16* U = reinterpret_cast<bt_bf16*>(p.U);
    bt_bf16* Ao = reinterpret_cast<bt_bf16*>(p.Aout);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 16 * j + lr;
      const int pix = col / p.Co, c = col - pix * p.Co;
      const float bi = p.bias[col], al = p.alpha[col];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + 16 * i + rg + r;
          if (row >= p.NBp) continue;
          float u = acc[i][j][r] + bi;
          float a = u > 0.f ? u : al * u;
          if (row >= p.Mreal) u = a = 0.f;                      // pad stamps: zeros, as everywhere in the stamp-inner tensors
          const size_t off = ((size_t)pix * p.NBp + row) * p.Co + c;
          if (U) U[off] = (bt_bf16)u;
          Ao[off] = (bt_bf16)a;
        }
    }
  } else {   // BGE_STAMP_GATE2
    const bt_bf16* a7 = reinterpret_cast<const bt_bf16*>(p.a7);
    const bt_bf16* u7 = reinterpret_cast<const bt_bf16*>(p.u7);
    bt_bf16* dU = reinterpret_cast<bt_bf16*>(p.dU);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 16 * j + lr;
      const int pix = col / p.Co, c = col - pix * p.Co;
      const float alf = p.alpha_flat[col], al7 = p.alpha7[col];
      float sflat = 0.f, s7 = 0.f, sb = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = m0 + 16 * i + rg + r;
          if (row >= p.NBp) continue;
          const size_t off = ((size_t)pix * p.NBp + row) * p.Co + c;
          const float d = row < p.Mreal ? acc[i][j][r] : 0.f;   // d(flatten PReLU output)
          const float av = (float)a7[off], uv = (float)u7[off];
          const float dA = d * (av > 0.f ? 1.f : alf);          // d(activation of the last encoder conv)
          const float du = dA * (uv > 0.f ? 1.f : al7);
          sflat += d * fminf(av, 0.f);
          s7 += dA * fminf(uv, 0.f);
          sb += du;
          dU[off] = (bt_bf16)du;
        }
      if (p.part_db) {
        // column sums over the tile's 32 rows: the four 16-lane groups hold rows rg .. rg + 3 of both blocks
        sflat += __shfl_xor(sflat, 16, 64); sflat += __shfl_xor(sflat, 32, 64);
        s7 += __shfl_xor(s7, 16, 64);       s7 += __shfl_xor(s7, 32, 64);
        sb += __shfl_xor(sb, 16, 64);       sb += __shfl_xor(sb, 32, 64);
        if (lane < 16) {
          const size_t o = (size_t)mt * p.N + col;
          p.part_dal_flat[o] = sflat;
          p.part_dal7[o] = s7;
          p.part_db[o] = sb;
        }
      }
    }
  }
}

// ---- dense kernel gradients: G[i][j] = sum_m X[m][i] * Y[m][j] ----------------------------------------------------
// operand element (m, e): STAMP / STAMP_PRELU  bf16 [(p * NBp + m) * C + c], e = p * C + c;  ROWS_F32  fp32 [m * ld + e]
template <int MODE>
__device__ __forceinline__ bt_bf16x8 bt_load_t(const void* base, int ld, int NBp, int C, const float* alpha, int Mreal,
                                               int Ereal, int e, int m) {
  bt_bf16x8 v;
  if constexpr (MODE == BGA_ROWS_F32) {
    const float* src = reinterpret_cast<const float*>(base);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float x = (e < Ereal && m + q < Mreal) ? src[(size_t)(m + q) * ld + e] : 0.f;
      v[q] = (bt_bf16)x;
    }
  } else {
    const int pix = e / C, c = e - pix * C;
    const bt_bf16* src = reinterpret_cast<const bt_bf16*>(base) + ((size_t)pix * NBp + m) * C + c;
    float al = 0.f;
    if constexpr (MODE == BGA_STAMP_PRELU) al = alpha[e];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      bt_bf16 x = m + q < NBp ? src[(size_t)q * C] : (bt_bf16)0.f;     // (a last block of 16 stamps: NBp % 32 == 16)
      if constexpr (MODE == BGA_STAMP_PRELU) {
        const float f = (float)x;
        x = (bt_bf16)(f > 0.f ? f : al * f);
      }
      v[q] = x;
    }
  }
  return v;
}

template <int XMODE, int YMODE>
__global__ __launch_bounds__(256) void bgemm_tn_kernel(const BGemmTnParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int IT = p.I >> 5, JT = p.J >> 5;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= IT * JT) return;
  const int jt = tile % JT, it = tile / JT;        // the four waves of a workgroup share their X rows
  const int i0 = it * 32, j0 = jt * 32;
  const int lr = lane & 15, mo = (lane >> 4) * 8;
  bt_f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = bt_f32x4{0.f, 0.f, 0.f, 0.f};
  for (int m = 0; m < p.NBp; m += 32) {
    bt_bf16x8 xa[2], yb[2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
      xa[a] = bt_load_t<XMODE>(p.X, p.ldx, p.NBp, p.Cx, p.x_alpha, p.Mreal, p.Ireal, i0 + 16 * a + lr, m + mo);
#pragma unroll
    for (int b = 0; b < 2; ++b)
      yb[b] = bt_load_t<YMODE>(p.Y, p.ldy, p.NBp, p.Cy, nullptr, p.Mreal, p.Jreal, j0 + 16 * b + lr, m + mo);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[a], yb[b], acc[a][b], 0, 0, 0);
  }
  const int rg = (lane >> 4) * 4;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + 16 * a + rg + r;
      if (i >= p.Ireal) continue;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = j0 + 16 * b + lr;
        if (j < p.Jreal) p.G[(size_t)i * p.ldg + j] = acc[a][b][r];
      }
    }
}

// t[b][i] = bias[i] + sum_s slab[s][b][i]: the finish of the encoder Dense for callers that do not run the sampler behind it
__global__ __launch_bounds__(256) void bt_finish_rows_kernel(const float* __restrict__ slab, int nslab, long slab_stride,
                                                             int lds, const float* __restrict__ bias, float* __restrict__ out,
                                                             int NB, int n, int ldo) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)NB * ldo) return;
  const int b = (int)(e / ldo), i = (int)(e - (long)b * ldo);
  float v = 0.f;
  if (i < n) {
    v = bias ? bias[i] : 0.f;
    for (int s = 0; s < nslab; ++s) v += slab[(size_t)s * slab_stride + (size_t)b * lds + i];
  }
  out[e] = v;
}

}  // namespace

int launch_bgemm(const BGemmParams& p0, hipStream_t s) {
  BGemmParams p = p0;
  if (p.M <= 0 || (p.M & 15) || p.N <= 0 || (p.N & 31) || p.K <= 0 || (p.K & 63) || (p.ldb & 7) || p.ldb < p.K ||
      p.nslab < 1 || (p.wg_ksplit != 1 && p.wg_ksplit != 2 && p.wg_ksplit != 4)) {
    set_error("bgemm: bad geometry (M %d, N %d, K %d, ldb %d, slabs %d, in-workgroup K split %d)", p.M, p.N, p.K, p.ldb,
              p.nslab, p.wg_ksplit);
    return E_INVALID;
  }
  if (p.amode == BGA_ROWS_F32) {
    if ((p.lda & 3) || (p.Kreal & 3) || p.Kreal > p.K || p.Mreal > p.M) {
      set_error("bgemm: fp32 row operand needs a 16-byte row stride (lda %d, K %d of %d)", p.lda, p.Kreal, p.K);
      return E_INVALID;
    }
  } else if ((p.C & 63) || p.C <= 0 || p.K % p.C || p.NBp != p.M) {
    set_error("bgemm: stamp-inner operand needs channel counts that are multiples of 64 (C %d, K %d)", p.C, p.K);
    return E_INVALID;
  }
  if (p.epi != BGE_SLAB && (p.nslab != 1 || p.Co <= 0 || p.N % p.Co || p.NBp != p.M)) {
    set_error("bgemm: stamp-inner epilogues take the whole K in one workgroup (slabs %d, Co %d, N %d)", p.nslab, p.Co, p.N);
    return E_INVALID;
  }
  const int MT = (p.M + 31) / 32, NT = p.N / 32;
  const int tiles = MT * NT, tpw = 4 / p.wg_ksplit;
  const dim3 grid((unsigned)((tiles + tpw - 1) / tpw), (unsigned)p.nslab);
#define DV_BGEMM(am, ep)                                                            \
  if (p.amode == am && p.epi == ep) {                                               \
    hipLaunchKernelGGL((bgemm_kernel<am, ep>), grid, dim3(256), 0, s, p);           \
    DV_HIP(hipGetLastError());                                                      \
    return OK;                                                                      \
  }
  DV_BGEMM(BGA_STAMP_PRELU, BGE_SLAB)
  DV_BGEMM(BGA_STAMP, BGE_SLAB)
  DV_BGEMM(BGA_ROWS_F32, BGE_SLAB)
  DV_BGEMM(BGA_ROWS_F32, BGE_STAMP_BIAS_PRELU)
  DV_BGEMM(BGA_ROWS_F32, BGE_STAMP_GATE2)
#undef DV_BGEMM
  set_error("bgemm: combination of operand mode %d and epilogue %d is not built", p.amode, p.epi);
  return E_INVALID;
}

int launch_bgemm_tn(const BGemmTnParams& p, hipStream_t s) {
  if (p.NBp <= 0 || (p.NBp & 31 && p.NBp & 15) || (p.I & 31) || (p.J & 31) || p.Ireal > p.I || p.Jreal > p.J || p.I <= 0 ||
      p.J <= 0) {
    set_error("bgemm_tn: bad geometry (stamps %d, I %d, J %d)", p.NBp, p.I, p.J);
    return E_INVALID;
  }
  if ((p.xmode != BGA_ROWS_F32 && (p.Cx <= 0 || p.I % p.Cx)) || (p.ymode != BGA_ROWS_F32 && (p.Cy <= 0 || p.J % p.Cy))) {
    set_error("bgemm_tn: stamp-inner operand with a channel count that does not divide its width");
    return E_INVALID;
  }
  const int tiles = (p.I / 32) * (p.J / 32);
  const dim3 grid((unsigned)((tiles + 3) / 4));
#define DV_BTN(xm, ym)                                                              \
  if (p.xmode == xm && p.ymode == ym) {                                             \
    hipLaunchKernelGGL((bgemm_tn_kernel<xm, ym>), grid, dim3(256), 0, s, p);        \
    DV_HIP(hipGetLastError());                                                      \
    return OK;                                                                      \
  }
  DV_BTN(BGA_STAMP_PRELU, BGA_ROWS_F32)
  DV_BTN(BGA_ROWS_F32, BGA_STAMP)
  DV_BTN(BGA_ROWS_F32, BGA_ROWS_F32)
#undef DV_BTN
  set_error("bgemm_tn: combination of operand modes %d / %d is not built", p.xmode, p.ymode);
  return E_INVALID;
}

int launch_bt_finish_rows(const float* slab, int nslab, long slab_stride, int lds, const float* bias, float* out, int NB,
                          int n, int ldo, hipStream_t s) {
  const long total = (long)NB * ldo;
  if (total <= 0) return OK;
  hipLaunchKernelGGL(bt_finish_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slab, nslab, slab_stride,
                     lds, bias, out, NB, n, ldo);
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
