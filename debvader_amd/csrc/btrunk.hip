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

// A fragment as it arrives from memory (the conversion / PReLU runs when the fragment is used, so that the loads of
// several K steps are in flight together: these products are latency-bound)
template <int AMODE>
struct BtRawA {
  bt_bf16x8 v;
};
template <>
struct BtRawA<BGA_ROWS_F32> {
  bt_f32x4 lo, hi;
};

template <int AMODE>
__device__ __forceinline__ void bt_issue_a(const BGemmParams& p, int row, int k, BtRawA<AMODE>& r) {
  if constexpr (AMODE == BGA_ROWS_F32) {
    r.lo = bt_f32x4{0.f, 0.f, 0.f, 0.f};
    r.hi = bt_f32x4{0.f, 0.f, 0.f, 0.f};
    if (row < p.Mreal) {
      const float* src = reinterpret_cast<const float*>(p.A) + (size_t)row * p.lda + k;
      if (k + 4 <= p.Kreal) r.lo = *reinterpret_cast<const bt_f32x4*>(src);
      if (k + 8 <= p.Kreal) r.hi = *reinterpret_cast<const bt_f32x4*>(src + 4);
    }
  } else {
    const int pix = k / p.C, c = k - pix * p.C;
    r.v = *reinterpret_cast<const bt_bf16x8*>(reinterpret_cast<const bt_bf16*>(p.A) + ((size_t)pix * p.NBp + row) * p.C + c);
  }
}

template <int AMODE>
__device__ __forceinline__ bt_bf16x8 bt_finish_a(const BtRawA<AMODE>& r, bt_f32x4 a0, bt_f32x4 a1) {
  if constexpr (AMODE == BGA_ROWS_F32) {
    return bt_pack8(r.lo, r.hi);
  } else if constexpr (AMODE == BGA_STAMP_PRELU) {
    bt_bf16x8 v = r.v;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float x = (float)v[i];
      const float al = i < 4 ? a0[i] : a1[i - 4];
      v[i] = (bt_bf16)(x > 0.f ? x : al * x);
    }
    return v;
  } else {
    return r.v;
  }
}

template <int AMODE>
struct BtStep {                 // operands of one 64-wide K step of a 32 x 32 wave tile
  BtRawA<AMODE> a[2][2];        // [row block][k half]
  bt_bf16x8 b[2][2];            // [column block][k half]
  bt_f32x4 al[2][2];            // STAMP_PRELU: slopes of the eight k of this lane, per half
};

constexpr int BT_PF = 3;        // K steps in flight per wave (register-direct: 3 x 16 fragment loads of 16 B per lane)

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

  auto issue = [&](int s, BtStep<AMODE>& f) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // the two 64-byte halves of a row's 128-byte line, back to back
      const int k = s * 64 + h * 32 + ko;
#pragma unroll
      for (int i = 0; i < 2; ++i) bt_issue_a<AMODE>(p, rowa[i], k, f.a[i][h]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        f.b[j][h] = *reinterpret_cast<const bt_bf16x8*>(Bw + (size_t)(n0 + 16 * j + lr) * p.ldb + k);
      if constexpr (AMODE == BGA_STAMP_PRELU) {
        f.al[h][0] = *reinterpret_cast<const bt_f32x4*>(p.a_alpha + k);
        f.al[h][1] = *reinterpret_cast<const bt_f32x4*>(p.a_alpha + k + 4);
      }
    }
  };
  auto mfma = [&](const BtStep<AMODE>& f) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      bt_bf16x8 a[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = bt_finish_a<AMODE>(f.a[i][h], f.al[h][0], f.al[h][1]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], f.b[j][h], acc[i][j], 0, 0, 0);
    }
  };
  BtStep<AMODE> f[BT_PF];
#pragma unroll
  for (int q = 0; q < BT_PF; ++q)
    if (s0 + q < s1) issue(s0 + q, f[q]);
  for (int s = s0; s < s1; s += BT_PF) {
#pragma unroll
    for (int q = 0; q < BT_PF; ++q) {
      if (s + q < s1) {
        mfma(f[q]);
        if (s + q + BT_PF < s1) issue(s + q + BT_PF, f[q]);
      }
    }
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
// The contraction index is the stamp, and both operands are stored stamp-major: a workgroup brings [128 stamps][64 e]
// tiles of X and Y into LDS with full 128-byte rows (four 16-byte pieces per thread and operand, all issued before the
// first is used; PReLU / fp32 -> bf16 on the way), and the MFMA fragments - eight consecutive stamps of one column - are
// read back TRANSPOSED with ds_read_b64_tr_b16 (cdna_hip_programming.md T10).  64 x 64 outputs per workgroup, 32 x 32 per wave.
constexpr int TN_CH = 128;      // stamps per LDS chunk
constexpr int TN_PITCH = 72;    // bf16 elements per LDS row (64 + 8: 144-byte rows keep the transposed reads off one bank group)

typedef unsigned bt_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bt_u32x2 bt_tr_read(unsigned lds_addr) {
  bt_u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(lds_addr));
  return v;
}
__device__ __forceinline__ bt_bf16x8 bt_tr_join(bt_u32x2 lo, bt_u32x2 hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 q = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bt_bf16x8, q);
}

// one 16-byte piece (eight columns e0 .. e0 + 7 of stamp m) of an operand tile, as it arrives
template <int MODE>
struct TnRaw {
  bt_bf16x8 v;
};
template <>
struct TnRaw<BGA_ROWS_F32> {
  bt_f32x4 lo, hi;
};
template <int MODE>
__device__ __forceinline__ void tn_issue(const void* base, int ld, int NBp, int C, int Mreal, int Ereal, int m, int e0,
                                         TnRaw<MODE>& r) {
  if constexpr (MODE == BGA_ROWS_F32) {
    r.lo = bt_f32x4{0.f, 0.f, 0.f, 0.f};
    r.hi = bt_f32x4{0.f, 0.f, 0.f, 0.f};
    if (m < Mreal) {
      const float* src = reinterpret_cast<const float*>(base) + (size_t)m * ld + e0;
      if (e0 + 8 <= Ereal && !(ld & 3)) {
        r.lo = *reinterpret_cast<const bt_f32x4*>(src);
        r.hi = *reinterpret_cast<const bt_f32x4*>(src + 4);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (e0 + q < Ereal) r.lo[q] = src[q];
          if (e0 + 4 + q < Ereal) r.hi[q] = src[4 + q];
        }
      }
    }
  } else {
    r.v = bt_zero8();
    if (m < NBp && e0 < Ereal) {
      const int pix = e0 / C, c = e0 - pix * C;
      r.v = *reinterpret_cast<const bt_bf16x8*>(reinterpret_cast<const bt_bf16*>(base) + ((size_t)pix * NBp + m) * C + c);
    }
  }
}
template <int MODE>
__device__ __forceinline__ bt_bf16x8 tn_finish(const TnRaw<MODE>& r, const float* alpha, int e0) {
  if constexpr (MODE == BGA_ROWS_F32) {
    return bt_pack8(r.lo, r.hi);
  } else if constexpr (MODE == BGA_STAMP_PRELU) {
    bt_bf16x8 v = r.v;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float x = (float)v[q];
      v[q] = (bt_bf16)(x > 0.f ? x : alpha[e0 + q] * x);
    }
    return v;
  } else {
    return r.v;
  }
}

template <int XMODE, int YMODE>
__global__ __launch_bounds__(256) void bgemm_tn_kernel(const BGemmTnParams p) {
  __shared__ __attribute__((aligned(16))) bt_bf16 xs[TN_CH * TN_PITCH];
  __shared__ __attribute__((aligned(16))) bt_bf16 ys[TN_CH * TN_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int JT = (p.J + 63) >> 6;
  const int jt = blockIdx.x % JT, it = blockIdx.x / JT;
  const int i0 = it * 64, j0 = jt * 64;
  const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
  bt_f32x4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = bt_f32x4{0.f, 0.f, 0.f, 0.f};
  // transposed-read address of this lane inside a [32 stamps][16 columns] block: group g = lane >> 4 takes stamps
  // 8g .. 8g + 7 (two reads of four rows), lane 4q + r of the group supplies row q, columns 4r .. 4r + 3
  const int g = lane >> 4, q4 = (lane & 15) >> 2, r4 = lane & 3;
  typedef __attribute__((address_space(3))) void* lds_ptr_t;
  const unsigned xs_base = (unsigned)(size_t)(lds_ptr_t)xs, ys_base = (unsigned)(size_t)(lds_ptr_t)ys;   // LDS byte addresses
  const unsigned lane_off = (unsigned)(((8 * g + q4) * TN_PITCH + 4 * r4) * 2);
  for (int mc = 0; mc < p.NBp; mc += TN_CH) {
    // ---- tiles of this chunk into LDS: piece id = thread + 256 * r -> (stamp row, 8-column segment) ----
    TnRaw<XMODE> rx[4];
    TnRaw<YMODE> ry[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int id = threadIdx.x + 256 * r, row = id >> 3, seg = id & 7;
      tn_issue<XMODE>(p.X, p.ldx, p.NBp, p.Cx, p.Mreal, p.Ireal, mc + row, i0 + seg * 8, rx[r]);
      tn_issue<YMODE>(p.Y, p.ldy, p.NBp, p.Cy, p.Mreal, p.Jreal, mc + row, j0 + seg * 8, ry[r]);
    }
    if (mc) __syncthreads();                       // the previous chunk's fragments have been read
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int id = threadIdx.x + 256 * r, row = id >> 3, seg = id & 7;
      *reinterpret_cast<bt_bf16x8*>(xs + row * TN_PITCH + seg * 8) = tn_finish<XMODE>(rx[r], p.x_alpha, i0 + seg * 8);
      *reinterpret_cast<bt_bf16x8*>(ys + row * TN_PITCH + seg * 8) = tn_finish<YMODE>(ry[r], nullptr, j0 + seg * 8);
    }
    __syncthreads();
    const int nst = (min(TN_CH, p.NBp - mc) + 31) >> 5;   // (NBp % 32 == 16: the last half step reads zero rows, see tn_issue)
    for (int st = 0; st < nst; ++st) {
      const unsigned so = (unsigned)(st * 32 * TN_PITCH * 2) + lane_off;
      bt_u32x2 xr[2][2], yr[2][2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const unsigned ad = xs_base + so + (unsigned)((wi + 16 * a) * 2);
        xr[a][0] = bt_tr_read(ad);
        xr[a][1] = bt_tr_read(ad + 4 * TN_PITCH * 2);
      }
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const unsigned ad = ys_base + so + (unsigned)((wj + 16 * b) * 2);
        yr[b][0] = bt_tr_read(ad);
        yr[b][1] = bt_tr_read(ad + 4 * TN_PITCH * 2);
      }
      // "the reads have landed": the wait carries the registers, so that nothing that uses them is scheduled in front of it
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(xr[0][0]), "+v"(xr[0][1]), "+v"(xr[1][0]), "+v"(xr[1][1]), "+v"(yr[0][0]), "+v"(yr[0][1]),
                     "+v"(yr[1][0]), "+v"(yr[1][1]));
      bt_bf16x8 xa[2], yb[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) xa[a] = bt_tr_join(xr[a][0], xr[a][1]);
#pragma unroll
      for (int b = 0; b < 2; ++b) yb[b] = bt_tr_join(yr[b][0], yr[b][1]);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[a], yb[b], acc[a][b], 0, 0, 0);
    }
  }
  const int rg = (lane >> 4) * 4, lr = lane & 15;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = i0 + wi + 16 * a + rg + r;
      if (i >= p.Ireal) continue;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int j = j0 + wj + 16 * b + lr;
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

__device__ __forceinline__ float bt_softplus(float x) { return x > 20.f ? x : log1pf(expf(x)); }
// position of L[i][j] (j <= i) in the params_size vector behind the d means: tfp.math.fill_triangular (model.py:49-52)
__device__ __forceinline__ int bt_tril_src(int d, int i, int j) {
  const int m = d * (d + 1) / 2;
  const int q = i * d + j;
  return q < m - d ? d + q : d * d - 1 - q;
}

template <int KC>   // KC = ceil(hid / 64) values of the hidden row per lane
__global__ __launch_bounds__(256) void bt_mid_bwd_kernel(const BMidBwdParams p) {
  __shared__ float dzs[64];
  __shared__ float st[64 + 64 * 65 / 2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const int hid = p.hid, d = p.d;
  // ---- d(hidden pre-activation): every wave holds the row (lane + 64 t), wave 0 writes it ----
  float dv[KC];
  {
    const float* sb = p.slab + (size_t)b * p.lds;
    const size_t ss = (size_t)p.slab_stride;
    const int ns = p.nslab;
#pragma unroll
    for (int t = 0; t < KC; ++t) {
      const int i = lane + 64 * t;
      dv[t] = 0.f;
      if (i < hid) {
        const float v0 = sb[i], v1 = ns > 1 ? sb[ss + i] : 0.f, v2 = ns > 2 ? sb[2 * ss + i] : 0.f, v3 = ns > 3 ? sb[3 * ss + i] : 0.f;
        float dsum = ((v0 + v1) + v2) + v3;
        for (int sl = 4; sl < ns; ++sl) dsum += sb[(size_t)sl * ss + i];
        const float u = p.uh[(size_t)b * hid + i];
        const bool pos = u > 0.f;
        dv[t] = pos ? dsum : dsum * p.alpha_h[i];
        if (w == 0) {
          p.duh[(size_t)b * hid + i] = dv[t];
          if (p.dalh) p.dalh[(size_t)b * hid + i] = pos ? 0.f : dsum * u;
        }
      }
    }
  }
  // ---- d(z')[n] = sum_i duh[i] * W0[n][i]: wave w owns n = w, w + 4, ...; four outputs per trip (dense_narrow_kernel) ----
  for (int n0 = w; n0 < d; n0 += 16) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 4 * j;
      const float* wr = p.W0 + (size_t)(n < d ? n : 0) * hid;
#pragma unroll
      for (int t = 0; t < KC; ++t) {
        const int k = lane + 64 * t;
        acc[j] = fmaf(dv[t], k < hid ? wr[k] : 0.f, acc[j]);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], o);
    }
    if (lane < 4 && n0 + 4 * lane < d) dzs[n0 + 4 * lane] = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
  }
  __syncthreads();
  if (w != 0) return;
  // ---- wave 0: PReLU gate of z, then the sampler backward (sampler_bwd_kernel, pointwise.hip) ----
  const int tw = d + d * (d + 1) / 2;
  const float* tb = p.t + (size_t)b * p.ldt;
  float* dtb = p.dt + (size_t)b * p.ldt;
  const size_t zo = (size_t)b * p.ldz;
  float e = 0.f, g = 0.f, raw = 0.f;
  if (lane < p.ldz) {
    float dzv = 0.f, dal = 0.f;
    if (lane < d) {
      const float zv = p.z[zo + lane], dzp = dzs[lane];
      const bool pos = zv > 0.f;
      dzv = pos ? dzp : dzp * p.alpha_in[lane];
      dal = pos ? 0.f : dzp * zv;
      e = p.eps[zo + lane];
      g = dzv + p.kls * zv;
      raw = tb[d + bt_tril_src(d, lane, lane)];
      st[lane] = g;
    }
    p.dz[zo + lane] = dzv;
    if (p.dalin) p.dalin[zo + lane] = dal;
  }
  float ldiag = 1.f, sg = 0.f;
  if (lane < d) {
    ldiag = bt_softplus(raw) + p.diag_shift;
    sg = 1.0f / (1.0f + expf(-raw));
  }
  for (int j = 0; j < d; ++j) {
    const float ej = __shfl(e, j, 64);
    if (lane < d && j <= lane) {
      const int src = d + bt_tril_src(d, lane, j);
      float v = g * ej;
      if (j == lane) v = (v - p.kls / ldiag) * sg;
      st[src] = v;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < p.ldt; i += 64) dtb[i] = i < tw ? st[i] : 0.f;
}

__global__ __launch_bounds__(256) void bt_colsums_kernel(const BColsums c) {
  const int q = blockIdx.y;
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (q >= c.count || col >= c.n[q]) return;
  const float* x = c.x[q] + col;
  const size_t ld = (size_t)c.ld[q];
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;      // four accumulators: loads in flight together, fixed order
  int b = 0;
  for (; b + 4 <= c.NB; b += 4) {
    const float v0 = x[(size_t)b * ld], v1 = x[(size_t)(b + 1) * ld], v2 = x[(size_t)(b + 2) * ld], v3 = x[(size_t)(b + 3) * ld];
    s0 += v0; s1 += v1; s2 += v2; s3 += v3;
  }
  for (; b < c.NB; ++b) s0 += x[(size_t)b * ld];
  c.out[q][col] = (float)((s0 + s1) + (s2 + s3));
}

}  // namespace

int launch_bt_mid_bwd(const BMidBwdParams& p, hipStream_t s) {
  if (p.NB <= 0) return OK;
  if (p.d < 1 || p.d > 64 || p.ldz < p.d || p.ldz > 64 || p.hid < 1 || p.hid > 1024 || p.nslab < 1 ||
      p.ldt < p.d + p.d * (p.d + 1) / 2) {
    set_error("trunk backward: latent_dim must be in [1,64] and the hidden width <= 1024 (%d, %d)", p.d, p.hid);
    return E_INVALID;
  }
  if (p.hid <= 256) hipLaunchKernelGGL(bt_mid_bwd_kernel<4>, dim3((unsigned)p.NB), dim3(256), 0, s, p);
  else if (p.hid <= 576) hipLaunchKernelGGL(bt_mid_bwd_kernel<9>, dim3((unsigned)p.NB), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(bt_mid_bwd_kernel<16>, dim3((unsigned)p.NB), dim3(256), 0, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

int launch_bt_colsums(const BColsums& c, hipStream_t s) {
  if (c.count <= 0 || c.NB <= 0) return OK;
  int mx = 0;
  for (int q = 0; q < c.count; ++q) mx = std::max(mx, c.n[q]);
  hipLaunchKernelGGL(bt_colsums_kernel, dim3((unsigned)((mx + 255) / 256), (unsigned)c.count), dim3(256), 0, s, c);
  DV_HIP(hipGetLastError());
  return OK;
}

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
  const dim3 grid((unsigned)(((p.I + 63) / 64) * ((p.J + 63) / 64)));
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
