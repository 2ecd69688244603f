// Bandwidth-bound kernels of the bf16 family (see bf16.h): layout changes at the fp32 <-> bf16 seams, the relu / crop /
// Normal-NLL head (model.py:137-159, metrics.py:16-26) over stamp-inner tensors, the unfused PReLU backward, and the
// fp32 -> bf16 weight matrices.
#include "common.h"
#include "bf16.h"

namespace dv {

typedef __bf16 bp_bf16;
typedef __bf16 bp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bp_bf16x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ float bp_block_sum(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  __syncthreads();
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}
}  // namespace

// ---- input: dataset rows -> normalised, stamp-inner, 16 channels ------------------------------------------------
// A workgroup moves 32 pixels x 16 stamps through LDS: the dataset side is read in whole stamp rows (32 pixels x C
// floats, contiguous), the stamp-inner side is written 16 stamps x 32 bytes = 512 contiguous bytes per pixel.
// (One thread per (pixel, stamp) without the tile read 24-byte pieces 83 KB apart: 92 us instead of ~20 for 256 stamps.)
constexpr int BI_PX = 32, BI_ST = 16, BI_MAXC = 15;   // 1 .. 15 bands + the constant-one channel fill the 16 (train.py:86,104-107 takes any)
__global__ __launch_bounds__(256) void bf_input_kernel(const float* __restrict__ x, const int* __restrict__ idx, int first,
                                                       int NB, int NBp, int HW, int C, const float* __restrict__ bn,
                                                       bp_bf16* __restrict__ xh) {
  __shared__ float tile[BI_ST][BI_PX * BI_MAXC + 1];
  const int p0 = blockIdx.x * BI_PX, b0 = blockIdx.y * BI_ST;
  const int npx = min(BI_PX, HW - p0);
  const int rowf = npx * C;                               // floats of one stamp's pixel run
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int sr = wave; sr < BI_ST; sr += 4) {
    const int b = b0 + sr;
    if (b >= NB) continue;
    const long row = idx ? idx[b] : first + b;
    const float* src = x + (row * HW + p0) * C;
    for (int f = lane; f < rowf; f += 64) tile[sr][f] = src[f];
  }
  __syncthreads();
  const int sr = threadIdx.x & 15;
  const int b = b0 + sr;
  for (int pp = threadIdx.x >> 4; pp < npx; pp += 16) {
    bp_bf16x8 lo, hi;
#pragma unroll
    for (int j = 0; j < 8; ++j) lo[j] = hi[j] = (bp_bf16)0.f;
    if (b < NB) {
#pragma unroll
      for (int c = 0; c < 16; ++c) {                     // (static register indexing: the band count is a run-time value)
        bp_bf16 v = (bp_bf16)0.f;
        if (c < C && c < BI_MAXC) v = (bp_bf16)((tile[sr][pp * C + c] - bn[2 * DV_BN_MAXC + c]) * bn[3 * DV_BN_MAXC + c]);   // bnstate rows are DV_BN_MAXC wide
        if (c == C) v = (bp_bf16)1.f;
        if (c < 8) lo[c] = v; else hi[c - 8] = v;
      }
    }
    if (b < NBp) {
      bp_bf16x8* o = reinterpret_cast<bp_bf16x8*>(xh + ((size_t)(p0 + pp) * NBp + b) * 16);
      o[0] = lo;
      o[1] = hi;
    }
  }
}

int launch_bf_input(const float* x, const int* idx, int first, int NB, int NBp, int HW, int C, const float* bnstate,
                    void* xh, hipStream_t s) {
  if (C > BI_MAXC) {
    set_error("bf_input: at most 15 bands");
    return E_INVALID;
  }
  hipLaunchKernelGGL(bf_input_kernel, dim3((unsigned)((HW + BI_PX - 1) / BI_PX), (unsigned)(NBp / BI_ST)), dim3(256), 0, s,
                     x, idx, first, NB, NBp, HW, C, bnstate, reinterpret_cast<bp_bf16*>(xh));
  DV_HIP(hipGetLastError());
  return OK;
}

// ---- stamp-inner bf16 <-> row-major fp32 (the dense trunk's side) -------------------------------------------------
// one thread per 8 channels; consecutive threads walk channels, then stamps (the bf16 side is the coalesced one)
__global__ __launch_bounds__(256) void bf_to_rows_kernel(const bp_bf16* __restrict__ src, float* __restrict__ dst, int NB,
                                                         int NBp, int P, int C) {
  const int c8n = C >> 3;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)P * NB * c8n) return;
  const int c8 = (int)(e % c8n);
  const long r = e / c8n;
  const int b = (int)(r % NB), pix = (int)(r / NB);
  const bp_bf16x8 v = *reinterpret_cast<const bp_bf16x8*>(src + ((size_t)pix * NBp + b) * C + c8 * 8);
  float* o = dst + (size_t)b * P * C + (size_t)pix * C + c8 * 8;
  f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
  f32x4 b4 = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
  reinterpret_cast<f32x4*>(o)[0] = a;
  reinterpret_cast<f32x4*>(o)[1] = b4;
}

__global__ __launch_bounds__(256) void bf_from_rows_kernel(const float* __restrict__ src, bp_bf16* __restrict__ dst, int NB,
                                                           int NBp, int P, int C) {
  const int c8n = C >> 3;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)P * NBp * c8n) return;
  const int c8 = (int)(e % c8n);
  const long r = e / c8n;
  const int b = (int)(r % NBp), pix = (int)(r / NBp);
  bp_bf16x8 v;
  if (b < NB) {
    const float* i = src + (size_t)b * P * C + (size_t)pix * C + c8 * 8;
    const f32x4 a = reinterpret_cast<const f32x4*>(i)[0], b4 = reinterpret_cast<const f32x4*>(i)[1];
    v[0] = (bp_bf16)a[0]; v[1] = (bp_bf16)a[1]; v[2] = (bp_bf16)a[2]; v[3] = (bp_bf16)a[3];
    v[4] = (bp_bf16)b4[0]; v[5] = (bp_bf16)b4[1]; v[6] = (bp_bf16)b4[2]; v[7] = (bp_bf16)b4[3];
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (bp_bf16)0.f;
  }
  *reinterpret_cast<bp_bf16x8*>(dst + ((size_t)pix * NBp + b) * C + c8 * 8) = v;
}

int launch_bf_to_rows(const void* src, float* dst, int NB, int NBp, int P, int C, hipStream_t s) {
  if (C & 7) {
    set_error("bf_to_rows: channels must be a multiple of 8");
    return E_INVALID;
  }
  const long n = (long)P * NB * (C >> 3);
  if (n == 0) return OK;
  hipLaunchKernelGGL(bf_to_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const bp_bf16*>(src), dst, NB, NBp, P, C);
  DV_HIP(hipGetLastError());
  return OK;
}

int launch_bf_from_rows(const float* src, void* dst, int NB, int NBp, int P, int C, hipStream_t s) {
  if (C & 7) {
    set_error("bf_from_rows: channels must be a multiple of 8");
    return E_INVALID;
  }
  const long n = (long)P * NBp * (C >> 3);
  hipLaunchKernelGGL(bf_from_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src,
                     reinterpret_cast<bp_bf16*>(dst), NB, NBp, P, C);
  DV_HIP(hipGetLastError());
  return OK;
}

// ---- unfused PReLU backward: one workgroup per pixel, thread (channel octet, stamp lane) --------------------------
__global__ __launch_bounds__(256) void bf_prelu_bwd_kernel(const bp_bf16* __restrict__ da, const bp_bf16* __restrict__ u,
                                                           const float* __restrict__ alpha, bp_bf16* __restrict__ du,
                                                           float* __restrict__ dalpha, float* __restrict__ db_rows, int NBp,
                                                           int C) {
  extern __shared__ float sh[];                       // [2][lanes][C]
  const int pix = blockIdx.x;
  const int c8n = C >> 3;
  const int nl = 256 / c8n;                            // stamp lanes (threads past nl * c8n idle when c8n does not divide 256)
  const int c8 = threadIdx.x % c8n, sl = threadIdx.x / c8n;
  float al[8], sa[8], sb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    al[j] = alpha[(size_t)pix * C + c8 * 8 + j];
    sa[j] = sb[j] = 0.f;
  }
  // Eight stamps per trip with all sixteen loads issued before the first use: the deep seam tensor (4 x 4 pixels x 256
  // channels) is sixteen workgroups whose threads each walked 32 stamps as a chain of dependent load - store trips (21 us
  // for 2 MB; round 5: same sums in the same order, ~4x fewer exposed latencies).  du may alias da (in place): a trip's
  // stores follow all of its loads, and trips of one thread touch disjoint rows.
  if (sl < nl) {
    constexpr int UNR = 8;
    for (int b0 = sl; b0 < NBp; b0 += nl * UNR) {
      bp_bf16x8 dv[UNR], uv[UNR];
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const int b = b0 + k * nl;
        if (b < NBp) {
          const size_t e = ((size_t)pix * NBp + b) * C + c8 * 8;
          dv[k] = *reinterpret_cast<const bp_bf16x8*>(da + e);
          uv[k] = *reinterpret_cast<const bp_bf16x8*>(u + e);
        }
      }
#pragma unroll
      for (int k = 0; k < UNR; ++k) {
        const int b = b0 + k * nl;
        if (b < NBp) {
          const size_t e = ((size_t)pix * NBp + b) * C + c8 * 8;
          bp_bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float d = (float)dv[k][j], uu = (float)uv[k][j];
            const float g = d * (uu > 0.f ? 1.f : al[j]);
            sa[j] += d * fminf(uu, 0.f);
            sb[j] += g;
            o[j] = (bp_bf16)g;
          }
          *reinterpret_cast<bp_bf16x8*>(du + e) = o;
        }
      }
    }
  }
  if (!dalpha && !db_rows) return;
  float* s0 = sh;
  float* s1 = sh + nl * C;
  if (sl < nl)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      s0[sl * C + c8 * 8 + j] = sa[j];
      s1[sl * C + c8 * 8 + j] = sb[j];
    }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
    for (int l = 0; l < nl; ++l) {
      a += s0[l * C + c];
      b += s1[l * C + c];
    }
    if (dalpha) dalpha[(size_t)pix * C + c] = a;
    if (db_rows) db_rows[(size_t)pix * C + c] = b;
  }
}

int launch_bf_prelu_bwd(const void* da, const void* u, const float* alpha, void* du, float* dalpha, float* db_rows,
                        int NBp, int P, int C, hipStream_t s) {
  const int c8n = C >> 3;
  if ((C & 7) || c8n < 1 || c8n > 256) {
    set_error("bf_prelu_bwd: unsupported channel count %d", C);
    return E_INVALID;
  }
  const int nl = 256 / c8n;
  const size_t lds = (size_t)2 * nl * C * sizeof(float);
  hipLaunchKernelGGL(bf_prelu_bwd_kernel, dim3((unsigned)P), dim3(256), lds, s, reinterpret_cast<const bp_bf16*>(da),
                     reinterpret_cast<const bp_bf16*>(u), alpha, reinterpret_cast<bp_bf16*>(du), dalpha, db_rows, NBp, C);
  DV_HIP(hipGetLastError());
  return OK;
}

// ---- column sums of a bf16 [rows][CW] tensor (CW = 16, or 32 for 8 - 15 bands): d(bias) of the head -----------------
template <int CW>
__global__ __launch_bounds__(256) void bf_colsum16_kernel(const bp_bf16* __restrict__ x, long rows, float* __restrict__ part) {
  __shared__ float sh[256 * CW];
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long r0 = (long)blockIdx.x * per, r1 = min(rows, r0 + per);
  float a[CW];
#pragma unroll
  for (int j = 0; j < CW; ++j) a[j] = 0.f;
  for (long r = r0 + threadIdx.x; r < r1; r += 256) {
#pragma unroll
    for (int q = 0; q < CW / 8; ++q) {
      const bp_bf16x8 v = *reinterpret_cast<const bp_bf16x8*>(x + r * CW + 8 * q);
#pragma unroll
      for (int j = 0; j < 8; ++j) a[8 * q + j] += (float)v[j];
    }
  }
#pragma unroll
  for (int j = 0; j < CW; ++j) sh[threadIdx.x * CW + j] = a[j];
  __syncthreads();
  if (threadIdx.x < CW) {
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += sh[i * CW + threadIdx.x];
    part[blockIdx.x * CW + threadIdx.x] = t;
  }
}

int launch_bf_colsum(const void* x, long rows, int C, float* part, int* nrows_out, hipStream_t s) {
  if (C != 16 && C != 32) {
    set_error("bf_colsum: 16 or 32 channels only");
    return E_INVALID;
  }
  const int nb = (int)std::min<long>(512, std::max<long>(1, rows / 1024));
  if (C == 16) hipLaunchKernelGGL(bf_colsum16_kernel<16>, dim3(nb), dim3(256), 0, s, reinterpret_cast<const bp_bf16*>(x), rows, part);
  else hipLaunchKernelGGL(bf_colsum16_kernel<32>, dim3(nb), dim3(256), 0, s, reinterpret_cast<const bp_bf16*>(x), rows, part);
  if (nrows_out) *nrows_out = nb;
  DV_HIP(hipGetLastError());
  return OK;
}

// ---- head: relu / crop / sigma floor / Normal NLL and its gradient, one thread per (pixel of the 2^L grid, stamp) -----
// Same arithmetic as head_kernel (pointwise.hip): NLL uses (y - mu)/sigma; the gradient w.r.t. the head conv's
// pre-activation is stored in bf16 (it feeds bf16 MFMA operands), zero outside the crop and for the pad stamps.
// A workgroup is 16 consecutive pixels of one row of the 2^L grid x 16 stamps; the labels of the tile (16 stamps x a run of
// 16 pixels x nb floats each, contiguous per stamp) arrive through LDS.
template <int CW>   // columns of the head tensors: 16 (1 .. 7 bands) or 32 (8 .. 15)
__global__ __launch_bounds__(256) void bf_head_kernel(const BHeadParams p) {
  __shared__ float sh[4];
  __shared__ float ytile[16][16 * (CW / 2) + 1];
  const int chunks = (p.Hd * p.Hd) >> 4;
  const int pc = blockIdx.x % chunks, sc = blockIdx.x / chunks;
  const int pix0 = pc * 16, b0 = sc * 16;
  // grids that are a multiple of 16 wide (the reference's 64): a chunk is a run of ONE row and its labels go through
  // LDS; other widths (multiples of 4 / 8 for shallow nets) read their labels per thread
  const bool tiled = (p.Hd & 15) == 0;
  const int lin = pix0 + (threadIdx.x >> 4);
  const int oh = tiled ? pix0 / p.Hd : lin / p.Hd, ow0 = tiled ? pix0 - oh * p.Hd : 0;
  const int h = oh - p.crop0;
  const bool rowin = (unsigned)h < (unsigned)p.H;
  if (p.y && rowin && tiled) {
    // label run of stamp b: pixels w in [w_lo, w_hi) of row h
    const int w_lo = max(ow0 - p.crop0, 0), w_hi = min(ow0 + 16 - p.crop0, p.H);
    const int nf = (w_hi - w_lo) * p.nb;
    const int sr = threadIdx.x >> 4, l16 = threadIdx.x & 15;
    const int b = b0 + sr;
    if (b < p.NB && nf > 0) {
      const long row = p.idx ? p.idx[b] : p.first + b;
      const float* src = p.y + ((row * p.H + h) * p.H + w_lo) * p.nb;
      const int shift = (w_lo - (ow0 - p.crop0)) * p.nb;      // position of w_lo inside the tile row
      for (int f = l16; f < nf; f += 16) ytile[sr][shift + f] = src[f];
    }
  }
  __syncthreads();
  const int sr = threadIdx.x & 15, pp = threadIdx.x >> 4;
  const int b = b0 + sr;
  const int ow = tiled ? ow0 + pp : lin - oh * p.Hd, w = ow - p.crop0;
  const long e = (long)(pix0 + pp) * p.NBp + b;
  float nll = 0.f, se = 0.f;
  {
    const bool in = rowin && (unsigned)w < (unsigned)p.H && b < p.NB;
    float d[CW];
#pragma unroll
    for (int j = 0; j < CW; ++j) d[j] = 0.f;
    if (in) {
      const f32x4* tp = reinterpret_cast<const f32x4*>(p.tpre + e * CW);
      float t[CW];
#pragma unroll
      for (int q = 0; q < CW / 4; ++q) {
        const f32x4 tq = tp[q];
        t[4 * q] = tq[0]; t[4 * q + 1] = tq[1]; t[4 * q + 2] = tq[2]; t[4 * q + 3] = tq[3];
      }
      const long opix = ((long)b * p.H + h) * p.H + w;
#pragma unroll
      for (int c = 0; c < CW / 2; ++c) {
        if (c >= p.nb) break;
        float tl = 0.f, ts = 0.f;
#pragma unroll
        for (int j = 0; j < CW; ++j) {             // static register indexing (runtime index = scratch)
          if (j == c) tl = t[j];
          if (j == p.nb + c) ts = t[j];
        }
        const float loc = fmaxf(tl, 0.f);
        const float sig = p.sigma_floor + fmaxf(ts, 0.f);
        if (p.loc) p.loc[opix * p.nb + c] = loc;
        if (p.scale) p.scale[opix * p.nb + c] = sig;
        if (p.y) {
          const float inv = 1.0f / sig;
          const float yv = tiled ? ytile[sr][pp * p.nb + c]
                                 : p.y[(((p.idx ? (long)p.idx[b] : (long)p.first + b) * p.H + h) * p.H + w) * p.nb + c];
          const float df = yv - loc;
          const float r = df * inv;
          nll += 0.5f * r * r + logf(sig) + 0.91893853320467274178f;
          if (p.mse_sample) {
            const float dsm = df - sig * dv_philox_normal((unsigned)b, (unsigned)((h * p.H + w) * p.nb + c), p.mse_stream, p.mse_seed);
            se += dsm * dsm;
          } else {
            se += df * df;
          }
          const float dl = tl > 0.f ? -(r * inv) * p.gscale : 0.f;
          const float ds = ts > 0.f ? (inv - r * r * inv) * p.gscale : 0.f;
#pragma unroll
          for (int j = 0; j < CW; ++j) {
            if (j == c) d[j] = dl;
            if (j == p.nb + c) d[j] = ds;
          }
        }
      }
    }
    if (p.dt) {
      bp_bf16x8* o = reinterpret_cast<bp_bf16x8*>(reinterpret_cast<bp_bf16*>(p.dt) + e * CW);
#pragma unroll
      for (int q = 0; q < CW / 8; ++q) {
        bp_bf16x8 oq;
#pragma unroll
        for (int j = 0; j < 8; ++j) oq[j] = (bp_bf16)d[8 * q + j];
        o[q] = oq;
      }
    }
  }
  const float a = bp_block_sum(nll, sh);
  const float b2 = bp_block_sum(se, sh);
  if (threadIdx.x == 0 && p.part) {
    p.part[blockIdx.x * 2] = a;
    p.part[blockIdx.x * 2 + 1] = b2;
  }
}

int launch_bf_head(const BHeadParams& p, hipStream_t s, int* nblocks_out) {
  if (p.nb > 15 || p.nb < 1 || (p.cw != 16 && p.cw != 32) || 2 * p.nb > p.cw || ((p.Hd * p.Hd) & 15) || (p.NBp & 15)) {
    set_error("bf_head: 1 .. 15 bands in 16 or 32 columns, pixel count and stamp padding multiples of 16");
    return E_INVALID;
  }
  const int nb = ((p.Hd * p.Hd) >> 4) * (p.NBp >> 4);
  if (nblocks_out) *nblocks_out = nb;
  if (nb == 0) return OK;
  if (p.cw == 16) hipLaunchKernelGGL(bf_head_kernel<16>, dim3(nb), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(bf_head_kernel<32>, dim3(nb), dim3(256), 0, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

// ---- batched partial-slab sums (grid.y = entry) ---------------------------------------------------------------------
// entries without columns: out[e] = sum_p src[p][e] (d(alpha)).  Entries with columns (d(bias)): a block takes a range of
// rows of the [rows][cols] image, sums the slabs and its rows per column and leaves ONE partial row in `out`
// ([gridDim.x][cols]); bf_colsum_final_kernel adds those few rows in block order.  Fixed orders throughout.
constexpr int BR_BLOCKS = 64;
__global__ __launch_bounds__(256) void bf_reduce_batch_kernel(const BRedBatch b) {
  __shared__ float sh[256 * 4];
  const BRedEntry d = b.e[blockIdx.y];
  if (d.cols <= 0) {
    for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < d.n; e += (long)gridDim.x * 1024) {
      f32x4 a = *reinterpret_cast<const f32x4*>(d.src + e);
      for (int p = 1; p < d.nparts; ++p) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(d.src + (size_t)p * d.n + e);
        a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
      }
      *reinterpret_cast<f32x4*>(d.out + e) = a;
    }
    return;
  }
  const int cq = d.cols >> 2;                          // column quads (cols is a multiple of 4)
  const int rows = d.n / d.cols;
  const int rpb = (rows + gridDim.x - 1) / gridDim.x;
  const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
  for (int q0 = 0; q0 < cq; q0 += 256) {               // (cols <= 1024: one pass)
    const int nl = max(1, 256 / min(cq - q0, 256));    // row lanes
    const int q = q0 + threadIdx.x % min(cq - q0, 256), rl = threadIdx.x / min(cq - q0, 256);
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (rl < nl)
      for (int r = r0 + rl; r < r1; r += nl)
        for (int p = 0; p < d.nparts; ++p) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(d.src + (size_t)p * d.n + (size_t)r * d.cols + q * 4);
          a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
        }
    __syncthreads();
    *reinterpret_cast<f32x4*>(sh + threadIdx.x * 4) = a;
    __syncthreads();
    if (rl == 0) {
      for (int l = 1; l < nl; ++l) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(sh + (threadIdx.x + l * min(cq - q0, 256)) * 4);
        a[0] += v[0]; a[1] += v[1]; a[2] += v[2]; a[3] += v[3];
      }
      *reinterpret_cast<f32x4*>(d.out + (size_t)blockIdx.x * d.cols + q * 4) = a;
    }
  }
}
__global__ __launch_bounds__(256) void bf_colsum_final_kernel(const BRedBatch b, int nblocks) {
  const BRedEntry d = b.e[blockIdx.x];
  if (!d.final_out || d.cols <= 0) return;
  for (int c = threadIdx.x; c < d.cols; c += 256) {
    float t = 0.f;
    int r = 0;
    for (; r + 8 <= nblocks; r += 8) {                 // eight loads in flight, added in row order (64 rows: 19 -> ~4 us)
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = d.out[(size_t)(r + k) * d.cols + c];
#pragma unroll
      for (int k = 0; k < 8; ++k) t += v[k];
    }
    for (; r < nblocks; ++r) t += d.out[(size_t)r * d.cols + c];
    d.final_out[c] = t;
  }
}

int launch_bf_reduce_batch(const BRedBatch& b, hipStream_t s) {
  if (b.count <= 0) return OK;
  if (b.count > DV_BF_MAX_RED) {
    set_error("bf_reduce_batch: too many entries");
    return E_INVALID;
  }
  bool anyc = false;
  for (int i = 0; i < b.count; ++i) {
    if ((b.e[i].n & 3) || (b.e[i].cols & 3)) {       // (`out` of a column entry holds max(n, BR_BLOCKS * cols) floats)
      set_error("bf_reduce_batch: slab sizes / column counts must be multiples of 4");
      return E_INVALID;
    }
    anyc = anyc || (b.e[i].cols > 0 && b.e[i].final_out);
  }
  hipLaunchKernelGGL(bf_reduce_batch_kernel, dim3(BR_BLOCKS, (unsigned)b.count), dim3(256), 0, s, b);
  if (anyc) hipLaunchKernelGGL(bf_colsum_final_kernel, dim3((unsigned)b.count), dim3(256), 0, s, b, BR_BLOCKS);
  DV_HIP(hipGetLastError());
  return OK;
}

// ---- fp32 master tensors -> bf16 [N][Kpad] matrices -------------------------------------------------------------
// grid.y = descriptor; one thread per destination element
__global__ __launch_bounds__(256) void bf_cast_kernel(const BCastDesc* __restrict__ descs) {
  const BCastDesc d = descs[blockIdx.y];
  if (!d.src) return;
  bp_bf16* dst = reinterpret_cast<bp_bf16*>(d.dst);
  const long total = (long)d.N * d.Kpad;
  if (d.n_is_b && !d.gamma) {
    // TRANSPOSED kernels (dst[n][tap * Cin + c] = src[tap][c][n]: the forward forms of the Conv2D layers, the data-gradient
    // forms of the transposed ones, the 4096 x 560 matrices of the trunk): 64 x 64 tiles through LDS, tap by tap, so that the
    // fp32 reads run along n and the bf16 writes along c.  The element-per-thread loop below reads such a tensor with a
    // stride of B floats per thread (50 us for the trunk's two alone; a whole decoder bucket: 44 us for 66 MB).
    __shared__ float tile[64][65];
    const int taps = d.taps ? d.taps : 9;
    const int kt = (d.Cin + 63) >> 6, nt = (d.N + 63) >> 6, per = kt * nt;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int t = blockIdx.x; t < taps * per; t += gridDim.x) {
      const int tap = t / per, r0 = t - tap * per;
      const int c0 = (r0 % kt) * 64, n0 = (r0 / kt) * 64;
      const float* src = d.src + (size_t)tap * d.A * d.B;
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + r * 4 + ty, n = n0 + tx;
        tile[r * 4 + ty][tx] = (c < d.A && n < d.B) ? src[(size_t)c * d.B + n] : 0.f;
      }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + r * 4 + ty, c = c0 + tx;
        if (n < d.N && c < d.Cin) dst[(size_t)n * d.Kpad + tap * d.Cin + c] = (bp_bf16)tile[tx][r * 4 + ty];
      }
    }
    const int tail = d.Kpad - taps * d.Cin;              // K padding behind the last tap: zeros
    for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < (long)d.N * tail; e += (long)gridDim.x * 256) {
      const int n = (int)(e / tail), j = (int)(e - (long)n * tail);
      dst[(size_t)n * d.Kpad + taps * d.Cin + j] = (bp_bf16)0.f;
    }
    return;
  }
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int n = (int)(e / d.Kpad), k = (int)(e - (long)n * d.Kpad);
    float v = 0.f;
    const int nreal = d.n_is_b ? d.B : d.A, creal = d.n_is_b ? d.A : d.B;
    if (k < (d.taps ? d.taps : 9) * d.Cin && n < nreal) {
      const int tap = k / d.Cin, c = k - tap * d.Cin;
      if (d.gamma) {
        // first conv: master [9][nbands][B], n = B index; folded input BatchNorm
        if (c < d.nbands) {
          v = d.src[((size_t)tap * d.A + c) * d.B + n] * d.gamma[c];
        } else if (c == d.nbands) {
          for (int cc = 0; cc < d.nbands; ++cc) v += d.src[((size_t)tap * d.A + cc) * d.B + n] * d.beta[cc];
        }
      } else if (c < creal) {
        v = d.n_is_b ? d.src[((size_t)tap * d.A + c) * d.B + n] : d.src[((size_t)tap * d.A + n) * d.B + c];
      }
    }
    dst[e] = (bp_bf16)v;
  }
}

int launch_bf_cast_weights(const BCastDesc* descs_dev, const BCastDesc* descs_host, int n, hipStream_t s) {
  if (n <= 0) return OK;
  long mx = 0;
  for (int i = 0; i < n; ++i) mx = std::max(mx, (long)descs_host[i].N * descs_host[i].Kpad);
  const unsigned gx = (unsigned)std::min<long>(256, std::max<long>(1, (mx + 2047) / 2048));
  hipLaunchKernelGGL(bf_cast_kernel, dim3(gx, (unsigned)n), dim3(256), 0, s, descs_dev);
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
