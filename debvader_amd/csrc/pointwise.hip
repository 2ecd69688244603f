// Bandwidth-bound kernels of the conv-VAE step: input BatchNorm (reference model.py:79), per-element PReLU
// (model.py:84,92,95,113,115,118,128,135), the MultivariateNormalTriL sampler with its Monte-Carlo KL term
// (model.py:43-58,207-214), the relu + crop + Normal NLL head (model.py:137-159, metrics.py:16-26) and the
// legacy-Adam update (train.py:125-130).  All are written for 64-lane wavefronts: reductions go through
// __shfl_down across the full wave, then LDS across the 4 waves of a block, then a per-block partial that a
// final double-precision pass sums in a fixed order (bit-reproducible, no float atomics).
#include <algorithm>

#include "common.h"

namespace dv {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// sums `v` over the 256-thread block; result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh /*>=4 floats*/) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}

// ------------------------------------------------------------------------------------------------
// generic column reduction of a small partial matrix in double precision: out[c] = scale * sum_r part[r][c]
__global__ __launch_bounds__(256) void reduce_rows_f64_kernel(const float* __restrict__ part, int nrows, int ld,
                                                              float* __restrict__ out, float scale) {
  __shared__ double sh[256];
  const int c = blockIdx.x;
  double acc = 0.0;
  for (int r = threadIdx.x; r < nrows; r += 256) acc += (double)part[(long)r * ld + c];
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[c] = (float)(sh[0] * (double)scale);
}

int launch_reduce_rows_f64(const float* part, int nrows, int ncols, float* out, float scale, hipStream_t s, int ld) {
  if (ncols <= 0) return OK;
  if (ld <= 0) ld = ncols;
  hipLaunchKernelGGL(reduce_rows_f64_kernel, dim3(ncols), dim3(256), 0, s, part, nrows, ld, out, scale);
  DV_HIP(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// BatchNorm over the band axis
constexpr int BN_MAXC = DV_BN_MAXC;   // 16
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int BN_PIX_PER_BLOCK = 1024;

// Per-band sums of x and x^2 over a block's 1024 pixels.  Four pixels per thread, all of their loads issued before the
// first add (six bands: three 8-byte loads per pixel - a pixel is 24 bytes); one LDS stage for the 16 block sums.
// (Round 3: 4096 pixels per block were 218 blocks for 256 CUs, sixteen dependent trips of six scalar loads each and 32
// block-wide barriers at the end: 35 us for 21 MB; this form: see DESIGN 4.)
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                       int first, int NB, int HW, int C, float* __restrict__ part) {
  __shared__ float sh[4][2 * BN_MAXC];
  const int total = NB * HW;
  const int p0 = blockIdx.x * BN_PIX_PER_BLOCK + threadIdx.x;
  float s[BN_MAXC], ss[BN_MAXC];
#pragma unroll
  for (int c = 0; c < BN_MAXC; ++c) s[c] = ss[c] = 0.f;
  const float* px[4];
  bool ok[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pp = p0 + 256 * k;
    ok[k] = pp < total;
    const int b = ok[k] ? pp / HW : 0;
    const int pix = ok[k] ? pp - b * HW : 0;
    const long row = idx ? idx[b] : first + b;
    px[k] = x + (row * HW + pix) * C;
  }
  if (C == 6) {
    f32x2 v[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) v[k][j] = ok[k] ? reinterpret_cast<const f32x2*>(px[k])[j] : (f32x2){0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        s[2 * j] += v[k][j][0];
        ss[2 * j] += v[k][j][0] * v[k][j][0];
        s[2 * j + 1] += v[k][j][1];
        ss[2 * j + 1] += v[k][j][1] * v[k][j][1];
      }
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int c = 0; c < BN_MAXC; ++c)
        if (c < C && ok[k]) {
          const float v = px[k][c];
          s[c] += v;
          ss[c] += v * v;
        }
  }
#pragma unroll
  for (int c = 0; c < BN_MAXC; ++c)
    if (c < C) {                                       // (C is uniform: bands beyond it hold zeros and need no shuffle)
      s[c] = wave_sum(s[c]);
      ss[c] = wave_sum(ss[c]);
    }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < BN_MAXC; ++c) {
      sh[threadIdx.x >> 6][c] = s[c];
      sh[threadIdx.x >> 6][BN_MAXC + c] = ss[c];
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * BN_MAXC)
    part[blockIdx.x * 2 * BN_MAXC + threadIdx.x] =
        (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

int launch_bn_stats(const float* x, const int* idx, int first, int NB, int HW, int C, float* part, int* nblocks,
                    hipStream_t s) {
  if (C > BN_MAXC) {
    set_error("bn: at most %d bands supported", BN_MAXC);
    return E_INVALID;
  }
  long total = (long)NB * HW;
  if (total >= (1L << 31) - BN_PIX_PER_BLOCK) {
    set_error("bn: batch too large (pixel count must fit 31 bits)");
    return E_INVALID;
  }
  int nb = (int)((total + BN_PIX_PER_BLOCK - 1) / BN_PIX_PER_BLOCK);
  *nblocks = nb;
  hipLaunchKernelGGL(bn_stats_kernel, dim3(nb), dim3(256), 0, s, x, idx, first, NB, HW, C, part);
  DV_HIP(hipGetLastError());
  return OK;
}

// sums: [0..8) sum x, [8..16) sum x^2 (already reduced over blocks and ranks); count = global N*H*W
__global__ void bn_finalize_kernel(const float* __restrict__ sums, float count, int C, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ mmean,
                                   float* __restrict__ mvar, float eps, float momentum, int unbiased, int training,
                                   int update_moving, float* __restrict__ st) {
  int c = threadIdx.x;
  if (c >= BN_MAXC) return;
  float mean = 0.f, var = 1.f, scale = 0.f, shift = 0.f, inv = 0.f;
  if (c < C) {
    if (training) {
      double m = (double)sums[c] / (double)count;
      double v = (double)sums[BN_MAXC + c] / (double)count - m * m;
      if (v < 0) v = 0;
      mean = (float)m;
      var = (float)v;
      if (update_moving) {
        float vu = unbiased ? (float)(v * ((double)count / ((double)count - 1.0))) : var;
        mmean[c] = mmean[c] * momentum + mean * (1.f - momentum);
        mvar[c] = mvar[c] * momentum + vu * (1.f - momentum);
      }
    } else {
      mean = mmean[c];
      var = mvar[c];
    }
    inv = 1.0f / sqrtf(var + eps);
    scale = gamma[c] * inv;
    shift = beta[c] - mean * scale;
  }
  st[c] = scale;
  st[BN_MAXC + c] = shift;
  st[2 * BN_MAXC + c] = mean;
  st[3 * BN_MAXC + c] = inv;
}

int launch_bn_finalize(const float* sums, float count, int C, const float* gamma, const float* beta,
                       float* moving_mean, float* moving_var, float eps, float momentum, int unbiased, int training,
                       int update_moving, float* bnstate, hipStream_t s) {
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(64), 0, s, sums, count, C, gamma, beta, moving_mean,
                     moving_var, eps, momentum, unbiased, training, update_moving, bnstate);
  DV_HIP(hipGetLastError());
  return OK;
}

template <int CP>
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const int* __restrict__ idx,
                                                       int first, int NB, int HW, int C, const float* __restrict__ st,
                                                       float* __restrict__ xn) {
  long pp = (long)blockIdx.x * 256 + threadIdx.x;
  if (pp >= (long)NB * HW) return;
  int b = (int)(pp / HW);
  int pix = (int)(pp - (long)b * HW);
  long row = idx ? idx[b] : first + b;
  const float* px = x + (row * HW + pix) * C;
  // channels [0,C): xhat = (x - mean) * inv_std; channel C: 1 (carries beta through the folded first conv, see
  // fold_bn_w1_kernel); remaining pad channels: 0.  gamma/beta live in the folded weights, not here.
  float o[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c)
    o[c] = (c < C) ? (px[c] - st[2 * BN_MAXC + c]) * st[3 * BN_MAXC + c] : (c == C ? 1.f : 0.f);
  f32x4* dst = reinterpret_cast<f32x4*>(xn + pp * CP);
#pragma unroll
  for (int q = 0; q < CP / 4; ++q) dst[q] = (f32x4){o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
}

int launch_bn_apply(const float* x, const int* idx, int first, int NB, int HW, int C, int Cpad, const float* bnstate,
                    float* xn, hipStream_t s) {
  if ((Cpad != 8 && Cpad != 16) || C >= Cpad) {
    set_error("bn_apply: needs Cpad 8 or 16 and fewer bands than that (one pad channel carries the BN shift)");
    return E_INVALID;
  }
  long total = (long)NB * HW;
  if (Cpad == 8)
    hipLaunchKernelGGL(bn_apply_kernel<8>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, idx, first, NB, HW, C,
                       bnstate, xn);
  else
    hipLaunchKernelGGL(bn_apply_kernel<16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, idx, first, NB, HW, C,
                       bnstate, xn);
  DV_HIP(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// PReLU
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ u, const float* __restrict__ alpha,
                                                        float* __restrict__ a, long total4, int E4) {
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  int e = (int)(i % E4);
  f32x4 uv = reinterpret_cast<const f32x4*>(u)[i];
  f32x4 al = reinterpret_cast<const f32x4*>(alpha)[e];
  f32x4 o;
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = uv[k] > 0.f ? uv[k] : al[k] * uv[k];
  reinterpret_cast<f32x4*>(a)[i] = o;
}

int launch_prelu_fwd(const float* u, const float* alpha, float* a, long NB, int E, hipStream_t s) {
  if (E & 3) {
    set_error("prelu: E must be a multiple of 4");
    return E_INVALID;
  }
  long total4 = NB * (E / 4);
  if (total4 == 0) return OK;
  hipLaunchKernelGGL(prelu_fwd_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, u, alpha, a, total4,
                     E / 4);
  DV_HIP(hipGetLastError());
  return OK;
}

// Narrow dense contraction out[b][n] = sum_k x[b][k] * W[n][k] for N <= 64 (the data gradient of the decoder's first
// Dense layer, model.py:113-114: 560 -> latent_dim): one wave per row, lanes stride K, one butterfly per output.  As a
// 128 x 32-tile gather-GEMM this was two workgroups walking K serially (27 us alone, 49 us beside the weight-gradient
// stream); here NB waves read a 72 KB weight matrix from L2.
template <int KC>   // KC = ceil(K / 64) row values per lane, held in registers
__global__ __launch_bounds__(256) void dense_narrow_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                           float* __restrict__ out, int NB, int K, int N) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x;                      // one row per workgroup, wave w owns outputs n = w, w + 4, ...
  const float* xr = x + (size_t)b * K;
  float xv[KC];
#pragma unroll
  for (int t = 0; t < KC; ++t) xv[t] = lane + 64 * t < K ? xr[lane + 64 * t] : 0.f;
  for (int n0 = w; n0 < N; n0 += 16) {           // four outputs per trip: their loads are issued together
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 4 * j;
      const float* wr = W + (size_t)(n < N ? n : 0) * K;
#pragma unroll
      for (int t = 0; t < KC; ++t) {
        const int k = lane + 64 * t;
        acc[j] = fmaf(xv[t], k < K ? wr[k] : 0.f, acc[j]);
      }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] += __shfl_xor(acc[j], o);
    }
    if (lane < 4 && n0 + 4 * lane < N)
      out[(size_t)b * N + n0 + 4 * lane] = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
  }
}

int launch_dense_narrow(const float* x, const float* W, float* out, int NB, int K, int N, hipStream_t s) {
  if (N > 64 || N < 1 || K < 1 || K > 1024) {
    set_error("dense_narrow: N must be in 1..64 and K in 1..1024");
    return E_INVALID;
  }
  if (NB == 0) return OK;
  if (K <= 256)
    hipLaunchKernelGGL(dense_narrow_kernel<4>, dim3((unsigned)NB), dim3(256), 0, s, x, W, out, NB, K, N);
  else if (K <= 576)
    hipLaunchKernelGGL(dense_narrow_kernel<9>, dim3((unsigned)NB), dim3(256), 0, s, x, W, out, NB, K, N);
  else
    hipLaunchKernelGGL(dense_narrow_kernel<16>, dim3((unsigned)NB), dim3(256), 0, s, x, W, out, NB, K, N);
  DV_HIP(hipGetLastError());
  return OK;
}

// du = da * (u>0 ? 1 : alpha) in place; d(alpha)[e] = sum_n da*min(u,0); d(bias)[c] = sum_{n,hw} du.
// grid (ceil(E/1024), nsplit): thread owns 4 consecutive elements e, loops over its slice of the batch.
// dbias_mode 0: none; 1: E == C (dense), partial [nsplit][E]; 2: C % 4 == 0 and C <= 1024, partial [nsplit*gridDim.x][C]
__global__ __launch_bounds__(256) void prelu_bwd_kernel(float* __restrict__ da, const float* __restrict__ u,
                                                        const float* __restrict__ alpha, int NB, int E, int C,
                                                        int nper, float* __restrict__ dalpha_part,
                                                        float* __restrict__ dbias_part, int dbias_mode) {
  __shared__ f32x4 shv[256];
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int split = blockIdx.y;
  const bool act = e < E;
  f32x4 dal = {0.f, 0.f, 0.f, 0.f}, dbs = {0.f, 0.f, 0.f, 0.f};
  if (act) {
    const f32x4 al = *reinterpret_cast<const f32x4*>(alpha + e);
    const int nbeg = split * nper, nend = min(NB, nbeg + nper);
    auto one = [&](const f32x4& g, const f32x4& uv, size_t off) {
      f32x4 d;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        bool pos = uv[k] > 0.f;
        d[k] = pos ? g[k] : g[k] * al[k];
        dal[k] += pos ? 0.f : g[k] * uv[k];
        dbs[k] += d[k];
      }
      *reinterpret_cast<f32x4*>(da + off) = d;
    };
    // four stamps per trip, all eight loads issued before the first in-place store (the compiler will not move a load of
    // `da` above a store to `da`): same sums in the same order
    int n = nbeg;
    for (; n + 4 <= nend; n += 4) {
      const size_t o0 = (size_t)n * E + e, o1 = o0 + E, o2 = o1 + E, o3 = o2 + E;
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(da + o0), u0 = *reinterpret_cast<const f32x4*>(u + o0);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(da + o1), u1 = *reinterpret_cast<const f32x4*>(u + o1);
      const f32x4 g2 = *reinterpret_cast<const f32x4*>(da + o2), u2 = *reinterpret_cast<const f32x4*>(u + o2);
      const f32x4 g3 = *reinterpret_cast<const f32x4*>(da + o3), u3 = *reinterpret_cast<const f32x4*>(u + o3);
      one(g0, u0, o0);
      one(g1, u1, o1);
      one(g2, u2, o2);
      one(g3, u3, o3);
    }
    for (; n < nend; ++n) {
      const size_t off = (size_t)n * E + e;
      const f32x4 g = *reinterpret_cast<const f32x4*>(da + off), uv = *reinterpret_cast<const f32x4*>(u + off);
      one(g, uv, off);
    }
    if (dalpha_part) *reinterpret_cast<f32x4*>(dalpha_part + (size_t)split * E + e) = dal;
    if (dbias_mode == 1) *reinterpret_cast<f32x4*>(dbias_part + (size_t)split * E + e) = dbs;
  }
  if (dbias_mode == 2) {
    shv[threadIdx.x] = dbs;
    __syncthreads();
    const int cq = C / 4;
    if ((int)threadIdx.x < cq) {
      // thread t of the block holds channel quad (blockIdx.x * 256 + t) % cq (the block starts anywhere in a row when C
      // does not divide 1024): the threads of quad threadIdx.x, in order
      const int t0 = (int)(((long)threadIdx.x - (long)blockIdx.x * 256 % cq + cq) % cq);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int t = t0; t < 256; t += cq) acc += shv[t];
      size_t row = (size_t)split * gridDim.x + blockIdx.x;
      *reinterpret_cast<f32x4*>(dbias_part + row * C + threadIdx.x * 4) = acc;
    }
  }
}

int launch_prelu_bwd(float* da, const float* u, const float* alpha, int NB, int E, int C, int nsplit,
                     float* dalpha_part, float* dbias_part, int* dbias_rows, hipStream_t s) {
  if ((E & 3) || nsplit < 1) return E_INVALID;
  int mode = 0;
  int gx = (E + 1023) / 1024;
  if (dbias_part) {
    if (E == C) {
      mode = 1;
      *dbias_rows = nsplit;
    } else if (C <= 1024 && (C & 3) == 0 && (E % C) == 0) {
      mode = 2;
      *dbias_rows = nsplit * gx;
    } else {
      set_error("prelu_bwd: unsupported channel count %d for the fused bias gradient", C);
      return E_INVALID;
    }
  }
  int nper = (NB + nsplit - 1) / nsplit;
  hipLaunchKernelGGL(prelu_bwd_kernel, dim3(gx, nsplit), dim3(256), 0, s, da, u, alpha, NB, E, C, nper, dalpha_part,
                     dbias_part, mode);
  DV_HIP(hipGetLastError());
  return OK;
}

// column sums of x[rows][C] (C multiple of 4) -> part[block][C]; blockIdx.y walks column tiles of 1024
constexpr int COLSUM_ROWS = 2048;
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long rows, int C,
                                                     float* __restrict__ part, int rows_per_block) {
  __shared__ f32x4 shv[256];
  const int q0 = blockIdx.y * 256;
  const int cq = min(256, C / 4 - q0);             // column quads of this tile
  const int rpb = 256 / cq;
  const int t = threadIdx.x;
  const int q = t % cq, rr = t / cq;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long r0 = (long)blockIdx.x * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  if (rr < rpb)
    for (long r = r0 + rr; r < r1; r += rpb) acc += *reinterpret_cast<const f32x4*>(x + r * C + (q0 + q) * 4);
  shv[t] = acc;
  __syncthreads();
  if (t < cq) {
    f32x4 tot = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < rpb; ++k) tot += shv[k * cq + t];
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.x * C + (q0 + t) * 4) = tot;
  }
}

int launch_colsum(const float* x, long rows, int C, float* part, int* nrows_part, hipStream_t s) {
  if ((C & 3) || C < 4) {
    set_error("colsum: unsupported C=%d", C);
    return E_INVALID;
  }
  // at least ~64 blocks for short matrices (the 256 x 560 encoder-dense bias gradient used to run in one block)
  int rpb = (int)std::min<long>(COLSUM_ROWS, std::max<long>(4, (rows + 63) / 64));
  int nb = (int)((rows + rpb - 1) / rpb);
  *nrows_part = nb;
  hipLaunchKernelGGL(colsum_kernel, dim3(nb, (C / 4 + 255) / 256), dim3(256), 0, s, x, rows, C, part, rpb);
  DV_HIP(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// relu + crop + Normal(loc, floor + scale) head: loss partial sums, outputs and d(loss)/d(tpre)
constexpr int HEAD_MAXNB = 16;
__global__ __launch_bounds__(256) void head_kernel(const HeadParams p) {
  __shared__ float sh[4];
  const long total = (long)p.NB * p.Hd * p.Hd;
  const long pp = (long)blockIdx.x * 256 + threadIdx.x;
  float nll = 0.f, se = 0.f;
  if (pp < total) {
    const int HdHd = p.Hd * p.Hd;
    const int b = (int)(pp / HdHd);
    const int rem = (int)(pp - (long)b * HdHd);
    const int oh = rem / p.Hd, ow = rem - oh * p.Hd;
    const int h = oh - p.crop0, w = ow - p.crop0;
    const bool in = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.H;
    const float* tp = p.tpre + pp * p.ld;
    float* dtp = p.dt ? p.dt + pp * p.ld : nullptr;
    if (dtp)
      for (int c = 2 * p.nb; c < p.ld; ++c) dtp[c] = 0.f;
    if (!in) {
      if (dtp)
        for (int c = 0; c < 2 * p.nb; ++c) dtp[c] = 0.f;
    } else {
      const long opix = ((long)b * p.H + h) * p.H + w;
      const float* yp = nullptr;
      if (p.y) {
        long row = p.idx ? p.idx[b] : p.first + b;
        yp = p.y + ((row * p.H + h) * p.H + w) * p.nb;
      }
#pragma unroll
      for (int c = 0; c < HEAD_MAXNB; ++c) {
        if (c >= p.nb) break;
        const float tl = tp[c], ts = tp[p.nb + c];
        const float loc = fmaxf(tl, 0.f);
        const float sig = p.sigma_floor + fmaxf(ts, 0.f);
        if (p.loc) p.loc[opix * p.nb + c] = loc;
        if (p.scale) p.scale[opix * p.nb + c] = sig;
        if (yp) {
          const float inv = 1.0f / sig;
          const float d = yp[c] - loc;
          const float r = d * inv;
          nll += 0.5f * r * r + logf(sig) + 0.91893853320467274178f;
          if (p.mse_sample) {
            const float ds = d - sig * dv_philox_normal((unsigned)(p.mse_row0 + b), (unsigned)((h * p.H + w) * p.nb + c), p.mse_stream, p.mse_seed);
            se += ds * ds;
          } else {
            se += d * d;
          }
          if (dtp) {
            dtp[c] = tl > 0.f ? -(r * inv) * p.gscale : 0.f;
            dtp[p.nb + c] = ts > 0.f ? (inv - r * r * inv) * p.gscale : 0.f;
          }
        }
      }
    }
  }
  float a = block_sum(nll, sh);
  float b = block_sum(se, sh);
  if (threadIdx.x == 0 && p.part) {
    p.part[blockIdx.x * 2] = a;
    p.part[blockIdx.x * 2 + 1] = b;
  }
}

// 6-band specialisation (the reference's only exercised band count): 3 x 16-byte loads/stores per pixel for the
// 12-channel head tensors, 8-byte accesses for the 6-channel label / output rows.
__global__ __launch_bounds__(256) void head6_kernel(const HeadParams p) {
  __shared__ float sh[4];
  const long total = (long)p.NB * p.Hd * p.Hd;
  const long pp = (long)blockIdx.x * 256 + threadIdx.x;
  float nll = 0.f, se = 0.f;
  if (pp < total) {
    const int HdHd = p.Hd * p.Hd;
    const int b = (int)(pp / HdHd);
    const int rem = (int)(pp - (long)b * HdHd);
    const int oh = rem / p.Hd, ow = rem - oh * p.Hd;
    const int h = oh - p.crop0, w = ow - p.crop0;
    const bool in = (unsigned)h < (unsigned)p.H && (unsigned)w < (unsigned)p.H;
    f32x4* dtp = p.dt ? reinterpret_cast<f32x4*>(p.dt + pp * p.ld) : nullptr;
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (dtp && p.ld == 16) dtp[3] = z4;                 // pad channels 12..15 of the 16-wide rows
    if (!in) {
      if (dtp) {
        dtp[0] = z4;
        dtp[1] = z4;
        dtp[2] = z4;
      }
    } else {
      const f32x4* tp = reinterpret_cast<const f32x4*>(p.tpre + pp * p.ld);
      const f32x4 t0 = tp[0], t1 = tp[1], t2 = tp[2];
      const float t[12] = {t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3]};
      const long opix = ((long)b * p.H + h) * p.H + w;
      float yv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (p.y) {
        long row = p.idx ? p.idx[b] : p.first + b;
        const f32x2* yp = reinterpret_cast<const f32x2*>(p.y + ((row * p.H + h) * p.H + w) * 6);
        const f32x2 y0 = yp[0], y1 = yp[1], y2 = yp[2];
        yv[0] = y0[0]; yv[1] = y0[1]; yv[2] = y1[0]; yv[3] = y1[1]; yv[4] = y2[0]; yv[5] = y2[1];
      }
      float loc[6], sig[6], d[12];
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        loc[c] = fmaxf(t[c], 0.f);
        sig[c] = p.sigma_floor + fmaxf(t[6 + c], 0.f);
        d[c] = d[6 + c] = 0.f;
        if (p.y) {
          const float inv = 1.0f / sig[c];
          const float df = yv[c] - loc[c];
          const float r = df * inv;
          nll += 0.5f * r * r + logf(sig[c]) + 0.91893853320467274178f;
          if (p.mse_sample) {
            const float dsm = df - sig[c] * dv_philox_normal((unsigned)(p.mse_row0 + b), (unsigned)((h * p.H + w) * 6 + c), p.mse_stream, p.mse_seed);
            se += dsm * dsm;
          } else {
            se += df * df;
          }
          d[c] = t[c] > 0.f ? -(r * inv) * p.gscale : 0.f;
          d[6 + c] = t[6 + c] > 0.f ? (inv - r * r * inv) * p.gscale : 0.f;
        }
      }
      if (dtp) {
        dtp[0] = (f32x4){d[0], d[1], d[2], d[3]};
        dtp[1] = (f32x4){d[4], d[5], d[6], d[7]};
        dtp[2] = (f32x4){d[8], d[9], d[10], d[11]};
      }
      if (p.loc) {
        f32x2* lp = reinterpret_cast<f32x2*>(p.loc + opix * 6);
        lp[0] = (f32x2){loc[0], loc[1]};
        lp[1] = (f32x2){loc[2], loc[3]};
        lp[2] = (f32x2){loc[4], loc[5]};
      }
      if (p.scale) {
        f32x2* sp = reinterpret_cast<f32x2*>(p.scale + opix * 6);
        sp[0] = (f32x2){sig[0], sig[1]};
        sp[1] = (f32x2){sig[2], sig[3]};
        sp[2] = (f32x2){sig[4], sig[5]};
      }
    }
  }
  float a = block_sum(nll, sh);
  float b2 = block_sum(se, sh);
  if (threadIdx.x == 0 && p.part) {
    p.part[blockIdx.x * 2] = a;
    p.part[blockIdx.x * 2 + 1] = b2;
  }
}

int launch_head(const HeadParams& p, hipStream_t s, int* nblocks_out) {
  if (p.nb > HEAD_MAXNB) {
    set_error("head: at most %d bands", HEAD_MAXNB);
    return E_INVALID;
  }
  long total = (long)p.NB * p.Hd * p.Hd;
  int nb = (int)((total + 255) / 256);
  if (nblocks_out) *nblocks_out = nb;
  if (nb == 0) return OK;
  if (p.ld < 2 * p.nb || (p.ld & 3)) {
    set_error("head: bad row stride");
    return E_INVALID;
  }
  if (p.nb == 6 && (p.ld == 12 || p.ld == 16))
    hipLaunchKernelGGL(head6_kernel, dim3(nb), dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL(head_kernel, dim3(nb), dim3(256), 0, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// MultivariateNormalTriL sampler + MC KL; one 64-lane wave per stamp, lane i owns row i of L.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }

__device__ __forceinline__ int tril_src(int d, int i, int j) {
  // position inside t[d:] that tfp.math.fill_triangular puts at L[i][j] (j <= i)
  const int m = d * (d + 1) / 2;
  const int q = i * d + j;
  return q < m - d ? d + q : d * d - 1 - q;
}

__global__ __launch_bounds__(256) void sampler_fwd_kernel(const SamplerParams p) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= p.NB) return;
  const int d = p.d;
  const int tw = d + d * (d + 1) / 2;
  // Monte-Carlo replication (dv_infer_mc): row b is sample b / rep_nb of stamp b % rep_nb
  const int bs = p.rep_nb > 0 ? b % p.rep_nb : b;
  const unsigned long long seed =
      (p.seed_ptr ? *p.seed_ptr : p.seed) + (p.rep_nb > 0 ? (unsigned long long)(b / p.rep_nb) : 0ull);
  const float* t = p.t + (size_t)bs * p.ldt;
  const size_t zo = (size_t)b * p.ldz;
  // the row of t goes through LDS (coalesced): the lower-triangle gather below would otherwise be a chain of
  // d dependent, divergent global loads per stamp (31 us per step at d = 32)
  __shared__ float st_all[4][64 + 64 * 65 / 2];
  float* st = st_all[threadIdx.x >> 6];
  if (p.nslab > 0) {
    // (two passes: the sums go to LDS first, so that no global store sits between the loads of consecutive elements -
    // the compiler cannot tell that t_out does not alias the slabs - and all of a lane's slab loads are in flight together)
    const float* __restrict__ sb = p.slab + (size_t)bs * p.lds;
    const size_t ss = (size_t)p.slab_stride;
    const int ns = p.nslab;
#pragma unroll 3
    for (int i = lane; i < tw; i += 64) {
      // (up to four slabs as four independent loads - a run-time slab loop is a chain of dependent round trips: 18 us
      // per launch instead of 8 - added in slab order)
      const float v0 = sb[i], v1 = ns > 1 ? sb[ss + i] : 0.f, v2 = ns > 2 ? sb[2 * ss + i] : 0.f, v3 = ns > 3 ? sb[3 * ss + i] : 0.f;
      float v = (((p.tbias[i] + v0) + v1) + v2) + v3;
      for (int sl = 4; sl < ns; ++sl) v += sb[(size_t)sl * ss + i];
      st[i] = v;
    }
    if (p.rep_nb == 0 || b < p.rep_nb) {
      float* to = p.t_out + (size_t)bs * p.ldt;
      for (int i = lane; i < p.ldt; i += 64) to[i] = i < tw ? st[i] : 0.f;
    }
  } else {
    for (int i = lane; i < tw; i += 64) st[i] = t[i];
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  float e = 0.f;
  if (lane < d) {
    if (p.gen) {
      unsigned r[4];
      philox4x32_10(p.row0 + bs, lane >> 2, p.stream, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
      const int a = lane & 3;
      const float u1 = ((float)r[a & ~1] + 1.0f) * 2.3283064365386963e-10f;
      const float u2 = ((float)r[(a & ~1) + 1] + 1.0f) * 2.3283064365386963e-10f;
      // (r+1)*2^-32 in fp32 can round to exactly 1.0 -> log(1)=0, still finite
      const float rad = sqrtf(-2.0f * logf(fminf(u1, 1.0f)));
      float sn, cs;
      sincosf(6.283185307179586f * u2, &sn, &cs);
      e = (a & 1) ? rad * sn : rad * cs;
      p.eps[zo + lane] = e;
    } else {
      e = p.eps[zo + lane];
    }
  } else if (lane < p.ldz && p.gen) {
    p.eps[zo + lane] = 0.f;
  }
  float z = 0.f, logd = 0.f, l2 = 0.f, ldiag = 0.f;
  if (lane < d) {
    z = st[lane];
    // the diagonal's softplus / log once per lane, outside the row loop (inside it the divergent branch made every
    // iteration pay for the transcendental sequence of one lane: 33 -> 12 us per half batch)
    ldiag = softplus_f(st[d + tril_src(d, lane, lane)]) + p.diag_shift;
    logd = logf(ldiag);
  }
  for (int j = 0; j < d; ++j) {
    const float ej = __shfl(e, j, 64);
    if (lane < d && j <= lane) {
      const float l = j == lane ? ldiag : st[d + tril_src(d, lane, j)];
      z += l * ej;
      l2 += l * l;
    }
  }
  float k = (lane < d) ? (0.5f * z * z - 0.5f * e * e - logd) : 0.f;
  k = wave_sum(k);
  if (lane < p.ldz) {                                 // (pad columns d .. ldz - 1: zeros)
    p.z[zo + lane] = lane < d ? z : 0.f;
    if (p.ain) p.ain[zo + lane] = lane < d ? (z > 0.f ? z : p.alpha_in[lane] * z) : 0.f;
    if (p.stddev) p.stddev[zo + lane] = lane < d ? sqrtf(l2) : 0.f;
  }
  if (lane == 0) p.kl[b] = k;
}

int launch_sampler_fwd(const SamplerParams& p, hipStream_t s) {
  if (p.d > 64 || p.d < 1 || p.ldz < p.d || p.ldz > 64 || p.ldt < p.d + p.d * (p.d + 1) / 2) {
    set_error("sampler: latent_dim must be in [1,64] (row strides %d / %d)", p.ldt, p.ldz);
    return E_INVALID;
  }
  if (p.NB == 0) return OK;
  hipLaunchKernelGGL(sampler_fwd_kernel, dim3((p.NB + 3) / 4), dim3(256), 0, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

__global__ __launch_bounds__(256) void sampler_bwd_kernel(const float* __restrict__ t, const float* __restrict__ eps,
                                                          const float* __restrict__ z, const float* __restrict__ dz,
                                                          float* __restrict__ dt, int NB, int d, int ldt, int ldz,
                                                          float diag_shift, float kls) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= NB) return;
  const int tw = d + d * (d + 1) / 2;
  const float* tb = t + (size_t)b * ldt;
  float* dtb = dt + (size_t)b * ldt;
  const size_t zo = (size_t)b * ldz;
  // the gradient row is assembled in LDS and written out coalesced (every element of the row is produced exactly
  // once: fill_triangular is a bijection between the d(d+1)/2 inputs and the lower triangle)
  __shared__ float st_all[4][64 + 64 * 65 / 2];
  float* st = st_all[threadIdx.x >> 6];
  float e = 0.f, g = 0.f, raw = 0.f;
  if (lane < d) {
    e = eps[zo + lane];
    g = dz[zo + lane] + kls * z[zo + lane];
    raw = tb[d + tril_src(d, lane, lane)];
    st[lane] = g;
  }
  float ldiag = 1.f, sg = 0.f;
  if (lane < d) {                                    // diagonal terms once per lane (see sampler_fwd_kernel)
    ldiag = softplus_f(raw) + diag_shift;
    sg = 1.0f / (1.0f + expf(-raw));
  }
  for (int j = 0; j < d; ++j) {
    const float ej = __shfl(e, j, 64);
    if (lane < d && j <= lane) {
      const int src = d + tril_src(d, lane, j);
      float v = g * ej;
      if (j == lane) v = (v - kls / ldiag) * sg;
      st[src] = v;
    }
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  for (int i = lane; i < ldt; i += 64) dtb[i] = i < tw ? st[i] : 0.f;
}

int launch_sampler_bwd(const float* t, const float* eps, const float* z, const float* dz, float* dt, int NB, int d,
                       int ldt, int ldz, float diag_shift, float kls, hipStream_t s) {
  if (NB == 0) return OK;
  hipLaunchKernelGGL(sampler_bwd_kernel, dim3((NB + 3) / 4), dim3(256), 0, s, t, eps, z, dz, dt, NB, d, ldt, ldz, diag_shift,
                     kls);
  DV_HIP(hipGetLastError());
  return OK;
}

// ------------------------------------------------------------------------------------------------
// tf.optimizers.legacy.Adam (reference train.py:126): m += (g-m)(1-b1); v += (g^2-v)(1-b2); w -= lr_t m/(sqrt(v)+eps)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                                   const float* __restrict__ g, long n4, float lr_t, float b1,
                                                   float b2, float eps) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 gw = reinterpret_cast<const f32x4*>(g)[i];
    f32x4 mw = reinterpret_cast<f32x4*>(m)[i];
    f32x4 vw = reinterpret_cast<f32x4*>(v)[i];
    f32x4 ww = reinterpret_cast<f32x4*>(w)[i];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      mw[k] += (gw[k] - mw[k]) * (1.f - b1);
      vw[k] += (gw[k] * gw[k] - vw[k]) * (1.f - b2);
      ww[k] -= lr_t * mw[k] / (sqrtf(vw[k]) + eps);
    }
    reinterpret_cast<f32x4*>(m)[i] = mw;
    reinterpret_cast<f32x4*>(v)[i] = vw;
    reinterpret_cast<f32x4*>(w)[i] = ww;
  }
}

int launch_adam(float* w, float* m, float* v, const float* g, long n, float lr_t, float b1, float b2, float eps,
                hipStream_t s) {
  if (n & 3) return E_INVALID;
  long n4 = n / 4;
  if (n4 == 0) return OK;
  int blocks = (int)min((n4 + 255) / 256, (long)2048);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, s, w, m, v, g, n4, lr_t, b1, b2, eps);
  DV_HIP(hipGetLastError());
  return OK;
}

// First conv with the input BatchNorm folded in (reference model.py:79-83): conv(gamma*xhat + beta) over the
// zero-padded image equals conv'(xhat8) with
//   wp[t][c][co] = w[t][c][co] * gamma[c]  (c < cin),   wp[t][cin][co] = sum_c w[t][c][co] * beta[c],   0 beyond,
// where channel `cin` of xhat8 is 1 inside the image.  The weight gradient w.r.t. wp (G) then yields every
// gradient of the layer pair without a data-gradient pass: see bn_conv0_grads_kernel.
__global__ void fold_bn_w1_kernel(const float* __restrict__ w, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, float* __restrict__ wp, int taps, int cin, int cpad,
                                  int cout) {
  int i = blockIdx.x * 256 + threadIdx.x;
  int total = taps * cpad * cout;
  if (i >= total) return;
  int co = i % cout;
  int c = (i / cout) % cpad;
  int t = i / (cout * cpad);
  float v = 0.f;
  if (c < cin) {
    v = w[(t * cin + c) * cout + co] * gamma[c];
  } else if (c == cin) {
    for (int k = 0; k < cin; ++k) v += w[(t * cin + k) * cout + co] * beta[k];
  }
  wp[i] = v;
}

int launch_pad_w1(const float* w, const float* gamma, const float* beta, float* wp, int taps, int cin, int cpad,
                  int cout, hipStream_t s) {
  if (cin >= cpad) return E_INVALID;
  int total = taps * cpad * cout;
  hipLaunchKernelGGL(fold_bn_w1_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, gamma, beta, wp, taps, cin,
                     cpad, cout);
  DV_HIP(hipGetLastError());
  return OK;
}

// G[t][c8][co] = d(loss)/d(wp).  dW[t][c][co] = gamma[c] G[t][c][co] + beta[c] G[t][cin][co];
// d(gamma)[c] = sum_{t,co} w[t][c][co] G[t][c][co];  d(beta)[c] = sum_{t,co} w[t][c][co] G[t][cin][co].   One block.
__global__ __launch_bounds__(256) void bn_conv0_grads_kernel(const float* __restrict__ G, const float* __restrict__ w,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ dW,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int taps, int cin, int cpad, int cout) {
  __shared__ double sg[BN_MAXC][256];              // (32 KB: the d(gamma) sums, then the d(beta) sums)
  double ag[BN_MAXC], ab[BN_MAXC];
#pragma unroll
  for (int c = 0; c < BN_MAXC; ++c) ag[c] = ab[c] = 0.0;
  const int total = taps * cin * cout;
  for (int i = threadIdx.x; i < total; i += 256) {
    const int co = i % cout;
    const int c = (i / cout) % cin;
    const int t = i / (cout * cin);
    const float g = G[(t * cpad + c) * cout + co];
    const float g1 = G[(t * cpad + cin) * cout + co];
    const float wv = w[i];
    dW[i] = gamma[c] * g + beta[c] * g1;
#pragma unroll
    for (int k = 0; k < BN_MAXC; ++k)
      if (k == c) {
        ag[k] += (double)wv * g;
        ab[k] += (double)wv * g1;
      }
  }
#pragma unroll
  for (int round = 0; round < 2; ++round) {
#pragma unroll
    for (int c = 0; c < BN_MAXC; ++c) sg[c][threadIdx.x] = round ? ab[c] : ag[c];
    __syncthreads();
    // fixed-order sum of a band's 256 partials in two levels of 16 (one thread walking all 256 with a dependent LDS read and
    // a double add per term made this one-block kernel 13 us of the serial tail of every train step)
    const int c = threadIdx.x >> 4, j = threadIdx.x & 15;          // 16 bands x 16 runs
    double part = 0.0;
    if (c < cin)
      for (int k = 0; k < 16; ++k) part += sg[c][j * 16 + k];
    __syncthreads();
    if (c < cin) sg[c][j] = part;
    __syncthreads();
    if ((int)threadIdx.x < cin) {
      double acc = 0.0;
      for (int k = 0; k < 16; ++k) acc += sg[threadIdx.x][k];
      (round ? dbeta : dgamma)[threadIdx.x] = (float)acc;
    }
    __syncthreads();
  }
}

int launch_bn_conv0_grads(const float* G, const float* w, const float* gamma, const float* beta, float* dW,
                          float* dgamma, float* dbeta, int taps, int cin, int cpad, int cout, hipStream_t s) {
  hipLaunchKernelGGL(bn_conv0_grads_kernel, dim3(1), dim3(256), 0, s, G, w, gamma, beta, dW, dgamma, dbeta, taps, cin,
                     cpad, cout);
  DV_HIP(hipGetLastError());
  return OK;
}

// head conv kernel/bias padded from 2*bands to a multiple of 16 output channels (zeros), so that its output and the
// loss gradient are 16-channel rows the 32-wide-K gather-GEMM can consume
__global__ void pad_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int nsrc, int ndst) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * ndst) return;
  int r = i / ndst, c = i - r * ndst;
  dst[i] = c < nsrc ? src[r * nsrc + c] : 0.f;
}
int launch_pad_cols(const float* src, float* dst, int rows, int nsrc, int ndst, hipStream_t s) {
  int total = rows * ndst;
  hipLaunchKernelGGL(pad_cols_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, dst, rows, nsrc, ndst);
  DV_HIP(hipGetLastError());
  return OK;
}
__global__ void take_cols_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int nsrc, int ndst) {
  int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * ndst) return;
  int r = i / ndst, c = i - r * ndst;
  dst[i] = src[r * nsrc + c];
}
int launch_take_cols(const float* src, float* dst, int rows, int nsrc, int ndst, hipStream_t s) {
  int total = rows * ndst;
  hipLaunchKernelGGL(take_cols_kernel, dim3((total + 255) / 256), dim3(256), 0, s, src, dst, rows, nsrc, ndst);
  DV_HIP(hipGetLastError());
  return OK;
}

// Welford running mean / sum of squared deviations over Monte-Carlo decodes (epistemic uncertainty,
// reference field_deblender.py:303-313: np.std(deblend(net, [stamp]*100)[0], axis=0))
__global__ __launch_bounds__(256) void welford_update_kernel(const float* __restrict__ x, float* __restrict__ mean,
                                                             float* __restrict__ m2, long n4, int k) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 mu = k == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : reinterpret_cast<f32x4*>(mean)[i];
    f32x4 s2 = k == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : reinterpret_cast<f32x4*>(m2)[i];
    const float inv = 1.0f / (float)(k + 1);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = v[c] - mu[c];
      mu[c] += d * inv;
      s2[c] += d * (v[c] - mu[c]);
    }
    reinterpret_cast<f32x4*>(mean)[i] = mu;
    reinterpret_cast<f32x4*>(m2)[i] = s2;
  }
}
__global__ __launch_bounds__(256) void welford_update_multi_kernel(const float* __restrict__ x, float* __restrict__ mean,
                                                                   float* __restrict__ m2, long n, int reps, int k0) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float mu = k0 == 0 ? 0.f : mean[i];
    float s2 = k0 == 0 ? 0.f : m2[i];
    for (int r = 0; r < reps; ++r) {               // same fold order (and arithmetic) as one welford_update per sample
      const float v = x[(long)r * n + i];
      const float d = v - mu;
      mu += d * (1.0f / (float)(k0 + r + 1));
      s2 += d * (v - mu);
    }
    mean[i] = mu;
    m2[i] = s2;
  }
}
int launch_welford_update_multi(const float* x, float* mean, float* m2, long n, int reps, int k0, hipStream_t s) {
  if (reps < 1) return E_INVALID;
  if (n <= 0) return OK;
  hipLaunchKernelGGL(welford_update_multi_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, s,
                     x, mean, m2, n, reps, k0);
  DV_HIP(hipGetLastError());
  return OK;
}
__global__ __launch_bounds__(256) void welford_finish_kernel(float* __restrict__ m2, long n4, float inv_n) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    f32x4 s2 = reinterpret_cast<f32x4*>(m2)[i];
#pragma unroll
    for (int c = 0; c < 4; ++c) s2[c] = sqrtf(fmaxf(s2[c] * inv_n, 0.f));
    reinterpret_cast<f32x4*>(m2)[i] = s2;
  }
}
int launch_welford_update(const float* x, float* mean, float* m2, long n, int k, hipStream_t s) {
  if (n & 3) return E_INVALID;
  long n4 = n / 4;
  if (n4 == 0) return OK;
  hipLaunchKernelGGL(welford_update_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 4096)), dim3(256), 0, s, x,
                     mean, m2, n4, k);
  DV_HIP(hipGetLastError());
  return OK;
}
int launch_welford_finish(float* m2, long n, int count, hipStream_t s) {
  if ((n & 3) || count < 1) return E_INVALID;
  long n4 = n / 4;
  if (n4 == 0) return OK;
  hipLaunchKernelGGL(welford_finish_kernel, dim3((unsigned)std::min<long>((n4 + 255) / 256, 4096)), dim3(256), 0, s, m2,
                     n4, 1.0f / (float)count);
  DV_HIP(hipGetLastError());
  return OK;
}

// tanh(arcsinh(x)) stamp normalisation and its inverse (reference normalize/normalize.py:3-7) around inference
// (deblend_cutout/deblender.py:14-22).  tanh(arcsinh x) = x / sqrt(1 + x^2) and sinh(arctanh y) = y / sqrt(1 - y^2);
// the inverse clamps |y| below 1 as the host path does.
__global__ void normalise_kernel(float* __restrict__ x, long n, int inverse) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float v = x[i];
    if (inverse) {
      v = fminf(fmaxf(v, -0.99999988f), 0.99999988f);
      x[i] = v / sqrtf(1.0f - v * v);
    } else {
      x[i] = v / sqrtf(1.0f + v * v);
    }
  }
}
int launch_normalise(float* x, long n, bool inverse, hipStream_t s) {
  if (n <= 0) return OK;
  int blocks = (int)min((n + 255) / 256, (long)4096);
  hipLaunchKernelGGL(normalise_kernel, dim3(blocks), dim3(256), 0, s, x, n, inverse ? 1 : 0);
  DV_HIP(hipGetLastError());
  return OK;
}

__global__ void fill_kernel(float* p, long n, float v) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = v;
}
int launch_fill(float* p, long n, float v, hipStream_t s) {
  if (n <= 0) return OK;
  int blocks = (int)min((n + 255) / 256, (long)2048);
  hipLaunchKernelGGL(fill_kernel, dim3(blocks), dim3(256), 0, s, p, n, v);
  DV_HIP(hipGetLastError());
  return OK;
}

__global__ void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx, int first, int NB,
                                   long row_elems, float* __restrict__ dst) {
  long total = (long)NB * row_elems;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int b = (int)(i / row_elems);
    long e = i - (long)b * row_elems;
    long row = idx ? idx[b] : first + b;
    dst[i] = src[row * row_elems + e];
  }
}
int launch_gather_rows(const float* src, const int* idx, int first, int NB, long row_elems, float* dst,
                       hipStream_t s) {
  long total = (long)NB * row_elems;
  if (total <= 0) return OK;
  int blocks = (int)min((total + 255) / 256, (long)4096);
  hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, s, src, idx, first, NB, row_elems, dst);
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
