// Small-batch forward of a conv / dense layer stack in ONE cooperative launch (gfx950 / MI355X only).
//
// deblend() on a handful of stamps (deblend_cutout/deblender.py:18 with one detected object, the per-object calls of
// deblend/field_deblender.py:265-274) is a chain of ~45 dependent few-microsecond kernels in the batched engine and is
// bound by the GPU's dispatch-to-dispatch latency (~13 us per kernel), not by arithmetic: the whole forward pass of
// one stamp is 0.33 GMAC, microseconds of vector work on 256 CUs.  Here the encoder stack (8 convs + flatten / PReLU /
// Dense, model.py:79-98) and the decoder stack (PReLU / Dense / PReLU / Dense / 8 Conv2DTranspose / head conv,
// model.py:112-137,149) are each ONE kernel: every workgroup walks the layer list, takes the layer's tiles in a
// grid-stride loop and meets the others at a grid-wide barrier (cooperative launch: all workgroups are resident).
//
// A tile = one stamp x a few output pixels x up to 64 output channels.  The four waves of a workgroup split the nine
// taps (or the K range of a dense layer) and are summed through LDS in a fixed order; lanes run over output channels,
// so weight reads are coalesced (Conv2D, Dense: [k][n]) or 16-byte per lane (Conv2DTranspose: [tap][n][c]) and the
// activation reads are wave-uniform broadcasts.  Plain fp32 FMAs, same arithmetic as the batched kernels up to the
// order of the K sum.
#include "common.h"
#include <hip/hip_cooperative_groups.h>

namespace cg = cooperative_groups;

namespace dv {

namespace {

__device__ __forceinline__ float sm_prelu(float v, float al) { return v > 0.f ? v : al * v; }

// conv tile: form 0 (Conv2D): source = out*s + k - pb; form 1 (Conv2DTranspose as a gather): out + pb = s*source + k
__device__ void sm_conv_layer(const SmLayer& L, int NB, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int cw = 1;
  while (cw < L.cout && cw < 64) cw <<= 1;            // lanes per pixel (power of two)
  const int ppw = 64 / cw;                            // pixels per tile
  const int nchunks = (L.cout + cw - 1) / cw;
  const int npix = L.hout * L.hout;
  const int npg = (npix + ppw - 1) / ppw;
  const long ntiles = (long)NB * npg * nchunks;
  const int pi = lane / cw, nl = lane % cw;
  // taps of this wave: 9 taps over 4 waves = 3,2,2,2
  const int t0 = wave == 0 ? 0 : 1 + 2 * wave, t1 = wave == 0 ? 3 : 3 + 2 * wave;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int nc = (int)(t % nchunks);
    const int pg = (int)((t / nchunks) % npg);
    const int b = (int)(t / ((long)nchunks * npg));
    const int pix = pg * ppw + pi;
    const int n = nc * cw + nl;
    const bool live = pix < npix && n < L.cout;
    const int oy = live ? pix / L.hout : 0, ox = live ? pix % L.hout : 0;
    float acc = 0.f;
    if (live) {
      for (int tap = t0; tap < t1; ++tap) {
        const int ky = tap / 3, kx = tap % 3;
        int iy, ix;
        bool ok;
        if (L.form == 0) {
          iy = oy * L.s + ky - L.pb;
          ix = ox * L.s + kx - L.pb;
          ok = iy >= 0 && iy < L.hin && ix >= 0 && ix < L.hin;
        } else {
          const int ty = oy + L.pb - ky, tx = ox + L.pb - kx;
          ok = ty >= 0 && tx >= 0 && ty % L.s == 0 && tx % L.s == 0;
          iy = ty / L.s;
          ix = tx / L.s;
          ok = ok && iy < L.hin && ix < L.hin;
        }
        if (!ok) continue;
        const float* xr = L.in + ((size_t)(b * L.hin + iy) * L.hin + ix) * L.cin;
        if (L.nmajor) {
          const float* wr = L.W + ((size_t)tap * L.cout + n) * L.cin;
#pragma unroll 2
          for (int c = 0; c < L.cin; c += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + c);
            const float4 wv = *reinterpret_cast<const float4*>(wr + c);
            acc = fmaf(xv.x, wv.x, acc);
            acc = fmaf(xv.y, wv.y, acc);
            acc = fmaf(xv.z, wv.z, acc);
            acc = fmaf(xv.w, wv.w, acc);
          }
        } else {
          const float* wr = L.W + (size_t)tap * L.cin * L.cout + n;
#pragma unroll 2
          for (int c = 0; c < L.cin; c += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xr + c);
            const float w0 = wr[(size_t)(c + 0) * L.cout], w1 = wr[(size_t)(c + 1) * L.cout];
            const float w2 = wr[(size_t)(c + 2) * L.cout], w3 = wr[(size_t)(c + 3) * L.cout];
            acc = fmaf(xv.x, w0, acc);
            acc = fmaf(xv.y, w1, acc);
            acc = fmaf(xv.z, w2, acc);
            acc = fmaf(xv.w, w3, acc);
          }
        }
      }
    }
    if (wave) red[(wave - 1) * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && live) {
      float v = (L.bias ? L.bias[n] : 0.f) + acc;
      v += red[lane];
      v += red[64 + lane];
      v += red[128 + lane];
      const size_t o = ((size_t)b * npix + pix) * L.cout + n;
      if (L.alpha) v = sm_prelu(v, L.alpha[(size_t)pix * L.cout + n]);
      L.out[o] = v;
    }
    __syncthreads();
  }
}

// dense tile: out[b][n] = bias[n] + sum_k f(in[b][k]) * W[k][n], f = PReLU with in_alpha (or identity); K split over
// ksplit tiles x 4 waves; ksplit > 1 writes partials that the following SM_REDUCE entry sums in order
__device__ void sm_dense_layer(const SmLayer& L, int NB, float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int K = L.cin, N = L.cout;
  const int nchunks = (N + 63) / 64;
  const int ks = L.ksplit < 1 ? 1 : L.ksplit;
  const int kper = (K + 4 * ks - 1) / (4 * ks);
  const long ntiles = (long)NB * nchunks * ks;
  for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int nc = (int)(t % nchunks);
    const int kq = (int)((t / nchunks) % ks);
    const int b = (int)(t / ((long)nchunks * ks));
    const int n = nc * 64 + lane;
    const bool live = n < N;
    const int k0 = (kq * 4 + wave) * kper, k1 = min(K, k0 + kper);
    const float* xr = L.in + (size_t)b * K;
    float acc = 0.f;
    if (live) {
#pragma unroll 4
      for (int k = k0; k < k1; ++k) {
        float xv = xr[k];
        if (L.in_alpha) xv = sm_prelu(xv, L.in_alpha[k]);
        acc = fmaf(xv, L.W[(size_t)k * N + n], acc);
      }
    }
    if (wave) red[(wave - 1) * 64 + lane] = acc;
    __syncthreads();
    if (wave == 0 && live) {
      float v = acc;
      v += red[lane];
      v += red[64 + lane];
      v += red[128 + lane];
      if (ks > 1) {
        L.part[((size_t)kq * NB + b) * N + n] = v;
      } else {
        v += L.bias ? L.bias[n] : 0.f;
        if (L.alpha) v = sm_prelu(v, L.alpha[n]);
        L.out[(size_t)b * N + n] = v;
      }
    }
    __syncthreads();
  }
}

__device__ void sm_reduce_layer(const SmLayer& L, int NB) {
  const long total = (long)NB * L.cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i % L.cout);
    float v = L.bias ? L.bias[n] : 0.f;
    for (int q = 0; q < L.ksplit; ++q) v += L.part[(size_t)q * total + i];
    if (L.alpha) v = sm_prelu(v, L.alpha[n]);
    L.out[i] = v;
  }
}

__global__ void __launch_bounds__(256) small_stack_kernel(SmStack p) {
  __shared__ float red[3 * 64];
  cg::grid_group grid = cg::this_grid();
  for (int li = 0; li < p.n; ++li) {
    const SmLayer& L = p.L[li];
    if (p.dbg == 1) { if (li + 1 < p.n) grid.sync(); continue; }     // timing ablation: barriers only
    if (L.kind == SM_CONV) sm_conv_layer(L, p.NB, red);
    else if (L.kind == SM_DENSE) sm_dense_layer(L, p.NB, red);
    else sm_reduce_layer(L, p.NB);
    if (li + 1 < p.n && p.dbg != 2) grid.sync();         // every workgroup passes the same n - 1 barriers
  }
}

}  // namespace

int small_stack_max_grid(int device) {
  int per_cu = 0, cus = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, small_stack_kernel, 256, 0) != hipSuccess) return 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess) return 0;
  int coop = 0;
  if (hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, device) != hipSuccess || !coop) return 0;
  static const int want = getenv("DV_SMALL_WGS_PER_CU") ? atoi(getenv("DV_SMALL_WGS_PER_CU")) : 4;
  return std::max(0, std::min(per_cu, std::max(1, want))) * cus;
}

int launch_small_stack(const SmStack& p, int grid, hipStream_t s) {
  if (grid <= 0 || p.n <= 0 || p.n > DV_SM_MAX_LAYERS) return E_INVALID;
  for (int i = 0; i < p.n; ++i) {
    const SmLayer& L = p.L[i];
    if (L.kind == SM_CONV && ((L.cin & 3) || !L.in || !L.out || !L.W || (L.s != 1 && L.s != 2))) return E_INVALID;
    if (L.kind == SM_DENSE && (!L.in || !L.W || (L.ksplit > 1 ? !L.part : !L.out))) return E_INVALID;
    if (L.kind == SM_REDUCE && (!L.part || !L.out || L.ksplit < 1)) return E_INVALID;
  }
  SmStack copy = p;
  static const int dbg = getenv("DV_SMALL_DBG") ? atoi(getenv("DV_SMALL_DBG")) : 0;
  copy.dbg = dbg;
  void* args[] = {&copy};
  if (hipLaunchCooperativeKernel(reinterpret_cast<const void*>(small_stack_kernel), dim3(grid), dim3(256), args, 0, s) !=
      hipSuccess)
    return E_HIP;
  return OK;
}

}  // namespace dv
