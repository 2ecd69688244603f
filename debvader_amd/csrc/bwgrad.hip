// bf16 weight gradient over stamp-inner tensors (see bf16.h):
//   out[tap][cx][cy] = sum_{gy, stamp} X[gy*s + tap - pb][stamp][cx] * Y[gy][stamp][cy]
// Kernel gradients of Conv2D (X = layer input, Y = d(pre-activation)) and Conv2DTranspose (X = d(pre-activation),
// Y = layer input; kernel layout (kh,kw,cout,cin)) of model.py:81-91,121-137.
//
// The contraction index is the STAMP: for one Y pixel and one tap, both operands are [stamps][channels] blocks of the
// stamp-inner layout, and v_mfma_f32_16x16x32_bf16 wants 8 consecutive k per lane for a fixed channel - a transposed
// read, which ds_read_b64_tr_b16 delivers from the row-major LDS image (4 stamps x 16 channels per 16-lane group).
// A workgroup owns a (32*WX) x (32*WY) channel tile, NT of the nine taps and a range of Y pixels; per (pixel,
// 32-stamp k-step) the Y block and the X blocks of its valid taps arrive by LDS-DMA (double buffered), each wave
// multiplies its 32 x 32 sub-tile for every tap.  Taps outside the image are skipped (uniform per pixel).
// Results are fp32 partial slabs [split][9][Cx][Cy], summed in a fixed order by reduce_partials (deterministic).
#include "common.h"
#include "bf16.h"
#include <algorithm>
#include <stdlib.h>

namespace dv {

typedef const __attribute__((address_space(1))) void* bw_gptr_t;
typedef __attribute__((address_space(3))) void* bw_lptr_t;
typedef __bf16 bw_bf16;
typedef __bf16 bw_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bw_bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) bw_bf16x4* bw_l4ptr_t;

namespace {
struct BWGeom {
  int WX, WY;          // waves along cx / cy
  int ntx, nty;        // channel tiles
  int ntg;             // tap groups (9 / NT)
  int nsplit, pix_per_split;
  int kst;             // 32-stamp k-steps per pixel
  int px, py;          // 1-KiB DMA pieces per k-step: X per tap, Y
  int stage_bytes;
};

// two transposed reads = the 8 k values (stamps 4*grp+0..3 and 16+4*grp+0..3) of channel (lane & 15)
__device__ __forceinline__ bw_bf16x8 tr_frag(const unsigned char* p0, const unsigned char* p1) {
  const bw_bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bw_l4ptr_t)(p0));
  const bw_bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((bw_l4ptr_t)(p1));
  bw_bf16x8 r;
  r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
  r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
  return r;
}
}  // namespace

// XC16 / YC16: the operand has 16 channels (first conv's padded input, the head's 16-channel gradient): a DMA piece
// is then [32 stamps][16 ch]; otherwise [16 stamps][32 ch] with the two 32-byte halves of rows 4-7 / 12-15 swapped
// (on the source side) so that the transposed reads of a half-wave touch all 64 banks once.
template <bool XC16, bool YC16, int NT>
__global__ __launch_bounds__(256, 2) void bwgrad_kernel(const BWgradParams p, const BWGeom gm) {
  constexpr int BX = XC16 ? 1 : 2, BY = YC16 ? 1 : 2;     // 16-channel blocks per wave
  constexpr int CXW = 16 * BX, CYW = 16 * BY;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = gm.WX * gm.WY;
  const int wx = wave / gm.WY, wy = wave - wx * gm.WY;
  int bid = blockIdx.x;
  const int tcy = bid % gm.nty; bid /= gm.nty;
  const int tcx = bid % gm.ntx; bid /= gm.ntx;
  const int tg = bid % gm.ntg;
  const int split = bid / gm.ntg;
  const int cx0 = tcx * CXW * gm.WX, cy0 = tcy * CYW * gm.WY;
  const int npy = p.Hy * p.Hy;
  const int pix0 = split * gm.pix_per_split;
  const int pix1 = min(npy, pix0 + gm.pix_per_split);
  const int niter = (pix1 - pix0) * gm.kst;

  const bw_bf16* Xb = reinterpret_cast<const bw_bf16*>(p.X);
  const bw_bf16* Yb = reinterpret_cast<const bw_bf16*>(p.Y);
  const unsigned char* zlane = reinterpret_cast<const unsigned char*>(p.zero) + lane * 16;

  // per-lane element offset inside a DMA piece, relative to (first stamp of the k-step, first channel of the piece)
  // 32-ch piece: row = lane >> 2, 16-B quarter (lane & 3) with the halves swapped for rows 4-7, 12-15
  const int d32_row = lane >> 2;
  const int d32_q = (lane & 3) ^ ((((lane >> 2) >> 2) & 1) << 1);
  const int d16_row = lane >> 1, d16_h = lane & 1;

  // source pixel of tap t for Y pixel (gh, gw); -1 when outside
  auto xpix = [&](int gh, int gw, int t) -> int {
    const int kh = t / 3, kw = t - kh * 3;
    const int xh = gh * p.s + kh - p.pb, xw = gw * p.s + kw - p.pb;
    return (xh >= 0 && xh < p.Hx && xw >= 0 && xw < p.Hx) ? xh * p.Hx + xw : -1;
  };

  auto issue = [&](int it, int buf) {
    unsigned char* sY = smem + buf * gm.stage_bytes;
    unsigned char* sX = sY + gm.py * 1024;
    const int pl = it / gm.kst, ks = it - pl * gm.kst;
    const int gp = pix0 + pl;
    const int gh = gp / p.Hy, gw = gp - gh * p.Hy;
    const int st0 = ks * 32;                        // first stamp of the k-step
    const int total = gm.py + NT * gm.px;
    for (int q = wave; q < total; q += nw) {
      const bool isy = q < gm.py;
      int t = 0, pc = q;
      if (!isy) {
        t = (q - gm.py) / gm.px;
        pc = (q - gm.py) - t * gm.px;
      }
      int spix = gp;
      if (!isy) spix = xpix(gh, gw, tg * NT + t);
      if (spix < 0) continue;                       // tap outside the image: never read either
      const bool c16 = isy ? YC16 : XC16;
      const int C = isy ? p.Cy : p.Cx;
      const bw_bf16* base = (isy ? Yb : Xb) + (size_t)spix * p.NBp * C + (isy ? cy0 : cx0);
      const void* src;
      if (c16) {
        const int stamp = st0 + d16_row;
        src = stamp < p.NBp ? (const void*)(base + (size_t)stamp * C + d16_h * 8) : (const void*)zlane;
      } else {
        // piece pc = (32-channel block pc >> 1, stamp half pc & 1)
        const int stamp = st0 + (pc & 1) * 16 + d32_row;
        src = stamp < p.NBp ? (const void*)(base + (size_t)stamp * C + (pc >> 1) * 32 + d32_q * 8) : (const void*)zlane;
      }
      unsigned char* dst = (isy ? sY : sX + t * gm.px * 1024) + pc * 1024;
      __builtin_amdgcn_global_load_lds((bw_gptr_t)src, (bw_lptr_t)dst, 16, 0, 0);
    }
  };

  // transposed-read lane roles: 16-lane group grp reads stamps 4*grp + q, lane 4q+pp supplies row q, columns 4pp..
  const int grp = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
  const int trow = 4 * grp + tq;
  // byte offset inside a 32-ch piece for 16-channel half h: row*64 + ((h ^ (grp & 1)) * 32) + tp*8
  const int t32_0 = trow * 64 + ((0 ^ (grp & 1)) * 32) + tp * 8;
  const int t32_1 = trow * 64 + ((1 ^ (grp & 1)) * 32) + tp * 8;
  const int t16 = trow * 32 + tp * 8;               // 16-ch piece: second read at +16 rows = +512

  f32x4 acc[NT][BX][BY];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int bx = 0; bx < BX; ++bx)
#pragma unroll
      for (int by = 0; by < BY; ++by) acc[t][bx][by] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (niter > 0) issue(0, 0);
  for (int it = 0; it < niter; ++it) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (it + 1 < niter) issue(it + 1, (it + 1) & 1);
    const unsigned char* sY = smem + (it & 1) * gm.stage_bytes;
    const unsigned char* sX = sY + gm.py * 1024;
    const int pl = it / gm.kst;
    const int gp = pix0 + pl;
    const int gh = gp / p.Hy, gw = gp - gh * p.Hy;
    bw_bf16x8 b[BY];
    if constexpr (YC16) {
      b[0] = tr_frag(sY + t16, sY + t16 + 512);
    } else {
      // wave's 32-channel block wy: pieces (2*wy, 2*wy+1) = stamp halves
      const unsigned char* y0 = sY + (2 * wy) * 1024;
      b[0] = tr_frag(y0 + t32_0, y0 + 1024 + t32_0);
      b[1] = tr_frag(y0 + t32_1, y0 + 1024 + t32_1);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (xpix(gh, gw, tg * NT + t) < 0) continue;
      const unsigned char* xt = sX + t * gm.px * 1024;
      bw_bf16x8 a[BX];
      if constexpr (XC16) {
        a[0] = tr_frag(xt + t16, xt + t16 + 512);
      } else {
        const unsigned char* x0 = xt + (2 * wx) * 1024;
        a[0] = tr_frag(x0 + t32_0, x0 + 1024 + t32_0);
        a[1] = tr_frag(x0 + t32_1, x0 + 1024 + t32_1);
      }
#pragma unroll
      for (int bx = 0; bx < BX; ++bx)
#pragma unroll
        for (int by = 0; by < BY; ++by)
          acc[t][bx][by] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[bx], b[by], acc[t][bx][by], 0, 0, 0);
    }
  }

  // slab[split][(tap*Cx + cx)*Cy + cy]; lane (c = lane & 15 -> cy, g = lane >> 4 -> cx rows 4g..4g+3)
  float* slab = p.part + (size_t)split * 9 * p.Cx * p.Cy;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int bx = 0; bx < BX; ++bx)
#pragma unroll
      for (int by = 0; by < BY; ++by)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cx = cx0 + wx * CXW + bx * 16 + 4 * g + r;
          const int cy = cy0 + wy * CYW + by * 16 + c;
          slab[((size_t)(tg * NT + t) * p.Cx + cx) * p.Cy + cy] = acc[t][bx][by][r];
        }
}

int launch_bwgrad(const BWgradParams& p, hipStream_t s, int* nsplit_out) {
  const bool xc16 = p.Cx == 16, yc16 = p.Cy == 16;
  if ((!xc16 && p.Cx % 32) || (!yc16 && p.Cy % 32) || (p.NBp & 15)) {
    set_error("bwgrad: channels must be 16 or multiples of 32 (Cx %d, Cy %d), stamps padded to 16 (%d)", p.Cx, p.Cy,
              p.NBp);
    return E_INVALID;
  }
  BWGeom gm;
  const int cxw = xc16 ? 16 : 32, cyw = yc16 ? 16 : 32;
  const int bx_n = p.Cx / cxw, by_n = p.Cy / cyw;          // wave-sized blocks per axis
  gm.WX = std::min(2, bx_n);
  gm.WY = std::min(4 / gm.WX, by_n);
  if (gm.WX * gm.WY < 4 && bx_n >= 4 && gm.WY == 1) gm.WX = 4;
  while (bx_n % gm.WX) --gm.WX;
  while (by_n % gm.WY) --gm.WY;
  gm.ntx = bx_n / gm.WX;
  gm.nty = by_n / gm.WY;
  // taps per workgroup: all nine while the accumulators are few; deep layers split the taps over workgroups instead
  // of the pixels (their slabs are megabytes each)
  const long ce = (long)p.Cx * p.Cy;
  const int nt = ce <= 64 * 64 ? 9 : (ce <= 128 * 128 ? 3 : 1);
  gm.ntg = 9 / nt;
  gm.kst = (p.NBp + 31) / 32;
  gm.px = xc16 ? 1 : 2 * gm.WX;
  gm.py = yc16 ? 1 : 2 * gm.WY;
  gm.stage_bytes = (gm.py + nt * gm.px) * 1024;
  const int npy = p.Hy * p.Hy;
  const long slab = 9L * p.Cx * p.Cy;
  static const long target = getenv("DV_BWGRAD_TARGET") ? atol(getenv("DV_BWGRAD_TARGET")) : 1024;
  long ns = std::max(1L, target / ((long)gm.ntx * gm.nty * gm.ntg));
  ns = std::min<long>(ns, npy);
  ns = std::min<long>(ns, (long)(p.part_capacity / (size_t)slab));
  if (ns < 1) {
    set_error("bwgrad: slab workspace too small");
    return E_STATE;
  }
  gm.pix_per_split = (int)((npy + ns - 1) / ns);
  gm.nsplit = (npy + gm.pix_per_split - 1) / gm.pix_per_split;
  if (nsplit_out) *nsplit_out = gm.nsplit;
  const unsigned grid = (unsigned)((long)gm.nsplit * gm.ntg * gm.ntx * gm.nty);
  const unsigned threads = 64u * gm.WX * gm.WY;
  const size_t lds = (size_t)2 * gm.stage_bytes;
#define BW_LAUNCH(XC, YC, NT_)                                                                                  \
  do {                                                                                                          \
    static size_t attr = 0;                                                                                     \
    if (lds > attr) {                                                                                           \
      DV_HIP(hipFuncSetAttribute((const void*)bwgrad_kernel<XC, YC, NT_>,                                       \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
      attr = lds;                                                                                               \
    }                                                                                                           \
    hipLaunchKernelGGL((bwgrad_kernel<XC, YC, NT_>), dim3(grid), dim3(threads), lds, s, p, gm);                 \
  } while (0)
  if (xc16 && yc16) {
    BW_LAUNCH(true, true, 9);
  } else if (xc16) {
    if (nt == 9) BW_LAUNCH(true, false, 9); else if (nt == 3) BW_LAUNCH(true, false, 3); else BW_LAUNCH(true, false, 1);
  } else if (yc16) {
    if (nt == 9) BW_LAUNCH(false, true, 9); else if (nt == 3) BW_LAUNCH(false, true, 3); else BW_LAUNCH(false, true, 1);
  } else {
    if (nt == 9) BW_LAUNCH(false, false, 9); else if (nt == 3) BW_LAUNCH(false, false, 3); else BW_LAUNCH(false, false, 1);
  }
#undef BW_LAUNCH
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
