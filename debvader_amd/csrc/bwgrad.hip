// bf16 weight gradient over stamp-inner tensors (see bf16.h):
//   out[tap][cx][cy] = sum_{gy, stamp} X[gy*s + tap - pb][stamp][cx] * Y[gy][stamp][cy]
// Kernel gradients of Conv2D (X = layer input, Y = d(pre-activation)) and Conv2DTranspose (X = d(pre-activation),
// Y = layer input; kernel layout (kh,kw,cout,cin)) of model.py:81-91,121-137.
//
// The contraction index is the STAMP: for one Y pixel and one tap both operands are [16 stamps][32 channels] blocks
// of the stamp-inner layout, and the MFMA wants 8 consecutive k per lane for a fixed channel - a transposed read,
// which ds_read_b64_tr_b16 delivers from the row-major LDS image (4 stamps x 16 channels per 16-lane group).
//
// Every WAVE is a pipeline of its own (no workgroup barrier in the loop): it owns a 32 x 32 channel tile for all nine
// taps (9 x 16 accumulator registers, v_mfma_f32_32x32x16_bf16), 16 stamps, and a range of Y rows.  It walks a row
// pixel by pixel with a SLIDING WINDOW of X blocks in its private LDS (3 rows x 8 column slots of 1 KiB): a step
// brings in only the s new columns (3*s blocks) and the next Y block by LDS-DMA, issued one step ahead of their use
// (four pixels ahead at stride 1, two at stride 2; counted vmcnt), and multiplies the Y block with the nine X blocks
// of the window - 4 KiB of DMA per 18 kFLOP x 16 instead of 10 KiB.  Taps outside the image are skipped (uniform per pixel).  The four waves of a workgroup are four
// consecutive 16-stamp chunks; they sum their tiles through LDS in a fixed order and write one fp32 slab
// [9][Cx][Cy] per workgroup, which reduce_partials adds up (deterministic).
// Operands with 16 channels (the first conv's padded input, the head's gradient) fill half of the 32-wide tile; the
// lanes of the other half read a zeroed LDS region.
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
typedef float bw_f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) bw_bf16x4* bw_l4ptr_t;

namespace {
constexpr int BW_NCS = 8;                              // column slots of the X window
constexpr int BW_NYS = 8;                              // Y slots
constexpr int BW_WAVE_LDS = (3 * BW_NCS + BW_NYS) * 1024;
constexpr int BW_RED_BYTES = 9 * 16 * 64 * 4;          // one wave's accumulators

struct BWGeom {
  int ntx, nty;        // 32-channel tiles
  int nrseg, rows_per; // Y row segments
  int nsc4;            // groups of four 16-stamp chunks
};

// Transposed LDS reads go through inline asm: for the builtin (an LDS read with no alias information) hipcc waits
// vmcnt(0) - every LDS-DMA in flight - in front of each step's first read, which serialises the pipeline; an asm
// statement is invisible to that pass, and the counted vmcnt / lgkmcnt waits below are placed by hand
// (cdna_hip_programming.md 5.7: hipcc neither counts nor orders what is inside asm).
typedef unsigned bw_u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ bw_u32x2 tr_read(unsigned lds_addr) {
  bw_u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "i"(OFF));
  return v;
}
__device__ __forceinline__ bw_bf16x8 tr_join(bw_u32x2 lo, bw_u32x2 hi) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 q = {lo[0], lo[1], hi[0], hi[1]};
  return __builtin_bit_cast(bw_bf16x8, q);
}
}  // namespace

template <bool XC16, bool YC16>
__global__ __launch_bounds__(256, 1) void bwgrad_kernel(const BWgradParams p, const BWGeom gm) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  {
    // XCD-contiguous order (blocks b and b+8 share an XCD): an XCD's workgroups then walk NEIGHBOURING rows, whose
    // X rows overlap two thirds, so the re-reads come out of that XCD's L2 instead of crossing the fabric three times
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tcy = bid % gm.nty; bid /= gm.nty;
  const int tcx = bid % gm.ntx; bid /= gm.ntx;
  const int rseg = bid % gm.nrseg;
  const int sc4 = bid / gm.nrseg;
  const int cx0 = tcx * 32, cy0 = tcy * 32;
  const int st0 = (sc4 * 4 + wave) * 16;               // first stamp of this wave's chunk
  const bool active = st0 < p.NBp;
  const int r0 = rseg * gm.rows_per, r1 = min(p.Hy, r0 + gm.rows_per);

  unsigned char* wl = smem + wave * BW_WAVE_LDS;
  unsigned char* xwin = wl;                            // [3][BW_NCS] KiB
  unsigned char* ywin = wl + 3 * BW_NCS * 1024;        // [BW_NYS] KiB

  const bw_bf16* Xb = reinterpret_cast<const bw_bf16*>(p.X);
  const bw_bf16* Yb = reinterpret_cast<const bw_bf16*>(p.Y);
  const unsigned char* zlane = reinterpret_cast<const unsigned char*>(p.zero) + lane * 16;

  // DMA of one pixel block (16 stamps): 32-channel tile of a C-channel tensor, or a whole 16-channel tensor row set.
  // The per-lane part of the source address is computed once; a block adds a scalar pixel offset.
  const unsigned char* xlane = XC16 ? (lane < 32 ? reinterpret_cast<const unsigned char*>(Xb + ((size_t)st0 + (lane >> 1)) * 16 + (lane & 1) * 8) : zlane)
                                    : reinterpret_cast<const unsigned char*>(Xb + ((size_t)st0 + (lane >> 2)) * p.Cx + cx0 + (lane & 3) * 8);
  const unsigned char* ylane = YC16 ? (lane < 32 ? reinterpret_cast<const unsigned char*>(Yb + ((size_t)st0 + (lane >> 1)) * 16 + (lane & 1) * 8) : zlane)
                                    : reinterpret_cast<const unsigned char*>(Yb + ((size_t)st0 + (lane >> 2)) * p.Cy + cy0 + (lane & 3) * 8);
  const size_t xps = (size_t)p.NBp * p.Cx * 2, yps = (size_t)p.NBp * p.Cy * 2;   // bytes per pixel
  const size_t xps_l = (XC16 && lane >= 32) ? 0 : xps, yps_l = (YC16 && lane >= 32) ? 0 : yps;
  // A block that lies outside the image (or past the end of the row) is never multiplied - its taps are skipped - so
  // its DMA needs no zero page and no select: the indices are clamped to some valid pixel and whatever lands is ignored.
  auto dma = [&](bool isx, int pix, unsigned char* dst) {
    const void* src = isx ? (const void*)(xlane + (size_t)pix * xps_l) : (const void*)(ylane + (size_t)pix * yps_l);
    __builtin_amdgcn_global_load_lds((bw_gptr_t)src, (bw_lptr_t)dst, 16, 0, 0);
  };
  // column xc of the three window rows of Y row r (image rows r*s - pb + {0,1,2})
  auto load_col = [&](int r, int xc) {
    const int xcc = min(max(xc, 0), p.Hx - 1);
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int xr = min(max(r * p.s - p.pb + kh, 0), p.Hx - 1);
      dma(true, xr * p.Hx + xcc, xwin + (kh * BW_NCS + (xc & (BW_NCS - 1))) * 1024);
    }
  };
  auto load_y = [&](int r, int w) {
    dma(false, r * p.Hy + min(w, p.Hy - 1), ywin + (w & (BW_NYS - 1)) * 1024);
  };

  // transposed-read lane roles for a [16 stamps][32 ch] block: 16-lane group g = (channel half g & 1, stamp half g >> 1)
  const int g = lane >> 4, li = lane & 15;
  const int chalf = g & 1, khalf = g >> 1;
  const int tq = li >> 2, tp = li & 3;
  // 32-ch block: row*64 + chalf*32 + tp*8; 16-ch block: rows are 32 B, the second channel half reads zeros
  const int o32 = (khalf * 8 + tq) * 64 + chalf * 32 + tp * 8;
  const int o16 = (khalf * 8 + tq) * 32 + tp * 8;
  const unsigned wl_addr = (unsigned)(size_t)(bw_lptr_t)wl;   // LDS byte address of the wave region
  // lane part of the transposed-read addresses; a block adds its (scalar) slot offset, the window row and the second
  // read are immediates
  const unsigned xl_addr = XC16 ? wl_addr + o16 : wl_addr + o32;
  const unsigned yl_addr = (YC16 ? wl_addr + o16 : wl_addr + o32) + 3 * BW_NCS * 1024;
  constexpr int XSECOND = XC16 ? 4 * 32 : 4 * 64, YSECOND = YC16 ? 4 * 32 : 4 * 64;
  const bool xzero = XC16 && chalf, yzero = YC16 && chalf;

  bw_f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  if (active) {
    // prefetch distance in pixels: the window's 8 column slots hold the 3 live columns and D*s incoming ones
    const int D = p.s == 1 ? 4 : 2;
    // loads of pixel w: its new columns (three for pixel 0) and its Y block; past the row's end the same number of
    // DMA instructions is issued from the zero page (the counted waits below rely on fixed counts)
    auto load_pixel = [&](int r, int w) {
      const int nc = w * p.s - p.pb + 2;
      if (w == 0) {
        load_col(r, nc - 2);
        load_col(r, nc - 1);
      } else if (p.s == 2) {
        load_col(r, nc - 1);
      }
      load_col(r, nc);
      load_y(r, w);
    };
    for (int r = r0; r < r1; ++r) {
      for (int w = 0; w < D; ++w) load_pixel(r, w);
      for (int w = 0; w < p.Hy; ++w) {
        load_pixel(r, w + D);
        // everything but the D youngest pixels' loads (3*s + 1 each) has landed
        if (p.s == 2)
          asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else
          asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        // all twenty transposed reads of the step are issued back to back (taps outside the image read blocks that no
        // MFMA consumes), then one wait.  One vector add per window column; rows and second reads are immediates.
        const unsigned ya = yl_addr + (w & (BW_NYS - 1)) * 1024;
        bw_u32x2 b0 = tr_read<0>(ya), b1 = tr_read<YSECOND>(ya);
        const int xc0 = w * p.s - p.pb;
        bw_u32x2 a0[9], a1[9];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const unsigned xa = xl_addr + ((xc0 + kw) & (BW_NCS - 1)) * 1024;
          a0[0 + kw] = tr_read<0>(xa);
          a1[0 + kw] = tr_read<XSECOND>(xa);
          a0[3 + kw] = tr_read<BW_NCS * 1024>(xa);
          a1[3 + kw] = tr_read<BW_NCS * 1024 + XSECOND>(xa);
          a0[6 + kw] = tr_read<2 * BW_NCS * 1024>(xa);
          a1[6 + kw] = tr_read<2 * BW_NCS * 1024 + XSECOND>(xa);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // 16-channel operands: the lanes of the absent channel half read the present half's rows (the transposed read
        // wants every lane active with an address of its own) and are zeroed here, by a select, not a branch
        if (YC16) {
          b0 = yzero ? (bw_u32x2){0u, 0u} : b0;
          b1 = yzero ? (bw_u32x2){0u, 0u} : b1;
        }
        if (XC16) {
#pragma unroll
          for (int t = 0; t < 9; ++t) {
            a0[t] = xzero ? (bw_u32x2){0u, 0u} : a0[t];
            a1[t] = xzero ? (bw_u32x2){0u, 0u} : a1[t];
          }
        }
        const bw_bf16x8 b = tr_join(b0, b1);
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          const int xr = r * p.s - p.pb + kh;
          if (xr < 0 || xr >= p.Hx) continue;
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const int xc = xc0 + kw;
            if (xc < 0 || xc >= p.Hx) continue;
            acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(a0[kh * 3 + kw], a1[kh * 3 + kw]), b,
                                                                       acc[kh * 3 + kw], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing zero-page loads, before the next row reuses the slots
    }
  }

  // ---- sum the four waves' tiles through LDS in wave order, then one slab per workgroup ----
  float* red = reinterpret_cast<float*>(smem);         // reuses wave 0's window (all DMA has been waited for)
  for (int wv = 0; wv < 4; ++wv) {
    __syncthreads();
    if (wave == wv) {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          float* q = red + (t * 16 + i) * 64 + lane;
          *q = wv == 0 ? acc[t][i] : *q + acc[t][i];
        }
    }
  }
  __syncthreads();
  float* slab = p.part + (size_t)(sc4 * gm.nrseg + rseg) * 9 * p.Cx * p.Cy;
  for (int e = tid; e < 9 * 16 * 64; e += 256) {
    const int ln = e & 63, ti = e >> 6;
    const int i = ti & 15, t = ti >> 4;
    const int cx = cx0 + (i & 3) + 8 * (i >> 2) + 4 * (ln >> 5);
    const int cy = cy0 + (ln & 31);
    if (cx < p.Cx && cy < p.Cy) slab[((size_t)t * p.Cx + cx) * p.Cy + cy] = red[e];
  }
}

int launch_bwgrad(const BWgradParams& p, hipStream_t s, int* nsplit_out) {
  const bool xc16 = p.Cx == 16, yc16 = p.Cy == 16;
  if ((!xc16 && p.Cx % 32) || (!yc16 && p.Cy % 32) || (p.NBp & 15) || p.s < 1 || p.s > 2) {
    set_error("bwgrad: channels must be 16 or multiples of 32 (Cx %d, Cy %d), stamps padded to 16 (%d), stride 1 or 2",
              p.Cx, p.Cy, p.NBp);
    return E_INVALID;
  }
  BWGeom gm;
  gm.ntx = (p.Cx + 31) / 32;
  gm.nty = (p.Cy + 31) / 32;
  gm.nsc4 = (p.NBp + 63) / 64;
  const size_t slab = (size_t)9 * p.Cx * p.Cy;
  // Y rows per workgroup: as few as the slab budget allows (more workgroups), never fewer than what keeps the slabs of
  // one launch under ~24 MB
  const long budget = 24L << 20;
  long copies = std::max<long>(gm.nsc4, budget / (long)(slab * sizeof(float)));   // (one slab per 64-stamp chunk at least)
  copies = std::min<long>(copies, (long)(p.part_capacity / slab));
  if (copies < gm.nsc4) {
    set_error("bwgrad: slab workspace too small");
    return E_STATE;
  }
  // ... and no more row segments than fill the chip once: a workgroup's fixed costs (window fill, the LDS reduction of
  // its four waves, a 9 x 32 x 32 slab written and reduced again) are paid per segment
  const long want_wgs = 256;
  const long per_seg = (long)gm.nsc4 * gm.ntx * gm.nty;
  long nseg_want = std::max<long>(1, (want_wgs + per_seg - 1) / per_seg);
  int nrseg = (int)std::min<long>(std::min<long>(p.Hy, copies / gm.nsc4), nseg_want);
  gm.rows_per = (p.Hy + nrseg - 1) / nrseg;
  gm.nrseg = (p.Hy + gm.rows_per - 1) / gm.rows_per;
  if (nsplit_out) *nsplit_out = gm.nrseg * gm.nsc4;
  const unsigned grid = (unsigned)((long)gm.nsc4 * gm.nrseg * gm.ntx * gm.nty);
  const size_t lds = (size_t)4 * BW_WAVE_LDS;
  static_assert(BW_RED_BYTES <= 4 * BW_WAVE_LDS, "reduction scratch must fit the windows");
#define BW_LAUNCH(XC, YC)                                                                                       \
  do {                                                                                                          \
    static bool attr = false;                                                                                   \
    if (!attr) {                                                                                                \
      DV_HIP(hipFuncSetAttribute((const void*)bwgrad_kernel<XC, YC>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)lds));                                                                    \
      attr = true;                                                                                              \
    }                                                                                                           \
    hipLaunchKernelGGL((bwgrad_kernel<XC, YC>), dim3(grid), dim3(256), lds, s, p, gm);                          \
  } while (0)
  if (xc16 && yc16) BW_LAUNCH(true, true);
  else if (xc16) BW_LAUNCH(true, false);
  else if (yc16) BW_LAUNCH(false, true);
  else BW_LAUNCH(false, false);
#undef BW_LAUNCH
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
