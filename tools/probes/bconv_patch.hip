// ROUND-3 EXPERIMENT, NOT PART OF THE LIBRARY (measured slower than bconv_uni_kernel on every layer: profiles/
// r03_bf16_patch_experiment.txt).  To try it: copy into debvader_amd/csrc/, add to the Makefile's SRCS,
// declare launch_bconv_patch in bf16.h and call it at the top of launch_bconv for Cin % 32 == 0 launches.
//
// Patch form of the stride-1 3x3 layers of the bf16 engine (Conv2D / Conv2DTranspose with stride 1 and their data
// gradients, model.py:81-83,128-134,137): tap reuse in LDS.
//
// bconv_uni_kernel gives a workgroup ONE output pixel x 256 stamps: every K step (tap, 32-channel chunk) gathers a
// fresh (16 + NBLK) KiB stage from L2 for 16 x NBLK MFMAs, nine times per activation byte - about 80 bytes of L2 -> LDS
// traffic per MFMA cycle of a CU against the ~28 the path delivers, which is why the family sat at 9 % matrix-pipe
// occupancy.  Here a workgroup owns a 4 x 4 PATCH of output pixels x 16 stamps x 16 NBLK output channels: per 32-channel
// chunk the 6 x 6 input patch (36 blocks of [16 stamps][32 channels] = 1 KiB each) arrives ONCE by LDS-DMA, a wave keeps
// the 3 x 6 blocks its output row needs in registers and all nine taps read them there; the chunk's weights
// (9 taps x NBLK blocks) come through LDS as well.  L2 -> LDS bytes per MFMA cycle: (36 + 9 NBLK) KiB per 576 NBLK / 4
// cycles = 32 for NBLK = 4 (2.5x less), and the input is fetched 2.25x instead of 9x.
// Work is (cout tile, 64-stamp part, spatial tile) items, each 1-4 stamp groups x Cin / 32 chunks, walked by persistent
// workgroups through a two-deep LDS ring that runs across item boundaries; the four stamp groups of a part are walked
// by one workgroup so that the d(alpha) / d(bias) sums of the fused PReLU backward (over 64 stamps, as bconv_uni_kernel
// writes them) stay in registers.  Epilogues as in bconv.hip, per output pixel: a wave-private LDS tile [16 stamps][BN],
// 16-byte row pieces to memory.
#include "common.h"
#include "bf16.h"
#include <stdlib.h>
#include <algorithm>

namespace dv {

typedef const __attribute__((address_space(1))) void* bp_gptr_t;
typedef __attribute__((address_space(3))) void* bp_lptr_t;
typedef __bf16 bp_bf16;
typedef __bf16 bp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bp_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bp_bf16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int BP_PATCH = 36 * 1024;     // bytes of one patch chunk

template <int N>
__device__ __forceinline__ void bp_store_bf(bp_bf16* dst, const float* v) {
  if constexpr (N == 1) {
    *dst = (bp_bf16)v[0];
  } else if constexpr (N == 2) {
    bp_bf16x2 o;
    o[0] = (bp_bf16)v[0]; o[1] = (bp_bf16)v[1];
    *reinterpret_cast<bp_bf16x2*>(dst) = o;
  } else {
    bp_bf16x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (bp_bf16)v[j];
    *reinterpret_cast<bp_bf16x4*>(dst) = o;
  }
}
template <int N>
__device__ __forceinline__ void bp_load_bf(const bp_bf16* src, float* v) {
  if constexpr (N == 1) {
    v[0] = (float)*src;
  } else if constexpr (N == 2) {
    const bp_bf16x2 o = *reinterpret_cast<const bp_bf16x2*>(src);
    v[0] = (float)o[0]; v[1] = (float)o[1];
  } else {
    const bp_bf16x4 o = *reinterpret_cast<const bp_bf16x4*>(src);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (float)o[j];
  }
}

struct BpGeom {
  int ntx, ntiles, NSB, nparts, ntn, nchunk, items, items_per_wg;
};
}  // namespace

// FORM 0: source pixel = out + k - 1 (Conv2D forward, Conv2DTranspose data gradient); FORM 1: source = out + 1 - k
template <int NBLK, int FORM>
__global__ __launch_bounds__(256, 1) void bconv_patch_kernel(const BConvParams p, const BpGeom gm) {
  constexpr int BN = 16 * NBLK;
  constexpr int WCH = 9 * NBLK * 1024;
  constexpr int STAGE = BP_PATCH + WCH;
  constexpr int WREG = NBLK <= 2 ? 1024 : 2048;           // wave-private epilogue tile: [16][BN] bf16, or [16][16] fp32
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* wreg = smem + 2 * STAGE + (threadIdx.x >> 6) * WREG;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.Hout, Cin = p.Cin, Cout = p.Cout, NBp = p.NBp;
  const int nchunk = gm.nchunk;

  // DMA lane roles: LDS slot `lane` of a [16 rows][4 x 16 B] block = (row lane >> 2, piece (lane & 3) ^ G4[row >> 2])
  const int drow = lane >> 2;
  const int dq = (lane & 3) ^ ((4 - (drow >> 2)) & 3);
  const unsigned char* Xb = reinterpret_cast<const unsigned char*>(p.X);
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
  const unsigned char* zlane = reinterpret_cast<const unsigned char*>(p.zero) + lane * 16;
  const unsigned a_lane = (unsigned)((drow * Cin + dq * 8) * 2);
  const unsigned b_lane = (unsigned)(((NBLK * drow) * p.Kpad + dq * 8) * 2);
  const int fr = lane & 15, fq = lane >> 4;
  const int fragoff = (fr * 4 + (fq ^ ((4 - (fr >> 2)) & 3))) * 16;
  const int c15 = lane & 15, g4 = lane >> 4;

  // ---- cursors over (item, stamp group, chunk) ----
  struct Cur {
    int item, g, c, gcount;          // item index, stamp group inside its part, chunk
    int ty0, tx0, n0, st0;           // tile origin, first output channel, first stamp of the group
    bool valid;
  };
  const int item_end = min(gm.items, (int)(blockIdx.x + 1) * gm.items_per_wg);
  auto decode = [&](Cur& u) {
    u.valid = u.item < item_end;
    if (!u.valid) return;
    // item = (cout tile * nparts + part) * ntiles + tile: consecutive items of a workgroup are neighbouring spatial
    // tiles of the same stamps and channels (their halos overlap in L2, the weight slice stays hot)
    const int tile = u.item % gm.ntiles, rest = u.item / gm.ntiles;
    const int part = rest % gm.nparts, tn = rest / gm.nparts;
    u.ty0 = (tile / gm.ntx) * 4;
    u.tx0 = (tile % gm.ntx) * 4;
    u.n0 = tn * BN;
    u.gcount = min(4, gm.NSB - part * 4);
    u.st0 = (part * 4 + u.g) * 16;
  };
  auto advance = [&](Cur& u) {
    if (++u.c < nchunk) return;
    u.c = 0;
    if (++u.g < u.gcount) {
      u.st0 += 16;
      return;
    }
    u.g = 0;
    ++u.item;
    decode(u);
  };

  // DMA of (cursor u) into ring stage stg: this wave's patch blocks k = wave, wave + 4, ... < 36 and weight blocks
  // k = wave, wave + 4, ... < 9 NBLK
  auto issue = [&](const Cur& u, int stg) {
    unsigned char* sA = smem + stg * STAGE;
    unsigned char* sB = sA + BP_PATCH;
    const size_t pixbytes = (size_t)NBp * Cin * 2;
    const unsigned char* xbase = Xb + ((size_t)u.st0 * Cin + u.c * 32) * 2;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int k = wave + 4 * i;                     // 0 .. 35
      const int py = (k * 43) >> 8, px = k - py * 6;  // k / 6 for k < 36
      const int iy = u.ty0 - 1 + py, ix = u.tx0 - 1 + px;
      const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)H;   // wave-uniform
      const void* src = ok ? (const void*)(xbase + (size_t)(iy * H + ix) * pixbytes + a_lane) : (const void*)zlane;
      __builtin_amdgcn_global_load_lds((bp_gptr_t)src, (bp_lptr_t)(sA + k * 1024), 16, 0, 0);
    }
    const unsigned char* wbase = Wb + ((size_t)u.n0 * p.Kpad + u.c * 32) * 2;
#pragma unroll
    for (int i = 0; i < (9 * NBLK + 3) / 4; ++i) {
      const int k = wave + 4 * i;
      if (k < 9 * NBLK) {                             // wave-uniform
        const int t = k / NBLK, j = k - t * NBLK;
        __builtin_amdgcn_global_load_lds((bp_gptr_t)(wbase + ((size_t)j * p.Kpad + t * Cin) * 2 + b_lane),
                                         (bp_lptr_t)(sB + k * 1024), 16, 0, 0);
      }
    }
  };

  Cur cur, nxt;
  cur.item = blockIdx.x * gm.items_per_wg;
  cur.g = 0;
  cur.c = 0;
  decode(cur);
  if (!cur.valid) return;
  nxt = cur;
  issue(cur, 0);
  advance(nxt);

  f32x4 acc[4][NBLK];
  float dal[4][NBLK], db[4][NBLK];                        // fused PReLU backward: sums over the stamps of the part
  int stg = 0;
  while (cur.valid) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (nxt.valid) issue(nxt, stg ^ 1);
    if (cur.c == 0) {
      float bias[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) bias[j] = 0.f;
      if (p.bias && (p.epi == BEPI_FWD || p.epi == BEPI_RAW32)) {
#pragma unroll
        for (int j = 0; j < NBLK; ++j) bias[j] = p.bias[cur.n0 + NBLK * c15 + j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NBLK; ++j) acc[i][j] = (f32x4){bias[j], bias[j], bias[j], bias[j]};
      if (cur.g == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < NBLK; ++j) dal[i][j] = db[i][j] = 0.f;
      }
    }
    // ---- the 3 x 6 input blocks of this wave's output row into registers, then nine taps x 4 pixels x NBLK MFMAs ----
    const unsigned char* sA = smem + stg * STAGE + fragoff;
    const unsigned char* sB = smem + stg * STAGE + BP_PATCH + fragoff;
    bp_bf16x8 ar[3][6];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int x = 0; x < 6; ++x) ar[r][x] = *reinterpret_cast<const bp_bf16x8*>(sA + ((wave + r) * 6 + x) * 1024);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      bp_bf16x8 b[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) b[j] = *reinterpret_cast<const bp_bf16x8*>(sB + (t * NBLK + j) * 1024);
      const int kh = t / 3, kw = t - kh * 3;
      const int ry = FORM == 0 ? kh : 2 - kh, rx = FORM == 0 ? kw : 2 - kw;   // patch row / column offset of this tap
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[ry][i + rx], b[j], acc[i][j], 0, 0, 0);
    }
    if (cur.c == nchunk - 1) {
      // ---- epilogue of this wave's four pixels (tile row `wave`), one [16 stamps][BN] tile at a time ----
      const int oy = cur.ty0 + wave;
      const int ch0 = cur.n0 + NBLK * c15;
      bp_bf16* wt = reinterpret_cast<bp_bf16*>(wreg);
      const bool last_group = cur.g == cur.gcount - 1;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ox = cur.tx0 + i;
        if (oy >= H || ox >= H) continue;                // wave-uniform
        const int pix = oy * H + ox;
        const size_t rb0 = (size_t)pix * NBp + cur.st0;
        // tile -> memory in 16-byte row pieces; esz = bytes per element
        auto flush = [&](void* dst, int esz) {
          const int rowb = BN * esz, total = 16 * rowb;
          unsigned char* out = reinterpret_cast<unsigned char*>(dst) + (rb0 * Cout + cur.n0) * esz;
          const size_t rstride = (size_t)Cout * esz;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int byte = (k * 64 + lane) * 16;
            if (byte >= total) break;
            const int row = byte / rowb, colb = byte - row * rowb;
            *reinterpret_cast<f32x4*>(out + row * rstride + colb) = *reinterpret_cast<const f32x4*>(wreg + byte);
          }
        };
        auto sync_tile = [&]() {
          __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): wave-private region
          __builtin_amdgcn_wave_barrier();
        };
        if (p.epi == BEPI_RAW32) {
          if constexpr (NBLK == 1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) reinterpret_cast<float*>(wreg)[(4 * g4 + r) * BN + c15] = acc[i][0][r];
            sync_tile();
            flush(p.Uf, 4);
            sync_tile();
          } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int j = 0; j < NBLK; ++j) p.Uf[(rb0 + 4 * g4 + r) * Cout + ch0 + j] = acc[i][j][r];
          }
          continue;
        }
        if (p.epi == BEPI_RAWBF || (p.epi == BEPI_FWD && p.U)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v[NBLK];
#pragma unroll
            for (int j = 0; j < NBLK; ++j) v[j] = acc[i][j][r];
            bp_store_bf<NBLK>(wt + (4 * g4 + r) * BN + NBLK * c15, v);
          }
          sync_tile();
          flush(p.U, 2);
          sync_tile();
          if (p.epi == BEPI_RAWBF) continue;
        }
        float al[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) al[j] = p.alpha[(size_t)pix * Cout + ch0 + j];
        if (p.epi == BEPI_FWD) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float a[NBLK];
#pragma unroll
            for (int j = 0; j < NBLK; ++j) a[j] = fmaxf(acc[i][j][r], 0.f) + al[j] * fminf(acc[i][j][r], 0.f);
            bp_store_bf<NBLK>(wt + (4 * g4 + r) * BN + NBLK * c15, a);
          }
          sync_tile();
          flush(p.A, 2);
          sync_tile();
          continue;
        }
        // BEPI_BWD: d(pre-activation) = d(activation) * (u > 0 ? 1 : alpha); stamp sums for d(alpha) / d(bias)
        {
          const int rowb = BN * 2, total = 16 * rowb;
          const unsigned char* uin = reinterpret_cast<const unsigned char*>(p.Uin) + (rb0 * Cout + cur.n0) * 2;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int byte = (k * 64 + lane) * 16;
            if (byte >= total) break;
            const int row = byte / rowb, colb = byte - row * rowb;
            *reinterpret_cast<f32x4*>(wreg + byte) = *reinterpret_cast<const f32x4*>(uin + (size_t)row * Cout * 2 + colb);
          }
          sync_tile();
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            bp_bf16* q = wt + (4 * g4 + r) * BN + NBLK * c15;
            float u[NBLK], du[NBLK];
            bp_load_bf<NBLK>(q, u);
#pragma unroll
            for (int j = 0; j < NBLK; ++j) {
              const float v = acc[i][j][r];
              du[j] = v * (u[j] > 0.f ? 1.f : al[j]);
              dal[i][j] += v * fminf(u[j], 0.f);
              db[i][j] += du[j];
            }
            bp_store_bf<NBLK>(q, du);
          }
          sync_tile();
          flush(p.U, 2);
          sync_tile();
          if (p.dal_part && last_group) {
            // sums over the (up to) 64 stamps of the part: the four row quarters of the accumulator tiles, then slab st0 / 64
#pragma unroll
            for (int j = 0; j < NBLK; ++j) {
              float a = dal[i][j], b2 = db[i][j];
              a += __shfl_xor(a, 16);
              a += __shfl_xor(a, 32);
              b2 += __shfl_xor(b2, 16);
              b2 += __shfl_xor(b2, 32);
              if (g4 == 0) {
                const size_t o = ((size_t)(cur.st0 >> 6) * H * H + pix) * Cout + ch0 + j;
                p.dal_part[o] = a;
                p.db_part[o] = b2;
              }
            }
          }
        }
      }
    }
    cur = nxt;
    if (nxt.valid) advance(nxt);
    stg ^= 1;
  }
}

// Returns 1 when the launch is not one this kernel takes (the caller then uses the general kernels).
int launch_bconv_patch(const BConvParams& p, hipStream_t s) {
  static const bool off = getenv("DV_BCONV_NO_PATCH") != nullptr;
  if (off || p.s != 1 || p.pb != 1 || p.Hin != p.Hout || p.Cin % 32 || p.Cout % 16 || p.Hout < 4 || (p.NBp & 15)) return 1;
  if (p.epi == BEPI_BWD && p.dal_part && (p.NBp & 63)) return 1;
  const int nblk = p.Cout % 64 == 0 ? 4 : (p.Cout % 32 == 0 ? 2 : 1);
  if (p.epi == BEPI_RAW32 && nblk != 1 && nblk != 2 && nblk != 4) return 1;
  BpGeom g;
  g.ntx = (p.Hout + 3) / 4;
  g.ntiles = g.ntx * g.ntx;
  g.NSB = p.NBp >> 4;
  g.nparts = (g.NSB + 3) / 4;
  g.ntn = p.Cout / (16 * nblk);
  g.nchunk = p.Cin / 32;
  const long items = (long)g.ntn * g.nparts * g.ntiles;
  if (items > (1L << 30)) return 1;
  g.items = (int)items;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  g.items_per_wg = (g.items + cus - 1) / cus;
  const int grid = (g.items + g.items_per_wg - 1) / g.items_per_wg;
  const size_t lds = (size_t)2 * (BP_PATCH + 9 * nblk * 1024) + 4 * (nblk <= 2 ? 1024 : 2048);
#define BP_LAUNCH(NB_, FORM_)                                                                                  \
  do {                                                                                                         \
    static bool attr = false;                                                                                  \
    if (!attr) {                                                                                               \
      DV_HIP(hipFuncSetAttribute((const void*)bconv_patch_kernel<NB_, FORM_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      attr = true;                                                                                             \
    }                                                                                                          \
    hipLaunchKernelGGL((bconv_patch_kernel<NB_, FORM_>), dim3(grid), dim3(256), lds, s, p, g);                 \
  } while (0)
  if (p.form == 0) {
    if (nblk == 4) BP_LAUNCH(4, 0); else if (nblk == 2) BP_LAUNCH(2, 0); else BP_LAUNCH(1, 0);
  } else {
    if (nblk == 4) BP_LAUNCH(4, 1); else if (nblk == 2) BP_LAUNCH(2, 1); else BP_LAUNCH(1, 1);
  }
#undef BP_LAUNCH
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
