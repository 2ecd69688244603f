// bf16 gather-GEMM over stamp-inner tensors (see bf16.h): Conv2D / Conv2DTranspose forward and data gradients of the
// conv-VAE (model.py:79-98,112-137) on v_mfma_f32_16x16x32_bf16.
//
// Tile: 16 groups of 16 rows (a group = 16 consecutive stamps of one output pixel) x BN = 16*NBLK output channels;
// four waves, wave w owns groups 4w..4w+3 and all NBLK column blocks.  K walks (valid tap, 32-channel chunk); one
// chunk of one group is exactly one global_load_lds_dwordx4 (64 lanes x 16 B = 16 stamps x 32 channels), taps that
// fall outside the image read a zero page, taps that no group of the tile uses are skipped - so the stride-2
// transposed convolutions cost their real 1/2/2/4 taps per output parity without any class bookkeeping.
// LDS: three stages of (16 + NBLK) KiB, the DMA of step i+2 is issued while step i is multiplied (counted vmcnt,
// one raw barrier per step).  The LDS image of a block is [16 rows][4 x 16 B], the 16-B piece index XOR-ed with
// G4[row >> 2] (G4 = 0,3,2,1) on the SOURCE side of the DMA, which makes the ds_read_b128 fragment reads of the four
// 16-lane groups conflict free.
// Column block j of a wave holds channels n0 + NBLK*c + j (c = MFMA column), so a lane owns NBLK CONSECUTIVE
// channels of 4 stamps: epilogue loads / stores are 2*NBLK-byte vectors straight from the accumulators, and the
// stamp sums of the fused PReLU backward (d(alpha), d(bias)) are register adds plus two cross-lane steps.
#include "common.h"
#include "bf16.h"
#include <stdlib.h>
#include <type_traits>

namespace dv {

typedef const __attribute__((address_space(1))) void* bc_gptr_t;
typedef __attribute__((address_space(3))) void* bc_lptr_t;
typedef __bf16 bc_bf16;
typedef __bf16 bc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bc_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bc_bf16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int BC_GT = 16;      // groups per tile
#ifndef BC_NST_VALUE
#define BC_NST_VALUE 3
#endif
constexpr int BC_NST = BC_NST_VALUE;   // LDS stages: 3 = DMA two steps ahead; 2 = one step ahead, but twice the workgroups per CU
constexpr int BC_MAXT = 25;    // taps of the largest kernel (5 x 5)
constexpr int BC_TABW = 28;    // table row: up to 25 tap offsets, output row base, pixel index, spare

template <int N>
struct BfVec;
template <>
struct BfVec<1> { typedef bc_bf16 type; };
template <>
struct BfVec<2> { typedef bc_bf16x2 type; };
template <>
struct BfVec<4> { typedef bc_bf16x4 type; };

template <int N>
__device__ __forceinline__ void store_bf(bc_bf16* dst, const float* v) {
  if constexpr (N == 1) {
    *dst = (bc_bf16)v[0];
  } else {
    typename BfVec<N>::type o;
#pragma unroll
    for (int j = 0; j < N; ++j) o[j] = (bc_bf16)v[j];
    *reinterpret_cast<typename BfVec<N>::type*>(dst) = o;
  }
}
template <int N>
__device__ __forceinline__ void load_bf(const bc_bf16* src, float* v) {
  if constexpr (N == 1) {
    v[0] = (float)*src;
  } else {
    const typename BfVec<N>::type o = *reinterpret_cast<const typename BfVec<N>::type*>(src);
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = (float)o[j];
  }
}
template <int N>
__device__ __forceinline__ void load_f32(const float* src, float* v) {
#pragma unroll
  for (int j = 0; j < N; ++j) v[j] = src[j];
}
}  // namespace

// CINMODE 0: Cin % 32 == 0, a K step is one 32-channel chunk of one tap.
// CINMODE 1: Cin % 8 == 0 but not % 32 (first conv, data gradient of the 16-channel head, 48- / 80-... channel layers): a K step is four 16-byte pieces,
//            piece q of step i is channels (i*4+q) % (Cin/8) * 8.. of tap (i*4+q) / (Cin/8); all nine taps are walked.
template <int NBLK, int CINMODE>
__global__ __launch_bounds__(256, BC_NST == 3 ? 2 : 4) void bconv_kernel(const BConvParams p) {
  constexpr int STAGE = (BC_GT + NBLK) * 1024;
  constexpr int BN = 16 * NBLK;
  constexpr int NST = BC_NST;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  int* tab = reinterpret_cast<int*>(smem + NST * STAGE);   // [16][BC_TABW]
  int* anyv = tab + BC_GT * BC_TABW;                      // [BC_MAXT + 1]
  int* vlist = anyv + BC_MAXT + 1;                        // [BC_MAXT] valid taps in order
  const int ksz = p.ksz, ntap = ksz * ksz;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.Cout / BN;
  int bid = blockIdx.x;
  {
    // XCD-contiguous tile order: the blocks one XCD receives (every 8th) walk neighbouring pixels, whose taps
    // overlap, so the re-read of the input comes out of that XCD's L2
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / ntn, tile_n = bid - tile_m * ntn;
  const int n0 = tile_n * BN;
  const int NSB = p.NBp >> 4;
  // Output pixels are walked in 8 x 8 blocks (virtual pixel index -> block, then row / column inside it; positions past
  // the image edge are empty): neighbouring tiles of an XCD share their 3 x 3 input neighbourhoods in its L2.
  const int nbx = (p.Hout + 7) >> 3;
  const int M16 = nbx * nbx * 64 * NSB;
  auto vpixel = [&](int vp, int* oh, int* ow) {
    const int blk = vp >> 6, by = blk / nbx, bx = blk - by * nbx;
    *oh = by * 8 + ((vp >> 3) & 7);
    *ow = bx * 8 + (vp & 7);
  };
  // source pixel of tap t for output pixel (oh, ow), -1 when outside the image (or, form 1, between the strides)
  auto src_pixel = [&](int oh, int ow, int t) -> int {
    const int kh = ksz == 3 ? (t * 11) >> 5 : t / ksz, kw = t - kh * ksz;       // (t * 11) >> 5 = t / 3 for t < 9
    int ih, iw;
    bool ok = true;
    if (p.form == 0) {
      ih = oh * p.s + kh - p.pb;
      iw = ow * p.s + kw - p.pb;
    } else {
      const int nh = oh + p.pb - kh, nw = ow + p.pb - kw;
      ok = nh >= 0 && nw >= 0 && (p.s == 1 || ((nh | nw) & 1) == 0);
      ih = p.s == 1 ? nh : nh >> 1;
      iw = p.s == 1 ? nw : nw >> 1;
    }
    ok = ok && ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Hin;
    return ok ? ih * p.Hin + iw : -1;
  };

  // ---- which groups / taps: a tile that is ONE pixel (stamps padded to a multiple of 256) needs no table -------
  const bool uni = (NSB & 15) == 0;
  int u_oh = 0, u_ow = 0, u_sb0 = 0;                    // uniform tile: its pixel and first stamp block
  int nvalid = 0;
  if (uni) {
    const int tpp = NSB >> 4;
    const int vp = tile_m / tpp;
    u_sb0 = (tile_m - vp * tpp) * 16;
    vpixel(vp, &u_oh, &u_ow);
    if (u_oh >= p.Hout || u_ow >= p.Hout) return;      // padding of the 8 x 8 blocks
    if (tid == 0) {
      int n = 0;
      for (int t = 0; t < ntap; ++t)
        if (src_pixel(u_oh, u_ow, t) >= 0) vlist[n++] = t;
      anyv[BC_MAXT] = n;
    }
    __syncthreads();
    nvalid = anyv[BC_MAXT];
  } else {
    for (int e = tid; e < BC_GT * ntap; e += 256) {
      const int g = e / ntap, t = e - g * ntap;
      const int m16 = tile_m * BC_GT + g;
      int off = -1;
      const int vp = m16 / NSB, sb = m16 - vp * NSB;
      int oh, ow;
      vpixel(vp, &oh, &ow);
      if (m16 < M16 && oh < p.Hout && ow < p.Hout) {
        const int sp = src_pixel(oh, ow, t);
        // offset of the group's [16][Cin] block in units of 8 elements (16 bytes)
        if (sp >= 0) off = (sp * p.NBp + sb * 16) * (p.Cin >> 3);
        if (t == 0) {
          tab[g * BC_TABW + BC_MAXT] = (oh * p.Hout + ow) * p.NBp + sb * 16;
          tab[g * BC_TABW + BC_MAXT + 1] = oh * p.Hout + ow;
        }
      } else if (t == 0) {
        tab[g * BC_TABW + BC_MAXT] = -1;
        tab[g * BC_TABW + BC_MAXT + 1] = 0;
      }
      tab[g * BC_TABW + t] = off;
    }
    __syncthreads();
    if (tid <= ntap) {                                  // taps 0 .. ntap - 1; ntap: any group inside the image
      int a = 0;
#pragma unroll
      for (int g = 0; g < BC_GT; ++g) a |= tab[g * BC_TABW + (tid < ntap ? tid : BC_MAXT)] >= 0 ? 1 : 0;
      anyv[tid < ntap ? tid : BC_MAXT] = a;
    }
    __syncthreads();
    if (!anyv[BC_MAXT]) return;
    __syncthreads();
    if (tid == 0) {
      int n = 0;
      for (int t = 0; t < ntap; ++t)
        if (anyv[t]) vlist[n++] = t;
      anyv[BC_MAXT] = n;
    }
    __syncthreads();
    nvalid = anyv[BC_MAXT];
  }
  nvalid = __builtin_amdgcn_readfirstlane(nvalid);
  // block offset (units of 16 bytes) of group g for tap t, -1 outside; output row base and pixel of group g
  auto group_off = [&](int g, int t) -> int {
    if (uni) {
      const int sp = src_pixel(u_oh, u_ow, t);
      return sp >= 0 ? (sp * p.NBp + (u_sb0 + g) * 16) * (p.Cin >> 3) : -1;
    }
    return tab[g * BC_TABW + t];
  };
  auto group_rb = [&](int g) -> int { return uni ? (u_oh * p.Hout + u_ow) * p.NBp + (u_sb0 + g) * 16 : tab[g * BC_TABW + BC_MAXT]; };
  auto group_pix = [&](int g) -> int { return uni ? u_oh * p.Hout + u_ow : tab[g * BC_TABW + BC_MAXT + 1]; };

  const int cpt = p.Cin >> 5;         // chunks per tap (CINMODE 0)
  const int ppt = p.Cin >> 3;         // 16-byte pieces per tap (CINMODE 1: 1 or 2)
  const int nsteps = CINMODE == 0 ? nvalid * cpt : (ntap * ppt + 3) >> 2;

  // DMA lane roles: LDS slot `lane` of a block = (row, piece ^ G4[row>>2])
  const int drow = lane >> 2;
  const int dq = (lane & 3) ^ ((4 - (drow >> 2)) & 3);
  const bc_bf16* Xb = reinterpret_cast<const bc_bf16*>(p.X);
  const bc_bf16* Wb = reinterpret_cast<const bc_bf16*>(p.W);
  const unsigned char* zlane = reinterpret_cast<const unsigned char*>(p.zero) + lane * 16;
  const int jb = wave % NBLK;   // B block this wave loads (duplicates when NBLK < 4: same bytes, same slot)
  const bc_bf16* wrow = Wb + (size_t)(n0 + NBLK * drow + jb) * p.Kpad + dq * 8;
  const int arow = drow * p.Cin;

  // fused PReLU backward: this wave's [64 rows][BN] tile of the target layer's pre-activation is fetched now, 16 bytes
  // per lane, and parked in registers until the epilogue
  constexpr int NROWCH = BN / 8;                        // 16-byte pieces per lane that make up [64][BN] bf16
  f32x4 uin[NROWCH];
  if (p.epi == BEPI_BWD) {
#pragma unroll
    for (int k = 0; k < NROWCH; ++k) {
      const int byte = (k * 64 + lane) * 16;
      const int row = byte / (BN * 2), colb = byte - row * (BN * 2);
      const int rb = group_rb(wave * 4 + (row >> 4));
      uin[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (rb >= 0)
        uin[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(p.Uin) +
                                                 ((size_t)(rb + (row & 15)) * p.Cout + n0) * 2 + colb);
    }
  }

  // DMA issue state (CINMODE 0): steps walk (valid tap, chunk) in order, so the four group sources of the current
  // tap are worked out once per tap, not once per step
  int is_ti = -1, is_cc = 0, is_tap = 0;
  const unsigned char* gsrc[4] = {zlane, zlane, zlane, zlane};
  bool gok[4] = {false, false, false, false};
  auto issue = [&](int step, int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + BC_GT * 1024;
    if constexpr (CINMODE == 0) {
      if (is_ti < 0 || is_cc + 1 == cpt) {
        ++is_ti;
        is_cc = 0;
        is_tap = __builtin_amdgcn_readfirstlane(vlist[is_ti]);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
          const int off = group_off(wave * 4 + gi, is_tap);
          gok[gi] = off >= 0;
          gsrc[gi] = reinterpret_cast<const unsigned char*>(Xb + ((size_t)(off >= 0 ? off : 0) * 8 + arow + dq * 8));
        }
      } else {
        ++is_cc;
      }
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const void* src = gok[gi] ? (const void*)(gsrc[gi] + is_cc * 64) : (const void*)zlane;
        __builtin_amdgcn_global_load_lds((bc_gptr_t)src, (bc_lptr_t)(sA + (wave * 4 + gi) * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds((bc_gptr_t)(wrow + is_tap * p.Cin + is_cc * 32), (bc_lptr_t)(sB + jb * 1024), 16, 0, 0);
    } else {
      const int piece = step * 4 + dq;
      const int tap = ppt == 1 ? piece : ppt == 2 ? piece >> 1 : piece / ppt;
      const int sub = piece - tap * ppt;
      const bool pv = piece < ntap * ppt;
      const int lo = arow + sub * 8;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int g = wave * 4 + gi;
        const int off = pv ? group_off(g, tap) : -1;
        const void* src = off >= 0 ? (const void*)(Xb + ((size_t)off * 8 + lo)) : (const void*)zlane;
        __builtin_amdgcn_global_load_lds((bc_gptr_t)src, (bc_lptr_t)(sA + g * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds((bc_gptr_t)(wrow + step * 32), (bc_lptr_t)(sB + jb * 1024), 16, 0, 0);
    }
  };

  // fragment read offset: lane l reads (row l & 15, piece l >> 4)
  const int fr = lane & 15, fq = lane >> 4;
  const int fragoff = (fr * 4 + (fq ^ ((4 - (fr >> 2)) & 3))) * 16;

  f32x4 acc[4][NBLK];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi)
#pragma unroll
    for (int j = 0; j < NBLK; ++j) acc[gi][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nsteps > 0) issue(0, 0);
  if (NST == 3 && nsteps > 1) issue(1, 1);
  int buf = 0;
  for (int i = 0; i < nsteps; ++i) {
    // five DMA instructions per wave and stage: all but the youngest stage have landed
    if (NST == 3 && i + 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (NST == 3) {
      if (i + 2 < nsteps) issue(i + 2, buf >= 1 ? buf - 1 : 2);   // (buf + 2) % 3: the stage read in step i - 1
    } else {
      if (i + 1 < nsteps) issue(i + 1, buf ^ 1);
    }
    const unsigned char* sA = smem + buf * STAGE + wave * 4096 + fragoff;
    const unsigned char* sB = smem + buf * STAGE + BC_GT * 1024 + fragoff;
    bc_bf16x8 a[4], b[NBLK];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) a[gi] = *reinterpret_cast<const bc_bf16x8*>(sA + gi * 1024);
#pragma unroll
    for (int j = 0; j < NBLK; ++j) b[j] = *reinterpret_cast<const bc_bf16x8*>(sB + j * 1024);
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
      for (int j = 0; j < NBLK; ++j)
        acc[gi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[gi], b[j], acc[gi][j], 0, 0, 0);
    buf = NST == 3 ? (buf == 2 ? 0 : buf + 1) : (buf ^ 1);
  }

  // ---- epilogue ---------------------------------------------------------------------------------------------------
  // Lane (c, g4) owns channels n0 + NBLK*c .. +NBLK-1 of stamps 4*g4 .. 4*g4+3 of each of its wave's 4 groups.  The
  // values go through a per-wave LDS tile [64 rows][BN] (the stage buffers are free once every wave has left the
  // loop) so that global memory sees 16-byte accesses of whole rows: the 2*NBLK-byte stores straight from the
  // accumulators were the largest single cost of the kernel (store-issue bound, 32 instructions per wave and tensor).
  __builtin_amdgcn_s_barrier();
  constexpr int WREG = 64 * BN * (NBLK == 1 ? 4 : 2);
  unsigned char* wreg = smem + wave * WREG;
  const int c = lane & 15, g4 = lane >> 4;
  const int ch0 = n0 + NBLK * c;
  int rbs[4];
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) rbs[gi] = group_rb(wave * 4 + gi);
  // rows of the tile -> global memory, 16 bytes per lane; esz = bytes per element
  auto flush = [&](void* dst, int esz) {
    const int rowb = BN * esz;
    const int npc = 64 * rowb / 1024;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k >= npc) break;
      const int byte = (k * 64 + lane) * 16;
      const int row = byte / rowb, colb = byte - row * rowb;
      const int rb = rbs[0] < 0 && rbs[1] < 0 && rbs[2] < 0 && rbs[3] < 0 ? -1
                     : ((row >> 4) == 0 ? rbs[0] : (row >> 4) == 1 ? rbs[1] : (row >> 4) == 2 ? rbs[2] : rbs[3]);
      const f32x4 v = *reinterpret_cast<const f32x4*>(wreg + byte);
      if (rb >= 0)
        *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(dst) + ((size_t)(rb + (row & 15)) * p.Cout + n0) * esz + colb) = v;
    }
  };
  float bias[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) bias[j] = 0.f;
  if (p.bias && (p.epi == BEPI_FWD || p.epi == BEPI_RAW32)) load_f32<NBLK>(p.bias + ch0, bias);

  if (p.epi == BEPI_RAW32) {
    if constexpr (NBLK == 1) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          reinterpret_cast<float*>(wreg)[(gi * 16 + 4 * g4 + r) * BN + c] = acc[gi][0][r] + bias[0];
      flush(p.Uf, 4);
    } else {
      // fp32 output is only produced by the 16-channel head: wider tiles store straight from the accumulators
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        if (rbs[gi] < 0) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < NBLK; ++j)
            p.Uf[((size_t)rbs[gi] + 4 * g4 + r) * p.Cout + ch0 + j] = acc[gi][j][r] + bias[j];
      }
    }
    return;
  }
  bc_bf16* wt = reinterpret_cast<bc_bf16*>(wreg);
  if (p.epi == BEPI_RAWBF) {
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) v[j] = acc[gi][j][r];
        store_bf<NBLK>(wt + (gi * 16 + 4 * g4 + r) * BN + NBLK * c, v);
      }
    flush(p.U, 2);
    return;
  }
  if (p.epi == BEPI_FWD) {
    float al[4][NBLK];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi) {
#pragma unroll
      for (int j = 0; j < NBLK; ++j) al[gi][j] = 0.f;
      if (rbs[gi] >= 0) load_f32<NBLK>(p.alpha + (size_t)group_pix(wave * 4 + gi) * p.Cout + ch0, al[gi]);
    }
    if (p.U) {
#pragma unroll
      for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v[NBLK];
#pragma unroll
          for (int j = 0; j < NBLK; ++j) v[j] = acc[gi][j][r] + bias[j];
          store_bf<NBLK>(wt + (gi * 16 + 4 * g4 + r) * BN + NBLK * c, v);
        }
      flush(p.U, 2);
    }
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
          const float v = acc[gi][j][r] + bias[j];
          a[j] = v > 0.f ? v : al[gi][j] * v;
        }
        store_bf<NBLK>(wt + (gi * 16 + 4 * g4 + r) * BN + NBLK * c, a);
      }
    flush(p.A, 2);
    return;
  }
  // ---- BEPI_BWD: d(pre-activation) = d(activation) * (u > 0 ? 1 : alpha), stamp sums for d(alpha) / d(bias) ----
#pragma unroll
  for (int k = 0; k < NROWCH; ++k) *reinterpret_cast<f32x4*>(wreg + (k * 64 + lane) * 16) = uin[k];
  float dal[NBLK], db[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) dal[j] = db[j] = 0.f;
  int pix_w = 0, rb_w = -1;
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    if (rbs[gi] < 0) continue;
    pix_w = group_pix(wave * 4 + gi);
    rb_w = rbs[gi];
    float al[NBLK];
    load_f32<NBLK>(p.alpha + (size_t)pix_w * p.Cout + ch0, al);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bc_bf16* q = wt + (gi * 16 + 4 * g4 + r) * BN + NBLK * c;
      float u[NBLK], du[NBLK];
      load_bf<NBLK>(q, u);
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        const float v = acc[gi][j][r];
        du[j] = v * (u[j] > 0.f ? 1.f : al[j]);
        dal[j] += v * fminf(u[j], 0.f);
        db[j] += du[j];
      }
      store_bf<NBLK>(q, du);
    }
  }
  flush(p.U, 2);
  if (p.dal_part && rb_w >= 0) {
    // the wave's four groups are 64 consecutive stamps of ONE pixel (NBp % 64 == 0): slab = stamp block / 4
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
      dal[j] += __shfl_xor(dal[j], 16);
      dal[j] += __shfl_xor(dal[j], 32);
      db[j] += __shfl_xor(db[j], 16);
      db[j] += __shfl_xor(db[j], 32);
    }
    if (g4 == 0) {
      const int part = (rb_w - pix_w * p.NBp) >> 6;
      const size_t o = ((size_t)part * p.Hout * p.Hout + pix_w) * p.Cout + ch0;
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        p.dal_part[o + j] = dal[j];
        p.db_part[o + j] = db[j];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The form the training batches take: stamps padded to a multiple of 256, so a tile is ONE output pixel x 256 stamps.
// Everything that the general kernel looks up per group is then uniform over the workgroup (one pixel, nine taps that
// are inside the image or not) and lives in scalar registers; the general kernel's dual paths cost it more
// instructions than MFMAs (SQ_ACTIVE_INST_ANY 45 %, SQ_VALU_MFMA_BUSY 5-10 %).  Differences:
//   * DMA sources are  uniform base (scalar) + 32-bit lane offset: no per-instruction vector address arithmetic,
//     no zero-page select (CINMODE 0: only taps inside the image are walked);
//   * the accumulators start at the bias; alpha is one row for the whole tile;
//   * the wave's 64 output rows are consecutive in memory: the LDS tile goes out as plain 16-byte row pieces.
// GT = 16-stamp groups per tile (16, 8 or 4: 256, 128 or 64 stamps per workgroup; a wave owns GW = GT/4 of them).  The
// deep layers (16 x 16 pixels and below) have only a few hundred 256-stamp tiles, each walking 36-72 K steps at the
// latency of its own three-stage ring; smaller stamp tiles put 4-8 independent pipelines on a CU.
// CH = 32-channel chunks per K step.  With Cin >= 64 a row of an operand block is 64 bytes out of a >= 128-byte row
// of the tensor, i.e. HALF of a 128-byte cache line per row: an LDS-DMA instruction then costs the L2 / Infinity Cache /
// HBM path 16 line requests for 1 KiB of use, and measured operand bandwidth halves (tools/probes/dma_probe.hip:
// 17 vs 33 TB/s out of L2, 4.2 vs 8.4 out of the Infinity Cache, 3.5 vs 7.0 out of HBM).  CH = 2 issues the two halves
// of the same lines back to back (chunks 2i and 2i+1 of a tap), so the second request merges with the first in L1.
// NS = LDS stages of the ring (the DMA of step i + NS - 1 is issued while step i is multiplied).  3 is what ships; a
// six-stage ring on 256-stamp tiles for the deep layers was measured in round 4 (one workgroup per CU walking 36-72 K
// steps with five steps in flight) and changed nothing on the 8 x 8 layers (35.9 -> 38.3 us) while the 16 x 16 layers
// lost 40 % to the single resident workgroup: the tiles are not paced by the latency of their DMA.
// workgroups per CU the uniform kernel is compiled for (register budget): what its LDS footprint admits
constexpr int bconv_uni_occupancy(int nblk, int gt, int ch, int ns) {
  if (ns > 3) return 1;
  if (ns == 2) {
    if (gt == 16) return nblk <= 2 ? 4 : 3;
    if (gt == 8) return ch == 2 ? 3 : 4;
    return 4;
  }
  return gt == 16 ? (ch == 1 ? 2 : 1) : (ch == 1 ? 4 : 2);
}

template <int NBLK, int CINMODE, int GT, int CH = 1, int NS = 3>
__global__ __launch_bounds__(256, bconv_uni_occupancy(NBLK, GT, CH, NS)) void bconv_uni_kernel(const BConvParams p) {
#ifdef DV_DEBUG_EXPORTS
  if (p.exp == 6) return;                                 // (measurement: empty workgroups - launch and dispatch only)
#endif
  constexpr int GW = GT / 4;                              // groups per wave
  constexpr int RW = GW * 16;                             // output rows (stamps) per wave
  constexpr int SUB = (GT + NBLK) * 1024;                 // one chunk of a stage: GT A blocks, NBLK B blocks
  constexpr int STAGE = CH * SUB;
  static_assert(CH == 1 || CINMODE == 0, "paired chunks only for Cin % 32 == 0");
  constexpr int BN = 16 * NBLK;
  static_assert(RW * BN * 2 >= 1024, "a wave's bf16 tile must be at least one 1-KiB row piece");
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  int* stab = reinterpret_cast<int*>(smem + NS * STAGE);   // [0] valid taps, [1..25] their ids, [26..50] their source pixels
  const int ksz = p.ksz, ntap = ksz * ksz;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.Cout / BN;
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / ntn, tile_n = bid - tile_m * ntn;
  const int n0 = tile_n * BN;
  const int tpp = p.NBp / (GT * 16);                    // tiles per pixel
  const int vp = tile_m / tpp;
  const int st0 = (tile_m - vp * tpp) * (GT * 16);      // first stamp of the tile
  const int nbx = (p.Hout + 7) >> 3;
  const int blk = vp >> 6, by = blk / nbx, bx = blk - by * nbx;
  const int oh = by * 8 + ((vp >> 3) & 7), ow = bx * 8 + (vp & 7);
  if (oh >= p.Hout || ow >= p.Hout) return;             // padding of the 8 x 8 pixel blocks
  const int pix = oh * p.Hout + ow;

  auto src_pixel = [&](int t) -> int {                   // t may be a lane value (CINMODE 1)
    const int kh = ksz == 3 ? (t * 11) >> 5 : t / ksz, kw = t - kh * ksz;
    int ih, iw;
    bool ok = true;
    if (p.form == 0) {
      ih = oh * p.s + kh - p.pb;
      iw = ow * p.s + kw - p.pb;
    } else {
      const int nh = oh + p.pb - kh, nw = ow + p.pb - kw;
      ok = nh >= 0 && nw >= 0 && (p.s == 1 || ((nh | nw) & 1) == 0);
      ih = p.s == 1 ? nh : nh >> 1;
      iw = p.s == 1 ? nw : nw >> 1;
    }
    ok = ok && ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Hin;
    return ok ? ih * p.Hin + iw : -1;
  };
  if (CINMODE == 0) {
    if (tid == 0) {
      int n = 0;
      if (ksz == 3) {                                     // the form every BASELINE configuration takes: fully unrolled
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int sp = src_pixel(t);
          if (sp >= 0) {
            stab[1 + n] = t;
            stab[1 + BC_MAXT + n] = sp;
            ++n;
          }
        }
      } else {
        for (int t = 0; t < ntap; ++t) {
          const int sp = src_pixel(t);
          if (sp >= 0) {
            stab[1 + n] = t;
            stab[1 + BC_MAXT + n] = sp;
            ++n;
          }
        }
      }
      stab[0] = n;
    }
    __syncthreads();
  }
  const int cpt = p.Cin >> 5, ppt = p.Cin >> 3;
  const int nvalid = CINMODE == 0 ? __builtin_amdgcn_readfirstlane(stab[0]) : ntap;
  const int nsteps_all = CINMODE == 0 ? nvalid * (cpt / CH) : (ntap * ppt + 3) >> 2;      // (CH == 2: cpt is even)
#ifdef DV_DEBUG_EXPORTS
  const int nsteps = (p.exp == 1 || p.exp == 5) ? min(nsteps_all, 1) : nsteps_all;   // (measurement)
#else
  const int nsteps = nsteps_all;
#endif

  const int drow = lane >> 2;
  const int dq = (lane & 3) ^ ((4 - (drow >> 2)) & 3);
  const unsigned char* Xb = reinterpret_cast<const unsigned char*>(p.X);
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
  const unsigned char* zlane = reinterpret_cast<const unsigned char*>(p.zero) + lane * 16;
  const int jb = wave % NBLK;
  // lane parts (bytes) of the DMA sources; the uniform parts are added per instruction as scalars
  const unsigned a_lane = (unsigned)((((st0 + wave * RW + drow) * p.Cin) + dq * 8) * 2);
  const unsigned b_lane = (unsigned)(((NBLK * drow) * p.Kpad + dq * 8) * 2);
  const size_t pixbytes = (size_t)p.NBp * p.Cin * 2;     // one pixel of the input tensor
  const unsigned gstride = (unsigned)(16 * p.Cin * 2);   // one 16-stamp group
  const unsigned char* wbase = Wb + (size_t)(n0 + jb) * p.Kpad * 2;

  // CINMODE 1: the lane source of step i (piece i * 4 + dq of the (tap, 8-channel piece) list); null = outside the image
  auto piece_src = [&](int i) -> const unsigned char* {
    const int piece = i * 4 + dq;
    const int tap = ppt == 1 ? piece : ppt == 2 ? piece >> 1 : piece / ppt;
    const int sub = piece - tap * ppt;
    const int sp = piece < ntap * ppt ? src_pixel(tap) : -1;
    return sp >= 0 ? Xb + (size_t)sp * pixbytes + (unsigned)((((st0 + wave * RW + drow) * p.Cin) + sub * 8) * 2) : nullptr;
  };
  // at most five steps (3 x 3 kernels on 8 or 16 channels): worked out once; longer walks work them out at issue time
  const bool ptab = nsteps <= 5;
  const unsigned char* psrc[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if constexpr (CINMODE == 1) {
    if (ptab) {
#pragma unroll
      for (int i = 0; i < 5; ++i) psrc[i] = piece_src(i);
    }
  }

  int is_ti = -1, is_cc = 0, is_tap = 0;
  const unsigned char* tsrc = Xb;                          // uniform: input pixel of the current tap
  auto issue = [&](int step, int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + GT * 1024;
    if constexpr (CINMODE == 0) {
      if (is_ti < 0 || is_cc + CH == cpt) {
        ++is_ti;
        is_cc = 0;
        is_tap = __builtin_amdgcn_readfirstlane(stab[1 + is_ti]);
        tsrc = Xb + (size_t)__builtin_amdgcn_readfirstlane(stab[1 + BC_MAXT + is_ti]) * pixbytes;
      } else {
        is_cc += CH;
      }
      const unsigned char* cs = tsrc + is_cc * 64;
#pragma unroll
      for (int gi = 0; gi < GW; ++gi)
#pragma unroll
        for (int h = 0; h < CH; ++h)       // both halves of the same cache lines back to back
          __builtin_amdgcn_global_load_lds((bc_gptr_t)(cs + h * 64 + gi * gstride + a_lane),
                                           (bc_lptr_t)(sA + h * SUB + (wave * GW + gi) * 1024), 16, 0, 0);
#pragma unroll
      for (int h = 0; h < CH; ++h)
        __builtin_amdgcn_global_load_lds((bc_gptr_t)(wbase + (is_tap * p.Cin + (is_cc + h) * 32) * 2 + b_lane),
                                         (bc_lptr_t)(sB + h * SUB + jb * 1024), 16, 0, 0);
    } else {
      const unsigned char* ps = nullptr;
      if (ptab) {
#pragma unroll
        for (int i = 0; i < 5; ++i)
          if (i == step) ps = psrc[i];
      } else {
        ps = piece_src(step);
      }
#pragma unroll
      for (int gi = 0; gi < GW; ++gi) {
        const void* src = ps ? (const void*)(ps + gi * gstride) : (const void*)zlane;
        __builtin_amdgcn_global_load_lds((bc_gptr_t)src, (bc_lptr_t)(sA + (wave * GW + gi) * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds((bc_gptr_t)(wbase + step * 64 + b_lane), (bc_lptr_t)(sB + jb * 1024), 16, 0, 0);
    }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int fragoff = (fr * 4 + (fq ^ ((4 - (fr >> 2)) & 3))) * 16;
  const int c = lane & 15, g4 = lane >> 4;
  const int ch0 = n0 + NBLK * c;

  // fused PReLU backward: the wave's [64 rows][BN] tile of the pre-activation, parked in registers
  constexpr int NPC = RW * BN / 512;                      // 16-byte pieces per lane of the wave's [RW][BN] bf16 tile
  const size_t rb0 = (size_t)pix * p.NBp + st0 + wave * RW;   // first output row of this wave
  f32x4 uin[NPC];
  if (p.epi == BEPI_BWD) {
#pragma unroll
    for (int k = 0; k < NPC; ++k) {
      const int byte = (k * 64 + lane) * 16;
      const int row = byte / (BN * 2), colb = byte - row * (BN * 2);
      uin[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(p.Uin) +
                                               ((rb0 + row) * p.Cout + n0) * 2 + colb);
    }
  }
  float bias[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) bias[j] = 0.f;
  if (p.bias && (p.epi == BEPI_FWD || p.epi == BEPI_RAW32)) load_f32<NBLK>(p.bias + ch0, bias);
  f32x4 acc[GW][NBLK];
#pragma unroll
  for (int gi = 0; gi < GW; ++gi)
#pragma unroll
    for (int j = 0; j < NBLK; ++j) acc[gi][j] = (f32x4){bias[j], bias[j], bias[j], bias[j]};

  constexpr int PER_STAGE = CH * (GW + 1);                // DMA instructions per wave and stage
#pragma unroll
  for (int k = 0; k < NS - 1; ++k)
    if (k < nsteps) issue(k, k);
  int buf = 0;
  for (int i = 0; i < nsteps; ++i) {
    // stage i must have landed; the stages issued after it (at most NS - 2 of them) may still be in flight
    const int ahead = min(NS - 2, nsteps - 1 - i);
    if (NS == 3) {
      if (ahead >= 1)
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE) : "memory");
      else
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (ahead >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PER_STAGE) : "memory");
      else if (ahead == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PER_STAGE) : "memory");
      else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER_STAGE) : "memory");
      else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_STAGE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (i + NS - 1 < nsteps) issue(i + NS - 1, buf >= 1 ? buf - 1 : NS - 1);   // the stage read in step i - 1
#pragma unroll
    for (int h = 0; h < CH; ++h) {
      const unsigned char* sA = smem + buf * STAGE + h * SUB + wave * (GW * 1024) + fragoff;
      const unsigned char* sB = smem + buf * STAGE + h * SUB + GT * 1024 + fragoff;
      bc_bf16x8 a[GW], b[NBLK];
#pragma unroll
      for (int gi = 0; gi < GW; ++gi) a[gi] = *reinterpret_cast<const bc_bf16x8*>(sA + gi * 1024);
#pragma unroll
      for (int j = 0; j < NBLK; ++j) b[j] = *reinterpret_cast<const bc_bf16x8*>(sB + j * 1024);
#pragma unroll
      for (int gi = 0; gi < GW; ++gi)
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
          acc[gi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[gi], b[j], acc[gi][j], 0, 0, 0);
    }
    buf = buf == NS - 1 ? 0 : buf + 1;
  }

#ifdef DV_DEBUG_EXPORTS
  if (p.exp >= 4) return;                                 // (measurement: no epilogue at all; 5: after a one-step loop)
#endif
  // ---- epilogue: per-wave LDS tile [RW rows][BN]; the rows are consecutive rows of the output tensor ----
  __builtin_amdgcn_s_barrier();
  constexpr int WREG = RW * BN * (NBLK == 1 ? 4 : 2);
  unsigned char* wreg = smem + wave * WREG;
  auto flush = [&](void* dst, int esz) {
#ifdef DV_DEBUG_EXPORTS
    if (p.exp == 3) return;                               // (measurement: no epilogue stores)
#endif
    const int rowb = BN * esz;
    unsigned char* out = reinterpret_cast<unsigned char*>(dst) + (rb0 * p.Cout + n0) * esz;
    const size_t rstride = (size_t)p.Cout * esz;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k >= RW * rowb / 1024) break;
      const int byte = (k * 64 + lane) * 16;
      const int row = byte / rowb, colb = byte - row * rowb;
      *reinterpret_cast<f32x4*>(out + row * rstride + colb) = *reinterpret_cast<const f32x4*>(wreg + byte);
    }
  };
  if (p.epi == BEPI_RAW32) {
    if constexpr (NBLK == 1) {
#pragma unroll
      for (int gi = 0; gi < GW; ++gi)
#pragma unroll
        for (int r = 0; r < 4; ++r) reinterpret_cast<float*>(wreg)[(gi * 16 + 4 * g4 + r) * BN + c] = acc[gi][0][r];
      flush(p.Uf, 4);
    } else {
#pragma unroll
      for (int gi = 0; gi < GW; ++gi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < NBLK; ++j) p.Uf[(rb0 + gi * 16 + 4 * g4 + r) * p.Cout + ch0 + j] = acc[gi][j][r];
    }
    return;
  }
  bc_bf16* wt = reinterpret_cast<bc_bf16*>(wreg);
  if (p.epi == BEPI_RAWBF || (p.epi == BEPI_FWD && p.U)) {
#pragma unroll
    for (int gi = 0; gi < GW; ++gi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) v[j] = acc[gi][j][r];
        store_bf<NBLK>(wt + (gi * 16 + 4 * g4 + r) * BN + NBLK * c, v);
      }
    flush(p.U, 2);
    if (p.epi == BEPI_RAWBF) return;
  }
  float al[NBLK];
  load_f32<NBLK>(p.alpha + (size_t)pix * p.Cout + ch0, al);
  if (p.epi == BEPI_FWD) {
#pragma unroll
    for (int gi = 0; gi < GW; ++gi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float a[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) a[j] = fmaxf(acc[gi][j][r], 0.f) + al[j] * fminf(acc[gi][j][r], 0.f);
        store_bf<NBLK>(wt + (gi * 16 + 4 * g4 + r) * BN + NBLK * c, a);
      }
    flush(p.A, 2);
    return;
  }
  // BEPI_BWD
#pragma unroll
  for (int k = 0; k < NPC; ++k) *reinterpret_cast<f32x4*>(wreg + (k * 64 + lane) * 16) = uin[k];
  float dal[NBLK], db[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) dal[j] = db[j] = 0.f;
#pragma unroll
  for (int gi = 0; gi < GW; ++gi)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bc_bf16* q = wt + (gi * 16 + 4 * g4 + r) * BN + NBLK * c;
      float u[NBLK], du[NBLK];
      load_bf<NBLK>(q, u);
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        const float v = acc[gi][j][r];
        du[j] = v * (u[j] > 0.f ? 1.f : al[j]);
        dal[j] += v * fminf(u[j], 0.f);
        db[j] += du[j];
      }
      store_bf<NBLK>(q, du);
    }
  flush(p.U, 2);
  if (p.dal_part) {
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
      dal[j] += __shfl_xor(dal[j], 16);
      dal[j] += __shfl_xor(dal[j], 32);
      db[j] += __shfl_xor(db[j], 16);
      db[j] += __shfl_xor(db[j], 32);
    }
    if constexpr (GW < 4) {
      // a partial slab row covers 64 stamps = 4 / GW waves: sum them through LDS in wave order (p.dal_part is the same for
      // every wave, so all of them reach the barrier)
      float* pbuf = reinterpret_cast<float*>(smem + NS * STAGE + 1024);   // [4 waves][2][BN]
      if (g4 == 0) {
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
          pbuf[(wave * 2 + 0) * BN + NBLK * c + j] = dal[j];
          pbuf[(wave * 2 + 1) * BN + NBLK * c + j] = db[j];
        }
      }
      __syncthreads();
      constexpr int WPP = 4 / GW;                           // waves per partial row
      if (wave % WPP == 0 && g4 == 0) {
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
          float a = 0.f, b = 0.f;
#pragma unroll
          for (int w = 0; w < WPP; ++w) {
            a += pbuf[((wave + w) * 2 + 0) * BN + NBLK * c + j];
            b += pbuf[((wave + w) * 2 + 1) * BN + NBLK * c + j];
          }
          dal[j] = a;
          db[j] = b;
        }
      }
      if (wave % WPP != 0) return;
    }
    if (g4 == 0) {
      const int part = (st0 + wave * RW) >> 6;
      const size_t o = ((size_t)part * p.Hout * p.Hout + pix) * p.Cout + ch0;
#pragma unroll
      for (int j = 0; j < NBLK; ++j) {
        p.dal_part[o + j] = dal[j];
        p.db_part[o + j] = db[j];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------
// Row-strip form for the stride-1 3 x 3 layers (round 4): register-blocked, with halo reuse.
//
// What bounded bconv_uni_kernel (round-4 measurements): a workgroup there is ONE output pixel, so every input
// block (16 stamps x 32 channels, 1 KiB) is fetched by nine workgroups and every fetched KiB feeds NBLK <= 4 MFMAs - the
// kernel lives on the L2 -> CU operand path (about one KiB per 16-36 cycles and CU) and on the issue cost of its LDS-DMA
// pieces, matrix pipe 9 % busy.  Two round-4 probes confirmed it: the same tiles with both operands loaded straight into
// registers (no LDS, no barrier) ran 1.4-4x SLOWER (the weights then cross the vector-memory path once per wave), and a
// six-stage ring on 256-stamp tiles changed nothing (not latency).
//
// Here a wave owns 16 stamps x EIGHT consecutive output pixels of one row x 16*NBLK output channels and keeps all of their
// accumulators in registers.  In the stamp-inner layout an input block IS an MFMA A fragment (lane (row l & 15, piece
// l >> 4) holds 16 consecutive bytes of row `stamp`), and no other wave wants this wave's stamps: A goes global -> VGPR
// directly, ten blocks per input row (the strip plus its halo), and each of them is used for up to 3 taps x NBLK MFMAs -
// 24 NBLK MFMAs per ten KiB instead of NBLK per KiB.  The weights of a 32-channel chunk (9 taps x NBLK KiB, shared by the
// workgroup's four waves = four stamp groups of the same strip) arrive by LDS-DMA into a double-buffered stage, one
// barrier per CHUNK (not per K step), and are read as fragments per input row (3 NBLK ds_read_b128).  Input rows /
// columns outside the image are read from a zero page, so the instruction stream - and with it the counted vmcnt at the
// chunk boundary - is the same for every strip.
template <int NBLK, int P = 8>                          // P: output pixels per strip
__global__ __launch_bounds__(256, 1) void bconv_row_kernel(const BConvParams p) {
  constexpr int BN = 16 * NBLK;
  constexpr int BSTAGE = 9 * NBLK * 1024;                 // weights of one chunk: [tap][column block][1 KiB fragment image]
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  unsigned char* sB = smem;                               // 2 stages
  constexpr int WREG = 16 * BN * (NBLK == 1 ? 4 : 2);     // per-wave epilogue tile [16 rows][BN]
  unsigned char* wreg_base = smem + 2 * BSTAGE;
  float* pbuf = reinterpret_cast<float*>(smem + 2 * BSTAGE + 4 * WREG);   // [8 px][4 waves][2][BN] partial sums of the fused PReLU backward

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.Cout / BN;
  const int nq = p.NBp >> 6;                              // 64-stamp quads
  const int nsx = (p.Hout + P - 1) / P;                   // strips per row
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  // tile order: column tile fastest (the tiles of one strip share its input), then stamp quad, then strip (row-major)
  const int tile_n = bid % ntn;
  const int rest = bid / ntn;
  const int quad = rest % nq;
  const int strip = rest / nq;
  const int oh = strip / nsx, ow0 = (strip - oh * nsx) * P;
  const int n0 = tile_n * BN;
  const int st0 = quad * 64 + wave * 16;                  // first stamp of this wave
  const int npx = min(P, p.Hout - ow0);                   // valid pixels of the strip

  const int fr = lane & 15, fq = lane >> 4;
  const int c = fr, g4 = fq;
  const int ch0 = n0 + NBLK * c;
  const unsigned char* Xb = reinterpret_cast<const unsigned char*>(p.X);
  const unsigned char* Wb = reinterpret_cast<const unsigned char*>(p.W);
  const size_t pixbytes = (size_t)p.NBp * p.Cin * 2;
  const int cpt = p.Cin >> 5;
  // A fragment source of input pixel (ih, iw), chunk cc: Xb + (ih*Hin + iw) * pixbytes + a_lane + cc * 64
  const unsigned a_lane = (unsigned)((((st0 + fr) * p.Cin) + fq * 8) * 2);
  // weight-DMA roles (as in bconv_uni_kernel): LDS slot `lane` of a block = (row lane >> 2, piece (lane & 3) ^ G4[row >> 2])
  const int drow = lane >> 2;
  const int dq = (lane & 3) ^ ((4 - (drow >> 2)) & 3);
  const unsigned b_lane = (unsigned)((((n0 + NBLK * drow) * p.Kpad) + dq * 8) * 2);
  const int fragoff = (fr * 4 + (fq ^ ((4 - (fr >> 2)) & 3))) * 16;
  // input offset (dh, dw) of a weight tap t = kh * 3 + kw: form 0 (Conv2D forward / Conv2DTranspose data gradient, pad 1):
  // source = out + k - 1; form 1 (Conv2DTranspose forward / Conv2D data gradient): source = out + 1 - k
  const int sgn = p.form == 0 ? 1 : -1;

  // BEPI_HEAD: the labels this lane's share of the head needs (lane = stamp l >> 2, bands (l & 3) and (l & 3) + 4; eight
  // pixels) are requested here, ahead of the whole K loop, so that the epilogue finds them in registers
  float hy[P][2];
#pragma unroll
  for (int px = 0; px < P; ++px) hy[px][0] = hy[px][1] = 0.f;
  if constexpr (NBLK == 1) {
    if (p.epi == BEPI_HEAD) {
      const int hs = lane >> 2, hq = lane & 3;
      const int b = st0 + hs, h = oh - p.hd.crop0;
      if ((unsigned)h < (unsigned)p.hd.H && b < p.hd.NB) {
        const long row = p.hd.idx ? (long)p.hd.idx[b] : (long)p.hd.first + b;
        const float* yrow = p.hd.y + (row * p.hd.H + h) * p.hd.H * p.hd.nb;
#pragma unroll
        for (int px = 0; px < P; ++px) {
          const int w = ow0 + px - p.hd.crop0;
          if ((unsigned)w < (unsigned)p.hd.H && px < npx) {
            if (hq < p.hd.nb) hy[px][0] = yrow[w * p.hd.nb + hq];
            if (hq + 4 < p.hd.nb) hy[px][1] = yrow[w * p.hd.nb + hq + 4];
          }
        }
      }
    }
  }

  // weights of chunk cc -> stage buffer: 9 * NBLK pieces, dealt round-robin over the four waves
  auto issue_b = [&](int cc, int buf) {
    unsigned char* dst = sB + buf * BSTAGE;
#pragma unroll
    for (int k = 0; k < (9 * NBLK + 3) / 4; ++k) {
      const int piece = k * 4 + wave;                     // (tap, column block)
      if (piece < 9 * NBLK) {
        const int t = piece / NBLK, j = piece - t * NBLK;
        __builtin_amdgcn_global_load_lds((bc_gptr_t)(Wb + (size_t)((t * p.Cin + cc * 32) * 2) + (size_t)j * p.Kpad * 2 + b_lane),
                                         (bc_lptr_t)(dst + piece * 1024), 16, 0, 0);
      }
    }
  };

  f32x4 acc[P][NBLK];
  {
    float bias[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) bias[j] = 0.f;
    if (p.bias && (p.epi == BEPI_FWD || p.epi == BEPI_RAW32 || p.epi == BEPI_HEAD)) load_f32<NBLK>(p.bias + ch0, bias);
#pragma unroll
    for (int px = 0; px < P; ++px)
#pragma unroll
      for (int j = 0; j < NBLK; ++j) acc[px][j] = (f32x4){bias[j], bias[j], bias[j], bias[j]};
  }

  // the three input rows of the strip (dh = -1, 0, 1) and its ten input columns (ow0 - 1 .. ow0 + 8).  Every source is a
  // UNIFORM base (an input pixel's [NBp][Cin] block, or the zero page, which is at least one such block long) plus the
  // lane's constant offset a_lane: no per-load vector address arithmetic, no address registers held across the loop
  const unsigned char* zbase = reinterpret_cast<const unsigned char*>(p.zero);
  bool rowok[3];
  int rowpix[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int ih = oh + r - 1;
    rowok[r] = ih >= 0 && ih < p.Hin;
    rowpix[r] = (rowok[r] ? ih : 0) * p.Hin + ow0 - 1;
  }
  unsigned colmask = 0;
#pragma unroll
  for (int i = 0; i < P + 2; ++i) {
    const int iw = ow0 - 1 + i;
    if (iw >= 0 && iw < p.Hin && i <= npx + 1) colmask |= 1u << i;
  }
  bc_bf16x8 ra[2][P + 2];
  auto load_a = [&](auto setc, int r, int cc) {
    constexpr int S = decltype(setc)::value;
#pragma unroll
    for (int i = 0; i < P + 2; ++i) {
      const bool ok = rowok[r] && ((colmask >> i) & 1u);
      const unsigned long long u64 = (unsigned long long)(ok ? Xb + (size_t)(rowpix[r] + i) * pixbytes + cc * 64 : zbase);
      // (uniform by construction; said so explicitly, or the address is built per lane in vector registers)
      const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u64), hi = __builtin_amdgcn_readfirstlane((unsigned)(u64 >> 32));
      const unsigned char* ub = reinterpret_cast<const unsigned char*>(((unsigned long long)hi << 32) | lo);
      ra[S][i] = *reinterpret_cast<const bc_bf16x8*>(ub + a_lane);
    }
  };
  // one input row against the weights of its three taps: out pixel px takes input column px + 1 + dw
  auto row_mfma = [&](auto setc, int r, int buf) {
    constexpr int S = decltype(setc)::value;
    const int dh = r - 1;
    const int kh = sgn > 0 ? dh + 1 : 1 - dh;
    bc_bf16x8 rb[3][NBLK];
    const unsigned char* bs = sB + buf * BSTAGE + fragoff;
#pragma unroll
    for (int dwi = 0; dwi < 3; ++dwi) {
      const int kw = sgn > 0 ? dwi : 2 - dwi;             // dw = dwi - 1
#pragma unroll
      for (int j = 0; j < NBLK; ++j) rb[dwi][j] = *reinterpret_cast<const bc_bf16x8*>(bs + ((kh * 3 + kw) * NBLK + j) * 1024);
    }
    if (!rowok[r]) return;                                // (uniform) a row outside the image contributes nothing
#pragma unroll
    for (int i = 0; i < P + 2; ++i) {
#pragma unroll
      for (int dwi = 0; dwi < 3; ++dwi) {
        const int px = i - dwi;                           // input column i = px + 1 + (dwi - 1)
        if (px < 0 || px >= P) continue;
#pragma unroll
        for (int j = 0; j < NBLK; ++j)
          acc[px][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra[S][i], rb[dwi][j], acc[px][j], 0, 0, 0);
      }
    }
  };
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, 1>;

  issue_b(0, 0);
  load_a(S0{}, 0, 0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 2) : "memory");     // the weight pieces are older than the ten A loads
  __builtin_amdgcn_s_barrier();
  if (cpt > 1) issue_b(1, 1);
  // rows are walked (chunk, row) with the A loads one row ahead: row index q = 3 * cc + r, register set q & 1
  for (int cc = 0; cc < cpt; ++cc) {
    const int buf = cc & 1;
    // rows 0 and 1 of the chunk (sets alternate with the global row index; 3 rows per chunk -> the parity flips per chunk)
    if ((cc & 1) == 0) {
      load_a(S1{}, 1, cc);
      row_mfma(S0{}, 0, buf);
      load_a(S0{}, 2, cc);
      row_mfma(S1{}, 1, buf);
      if (cc + 1 < cpt) load_a(S1{}, 0, cc + 1);
      row_mfma(S0{}, 2, buf);
    } else {
      load_a(S0{}, 1, cc);
      row_mfma(S1{}, 0, buf);
      load_a(S1{}, 2, cc);
      row_mfma(S0{}, 1, buf);
      if (cc + 1 < cpt) load_a(S0{}, 0, cc + 1);
      row_mfma(S1{}, 2, buf);
    }
    if (cc + 1 < cpt) {
      // the weights of chunk cc + 1 were issued a chunk ago: everything but the ten youngest loads (the next row's A) is older
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P + 2) : "memory");
      __builtin_amdgcn_s_barrier();                       // every wave's pieces have landed, every wave is done with stage `buf`
      if (cc + 2 < cpt) issue_b(cc + 2, buf);
    }
  }

  // ---- epilogue, pixel by pixel: per-wave LDS tile [16 rows][BN], rows = the wave's 16 stamps of that pixel ----
  unsigned char* wreg = wreg_base + wave * WREG;
  bc_bf16* wt = reinterpret_cast<bc_bf16*>(wreg);
  auto flush = [&](void* dst, int esz, size_t rb0) {
    const int rowb = BN * esz;
    unsigned char* out = reinterpret_cast<unsigned char*>(dst) + (rb0 * p.Cout + n0) * esz;
    const size_t rstride = (size_t)p.Cout * esz;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k >= 16 * rowb / 1024) break;
      const int byte = (k * 64 + lane) * 16;
      const int row = byte / rowb, colb = byte - row * rowb;
      *reinterpret_cast<f32x4*>(out + row * rstride + colb) = *reinterpret_cast<const f32x4*>(wreg + byte);
    }
  };
  constexpr int NPC = 16 * BN / 512;                      // 16-byte pieces per lane of a [16][BN] bf16 tile (NBLK >= 2)
  float h_nll = 0.f, h_se = 0.f;                          // BEPI_HEAD: this lane's share of the two loss sums
#pragma unroll
  for (int px = 0; px < P; ++px) {
    if (px >= npx) break;                                 // (uniform)
    const int pix = oh * p.Hout + ow0 + px;
    const size_t rb0 = (size_t)pix * p.NBp + st0;
    if (p.epi == BEPI_HEAD) {
      if constexpr (NBLK == 1) {
        // the wave's [16 stamps][16 columns] fp32 tile of this pixel, then bf_head_kernel's arithmetic on it: lane (stamp
        // l >> 2, q = l & 3) takes the bands q and q + 4 (column c = loc, column nb + c = scale pre-activation)
        float* tile = reinterpret_cast<float*>(wreg);
#pragma unroll
        for (int r = 0; r < 4; ++r) tile[(4 * g4 + r) * 16 + c] = acc[px][0][r];
        bc_bf16* dtile = reinterpret_cast<bc_bf16*>(pbuf) + wave * 256;     // [16][16] bf16
        const int hs = lane >> 2, hq = lane & 3;
        const int b = st0 + hs;
        const int h = oh - p.hd.crop0, w = ow0 + px - p.hd.crop0;
        const bool in = (unsigned)h < (unsigned)p.hd.H && (unsigned)w < (unsigned)p.hd.H && b < p.hd.NB;
        {
          bc_bf16x4 z;
#pragma unroll
          for (int k = 0; k < 4; ++k) z[k] = (bc_bf16)0.f;
          *reinterpret_cast<bc_bf16x4*>(dtile + hs * 16 + 4 * hq) = z;
        }
        if (in) {
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int cb = hq + 4 * k;
            if (cb < p.hd.nb) {
              const float tl = tile[hs * 16 + cb], ts = tile[hs * 16 + p.hd.nb + cb];
              const float loc = fmaxf(tl, 0.f);
              const float sig = p.hd.sigma_floor + fmaxf(ts, 0.f);
              const float inv = 1.0f / sig;
              const float df = hy[px][k] - loc;
              const float rr = df * inv;
              h_nll += 0.5f * rr * rr + logf(sig) + 0.91893853320467274178f;
              if (p.hd.mse_sample) {
                const float dsm = df - sig * dv_philox_normal((unsigned)b, (unsigned)((h * p.hd.H + w) * p.hd.nb + cb),
                                                              p.hd.mse_stream, p.hd.mse_seed);
                h_se += dsm * dsm;
              } else {
                h_se += df * df;
              }
              dtile[hs * 16 + cb] = (bc_bf16)(tl > 0.f ? -(rr * inv) * p.hd.gscale : 0.f);
              dtile[hs * 16 + p.hd.nb + cb] = (bc_bf16)(ts > 0.f ? (inv - rr * rr * inv) * p.hd.gscale : 0.f);
            }
          }
        }
        if (lane < 32)                                     // the wave's 512 B of this pixel: 16 rows x 32 B
          *reinterpret_cast<f32x4*>(reinterpret_cast<unsigned char*>(p.hd.dt) + (rb0 + (lane >> 1)) * 32 + (lane & 1) * 16) =
              *reinterpret_cast<const f32x4*>(reinterpret_cast<const unsigned char*>(dtile) + lane * 16);
      }
      continue;
    }
    if (p.epi == BEPI_RAW32) {
      if constexpr (NBLK == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) reinterpret_cast<float*>(wreg)[(4 * g4 + r) * BN + c] = acc[px][0][r];
        flush(p.Uf, 4, rb0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < NBLK; ++j) p.Uf[(rb0 + 4 * g4 + r) * p.Cout + ch0 + j] = acc[px][j][r];
      }
      continue;
    }
    if constexpr (NBLK >= 2) {
      if (p.epi == BEPI_RAWBF || (p.epi == BEPI_FWD && p.U)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v[NBLK];
#pragma unroll
          for (int j = 0; j < NBLK; ++j) v[j] = acc[px][j][r];
          store_bf<NBLK>(wt + (4 * g4 + r) * BN + NBLK * c, v);
        }
        flush(p.U, 2, rb0);
        if (p.epi == BEPI_RAWBF) continue;
      }
      float al[NBLK];
      load_f32<NBLK>(p.alpha + (size_t)pix * p.Cout + ch0, al);
      if (p.epi == BEPI_FWD) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float a[NBLK];
#pragma unroll
          for (int j = 0; j < NBLK; ++j) a[j] = fmaxf(acc[px][j][r], 0.f) + al[j] * fminf(acc[px][j][r], 0.f);
          store_bf<NBLK>(wt + (4 * g4 + r) * BN + NBLK * c, a);
        }
        flush(p.A, 2, rb0);
        continue;
      }
      // BEPI_BWD: d(pre-activation) = d(activation) * gate(u), stamp sums for d(alpha) / d(bias)
#pragma unroll
      for (int k = 0; k < NPC; ++k) {
        const int byte = (k * 64 + lane) * 16;
        const int row = byte / (BN * 2), colb = byte - row * (BN * 2);
        *reinterpret_cast<f32x4*>(wreg + byte) = *reinterpret_cast<const f32x4*>(
            reinterpret_cast<const unsigned char*>(p.Uin) + ((rb0 + row) * p.Cout + n0) * 2 + colb);
      }
      float dal[NBLK], db[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) dal[j] = db[j] = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        bc_bf16* q = wt + (4 * g4 + r) * BN + NBLK * c;
        float u[NBLK], du[NBLK];
        load_bf<NBLK>(q, u);
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
          const float v = acc[px][j][r];
          du[j] = v * (u[j] > 0.f ? 1.f : al[j]);
          dal[j] += v * fminf(u[j], 0.f);
          db[j] += du[j];
        }
        store_bf<NBLK>(q, du);
      }
      flush(p.U, 2, rb0);
      if (p.dal_part) {
        // sums over the wave's 16 stamps (rows live on the lane groups g4); parked in LDS per (pixel, wave) until the end
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
          dal[j] += __shfl_xor(dal[j], 16);
          dal[j] += __shfl_xor(dal[j], 32);
          db[j] += __shfl_xor(db[j], 16);
          db[j] += __shfl_xor(db[j], 32);
        }
        if (g4 == 0) {
#pragma unroll
          for (int j = 0; j < NBLK; ++j) {
            pbuf[((px * 4 + wave) * 2 + 0) * BN + NBLK * c + j] = dal[j];
            pbuf[((px * 4 + wave) * 2 + 1) * BN + NBLK * c + j] = db[j];
          }
        }
      }
    }
  }
  if constexpr (NBLK == 1) {
    if (p.epi == BEPI_HEAD) {                             // (uniform) one row of partial sums per workgroup, waves in order
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        h_nll += __shfl_xor(h_nll, o);
        h_se += __shfl_xor(h_se, o);
      }
      __syncthreads();                                    // every wave is done with pbuf's tiles
      if (lane == 0) {
        pbuf[wave * 2] = h_nll;
        pbuf[wave * 2 + 1] = h_se;
      }
      __syncthreads();
      if (tid == 0) {
        p.hd.part[blockIdx.x * 2] = ((pbuf[0] + pbuf[2]) + pbuf[4]) + pbuf[6];
        p.hd.part[blockIdx.x * 2 + 1] = ((pbuf[1] + pbuf[3]) + pbuf[5]) + pbuf[7];
      }
    }
  }
  if constexpr (NBLK >= 2) {
    if (p.epi == BEPI_BWD && p.dal_part) {
      // ... then over the four waves = 64 stamps, in wave order: one partial row per 64-stamp quad, as bconv_uni_kernel
      // writes them (p.dal_part is uniform: every wave reaches the barrier)
      __syncthreads();
      for (int e = tid; e < npx * BN; e += 256) {
        const int px = e / BN, col = e - px * BN;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          a += pbuf[((px * 4 + w) * 2 + 0) * BN + col];
          b += pbuf[((px * 4 + w) * 2 + 1) * BN + col];
        }
        const size_t o = ((size_t)quad * p.Hout * p.Hout + oh * p.Hout + ow0 + px) * p.Cout + n0 + col;
        p.dal_part[o] = a;
        p.db_part[o] = b;
      }
    }
  }
}

static int bconv_head_strip() {     // pixels per strip of the head-carrying row launch
  static const int v = getenv("DV_BF_HEAD_STRIP") ? atoi(getenv("DV_BF_HEAD_STRIP")) : 4;
  return v == 8 ? 8 : (v == 2 ? 2 : 4);
}
long bconv_head_tiles(const BConvParams& p) {
  static const int row_mode = getenv("DV_BCONV_ROW") ? atoi(getenv("DV_BCONV_ROW")) : 1;
  const int ksz = p.ksz == 0 ? 3 : p.ksz;
  if (!row_mode || p.Cin % 32 || ksz != 3 || p.s != 1 || p.pb != 1 || p.Hin != p.Hout || (p.NBp & 63) || p.Cout != 16) return 0;
  const int P = bconv_head_strip();
  return (long)p.Hout * ((p.Hout + P - 1) / P) * (p.NBp >> 6);
}

int launch_bconv(const BConvParams& p_in, hipStream_t s) {
  BConvParams p = p_in;
  if (p.ksz == 0) p.ksz = 3;
#ifdef DV_DEBUG_EXPORTS
  {
    static const int exp_mode = DV_EXP_SWITCH("DV_EXP_BCONV");   // measurement only, development library (bf16.h)
    p.exp = exp_mode;
    if (exp_mode == 2) return OK;                       // (2: no conv launch at all)
  }
#endif
  if (p.NBp <= 0 || (p.NBp & 15) || p.Cout % 16 || p.Kpad % 32 || p.s < 1 || p.s > 2 || p.ksz < 1 || p.ksz > 5 ||
      p.Kpad < p.ksz * p.ksz * p.Cin) {
    set_error("bconv: bad geometry (NBp %d, Cout %d, Kpad %d, kernel size %d)", p.NBp, p.Cout, p.Kpad, p.ksz);
    return E_INVALID;
  }
  // mode 1 (K walked in 8-channel pieces across tap boundaries) was written for the 8- / 16-channel first conv and head
  // gradient; it is what every other channel count that is a multiple of 8 but not of 32 takes too (filters 48, 80, ...:
  // functional, the per-piece lane addressing is not tuned for long K)
  const int mode = p.Cin % 32 == 0 ? 0 : (p.Cin % 8 == 0 ? 1 : -1);
  if (mode < 0) {
    set_error("bconv: input channels must be a multiple of 8 (got %d)", p.Cin);
    return E_INVALID;
  }
  if (p.epi == BEPI_BWD && p.dal_part && !bconv_bwd_fusable(p.NBp)) {
    set_error("bconv: fused PReLU backward needs a stamp count padded to a multiple of 64");
    return E_INVALID;
  }
  if ((size_t)p.Hin * p.Hin * p.NBp * (p.Cin >> 3) >= ((size_t)1 << 31) ||
      (size_t)p.Hout * p.Hout * p.NBp >= ((size_t)1 << 31)) {
    set_error("bconv: tensor too large for 32-bit block offsets");
    return E_INVALID;
  }
  int nblk = p.Cout % 64 == 0 ? 4 : (p.Cout % 32 == 0 ? 2 : 1);
  const long nbx = (p.Hout + 7) / 8;
  const long M16 = nbx * nbx * 64 * (p.NBp >> 4);   // 8 x 8 pixel blocks, see the kernel
  // uniform-tile kernel: the stamp count must be a multiple of its stamp tile (256, 128 or 64 stamps); the first-layer
  // form (CINMODE 1) exists for 256-stamp tiles only, and a 64-stamp tile needs at least 32 output columns
  int gt0 = (p.NBp & 255) == 0 ? 16 : ((p.NBp & 127) == 0 ? 8 : ((p.NBp & 63) == 0 ? 4 : 0));
  if (mode == 1 && gt0 != 16) gt0 = 0;
  if (gt0 == 4 && nblk < 2) gt0 = 0;
  const bool uni = gt0 != 0;
  // Deep layers have few row tiles and a long K loop that one workgroup walks alone.  First smaller stamp tiles (uniform
  // kernel: 128 or 64 stamps per workgroup, 4-8 independent pipelines per CU; the weights are re-read from L2 once more
  // per halving), then narrower column tiles (the input is re-read once more per halving) put enough workgroups on the
  // chip; these launches are latency-bound and do not notice the extra L2 traffic.
  const long want_tiles = getenv("DV_BCONV_MIN_TILES") ? atol(getenv("DV_BCONV_MIN_TILES")) : 512;   // (read per call: the tests toggle it)
  const long gt_tiles = 1024;
  int gt = uni ? gt0 : BC_GT;
  auto ntiles = [&](int g, int nb) { return ((M16 + g - 1) / g) * (long)(p.Cout / (16 * nb)); };
  if (uni && mode == 0)
    while (gt > 4 && ntiles(gt, nblk) < gt_tiles && (gt / 2) * nblk >= 8) gt >>= 1;
  while (nblk > 1 && ntiles(gt, nblk) < want_tiles && gt * (nblk / 2) >= 8) nblk >>= 1;
  // paired chunks (K step = 64 channels) where an operand row is half a cache line per chunk (Cin >= 64) AND the launch
  // is one of the deep, L2-resident ones that the tile rule above gave small stamp tiles: -10..20 % there; the large
  // launches (256-stamp tiles) are bound elsewhere and lose with the doubled stage (measured, tools/bconv_launches.py)
  const bool pair = uni && mode == 0 && p.Cin % 64 == 0 && gt <= 8;
  const int ch = pair ? 2 : 1;
  const long tiles = ntiles(gt, nblk);
  // row-strip form (bconv_row_kernel): stride-1 3 x 3 layers with Cin % 32 == 0 on batches padded to 64 stamps.  Shipped
  // for the 16-column head conv only (measured, same box, serialised 256-stamp step: 52.4 -> 37.7 us); on the 32- to
  // 256-channel layers it is parity-green but 1.2-1.6x SLOWER than the one-pixel tiles (one wave per SIMD with a single
  // input row of register loads in flight does not cover the L2 / HBM latency: DESIGN 4b).  DV_BCONV_ROW=2 takes it for
  // every eligible layer (the A/B switch of that measurement), =0 never.
  static const int row_mode = getenv("DV_BCONV_ROW") ? atoi(getenv("DV_BCONV_ROW")) : 1;
  if (row_mode && mode == 0 && p.ksz == 3 && p.s == 1 && p.pb == 1 && p.Hin == p.Hout && (p.NBp & 63) == 0) {
    int nb = p.Cout % 64 == 0 ? 4 : (p.Cout % 32 == 0 ? 2 : 1);
    const long strips = (long)p.Hout * ((p.Hout + 7) / 8);
    auto rtiles = [&](int b) { return strips * (p.NBp >> 6) * (p.Cout / (16 * b)); };
    // at least one workgroup per CU before the column tile stays wide
    while (nb > 1 && rtiles(nb) < 256) nb >>= 1;
    if (nb == 1 && p.epi != BEPI_RAW32 && p.epi != BEPI_HEAD) nb = 0;   // the bf16 epilogues want >= 32 columns (1-KiB row pieces)
    if (p.epi == BEPI_HEAD && (nb != 1 || p.Cout != 16)) nb = 0;
    if (nb == 1 && p.Cout % 16) nb = 0;
    if (nb > 1 && row_mode < 2) nb = 0;
    // the head-carrying launch (BEPI_HEAD) takes FOUR-pixel strips: 57 us against 68 with eight (half the registers of input
    // fragments, twice the workgroups; the plain head conv keeps eight, not re-measured).  DV_BF_HEAD_STRIP=8 | 2: A/B
    const int head_p = bconv_head_strip();
    if (nb == 1 && p.epi == BEPI_HEAD && head_p != 8) {
      const long rtp = (long)p.Hout * ((p.Hout + head_p - 1) / head_p) * (p.NBp >> 6);
      const size_t rlp = (size_t)2 * 9 * 1024 + 4 * 16 * 16 * 4 + (size_t)8 * 4 * 2 * 16 * 4 + 1024;
#define BROW_HEAD(P_)                                                                                      \
      do {                                                                                                 \
        static bool attr = false;                                                                          \
        if (!attr) {                                                                                       \
          DV_HIP(hipFuncSetAttribute((const void*)bconv_row_kernel<1, P_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rlp)); \
          attr = true;                                                                                     \
        }                                                                                                  \
        hipLaunchKernelGGL((bconv_row_kernel<1, P_>), dim3((unsigned)rtp), dim3(256), rlp, s, p);          \
      } while (0)
      if (head_p == 4) BROW_HEAD(4);
      else BROW_HEAD(2);
#undef BROW_HEAD
      DV_HIP(hipGetLastError());
      return OK;
    }
    if (nb) {
      const long rt = rtiles(nb);
      const size_t rl = (size_t)2 * 9 * nb * 1024 + 4 * 16 * 16 * nb * (nb == 1 ? 4 : 2) + (size_t)8 * 4 * 2 * 16 * nb * 4 + 1024;
#define BROW_LAUNCH(NB_)                                                                                  \
      do {                                                                                                 \
        static size_t attr_lds = 0;                                                                        \
        if (attr_lds < rl) {                                                                               \
          DV_HIP(hipFuncSetAttribute((const void*)bconv_row_kernel<NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rl)); \
          attr_lds = rl;                                                                                   \
        }                                                                                                  \
        hipLaunchKernelGGL((bconv_row_kernel<NB_>), dim3((unsigned)rt), dim3(256), rl, s, p);             \
      } while (0)
      if (nb == 4) BROW_LAUNCH(4);
      else if (nb == 2) BROW_LAUNCH(2);
      else BROW_LAUNCH(1);
#undef BROW_LAUNCH
      DV_HIP(hipGetLastError());
      return OK;
    }
  }
  if (p.epi == BEPI_HEAD) {
    set_error("bconv: the fused head needs the row-strip form (ask bconv_head_tiles first)");
    return E_INVALID;
  }
  // TWO ring stages instead of three (round 4): half of a launch's time is prologue and epilogue (HBM-bound stores on
  // the 64 x 64 layers; launch, tap table and tile sums on the deep ones), which only ANOTHER resident workgroup can
  // overlap with a K loop - 36-48 KB per workgroup instead of 54-76 put three to five on a CU instead of two.  Same box,
  // alternating runs, 256-stamp step: three stages everywhere 2.04 ms; two stages for the 256-stamp tiles of <= 32
  // columns 1.97; also for the 128- / 64-stamp tiles 1.95 (the default); also for the 256-stamp x 64-column tile
  // (64 accumulators: three workgroups per CU spill) 1.99.  DV_BCONV_NS2=0|1|2|3 selects these (read per call: tests).
  const int ns2_mode = getenv("DV_BCONV_NS2") ? atoi(getenv("DV_BCONV_NS2")) : 2;
  const bool ns2 = uni && ((ns2_mode >= 1 && gt == 16 && nblk <= 2) || (ns2_mode >= 2 && gt < 16) || (ns2_mode >= 3 && gt == 16));
  const size_t lds = uni ? (size_t)(ns2 ? 2 : 3) * ch * (gt + nblk) * 1024 + 4096 : (size_t)BC_NST * (BC_GT + nblk) * 1024 + 2048;
#define BC_LAUNCH_K(KERNEL_)                                                                           \
  do {                                                                                                 \
    static size_t attr_lds = 0;                                                                        \
    if (attr_lds < lds) {                                                                              \
      DV_HIP(hipFuncSetAttribute((const void*)KERNEL_, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      attr_lds = lds;                                                                                  \
    }                                                                                                  \
    hipLaunchKernelGGL(KERNEL_, dim3((unsigned)tiles), dim3(256), lds, s, p);                          \
  } while (0)
#define BC_LAUNCH(NB_, MODE_)                                                                          \
  do {                                                                                                 \
    if (!uni) BC_LAUNCH_K((bconv_kernel<NB_, MODE_>));                                                 \
    else if constexpr (MODE_ == 1) {                                                                   \
      if (ns2) BC_LAUNCH_K((bconv_uni_kernel<NB_, 1, 16, 1, 2>));                                      \
      else BC_LAUNCH_K((bconv_uni_kernel<NB_, 1, 16, 1>));                                             \
    } else if (gt == 16) {                                                                             \
      if (ns2) BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 16, 1, 2>));                                      \
      else BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 16, 1>));                                             \
    } else if (gt == 8) {                                                                              \
      if (pair && ns2) BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 8, 2, 2>));                               \
      else if (pair) BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 8, 2>));                                    \
      else if (ns2) BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 8, 1, 2>));                                  \
      else BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 8, 1>));                                              \
    } else if constexpr (NB_ >= 2) {                                                                   \
      if (pair && ns2) BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 4, 2, 2>));                               \
      else if (pair) BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 4, 2>));                                    \
      else if (ns2) BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 4, 1, 2>));                                  \
      else BC_LAUNCH_K((bconv_uni_kernel<NB_, 0, 4, 1>));                                              \
    }                                                                                                  \
  } while (0)
  if (mode == 0) {
    if (nblk == 4) BC_LAUNCH(4, 0);
    else if (nblk == 2) BC_LAUNCH(2, 0);
    else BC_LAUNCH(1, 0);
  } else {
    if (nblk == 4) BC_LAUNCH(4, 1);
    else if (nblk == 2) BC_LAUNCH(2, 1);
    else BC_LAUNCH(1, 1);
  }
#undef BC_LAUNCH_K
#undef BC_LAUNCH
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
