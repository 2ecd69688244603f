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

namespace dv {

typedef const __attribute__((address_space(1))) void* bc_gptr_t;
typedef __attribute__((address_space(3))) void* bc_lptr_t;
typedef __bf16 bc_bf16;
typedef __bf16 bc_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bc_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bc_bf16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int BC_GT = 16;      // groups per tile
constexpr int BC_TABW = 12;    // table row: 9 tap offsets, output row base, pixel index, spare

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
// CINMODE 1: Cin == 8 or 16 (first conv, data gradient of the 16-channel head): a K step is four 16-byte pieces,
//            piece q of step i is channels (i*4+q) % (Cin/8) * 8.. of tap (i*4+q) / (Cin/8); all nine taps are walked.
template <int NBLK, int CINMODE>
__global__ __launch_bounds__(256, 2) void bconv_kernel(const BConvParams p) {
  constexpr int STAGE = (BC_GT + NBLK) * 1024;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  int* tab = reinterpret_cast<int*>(smem + 3 * STAGE);   // [16][12]
  int* anyv = tab + BC_GT * BC_TABW;                      // [16]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntn = p.Cout / (16 * NBLK);
  int bid = blockIdx.x;
  {
    // XCD-contiguous tile order: the blocks one XCD receives (every 8th) walk neighbouring pixels, whose taps
    // overlap, so the re-read of the input comes out of that XCD's L2
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tile_m = bid / ntn, tile_n = bid - tile_m * ntn;
  const int n0 = tile_n * 16 * NBLK;
  const int NSB = p.NBp >> 4;
  const int M16 = p.Hout * p.Hout * NSB;

  if (tid < BC_GT * 9) {
    const int g = tid / 9, t = tid - g * 9;
    const int m16 = tile_m * BC_GT + g;
    int off = -1;
    if (m16 < M16) {
      const int pix = m16 / NSB, sb = m16 - pix * NSB;
      const int oh = pix / p.Hout, ow = pix - oh * p.Hout;
      const int kh = t / 3, kw = t - kh * 3;
      int ih, iw;
      bool ok = true;
      if (p.form == 0) {
        ih = oh * p.s + kh - p.pb;
        iw = ow * p.s + kw - p.pb;
      } else {
        const int nh = oh + p.pb - kh, nw = ow + p.pb - kw;
        ok = nh >= 0 && nw >= 0 && (nh % p.s) == 0 && (nw % p.s) == 0;
        ih = nh / p.s;
        iw = nw / p.s;
      }
      ok = ok && ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Hin;
      // offset of the group's [16][Cin] block in units of 8 elements (16 bytes)
      if (ok) off = ((ih * p.Hin + iw) * p.NBp + sb * 16) * (p.Cin >> 3);
      if (t == 0) {
        tab[g * BC_TABW + 9] = pix * p.NBp + sb * 16;
        tab[g * BC_TABW + 10] = pix;
      }
    } else if (t == 0) {
      tab[g * BC_TABW + 9] = -1;
      tab[g * BC_TABW + 10] = 0;
    }
    tab[g * BC_TABW + t] = off;
  }
  __syncthreads();
  if (tid < 9) {
    int a = 0;
#pragma unroll
    for (int g = 0; g < BC_GT; ++g) a |= tab[g * BC_TABW + tid] >= 0 ? 1 : 0;
    anyv[tid] = a;
  }
  __syncthreads();
  unsigned long long vcode = 0;
  int nvalid = 0;
#pragma unroll
  for (int t = 0; t < 9; ++t)
    if (anyv[t]) {
      vcode |= (unsigned long long)t << (4 * nvalid);
      ++nvalid;
    }
  nvalid = __builtin_amdgcn_readfirstlane(nvalid);
  const unsigned vlo = __builtin_amdgcn_readfirstlane((unsigned)vcode);
  const unsigned vhi = __builtin_amdgcn_readfirstlane((unsigned)(vcode >> 32));
  vcode = ((unsigned long long)vhi << 32) | vlo;

  const int cpt = p.Cin >> 5;         // chunks per tap (CINMODE 0)
  const int ppt = p.Cin >> 3;         // 16-byte pieces per tap (CINMODE 1: 1 or 2)
  const int nsteps = CINMODE == 0 ? nvalid * cpt : (9 * ppt + 3) >> 2;

  // DMA lane roles: LDS slot `lane` of a block = (row, piece ^ G4[row>>2])
  const int drow = lane >> 2;
  const int dq = (lane & 3) ^ ((4 - (drow >> 2)) & 3);
  const bc_bf16* Xb = reinterpret_cast<const bc_bf16*>(p.X);
  const bc_bf16* Wb = reinterpret_cast<const bc_bf16*>(p.W);
  const unsigned char* zlane = reinterpret_cast<const unsigned char*>(p.zero) + lane * 16;
  const int jb = wave % NBLK;   // B block this wave loads (duplicates when NBLK < 4: same bytes, same slot)
  const bc_bf16* wrow = Wb + (size_t)(n0 + NBLK * drow + jb) * p.Kpad + dq * 8;
  const int arow = drow * p.Cin;

  auto issue = [&](int step, int buf) {
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + BC_GT * 1024;
    if constexpr (CINMODE == 0) {
      const int ti = step / cpt, cc = step - ti * cpt;
      const int tap = (int)((vcode >> (4 * ti)) & 15);
      const int lo = arow + cc * 32 + dq * 8;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int g = wave * 4 + gi;
        const int off = tab[g * BC_TABW + tap];
        const void* src = off >= 0 ? (const void*)(Xb + ((size_t)off * 8 + lo)) : (const void*)zlane;
        __builtin_amdgcn_global_load_lds((bc_gptr_t)src, (bc_lptr_t)(sA + g * 1024), 16, 0, 0);
      }
      __builtin_amdgcn_global_load_lds((bc_gptr_t)(wrow + tap * p.Cin + cc * 32), (bc_lptr_t)(sB + jb * 1024), 16, 0, 0);
    } else {
      const int piece = step * 4 + dq;
      const int tap = ppt == 1 ? piece : piece >> 1;
      const int sub = ppt == 1 ? 0 : piece & 1;
      const bool pv = piece < 9 * ppt;
      const int lo = arow + sub * 8;
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) {
        const int g = wave * 4 + gi;
        const int off = pv ? tab[g * BC_TABW + tap] : -1;
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
  if (nsteps > 1) issue(1, 1);
  int buf = 0;
  for (int i = 0; i < nsteps; ++i) {
    // five DMA instructions per wave and stage: all but the youngest stage have landed
    if (i + 1 < nsteps)
      asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (i + 2 < nsteps) issue(i + 2, buf >= 1 ? buf - 1 : 2);   // (buf + 2) % 3: the stage read in step i - 1
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
    buf = buf == 2 ? 0 : buf + 1;
  }

  // ---- epilogue: lane (c, g4) owns channels ch0 .. ch0+NBLK-1 of stamps 4*g4 .. 4*g4+3 of each of its 4 groups ----
  const int c = lane & 15, g4 = lane >> 4;
  const int ch0 = n0 + NBLK * c;
  float bias[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) bias[j] = 0.f;
  if (p.bias && (p.epi == BEPI_FWD || p.epi == BEPI_RAW32)) load_f32<NBLK>(p.bias + ch0, bias);
  float dal[NBLK], db[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) dal[j] = db[j] = 0.f;
  int pix_w = 0, rb_w = -1;
#pragma unroll
  for (int gi = 0; gi < 4; ++gi) {
    const int G = wave * 4 + gi;
    const int rb = tab[G * BC_TABW + 9];
    if (rb < 0) continue;
    const int pix = tab[G * BC_TABW + 10];
    pix_w = pix;
    rb_w = rb;
    float al[NBLK];
    if (p.epi == BEPI_FWD || p.epi == BEPI_BWD) load_f32<NBLK>(p.alpha + (size_t)pix * p.Cout + ch0, al);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t e = ((size_t)rb + 4 * g4 + r) * p.Cout + ch0;
      float v[NBLK];
#pragma unroll
      for (int j = 0; j < NBLK; ++j) v[j] = acc[gi][j][r] + bias[j];
      if (p.epi == BEPI_RAW32) {
#pragma unroll
        for (int j = 0; j < NBLK; ++j) p.Uf[e + j] = v[j];
      } else if (p.epi == BEPI_RAWBF) {
        store_bf<NBLK>(reinterpret_cast<bc_bf16*>(p.U) + e, v);
      } else if (p.epi == BEPI_FWD) {
        float a[NBLK];
#pragma unroll
        for (int j = 0; j < NBLK; ++j) a[j] = v[j] > 0.f ? v[j] : al[j] * v[j];
        if (p.U) store_bf<NBLK>(reinterpret_cast<bc_bf16*>(p.U) + e, v);
        store_bf<NBLK>(reinterpret_cast<bc_bf16*>(p.A) + e, a);
      } else {
        float u[NBLK], du[NBLK];
        load_bf<NBLK>(reinterpret_cast<const bc_bf16*>(p.Uin) + e, u);
#pragma unroll
        for (int j = 0; j < NBLK; ++j) {
          du[j] = v[j] * (u[j] > 0.f ? 1.f : al[j]);
          dal[j] += v[j] * fminf(u[j], 0.f);
          db[j] += du[j];
        }
        store_bf<NBLK>(reinterpret_cast<bc_bf16*>(p.U) + e, du);
      }
    }
  }
  if (p.epi == BEPI_BWD && p.dal_part && rb_w >= 0) {
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

int launch_bconv(const BConvParams& p, hipStream_t s) {
  if (p.NBp <= 0 || (p.NBp & 15) || p.Cout % 16 || p.Kpad % 32) {
    set_error("bconv: bad geometry (NBp %d, Cout %d, Kpad %d)", p.NBp, p.Cout, p.Kpad);
    return E_INVALID;
  }
  const int mode = p.Cin % 32 == 0 ? 0 : ((p.Cin == 8 || p.Cin == 16) ? 1 : -1);
  if (mode < 0) {
    set_error("bconv: input channels must be 8, 16 or a multiple of 32 (got %d)", p.Cin);
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
  const int nblk = p.Cout % 64 == 0 ? 4 : (p.Cout % 32 == 0 ? 2 : 1);
  const long M16 = (long)p.Hout * p.Hout * (p.NBp >> 4);
  const long tiles = ((M16 + BC_GT - 1) / BC_GT) * (p.Cout / (16 * nblk));
  const size_t lds = (size_t)3 * (BC_GT + nblk) * 1024 + 1024;
#define BC_LAUNCH(NB_, MODE_)                                                                          \
  do {                                                                                                 \
    static bool attr_done = false;                                                                     \
    if (!attr_done) {                                                                                  \
      DV_HIP(hipFuncSetAttribute((const void*)bconv_kernel<NB_, MODE_>,                                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));               \
      attr_done = true;                                                                                \
    }                                                                                                  \
    hipLaunchKernelGGL((bconv_kernel<NB_, MODE_>), dim3((unsigned)tiles), dim3(256), lds, s, p);       \
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
#undef BC_LAUNCH
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
