// Stride-2 data-gradient-form gather-GEMM with the four output-parity classes fused in one workgroup.
//
// A stride-2 Conv2DTranspose forward (reference: src/debvader/model/model.py:86-110, the decoder's upsampling
// layers) and the data gradient of a stride-2 Conv2D (model.py:33-70, the encoder's downsampling layers) are the
// same contraction: output pixel (2i+ph, 2j+pw) gathers the source pixels (i+dh, j+dw) for the kernel taps of
// matching parity.  Per dimension one parity has two taps (dh in {0, x}) and the other has one (dh = 0), so the
// four classes (ph, pw) have 4 / 2 / 2 / 1 taps and K = ntaps*Cin differs per class.  gconv2.hip runs the classes as
// separate tiles of one launch; their K loops are 1-4 taps short and the prologue / epilogue dominates (50-60 TF
// against 100+ for the stride-1 layers).
//
// Here a workgroup owns a tile of BASE pixels (i, j) and keeps accumulators for all four classes.  The K loop walks
// the 2x2 source neighbourhood {0,x}x{0,x}: one gathered A chunk (neighbour e, 32 channels) feeds every class that
// uses that neighbour (1, 2, 2 or 4 of them) against that class's weight tap, so
//   * the K loop has 4*Cin/32 steps carrying 9*Cin/32 tile-products: as long as a stride-1 layer's loop;
//   * A is gathered 4 times per base pixel instead of 9;
//   * the epilogue writes complete 2x2 output blocks.
// Classes are numbered c = 2*[ph has two taps] + [pw has two taps]; neighbour e = 2*[dh == x] + [dw == x] is used
// by class c iff (e & ~c) == 0.  Phases run e = 3, 2, 1, 0 (lightest first: the exposed prologue load is the small
// one), each phase a compile-time class set so loads and MFMAs are straight-line code.
#include "common.h"
#include <stdlib.h>

namespace dv {

namespace {
constexpr int BKS = 32;
template <int V>
struct IC {
  static constexpr int value = V;
};
__host__ __device__ constexpr int phase_nbr(int ph) { return 3 - ph; }
__host__ __device__ constexpr int nbr_mask(int e) {   // classes that use neighbour e
  return (e == 0) ? 15 : (e == 1) ? 10 : (e == 2) ? 12 : 8;
}
__host__ __device__ constexpr int popc4(int m) { return (m & 1) + ((m >> 1) & 1) + ((m >> 2) & 1) + ((m >> 3) & 1); }
}  // namespace

// TMW: 16-row MFMA blocks per wave along M (2: 32 base pixels per wave; 1: 16, half the LDS and accumulators, so
// three workgroups fit a CU)
template <int WGM, int WGN, int TMW = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void gconv_s2_kernel(const GConvS2Params p) {
  constexpr int NW = WGM * WGN, NT = 64 * NW;
  constexpr int WMR = 16 * TMW;                   // rows per wave
  constexpr int BM = WMR * WGM, BN = 32 * WGN;
  constexpr int RP = NT / 8;                      // tile rows filled per pass (8 lanes x 16 B per row)
  constexpr int AROWS = BM / RP, BROWS = BN / RP;
  constexpr int A_ELEMS = BM * BKS, B_ELEMS = BN * BKS;
  constexpr int STAGE = A_ELEMS + 4 * B_ELEMS;
  constexpr int LDC = 36;                         // epilogue staging row stride (floats)
  static_assert(NW * WMR * LDC <= 2 * STAGE, "staging must fit in the operand buffers");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  int* s_in = reinterpret_cast<int*>(smem + 2 * STAGE);   // [BM] gather base (elements)
  int* s_nm = s_in + BM;                                   // [BM] bits 0-3 neighbour valid, bits 4-7 class valid
  int* s_out = s_nm + BM;                                  // [BM] output offset of the 2x2 block's (0,0) pixel
  int* s_al = s_out + BM;                                  // [BM] same, inside one stamp (alpha)

  const unsigned long long tl0 = p.dbg_out ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * WMR, wn0 = (wave % WGN) * 32;
  const int l15 = lane & 15, lg = lane >> 4;

  int bid = blockIdx.x;
  {  // contiguous tile range per XCD (see gconv2.hip)
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  }
  const int ntn = (p.Cout + BN - 1) / BN;
  const int m0 = (bid / ntn) * BM;
  const int n0 = (bid % ntn) * BN;

  if (tid < BM) {
    const int m = m0 + tid;
    int in_off = 0, nm = 0, out_off = 0, al_off = 0;
    if (m < p.M) {
      const int HW = p.Hc * p.Wc;
      const int nb = m / HW, rem = m - nb * HW;
      const int ii = rem / p.Wc, jj = rem - ii * p.Wc;
      in_off = ((nb * p.Hin + ii) * p.Win + jj) * p.Cin;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ih = ii + p.ndh[e], iw = jj + p.ndw[e];
        if ((unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) nm |= 1 << e;
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (2 * ii + p.cph[c] < p.Hout && 2 * jj + p.cpw[c] < p.Wout) nm |= 16 << c;
      al_off = (2 * ii * p.Wout + 2 * jj) * p.Cout;
      out_off = nb * p.Hout * p.Wout * p.Cout + al_off;
    }
    s_in[tid] = in_off;
    s_nm[tid] = nm;
    s_out[tid] = out_off;
    s_al[tid] = al_off;
  }
  __syncthreads();

  const int kq = tid & 7, r0 = tid >> 3;
  int rin[AROWS];
  unsigned rnm = 0;                                // 4 neighbour bits per A row of this thread
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    rin[i] = s_in[r0 + RP * i] + kq * 4;
    rnm |= (unsigned)(s_nm[r0 + RP * i] & 15) << (4 * i);
  }
  int wthr[BROWS];
  unsigned wok = 0;
#pragma unroll
  for (int i = 0; i < BROWS; ++i) {
    const int n = n0 + r0 + RP * i;
    const bool ok = n < p.Cout;
    wthr[i] = ok ? n * p.Cin + kq * 4 : 0;
    wok |= (ok ? 1u : 0u) << i;
  }

  f32x4 areg[AROWS];
  f32x4 breg[4][BROWS];
  unsigned amask = 0;
  const int cpt = p.Cin / BKS;
  const int CC = p.Cout * p.Cin;

  auto load_global = [&](auto phase, int cc) {
    constexpr int E = phase_nbr(decltype(phase)::value);
    constexpr int MK = nbr_mask(E);
    const int tapoff = (p.ndh[E] * p.Win + p.ndw[E]) * p.Cin + cc * BKS;
    amask = 0;
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      const bool ok = (rnm >> (4 * i + E)) & 1;
      const unsigned off = ok ? (unsigned)(rin[i] + tapoff) : 0u;
      areg[i] = *reinterpret_cast<const f32x4*>(p.X + off);
      amask |= (ok ? 1u : 0u) << i;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if ((MK >> c) & 1) {
        const int wbase = p.wt[E][c] * CC + cc * BKS;
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
          const unsigned off = ((wok >> i) & 1u) ? (unsigned)(wbase + wthr[i]) : 0u;
          breg[c][i] = *reinterpret_cast<const f32x4*>(p.W + off);
        }
      }
    }
  };

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int sw_w = (r0 >> 1) & 7;                 // RP is a multiple of 16: same swizzle for every pass
  auto store_lds = [&](auto phase, int buf) {
    constexpr int MK = nbr_mask(phase_nbr(decltype(phase)::value));
    float* a = smem + buf * STAGE;
#pragma unroll
    for (int i = 0; i < AROWS; ++i)
      *reinterpret_cast<f32x4*>(a + (r0 + RP * i) * BKS + ((kq ^ sw_w) << 2)) = ((amask >> i) & 1u) ? areg[i] : zero4;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if ((MK >> c) & 1) {
        float* b = a + A_ELEMS + c * B_ELEMS;
#pragma unroll
        for (int i = 0; i < BROWS; ++i)
          *reinterpret_cast<f32x4*>(b + (r0 + RP * i) * BKS + ((kq ^ sw_w) << 2)) = ((wok >> i) & 1u) ? breg[c][i] : zero4;
      }
    }
  };

  f32x4 acc[4][TMW][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int a = 0; a < TMW; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[c][a][b] = zero4;

  const int sw_r = l15 >> 1;
  auto compute = [&](auto phase, int buf) {
    constexpr int MK = nbr_mask(phase_nbr(decltype(phase)::value));
    const float* a = smem + buf * STAGE;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int quad = ((q * 4 + lg) ^ sw_r) << 2;
      f32x4 af[TMW], bf[4][2];
#pragma unroll
      for (int tm = 0; tm < TMW; ++tm) af[tm] = *reinterpret_cast<const f32x4*>(a + (wm0 + tm * 16 + l15) * BKS + quad);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if ((MK >> c) & 1) {
#pragma unroll
          for (int tn = 0; tn < 2; ++tn)
            bf[c][tn] = *reinterpret_cast<const f32x4*>(a + A_ELEMS + c * B_ELEMS + (wn0 + tn * 16 + l15) * BKS + quad);
        }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if ((MK >> c) & 1) {
#pragma unroll
            for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
              for (int tn = 0; tn < 2; ++tn)
                acc[c][tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm][jj], bf[c][tn][jj], acc[c][tm][tn], 0, 0, 0);
          }
    }
  };

  // one K step of phase PH whose prefetch belongs to phase PN: a single basic block, the gather's VALU / VMEM
  // instructions spread between the MFMAs
  auto step = [&](auto ph, auto pn, int cc_next, int cur) {
    constexpr int NMF = 16 * TMW * popc4(nbr_mask(phase_nbr(decltype(ph)::value)));
    load_global(pn, cc_next);
    compute(ph, cur);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, NMF / 8, 0);   // MFMA
      __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);         // VMEM read
      __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);        // VALU
    }
    store_lds(pn, cur ^ 1);
    __syncthreads();
  };

  int cur = 0;
  load_global(IC<0>{}, 0);
  store_lds(IC<0>{}, 0);
  __syncthreads();
  const unsigned long long tl1 = p.dbg_out ? __builtin_amdgcn_s_memrealtime() : 0ull;
  auto run_phase = [&](auto ph, auto pn, bool last) {
    for (int cc = 0; cc + 1 < cpt; ++cc) {
      step(ph, ph, cc + 1, cur);
      cur ^= 1;
    }
    if (!last) {
      step(ph, pn, 0, cur);
      cur ^= 1;
    } else {
      compute(ph, cur);
      __syncthreads();
    }
  };
  run_phase(IC<0>{}, IC<1>{}, false);
  run_phase(IC<1>{}, IC<2>{}, false);
  run_phase(IC<2>{}, IC<3>{}, false);
  run_phase(IC<3>{}, IC<3>{}, true);
  const unsigned long long tl2 = p.dbg_out ? __builtin_amdgcn_s_memrealtime() : 0ull;

  // ---- epilogue: per class, accumulators -> per-wave LDS staging (32 x 32) -> float4 rows -------------------
  // The PReLU slopes of class c+1 are loaded while class c is staged and stored (the loop is unrolled, the two
  // register sets alternate): a class's epilogue otherwise starts with a global-load latency nothing hides.
  float* stg = smem + wave * (WMR * LDC);
  constexpr int RPLE = WMR / 8;                    // rows per lane in the epilogue
  const int f4 = lane & 7;
  const int col = n0 + wn0 + f4 * 4;
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (p.epi >= 1 && col < p.Cout) bias4 = *reinterpret_cast<const f32x4*>(p.bias + col);
  int orow[RPLE], arow[RPLE];                            // output / alpha offsets of this lane's four rows (class (0,0) pixel)
  unsigned rowok = 0;                              // bits 4*i + c: row i stores class c
#pragma unroll
  for (int i = 0; i < RPLE; ++i) {
    const int row = wm0 + (lane >> 3) + 8 * i;
    orow[i] = s_out[row];
    arow[i] = s_al[row];
    rowok |= (unsigned)((s_nm[row] >> 4) & 15) << (4 * i);
  }
  if (col >= p.Cout) rowok = 0;
  f32x4 alr[2][RPLE];
  auto load_alpha = [&](int c, f32x4 (&dst)[RPLE]) {
    const int coff = (p.cph[c] * p.Wout + p.cpw[c]) * p.Cout;
#pragma unroll
    for (int i = 0; i < RPLE; ++i) {
      const bool ok = (rowok >> (4 * i + c)) & 1;
      dst[i] = *reinterpret_cast<const f32x4*>(p.alpha + (ok ? (unsigned)(arow[i] + coff + col) : 0u));
    }
  };
  if (p.epi == 2) load_alpha(0, alr[0]);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    if (p.epi == 2 && c + 1 < 4) load_alpha(c + 1, alr[(c + 1) & 1]);
#pragma unroll
    for (int tm = 0; tm < TMW; ++tm)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 4; ++r) stg[(tm * 16 + lg * 4 + r) * LDC + tn * 16 + l15] = acc[c][tm][tn][r];
    __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): wave-private region
    __builtin_amdgcn_wave_barrier();
    const int coff = (p.cph[c] * p.Wout + p.cpw[c]) * p.Cout;
#pragma unroll
    for (int i = 0; i < RPLE; ++i) {
      const int rr = (lane >> 3) + 8 * i;
      if (!((rowok >> (4 * i + c)) & 1)) continue;
      const unsigned ooff = (unsigned)(orow[i] + coff + col);
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + rr * LDC + f4 * 4);
      v += bias4;
      if (p.U) *reinterpret_cast<f32x4*>(p.U + ooff) = v;
      if (p.epi == 2) {
        const f32x4 al = alr[c & 1][i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? v[k] : al[k] * v[k];
        *reinterpret_cast<f32x4*>(p.A + ooff) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (p.dbg_out && tid == 0 && blockIdx.x < (1 << 14)) {
    __builtin_amdgcn_s_waitcnt(0);
    unsigned* d = p.dbg_out + (size_t)blockIdx.x * 4;
    d[0] = (unsigned)tl0; d[1] = (unsigned)tl1; d[2] = (unsigned)tl2; d[3] = (unsigned)__builtin_amdgcn_s_memrealtime();
  }
}

template <int WGM, int WGN, int TMW = 2>
static int launch_s2_cfg(const GConvS2Params& p, hipStream_t s) {
  constexpr int BM = 16 * TMW * WGM, BN = 32 * WGN;
  constexpr size_t smem = (size_t)2 * (BM * BKS + 4 * BN * BKS) * sizeof(float) + 4 * BM * sizeof(int);
  static bool attr_set = false;
  auto kern = gconv_s2_kernel<WGM, WGN, TMW>;
  if (!attr_set) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr_set = true;
  }
  const long tiles = (long)((p.M + BM - 1) / BM) * ((p.Cout + BN - 1) / BN);
  if (tiles == 0) return OK;
  hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64 * WGM * WGN), smem, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

static int s2_tile_override = -1;
void debug_set_gconv_s2_tile(int code) { s2_tile_override = code; }
static unsigned* s2_dbg_out = nullptr;
void debug_set_gconv_s2_dbg(unsigned* out) { s2_dbg_out = out; }

// tile codes: 0 = 64 x 64 (2x2 waves), 1 = 128 x 32 (4x1), 2 = 64 x 32 (2x1), 3 = 32 x 64 (1x2)
int launch_gconv_s2(const GConvS2Params& p0, hipStream_t s) {
  GConvS2Params p = p0;
  p.dbg_out = s2_dbg_out;
  if ((p.Cin % BKS) || (p.Cout & 3) || p.M <= 0) {
    set_error("gconv_s2: unsupported shape (Cin=%d Cout=%d M=%d)", p.Cin, p.Cout, p.M);
    return E_INVALID;
  }
  if ((long)p.NB * p.Hin * p.Win * p.Cin >= (1L << 30) || (long)p.NB * p.Hout * p.Wout * p.Cout >= (1L << 30) ||
      (long)9 * p.Cin * p.Cout >= (1L << 30)) {
    set_error("gconv_s2: tensor too large for 32-bit offsets; lower max_batch");
    return E_INVALID;
  }
  if (p.epi == 2 && (!p.alpha || !p.A)) {
    set_error("gconv_s2: PReLU epilogue needs alpha and A");
    return E_INVALID;
  }
  if (p.epi >= 1 && !p.bias) {
    set_error("gconv_s2: bias epilogue without bias");
    return E_INVALID;
  }
  int t = s2_tile_override;
  // 128 x 32 (four waves stacked along M, all sharing the class's 32-column weight tile) measured fastest on every
  // stride-2 layer of the network (tools/layer_bench.py over the tile forms): 64 x 64 is 20-45 % slower
  if (t < 0) {
    // deep layers with few base pixels: half-height workgroups (64 x 32, three per CU) double the workgroup count
    // (4x4x256 -> 8x8x256: 66 -> 54 us); everywhere else the 128 x 32 tile wins
    const long wgs128 = (long)((p.M + 127) / 128) * ((p.Cout + 31) / 32);
    t = wgs128 < 512 ? 4 : 1;
  }
  switch (t) {
    case 0: return launch_s2_cfg<2, 2>(p, s);
    case 1: return launch_s2_cfg<4, 1>(p, s);
    case 2: return launch_s2_cfg<2, 1>(p, s);
    case 4: return launch_s2_cfg<4, 1, 1>(p, s);
    default: return launch_s2_cfg<1, 2>(p, s);
  }
}

}  // namespace dv
