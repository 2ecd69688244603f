// Gather-GEMM, second generation: the main-loop kernel for every layer whose input channel count is a
// multiple of 32 (all but the first conv, the head's data gradient and the 560-wide dense operands).
//
// Same contraction as gconv.hip (see common.h); what changes is everything around the MFMAs:
//  * one launch covers all output-parity classes of a stride-2 Conv2DTranspose / conv data gradient
//    (blockIdx -> class, heaviest class first), instead of four launches with four tails;
//  * a K chunk of 32 is one (tap, 32-channel slab), so the tap is wave-uniform: per-row gather state is a
//    32-bit element offset plus a 9-bit tap-validity mask computed ONCE per workgroup (LDS row table), and the
//    per-chunk address work is one add per row;
//  * A (and n-major B) tiles sit in LDS as [row][32] with the 16-byte quad index XOR-swizzled by (row>>1)&7:
//    no padding, and the ds_read_b128 fragment reads of v_mfma_f32_16x16x4_f32 are bank-conflict free;
//  * the epilogue goes through a per-wave LDS staging tile so that bias / PReLU / stores run on float4 rows
//    (256-byte contiguous segments per output pixel) instead of 64-byte column fragments.
#include "common.h"

namespace dv {

namespace {
constexpr int BK2 = 32;
template <int V>
struct SetC {
  static constexpr int value = V;
};

__device__ __forceinline__ int t_dh(unsigned long long code, int t) { return (int)((code >> (4 * t)) & 3) - 1; }
__device__ __forceinline__ int t_dw(unsigned long long code, int t) { return (int)((code >> (4 * t + 2)) & 3) - 1; }
__device__ __forceinline__ int t_wt(unsigned long long code, int t) { return (int)((code >> (4 * t)) & 15); }
}  // namespace

// SUB = taps per 32-wide K chunk: 1 for Cin % 32 == 0, 2 for Cin == 16, 4 for Cin == 8 (tap then varies per lane)
// RAG: Cin is a multiple of 4 but not of 32 (the 560-wide dense operands, model.py:96-98,113-117): the last chunk of a tap
// is ragged, its missing quads / rows are zero-filled through the same masks that handle out-of-image rows
template <int BM, int BN, int WGM, int WGN, bool NMAJOR, int SUB = 1, bool RAG = false>
// occupancy target: two workgroups per CU for the 128 x 128 tile (LDS-limited anyway), three for the smaller ones -
// with the second register set the compiler otherwise settles just above the 168-register line of three waves / SIMD
__global__ __launch_bounds__(256, (BM * BN >= 128 * 128 || BM >= 256) ? 2 : 3) void gconv2_kernel(const GConv2Params p) {
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 16, TN = WN / 16;
  constexpr int AROWS = BM / 32;
  constexpr int LDBK = BN + 4;
  constexpr int A_ELEMS = BM * BK2;
  constexpr int B_ELEMS = NMAJOR ? BN * BK2 : BK2 * LDBK;
  constexpr int BROWS_N = (BN + 31) / 32;
  constexpr int BQ = BN / 4;
  constexpr int BKR = 256 / BQ;
  constexpr int BPASS = (BK2 + BKR - 1) / BKR;
  constexpr int LDC = WN + 4;                    // staging row stride (floats)
  constexpr int STG_ROWS = 32;                   // rows staged per pass and wave (2 MFMA row blocks)
  constexpr int CINS = BK2 / SUB;                // channels per tap inside a chunk (SUB > 1: equals Cin)
  constexpr int QPT = 8 / SUB;                   // 16-byte quads per tap inside a chunk
  static_assert(4 * STG_ROWS * LDC <= 2 * A_ELEMS + 2 * B_ELEMS, "staging must fit in the operand buffers");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * A_ELEMS;
  int* s_in = reinterpret_cast<int*>(smem + 2 * A_ELEMS + 2 * B_ELEMS);  // [BM] gather base (elements)
  int* s_mask = s_in + BM;                                                // [BM] tap validity bits
  int* s_out = s_mask + BM;                                               // [BM] output pixel offset (elements) or -1
  int* s_al = s_out + BM;                                                 // [BM] alpha pixel offset

  const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
  const int l15 = lane & 15, lg = lane >> 4;

  // ---- which class / tile ---------------------------------------------------------------------
  // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2).  Neighbouring tiles share input
  // rows (3x3 halo, the N tiles of one M panel), so give every XCD a contiguous range of tiles: block b of
  // XCD group b%8 takes tile (b%8)*ceil + b/8 (bijective form for grids that are not a multiple of 8).
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, loc = bid >> 3;
    bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  }
  int c = 0;
#pragma unroll
  for (int k = 1; k < 4; ++k)
    if (k < p.nclass && bid >= p.cls[k].tile0) c = k;
  const GClass2 cl = p.cls[c];
  const int ntn = (p.Cout + BN - 1) / BN;
  const int t = bid - cl.tile0;
  const int m0 = (t / ntn) * BM;
  const int n0 = (t % ntn) * BN;

  // ---- row table (one division pair per ROW of the tile, not per thread and chunk) -------------
  if (tid < BM) {
    const int m = m0 + tid;
    int in_off = 0, mask = 0, out_off = -1, al_off = 0;
    if (m < cl.M) {
      const int HcWc = cl.Hc * cl.Wc;
      // row-major tiles walk (stamp, pixel); batch-major tiles walk (pixel, stamp): all rows of a tile then share
      // one output pixel, which lets the epilogue reduce d(alpha) / d(bias) over the stamps of the tile
      const int nb = p.batch_major ? m % p.NB : m / HcWc;
      const int rem = p.batch_major ? m / p.NB : m - nb * HcWc;
      const int ii = rem / cl.Wc;
      const int jj = rem - ii * cl.Wc;
      const int ih0 = ii * p.sin, iw0 = jj * p.sin;
      in_off = ((nb * p.Hin + ih0) * p.Win + iw0) * p.Cin;
      for (int tp = 0; tp < cl.ntaps; ++tp) {
        const int ih = ih0 + t_dh(cl.tapcode, tp), iw = iw0 + t_dw(cl.tapcode, tp);
        if ((unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) mask |= 1 << tp;
      }
      const int oh = ii * p.sout + cl.ph, ow = jj * p.sout + cl.pw;
      al_off = (oh * p.Wout + ow) * p.Cout;
      out_off = nb * p.Hout * p.Wout * p.Cout + al_off;
    }
    s_in[tid] = in_off;
    s_mask[tid] = mask;
    s_out[tid] = out_off;
    s_al[tid] = al_off;
  }
  __syncthreads();

  const int kq = tid & 7, r0 = tid >> 3;
  int rin[AROWS], rmask[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    rin[i] = s_in[r0 + 32 * i] + (SUB == 1 ? kq : kq % QPT) * 4;
    rmask[i] = s_mask[r0 + 32 * i];
  }
  // weight addressing: thread-constant part
  int wthr[NMAJOR ? BROWS_N : BPASS];
  unsigned wok = 0;
  if (NMAJOR) {
#pragma unroll
    for (int i = 0; i < BROWS_N; ++i) {
      const int n = n0 + r0 + 32 * i;
      const bool ok = n < p.Cout && r0 + 32 * i < BN;
      wthr[i] = ok ? n * p.Cin + (SUB == 1 ? kq : kq % QPT) * 4 : 0;
      wok |= (ok ? 1u : 0u) << i;
    }
  } else {
    const int nq = tid % BQ, kr0 = tid / BQ;
#pragma unroll
    for (int i = 0; i < BPASS; ++i) {
      const int kr = kr0 + BKR * i;
      const int n = n0 + nq * 4;
      const bool ok = kr < BK2 && n < p.Cout;
      wthr[i] = ok ? (SUB == 1 ? kr : kr % CINS) * p.Cout + n : 0;
      wok |= (ok ? 1u : 0u) << i;
    }
  }

  // two register sets: a chunk's global loads are issued TWO iterations before its LDS store (SUB == 1 loop), so a
  // lone workgroup is not paced by the L2 / HBM latency of its gather
  f32x4 areg[2][AROWS];
  f32x4 breg[2][NMAJOR ? BROWS_N : BPASS];
  unsigned amask2[2] = {0, 0}, bmaskv2[2] = {0xffffffffu, 0xffffffffu};
  const int cpt = SUB == 1 ? (RAG ? (p.Cin + BK2 - 1) / BK2 : p.Cin / BK2) : 1;     // chunks per tap
  int tap = 0, cc = 0;                            // chunk -> (tap, channel slab), advanced incrementally

  auto load_global = [&](auto setc) {             // loads chunk (tap, cc) into register set S, then advances
    constexpr int S = decltype(setc)::value;
    unsigned amask = 0, bmaskv = 0xffffffffu;
    if constexpr (SUB == 1) {
      const int dh = t_dh(cl.tapcode, tap), dw = t_dw(cl.tapcode, tap);
      const int wt = t_wt(cl.wtcode, tap);
      const int tapoff = (dh * p.Win + dw) * p.Cin + cc * BK2;
      const bool kv = !RAG || cc * BK2 + kq * 4 < p.Cin;       // this lane's channel quad exists
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const bool ok = ((rmask[i] >> tap) & 1) && kv;
        const unsigned off = ok ? (unsigned)(rin[i] + tapoff) : 0u;
        areg[S][i] = *reinterpret_cast<const f32x4*>(p.X + off);
        amask |= (ok ? 1u : 0u) << i;
      }
      if (NMAJOR) {
        const int wbase = wt * p.Cout * p.Cin + cc * BK2;
        if (RAG) bmaskv = kv ? 0xffffffffu : 0u;
#pragma unroll
        for (int i = 0; i < BROWS_N; ++i) {
          const unsigned off = (((wok >> i) & 1u) && kv) ? (unsigned)(wbase + wthr[i]) : 0u;
          breg[S][i] = *reinterpret_cast<const f32x4*>(p.W + off);
        }
      } else {
        const int wbase = (wt * p.Cin + cc * BK2) * p.Cout;
        if (RAG) bmaskv = 0;
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
          const bool kvb = !RAG || cc * BK2 + (int)(tid / BQ) + BKR * i < p.Cin;   // this pass's weight row exists
          const unsigned off = (((wok >> i) & 1u) && kvb) ? (unsigned)(wbase + wthr[i]) : 0u;
          breg[S][i] = *reinterpret_cast<const f32x4*>(p.W + off);
          if (RAG) bmaskv |= (kvb ? 1u : 0u) << i;
        }
      }
      if (++cc == cpt) {
        cc = 0;
        ++tap;
      }
    } else {
      // several taps share one chunk: this lane's quad belongs to tap (tap + kq / QPT); taps past the table read 0
      const int mytap = tap + kq / QPT;
      const int tcl = min(mytap, 8);
      const int dh = t_dh(cl.tapcode, tcl), dw = t_dw(cl.tapcode, tcl);
      const int tapoff = (dh * p.Win + dw) * p.Cin;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        const bool ok = mytap < cl.ntaps && ((rmask[i] >> tcl) & 1);
        const unsigned off = ok ? (unsigned)(rin[i] + tapoff) : 0u;
        areg[S][i] = *reinterpret_cast<const f32x4*>(p.X + off);
        amask |= (ok ? 1u : 0u) << i;
      }
      bmaskv = 0;
      if (NMAJOR) {
        const int wt = t_wt(cl.wtcode, tcl);
#pragma unroll
        for (int i = 0; i < BROWS_N; ++i) {
          const bool ok = ((wok >> i) & 1u) && mytap < cl.ntaps;
          const unsigned off = ok ? (unsigned)(wt * p.Cout * p.Cin + wthr[i]) : 0u;
          breg[S][i] = *reinterpret_cast<const f32x4*>(p.W + off);
          bmaskv |= (ok ? 1u : 0u) << i;
        }
      } else {
        const int kr0b = tid / BQ;
#pragma unroll
        for (int i = 0; i < BPASS; ++i) {
          const int kr = kr0b + BKR * i;
          const int btap = tap + kr / CINS;
          const int bcl = min(btap, 8);
          const bool ok = ((wok >> i) & 1u) && btap < cl.ntaps;
          const unsigned off = ok ? (unsigned)(t_wt(cl.wtcode, bcl) * p.Cin * p.Cout + wthr[i]) : 0u;
          breg[S][i] = *reinterpret_cast<const f32x4*>(p.W + off);
          bmaskv |= (ok ? 1u : 0u) << i;
        }
      }
      tap += SUB;
    }
    amask2[S] = amask;
    bmaskv2[S] = bmaskv;
  };

  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const int sw_w = (r0 >> 1) & 7;                 // swizzle of the rows this thread writes (r0 + 32 i: same value)
  auto store_lds = [&](auto setc, int buf) {
    constexpr int S = decltype(setc)::value;
    const unsigned amask = amask2[S], bmaskv = bmaskv2[S];
    float* a = As + buf * A_ELEMS;
#pragma unroll
    for (int i = 0; i < AROWS; ++i)
      *reinterpret_cast<f32x4*>(a + (r0 + 32 * i) * BK2 + ((kq ^ sw_w) << 2)) = ((amask >> i) & 1u) ? areg[S][i] : zero4;
    float* b = Bs + buf * B_ELEMS;
    if (NMAJOR) {
#pragma unroll
      for (int i = 0; i < BROWS_N; ++i)
        if (r0 + 32 * i < BN)
          *reinterpret_cast<f32x4*>(b + (r0 + 32 * i) * BK2 + ((kq ^ sw_w) << 2)) = (((wok & bmaskv) >> i) & 1u) ? breg[S][i] : zero4;
    } else {
      const int nq = tid % BQ, kr0 = tid / BQ;
#pragma unroll
      for (int i = 0; i < BPASS; ++i) {
        const int kr = kr0 + BKR * i;
        if (kr < BK2) *reinterpret_cast<f32x4*>(b + kr * LDBK + nq * 4) = (((wok & bmaskv) >> i) & 1u) ? breg[S][i] : zero4;
      }
    }
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int a = 0; a < TM; ++a)
#pragma unroll
    for (int b = 0; b < TN; ++b) acc[a][b] = zero4;

  const int sw_r = l15 >> 1;                      // swizzle of the fragment rows this lane reads
  auto compute = [&](int buf) {
    const float* a = As + buf * A_ELEMS;
    const float* b = Bs + buf * B_ELEMS;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int quad = ((q * 4 + lg) ^ sw_r) << 2;
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        af[tm] = *reinterpret_cast<const f32x4*>(a + (wm0 + tm * 16 + l15) * BK2 + quad);
#pragma unroll
      for (int tn = 0; tn < TN; ++tn) {
        if (NMAJOR) {
          bf[tn] = *reinterpret_cast<const f32x4*>(b + (wn0 + tn * 16 + l15) * BK2 + quad);
        } else {
          const float* bp = b + (q * 16 + lg * 4) * LDBK + wn0 + tn * 16 + l15;
          bf[tn] = (f32x4){bp[0], bp[LDBK], bp[2 * LDBK], bp[3 * LDBK]};
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[tm][jj], bf[tn][jj], acc[tm][tn], 0, 0, 0);
    }
  };

  // split-K (dense layers with few output tiles): blockIdx.y owns a contiguous range of chunks and writes a raw
  // partial tile into slab blockIdx.y of U; splitk_finish_kernel sums the slabs in order and applies the epilogue
  int nchunks = SUB == 1 ? cl.ntaps * cpt : (cl.ntaps + SUB - 1) / SUB;
  float* Uout = p.U;
  if (p.ksplit > 1) {
    const int cps = (nchunks + p.ksplit - 1) / p.ksplit;
    const int kbeg = blockIdx.y * cps;
    nchunks = max(0, min(nchunks, kbeg + cps) - kbeg);
    tap = kbeg / cpt;
    cc = kbeg - tap * cpt;
    Uout += (size_t)blockIdx.y * p.NB * p.Hout * p.Wout * p.Cout;
  }
  const bool deep = p.dbg != 1 && p.prio != 4;   // distance-2 register prefetch (see areg / breg)
  if (nchunks > 0) {
    load_global(SetC<0>{});
    store_lds(SetC<0>{}, 0);
    if (deep && nchunks > 1) load_global(SetC<1>{});
  }
  __syncthreads();
  unsigned long long tl1 = 0, tl2 = 0;
  if (p.dbg == 2) tl1 = __builtin_amdgcn_s_memrealtime();
  if (deep) {
    // Steady-state iterations are one basic block: the global loads of chunk k+2 (into the register set chunk k
    // vacated), the MFMAs of chunk k, then the LDS store of chunk k+1, whose loads were issued one iteration
    // earlier.  The scheduler is asked to spread the gather's VALU / VMEM instructions between the MFMAs.
    // (Loads one chunk ahead paced a lone workgroup at 8.6k cycles per chunk against 4.1k of MFMA work.)
    auto iter = [&](auto ld, auto st, int cur) {
      load_global(ld);
      compute(cur);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);   // MFMA
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);         // VMEM read
        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);         // VALU
      }
      store_lds(st, cur ^ 1);
      __syncthreads();
    };
    int kc = 0;                                    // even here and after every pass of the loop
    for (; kc + 3 < nchunks; kc += 2) {
      iter(SetC<0>{}, SetC<1>{}, 0);               // chunk kc in buffer 0: load kc+2 -> set 0, store kc+1 (set 1)
      iter(SetC<1>{}, SetC<0>{}, 1);               // chunk kc+1 in buffer 1: load kc+3 -> set 1, store kc+2 (set 0)
    }
    int r = nchunks - kc;                          // 0 (nchunks == 0), 1, 2 or 3 chunks left, chunk kc in buffer 0
    if (r == 3) {
      iter(SetC<0>{}, SetC<1>{}, 0);
      compute(1);
      store_lds(SetC<0>{}, 0);
      __syncthreads();
      compute(0);
    } else if (r == 2) {
      compute(0);
      store_lds(SetC<1>{}, 1);
      __syncthreads();
      compute(1);
    } else if (r == 1) {
      compute(0);
    }
    __syncthreads();
  } else if (p.dbg != 1) {
    for (int kc = 0; kc < nchunks; ++kc) {
      const int cur = kc & 1;
      if (kc + 1 < nchunks) load_global(SetC<0>{});
      compute(cur);
      if (kc + 1 < nchunks) store_lds(SetC<0>{}, cur ^ 1);
      __syncthreads();
    }
  } else {  // timing build path: same loop with s_memtime stamps per phase (block 5 reports)
    unsigned long long tl = 0, tc = 0, ts = 0, tb = 0, t0 = __builtin_amdgcn_s_memtime(), t1;
    const unsigned long long tbeg = t0, rbeg = __builtin_amdgcn_s_memrealtime();
    for (int kc = 0; kc < nchunks; ++kc) {
      const int cur = kc & 1;
      if (kc + 1 < nchunks) load_global(SetC<0>{});
      t1 = __builtin_amdgcn_s_memtime(); tl += t1 - t0; t0 = t1;
      compute(cur);
      asm volatile("" ::"v"(acc[0][0]), "v"(acc[TM - 1][TN - 1]));
      t1 = __builtin_amdgcn_s_memtime(); tc += t1 - t0; t0 = t1;
      if (kc + 1 < nchunks) store_lds(SetC<0>{}, cur ^ 1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t1 = __builtin_amdgcn_s_memtime(); ts += t1 - t0; t0 = t1;
      __syncthreads();
      t1 = __builtin_amdgcn_s_memtime(); tb += t1 - t0; t0 = t1;
    }
    if (bid == 5 && lane == 0 && p.dbg_out) {
      float* d = p.dbg_out + wave * 8;
      d[0] = (float)tl; d[1] = (float)tc; d[2] = (float)ts; d[3] = (float)tb; d[4] = (float)nchunks;
      d[5] = (float)(__builtin_amdgcn_s_memtime() - tbeg); d[6] = (float)(__builtin_amdgcn_s_memrealtime() - rbeg);
    }
  }

  if (p.dbg == 2) tl2 = __builtin_amdgcn_s_memrealtime();
  // ---- epilogue: accumulators -> per-wave LDS staging -> float4 rows --------------------------
  float* stg = smem + wave * (STG_ROWS * LDC);    // operand buffers are free after the last barrier
  constexpr int F4_PER_ROW = WN / 4;
  constexpr int ROWS_PER_IT = 64 / F4_PER_ROW;
  constexpr int TM_PER_PASS = STG_ROWS / 16;
  f32x4 pdal = {0.f, 0.f, 0.f, 0.f}, pdb = {0.f, 0.f, 0.f, 0.f};   // epi 3: this lane's partial sums over its rows
#pragma unroll
  for (int pass = 0; pass < (TM + TM_PER_PASS - 1) / TM_PER_PASS; ++pass) {
#pragma unroll
    for (int tl = 0; tl < TM_PER_PASS; ++tl) {
      const int tm = pass * TM_PER_PASS + tl;
      if (tm < TM) {
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
          for (int r = 0; r < 4; ++r) stg[(tl * 16 + lg * 4 + r) * LDC + tn * 16 + l15] = acc[tm][tn][r];
      }
    }
    // wave-private region: only this wave's own LDS writes have to land before it reads them back
    __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    const int f4 = lane % F4_PER_ROW;
    const int col = n0 + wn0 + f4 * 4;
#pragma unroll
    for (int rr = lane / F4_PER_ROW; rr < STG_ROWS; rr += ROWS_PER_IT) {
      const int tm = pass * TM_PER_PASS + rr / 16;
      if (tm >= TM) break;
      const int row = wm0 + tm * 16 + (rr & 15);  // row inside the workgroup tile
      const int ooff = s_out[row];
      if (ooff < 0 || col >= p.Cout) continue;
      f32x4 v = *reinterpret_cast<const f32x4*>(stg + rr * LDC + f4 * 4);
      if (p.epi == 3) {
        // fused PReLU backward of the layer that produced this tensor: v is d(activation); write
        // d(pre-activation) = v * (u > 0 ? 1 : alpha) and accumulate the parameter-gradient partials
        const f32x4 uu = *reinterpret_cast<const f32x4*>(p.Uin + (unsigned)ooff + col);
        const f32x4 al = *reinterpret_cast<const f32x4*>(p.alpha + (unsigned)s_al[row] + col);
        f32x4 d;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool pos = uu[k] > 0.f;
          d[k] = pos ? v[k] : v[k] * al[k];
          pdal[k] += pos ? 0.f : v[k] * uu[k];
          pdb[k] += d[k];
        }
        *reinterpret_cast<f32x4*>(Uout + (unsigned)ooff + col) = d;
        continue;
      }
      if (p.epi >= 1) v += *reinterpret_cast<const f32x4*>(p.bias + col);
      if (Uout) *reinterpret_cast<f32x4*>(Uout + (unsigned)ooff + col) = v;
      if (p.epi == 2) {
        const f32x4 al = *reinterpret_cast<const f32x4*>(p.alpha + (unsigned)s_al[row] + col);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? v[k] : al[k] * v[k];
        *reinterpret_cast<f32x4*>(p.A + (unsigned)ooff + col) = o;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (p.epi == 3 && p.dal_part) {
    // lanes with the same column quad (lane % F4_PER_ROW) hold partial sums over different rows: butterfly them.
    // Every row of the tile is the same output pixel (batch-major tiles), so the sums are that pixel's
    // d(alpha) / d(bias) contributions of this wave's stamps; slots are summed later in a fixed order.
#pragma unroll
    for (int o = F4_PER_ROW; o < 64; o <<= 1) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pdal[k] += __shfl_xor(pdal[k], o, 64);
        pdb[k] += __shfl_xor(pdb[k], o, 64);
      }
    }
    const int colq = n0 + wn0 + (lane % F4_PER_ROW) * 4;
    if (lane < F4_PER_ROW && colq < p.Cout && m0 < cl.M) {
      const int wmi = wave / WGN;
      const int slot = ((m0 % p.NB) / BM) * WGM + wmi;
      *reinterpret_cast<f32x4*>(p.dal_part + (size_t)slot * p.alpha_elems + (unsigned)s_al[0] + colq) = pdal;
      if (p.db_part)
        *reinterpret_cast<f32x4*>(p.db_part + ((size_t)(bid / ntn) * WGM + wmi) * p.Cout + colq) = pdb;
    }
  }
  if (p.dbg == 2 && p.dbg_out && tid == 0 && blockIdx.y == 0) {
    __builtin_amdgcn_s_waitcnt(0);      // this wave's stores have been acknowledged
    unsigned* d = reinterpret_cast<unsigned*>(p.dbg_out) + (size_t)blockIdx.x * 4;
    d[0] = (unsigned)tl_start; d[1] = (unsigned)tl1; d[2] = (unsigned)tl2; d[3] = (unsigned)__builtin_amdgcn_s_memrealtime();
  }
}

template <int BM, int BN, int WGM, int WGN, bool NMAJOR, int SUB = 1, bool RAG = false>
static int launch2_cfg(GConv2Params p, hipStream_t s) {
  constexpr int A_ELEMS = BM * BK2;
  constexpr int B_ELEMS = NMAJOR ? BN * BK2 : BK2 * (BN + 4);
  constexpr size_t smem = (size_t)(2 * A_ELEMS + 2 * B_ELEMS) * sizeof(float) + 4 * BM * sizeof(int);
  static bool attr_set = false;
  auto kern = gconv2_kernel<BM, BN, WGM, WGN, NMAJOR, SUB, RAG>;
  if (!attr_set) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr_set = true;
  }
  const int ntn = (p.Cout + BN - 1) / BN;
  int tiles = 0;
  for (int c = 0; c < p.nclass; ++c) {
    p.cls[c].tile0 = tiles;
    tiles += ((p.cls[c].M + BM - 1) / BM) * ntn;
  }
  if (tiles == 0) return OK;
  hipLaunchKernelGGL(kern, dim3(tiles, p.ksplit > 1 ? p.ksplit : 1), dim3(256), smem, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

static int g2_tile_override = -1;
void debug_set_gconv2_tile(int code) { g2_tile_override = code; }

// tile choice (code -> BM x BN, waves along M): 0 128x128 (2), 1 128x64 (2), 2 64x64 (2), 3 128x32 (4), 4 64x128 (2),
// 5 128x16 (4), 6 256x32 (4), 7 256x16 (4)
static int choose_tile(const GConv2Params& p) {
  const int N = p.Cout;
  if (p.Cin % BK2 && p.Cin > 32) return 2;       // ragged-K form exists for the 64 x 64 tile only
  if (p.Cin == 8 || p.Cin == 16) return N <= 16 ? 5 : 3;
  if (g2_tile_override >= 0 && g2_tile_override <= 7) return g2_tile_override;
  long Mtot = 0;
  for (int c = 0; c < p.nclass; ++c) Mtot += p.cls[c].M;
  if (N <= 16) return 5;
  if (N <= 32) return 3;
  if (N <= 64) return (Mtot + 127) / 128 >= 384 ? 1 : 2;
  // aim for at least two resident workgroups per CU before growing the tile; parity-class launches have
  // short K loops (1-4 taps), so they want twice as many, smaller tiles (measured per layer, tools/layer_bench.py)
  const long want = p.nclass > 1 ? 1024 : 384;   // 1.5 workgroups per CU (re-measured with the two-chunk prefetch)
  const long t128 = ((Mtot + 127) / 128) * ((N + 127) / 128);
  if (t128 >= want) return 0;
  const long t12864 = ((Mtot + 127) / 128) * ((N + 63) / 64);
  if (t12864 >= want) return 1;
  return 2;
}

void gconv2_tile_geometry(const GConv2Params& p, int* bm, int* wgm, long* mtiles) {
  static const int BMs[8] = {128, 128, 64, 128, 64, 128, 256, 256};
  static const int WGMs[8] = {2, 2, 2, 4, 2, 4, 4, 4};
  const int t = choose_tile(p);
  *bm = BMs[t];
  *wgm = WGMs[t];
  long mt = 0;
  for (int c = 0; c < p.nclass; ++c) mt += (p.cls[c].M + BMs[t] - 1) / BMs[t];
  *mtiles = mt;
}

template <bool NMAJOR>
static int dispatch2(const GConv2Params& p, hipStream_t s) {
  const int t = choose_tile(p);
  if (p.Cin % BK2 && p.Cin > 32) return launch2_cfg<64, 64, 2, 2, NMAJOR, 1, true>(p, s);   // ragged K (dense, 560 wide)
  if (p.Cin == 8) return t == 5 ? launch2_cfg<128, 16, 4, 1, NMAJOR, 4>(p, s) : launch2_cfg<128, 32, 4, 1, NMAJOR, 4>(p, s);
  if (p.Cin == 16) return t == 5 ? launch2_cfg<128, 16, 4, 1, NMAJOR, 2>(p, s) : launch2_cfg<128, 32, 4, 1, NMAJOR, 2>(p, s);
  switch (t) {
    case 0: return launch2_cfg<128, 128, 2, 2, NMAJOR>(p, s);
    case 1: return launch2_cfg<128, 64, 2, 2, NMAJOR>(p, s);
    case 2: return launch2_cfg<64, 64, 2, 2, NMAJOR>(p, s);
    case 3: return launch2_cfg<128, 32, 4, 1, NMAJOR>(p, s);
    case 4: return launch2_cfg<64, 128, 2, 2, NMAJOR>(p, s);
    case 5: return launch2_cfg<128, 16, 4, 1, NMAJOR>(p, s);
    case 6: return launch2_cfg<256, 32, 4, 1, NMAJOR>(p, s);
    default: return launch2_cfg<256, 16, 4, 1, NMAJOR>(p, s);
  }
}

// out = sum_k slab[k] (+bias) -> U ; PReLU(alpha) -> A.  total4 = M*N/4 float4 elements, N4 = N/4.
__global__ __launch_bounds__(256) void splitk_finish_kernel(const float* __restrict__ slabs, int ksplit, int total4,
                                                            int N4, const float* __restrict__ bias,
                                                            const float* __restrict__ alpha, int alpha4_per_stamp,
                                                            float* __restrict__ U, float* __restrict__ A) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total4) return;
  const f32x4* sp = reinterpret_cast<const f32x4*>(slabs) + e;
  f32x4 v = sp[0];
  for (int k = 1; k < ksplit; ++k) v += sp[(size_t)k * total4];
  if (bias) v += reinterpret_cast<const f32x4*>(bias)[e % N4];
  if (U) reinterpret_cast<f32x4*>(U)[e] = v;
  if (A) {
    const f32x4 al = reinterpret_cast<const f32x4*>(alpha)[e % alpha4_per_stamp];
    f32x4 o;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = v[k] > 0.f ? v[k] : al[k] * v[k];
    reinterpret_cast<f32x4*>(A)[e] = o;
  }
}

int launch_splitk_finish(const float* slabs, int ksplit, long total, int N, const float* bias, const float* alpha,
                         long alpha_per_stamp, float* U, float* A, hipStream_t s) {
  if ((total & 3) || (N & 3) || total / 4 >= (1L << 31)) return E_INVALID;
  int total4 = (int)(total / 4);
  hipLaunchKernelGGL(splitk_finish_kernel, dim3((total4 + 255) / 256), dim3(256), 0, s, slabs, ksplit, total4, N / 4,
                     bias, alpha, (int)(alpha_per_stamp / 4), U, A);
  DV_HIP(hipGetLastError());
  return OK;
}

static int g2_prio = 0;
void debug_set_gconv2_prio(int v) { g2_prio = v; }
static int g2_dbg = 0;
static float* g2_dbg_out = nullptr;
void debug_set_gconv2_dbg(int v, float* out) {
  g2_dbg = v;
  g2_dbg_out = out;
}

int launch_gconv2(const GConv2Params& p0, hipStream_t s) {
  GConv2Params p = p0;
  p.dbg = g2_dbg;
  p.dbg_out = g2_dbg_out;
  p.prio = g2_prio;
  const bool small_cin = (p.Cin == 8 || p.Cin == 16) && p.Cout <= 32 && p.ksplit <= 1;
  const bool ragged = p.Cin > 32 && (p.Cin & 3) == 0 && p.nclass == 1 && p.cls[0].ntaps == 1 && p.epi != 3;
  if (p.nclass < 1 || p.nclass > 4 || ((p.Cin % BK2) && !small_cin && !ragged) || (p.Cout & 3)) {
    set_error("gconv2: unsupported shape (Cin=%d Cout=%d nclass=%d)", p.Cin, p.Cout, p.nclass);
    return E_INVALID;
  }
  if ((long)p.NB * p.Hin * p.Win * p.Cin >= (1L << 30) || (long)p.NB * p.Hout * p.Wout * p.Cout >= (1L << 30) ||
      (long)9 * p.Cin * p.Cout >= (1L << 30)) {
    set_error("gconv2: tensor too large for 32-bit byte offsets; lower max_batch");
    return E_INVALID;
  }
  if (p.epi == 2 && (!p.alpha || !p.A)) {
    set_error("gconv2: PReLU epilogue needs alpha and A");
    return E_INVALID;
  }
  if (p.epi == 3 && (!p.batch_major || !p.Uin || !p.alpha || !p.U || p.ksplit > 1)) {
    set_error("gconv2: the fused PReLU-backward epilogue needs batch-major tiles, u, alpha and an output");
    return E_INVALID;
  }
  if (p.ksplit > 1 && p.epi != 0) {
    set_error("gconv2: split-K launches write raw partial slabs (epi 0)");
    return E_INVALID;
  }
  return p.w_nmajor ? dispatch2<true>(p, s) : dispatch2<false>(p, s);
}

}  // namespace dv
