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
// taps (9 x 16 accumulator registers, v_mfma_f32_32x32x16_bf16), 16 stamps, and a range of Y rows.  It walks the pixels
// of its rows as ONE stream with a SLIDING WINDOW of X blocks in its private LDS (3 rows x a ring of 10 column slots of
// 1 KiB): a step brings in only the s new columns (3*s blocks; three columns at a row start) and the next Y block by
// LDS-DMA, issued four to six pixels ahead of their use (counted vmcnt), and multiplies the Y block with the nine X
// blocks of the window - 4 KiB of DMA per 18 kFLOP x 16 instead of 10 KiB.  Blocks outside the image come from the zero
// page (scalar address arithmetic), so the loop has no branch; the operands of pixel t + 1 are read from LDS into a
// second register set between the MFMAs of pixel t.  The four waves of a workgroup are four consecutive 16-stamp
// chunks; each parks its tile in its own LDS region, the workgroup sums them in wave order and writes one fp32 slab
// [9][Cx][Cy], which reduce_partials adds up (deterministic).
// Operands with 16 channels (the first conv's padded input, the head's gradient) fill half of the 32-wide tile; the
// lanes of the other half are zeroed by a select.
//
// Round-4 measurements behind this form: with one wave per SIMD the loop is bound by the LDS-DMA path
// (about 16 B/clk/CU with four waves in flight: a launch without its DMA takes 60 % of the time, without its MFMAs or
// without its LDS reads 95 %), and 9 us of every launch are fixed (launch, accumulator reduction, slab).
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
constexpr int BW_NCS = 10;                             // column slots of the X window (a ring, allocated column by column)
constexpr int BW_NYS = 8;                              // Y slots
constexpr int BW_WAVE_LDS = (3 * BW_NCS + BW_NYS) * 1024;
constexpr int BW_RED_BYTES = 9 * 16 * 64 * 4;          // one wave's accumulators
static_assert(BW_RED_BYTES <= BW_WAVE_LDS, "a wave parks its accumulators in its own window region");

struct BWGeom {
  int ntx, nty;        // 32-channel tiles
  int nrseg, rows_per; // Y row segments
  int nsc4;            // groups of four 16-stamp chunks
  int D;               // prefetch distance in pixels (launcher: the window ring must hold pixels t+1 .. t+D)
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
// One pixel's operands: the Y block and the nine X blocks of the window, as the transposed reads deliver them.
struct BWOps {
  bw_u32x2 b0, b1, a0[9], a1[9];
};
// "The reads of this set have landed": the wait carries the registers as in/out operands, so that nothing that uses
// them (not even a register copy) can be scheduled in front of it.
__device__ __forceinline__ void tr_wait(BWOps& o) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(o.b0), "+v"(o.b1), "+v"(o.a0[0]), "+v"(o.a1[0]), "+v"(o.a0[1]), "+v"(o.a1[1]), "+v"(o.a0[2]),
                 "+v"(o.a1[2]), "+v"(o.a0[3]), "+v"(o.a1[3]));
  asm volatile("" : "+v"(o.a0[4]), "+v"(o.a1[4]), "+v"(o.a0[5]), "+v"(o.a1[5]), "+v"(o.a0[6]), "+v"(o.a1[6]),
               "+v"(o.a0[7]), "+v"(o.a1[7]), "+v"(o.a0[8]), "+v"(o.a1[8]));
}
__device__ __forceinline__ int bw_wrap(int x) { return x >= BW_NCS ? x - BW_NCS : x; }
// The nine accumulator blocks live in the accumulation registers for the whole loop.  Left to itself the register allocator
// keeps six of them in vector registers across the loop's back edge and copies them over and back around the MFMAs that use
// them - 96 v_accvgpr_write + 96 v_accvgpr_read per two steps, a quarter of the loop's instructions, in a kernel whose
// waves spend 68 % of their cycles issuing (SQ_ACTIVE_INST_ANY).  An empty asm with an "a" operand per block at the top of
// a step fixes the class.
#define BW_PIN_ACC()                                                                                            \
  asm volatile("" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]), "+a"(acc[3]), "+a"(acc[4]), "+a"(acc[5]), "+a"(acc[6]), \
               "+a"(acc[7]), "+a"(acc[8]))
}  // namespace

// S = stride.  WAIT = DMA instructions of the D - 2 youngest issued pixels, (D - 2) * (3 * S + 1): the immediate of the
// counted vmcnt at the top of a step (pixels up to t + D - 1 are in flight there, pixel t + 1 is needed).  A pixel that
// starts a row issues more (three columns instead of S), which only makes the wait conservative.
template <bool XC16, bool YC16, int S, int WAIT>
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

  // DMA of one pixel block (16 stamps): 32-channel tile of a C-channel tensor, or a whole 16-channel tensor row set.
  // Source address = a SCALAR base (tensor + stamp chunk + channel tile + pixel, or the zero page for a block outside
  // the image) + a per-lane offset computed once: the choice between the two is scalar arithmetic, no branch and no
  // per-lane select.  The zero page is at least one pixel's [NBp][C] block long, i.e. longer than any lane offset.
  // 16-channel tensors: a block is 512 B, lanes 32..63 fetch the same rows again into the unused half of the slot.
  typedef unsigned long long bw_u64;
  const bw_u64 zu = (bw_u64)p.zero;
  const bw_u64 xu = (bw_u64)(Xb + (size_t)st0 * (XC16 ? 16 : p.Cx) + (XC16 ? 0 : cx0));
  const bw_u64 yu = (bw_u64)(Yb + (size_t)st0 * (YC16 ? 16 : p.Cy) + (YC16 ? 0 : cy0));
  const unsigned xoff_l = XC16 ? ((lane & 31) >> 1) * 32 + (lane & 1) * 16 : (lane >> 2) * p.Cx * 2 + (lane & 3) * 16;
  const unsigned yoff_l = YC16 ? ((lane & 31) >> 1) * 32 + (lane & 1) * 16 : (lane >> 2) * p.Cy * 2 + (lane & 3) * 16;
  const bw_u64 xps = (bw_u64)p.NBp * p.Cx * 2, yps = (bw_u64)p.NBp * p.Cy * 2;   // bytes per pixel
  auto dma = [&](bw_u64 base, bw_u64 pixoff, bool ok, unsigned off_l, unsigned char* dst) {
    const bw_u64 m = (bw_u64)0 - (bw_u64)ok;
    const bw_u64 ub = zu + (m & (base - zu + pixoff));
    __builtin_amdgcn_global_load_lds((bw_gptr_t)(ub + off_l), (bw_lptr_t)dst, 16, 0, 0);
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
    // The Y pixels of the row segment are ONE stream t = 0 .. T-1 (rows back to back): no drain and refill per row.
    // A pixel's three window columns sit in consecutive slots of the column ring, first slot base(t); base advances by
    // s inside a row and by 3 at a row start (whose pixel brings three new columns instead of s).  X blocks outside the
    // image come from the zero page, so every pixel multiplies all nine taps - no branch in the loop - and adding an
    // exact zero product leaves the sums what skipping the tap left them.
    const int T = (r1 - r0) * p.Hy;
    const int D = gm.D;
    int l_t = 0, l_r = r0, l_w = 0, l_base = 0;         // loader: next pixel, its row / column, its first column slot
    auto load_col = [&](int j, bool real) {            // column j (0..2) of the loader's pixel: three window rows
      const int xc = l_w * S - p.pb + j;
      unsigned char* dst = xwin + bw_wrap(l_base + j) * 1024;
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int xr = l_r * S - p.pb + kh;
        const bool ok = real & ((unsigned)xr < (unsigned)p.Hx) & ((unsigned)xc < (unsigned)p.Hx);
        dma(xu, (bw_u64)(unsigned)(xr * p.Hx + xc) * xps, ok, xoff_l, dst + kh * BW_NCS * 1024);
      }
    };
    auto load_pixel = [&]() {
      const bool real = l_t < T;                        // past the end: the same instructions from the zero page
      if (l_w == 0) {                                   // (the counted waits rely on their number)
#pragma unroll
        for (int j = 0; j < 3 - S; ++j) load_col(j, real);
      }
#pragma unroll
      for (int j = 3 - S; j < 3; ++j) load_col(j, real);
      dma(yu, (bw_u64)(unsigned)(l_r * p.Hy + l_w) * yps, real, yoff_l, ywin + (l_t & (BW_NYS - 1)) * 1024);
      ++l_t;
      if (++l_w == p.Hy) {
        l_w = 0;
        ++l_r;
        l_base = bw_wrap(l_base + 3);
      } else {
        l_base = bw_wrap(l_base + S);
      }
    };
    int c_t = 0, c_w = 0, c_base = 0;                   // reader: next pixel to fetch from LDS
    // One step: fetch pixel t + 1's operands from LDS into the other register set WHILE the nine MFMAs of pixel t run
    // (one wave per SIMD: nobody else would hide the reads) - a pair of transposed reads behind every MFMA, pinned there
    // by scheduling barriers - and bring pixel t + D in from memory in the shadow of the third MFMA.
    auto step = [&](BWOps& cur, BWOps& nxt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");   // pixel t + 1 has landed (pixels .. t+D-1 are issued)
      BW_PIN_ACC();
      const unsigned ya = yl_addr + (c_t & (BW_NYS - 1)) * 1024;
      unsigned xa[3];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xa[kw] = xl_addr + bw_wrap(c_base + kw) * 1024;
      __builtin_amdgcn_sched_barrier(0);
      nxt.b0 = tr_read<0>(ya);
      nxt.b1 = tr_read<YSECOND>(ya);
      // 16-channel operands: the lanes of the absent channel half read the present half's rows (the transposed read
      // wants every lane active with an address of its own) and are zeroed here, by a select, not a branch
      if (YC16) {
        cur.b0 = yzero ? (bw_u32x2){0u, 0u} : cur.b0;
        cur.b1 = yzero ? (bw_u32x2){0u, 0u} : cur.b1;
      }
      const bw_bf16x8 b = tr_join(cur.b0, cur.b1);
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (XC16) {
          cur.a0[t] = xzero ? (bw_u32x2){0u, 0u} : cur.a0[t];
          cur.a1[t] = xzero ? (bw_u32x2){0u, 0u} : cur.a1[t];
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(cur.a0[t], cur.a1[t]), b, acc[t], 0, 0, 0);
        const int kh = t / 3, kw = t % 3;
        if (kh == 0) {
          nxt.a0[t] = tr_read<0>(xa[kw]);
          nxt.a1[t] = tr_read<XSECOND>(xa[kw]);
        } else if (kh == 1) {
          nxt.a0[t] = tr_read<BW_NCS * 1024>(xa[kw]);
          nxt.a1[t] = tr_read<BW_NCS * 1024 + XSECOND>(xa[kw]);
        } else {
          nxt.a0[t] = tr_read<2 * BW_NCS * 1024>(xa[kw]);
          nxt.a1[t] = tr_read<2 * BW_NCS * 1024 + XSECOND>(xa[kw]);
        }
        if (t == 2) load_pixel();
        __builtin_amdgcn_sched_barrier(0);
      }
      ++c_t;
      if (++c_w == p.Hy) {
        c_w = 0;
        c_base = bw_wrap(c_base + 3);
      } else {
        c_base = bw_wrap(c_base + S);
      }
      tr_wait(nxt);
    };
    BWOps A, B;
    for (int w = 0; w < D; ++w) load_pixel();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WAIT) : "memory");
    {                                                  // operands of pixel 0
      const unsigned ya = yl_addr;
      A.b0 = tr_read<0>(ya);
      A.b1 = tr_read<YSECOND>(ya);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const unsigned xa = xl_addr + kw * 1024;
        A.a0[0 + kw] = tr_read<0>(xa);
        A.a1[0 + kw] = tr_read<XSECOND>(xa);
        A.a0[3 + kw] = tr_read<BW_NCS * 1024>(xa);
        A.a1[3 + kw] = tr_read<BW_NCS * 1024 + XSECOND>(xa);
        A.a0[6 + kw] = tr_read<2 * BW_NCS * 1024>(xa);
        A.a1[6 + kw] = tr_read<2 * BW_NCS * 1024 + XSECOND>(xa);
      }
      c_t = 1;
      c_w = p.Hy == 1 ? 0 : 1;
      c_base = p.Hy == 1 ? 3 : S;
      tr_wait(A);
    }
    for (int t = 0; t < T; t += 2) {
      step(A, B);
      if (t + 1 < T) step(B, A);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing zero-page loads, before the region is reused below
  }

  // ---- every wave parks its tile in its own window region; sums in wave order, one slab per workgroup ----
  {
    float* mine = reinterpret_cast<float*>(wl);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[(t * 16 + i) * 64 + lane] = acc[t][i];
  }
  __syncthreads();
  float* slab = p.part + (size_t)(sc4 * gm.nrseg + rseg) * 9 * p.Cx * p.Cy;
  typedef float bw_f32x4 __attribute__((ext_vector_type(4)));
  const bw_f32x4* red0 = reinterpret_cast<const bw_f32x4*>(smem);
  constexpr int WSTRIDE = BW_WAVE_LDS / 16;
  // four consecutive lanes = four consecutive cy of one (tap, cx): 16-byte LDS reads and slab stores
  for (int q = tid; q < 9 * 16 * 16; q += 256) {
    const int l4 = q & 15, ti = q >> 4;
    const int i = ti & 15, t = ti >> 4;
    const int cx = cx0 + (i & 3) + 8 * (i >> 2) + 4 * (l4 >> 3);
    const int cy = cy0 + (l4 & 7) * 4;
    const bw_f32x4 v = ((red0[q] + red0[WSTRIDE + q]) + red0[2 * WSTRIDE + q]) + red0[3 * WSTRIDE + q];
    if (cx < p.Cx && cy < p.Cy) *reinterpret_cast<bw_f32x4*>(slab + ((size_t)t * p.Cx + cx) * p.Cy + cy) = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two Y rows per wave (round 6, stride-1 layers).  Every stride-1 launch of the 59-px net takes the same ~48 us whatever its
// resolution (64 x 64 x 32 channels ... 8 x 8 x 256): the same 65536 wave-steps of 4 KiB of LDS-DMA each, and the ablation
// of round 4 put 40 % of a launch on that path.  A wave that walks a PAIR of Y rows shares the X window between them: the
// window is four rows high, a step brings in one new column of four blocks and two Y blocks (6 KiB for two pixels instead of
// 8) and multiplies twice nine taps - X row kh + 1 is tap kh of the lower pixel and tap kh + 1 of the upper one - from 14
// blocks read out of LDS instead of 20.  Same accumulators, same reduction, same slabs; the loop bookkeeping is paid per
// column instead of per pixel.  A wave's region is 40 KiB (4 x 8 column slots + 2 x 4 Y slots): with a prefetch distance of
// four columns as many bytes are in flight as in the one-row form.  Launches whose row pairs alone would not fill the chip
// (64 x 64 x 32: 32 pairs x 4 stamp groups) are also cut along the columns.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int B2_NCS = 8;                              // column slots of the X window (a power of two: wrap by mask)
constexpr int B2_NYS = 4;                              // Y slots per row
constexpr int B2_WAVE_LDS = (4 * B2_NCS + 2 * B2_NYS) * 1024;
static_assert(BW_RED_BYTES <= B2_WAVE_LDS, "a wave parks its accumulators in its own window region");
constexpr int B2_D = 4;                                // prefetch distance in columns: 3 + (D - 1) + 2 <= B2_NCS, D <= B2_NYS
constexpr int B2_WAIT = (B2_D - 2) * 6;                // DMA instructions of the D - 2 youngest columns (4 X blocks + 2 Y)

struct BW2Geom {
  int ntx, nty;        // 32-channel tiles
  int nrseg, rows_per; // Y row segments (rows_per is even)
  int ncseg, cols_per; // Y column segments
  int nsc4;            // groups of four 16-stamp chunks
};
struct BW2Ops {        // one column's operands: two Y blocks, 4 x 3 X blocks
  bw_u32x2 b0[2], b1[2], a0[12], a1[12];
};
__device__ __forceinline__ void tr_wait2(BW2Ops& o) {
  asm volatile("s_waitcnt lgkmcnt(0)"
               : "+v"(o.b0[0]), "+v"(o.b1[0]), "+v"(o.b0[1]), "+v"(o.b1[1]), "+v"(o.a0[0]), "+v"(o.a1[0]), "+v"(o.a0[1]),
                 "+v"(o.a1[1]), "+v"(o.a0[2]), "+v"(o.a1[2]), "+v"(o.a0[3]), "+v"(o.a1[3]));
  asm volatile("" : "+v"(o.a0[4]), "+v"(o.a1[4]), "+v"(o.a0[5]), "+v"(o.a1[5]), "+v"(o.a0[6]), "+v"(o.a1[6]),
               "+v"(o.a0[7]), "+v"(o.a1[7]), "+v"(o.a0[8]), "+v"(o.a1[8]), "+v"(o.a0[9]), "+v"(o.a1[9]), "+v"(o.a0[10]),
               "+v"(o.a1[10]), "+v"(o.a0[11]), "+v"(o.a1[11]));
}
}  // namespace

template <bool XC16, bool YC16>
__global__ __launch_bounds__(256, 1) void bwgrad2_kernel(const BWgradParams p, const BW2Geom gm) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  {   // XCD-contiguous order, as in bwgrad_kernel
    const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7;
    bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
  const int tcy = bid % gm.nty; bid /= gm.nty;
  const int tcx = bid % gm.ntx; bid /= gm.ntx;
  const int cseg = bid % gm.ncseg; bid /= gm.ncseg;
  const int rseg = bid % gm.nrseg;
  const int sc4 = bid / gm.nrseg;
  const int cx0 = tcx * 32, cy0 = tcy * 32;
  const int st0 = (sc4 * 4 + wave) * 16;               // first stamp of this wave's chunk
  const bool active = st0 < p.NBp;
  const int r0 = rseg * gm.rows_per, r1 = min(p.Hy, r0 + gm.rows_per);
  const int c0 = cseg * gm.cols_per, c1 = min(p.Hy, c0 + gm.cols_per);

  unsigned char* wl = smem + wave * B2_WAVE_LDS;
  unsigned char* xwin = wl;                            // [4][B2_NCS] KiB
  unsigned char* ywin = wl + 4 * B2_NCS * 1024;        // [2][B2_NYS] KiB

  const bw_bf16* Xb = reinterpret_cast<const bw_bf16*>(p.X);
  const bw_bf16* Yb = reinterpret_cast<const bw_bf16*>(p.Y);
  typedef unsigned long long bw_u64;
  const bw_u64 zu = (bw_u64)p.zero;
  const bw_u64 xu = (bw_u64)(Xb + (size_t)st0 * (XC16 ? 16 : p.Cx) + (XC16 ? 0 : cx0));
  const bw_u64 yu = (bw_u64)(Yb + (size_t)st0 * (YC16 ? 16 : p.Cy) + (YC16 ? 0 : cy0));
  const unsigned xoff_l = XC16 ? ((lane & 31) >> 1) * 32 + (lane & 1) * 16 : (lane >> 2) * p.Cx * 2 + (lane & 3) * 16;
  const unsigned yoff_l = YC16 ? ((lane & 31) >> 1) * 32 + (lane & 1) * 16 : (lane >> 2) * p.Cy * 2 + (lane & 3) * 16;
  const bw_u64 xps = (bw_u64)p.NBp * p.Cx * 2, yps = (bw_u64)p.NBp * p.Cy * 2;   // bytes per pixel

  const int g = lane >> 4, li = lane & 15;
  const int chalf = g & 1, khalf = g >> 1;
  const int tq = li >> 2, tp = li & 3;
  const int o32 = (khalf * 8 + tq) * 64 + chalf * 32 + tp * 8;
  const int o16 = (khalf * 8 + tq) * 32 + tp * 8;
  const unsigned wl_addr = (unsigned)(size_t)(bw_lptr_t)wl;
  const unsigned xl_addr = XC16 ? wl_addr + o16 : wl_addr + o32;
  const unsigned yl_addr = (YC16 ? wl_addr + o16 : wl_addr + o32) + 4 * B2_NCS * 1024;
  constexpr int XSECOND = XC16 ? 4 * 32 : 4 * 64, YSECOND = YC16 ? 4 * 32 : 4 * 64;
  constexpr int XROW = B2_NCS * 1024, YROW = B2_NYS * 1024;
  const bool xzero = XC16 && chalf, yzero = YC16 && chalf;

  bw_f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  if (active && r0 < r1 && c0 < c1) {
    // The columns of the segment's row pairs are ONE stream t = 0 .. T-1.  A column's three window columns sit in
    // consecutive slots of the ring, first slot base(t); base advances by 1 inside a pair and by 3 at a pair's first column
    // (which brings three new columns).  Blocks outside the image - and the whole lower row of a last, odd pair - come from
    // the zero page: every column multiplies all eighteen taps, no branch in the loop.
    const int ncols = c1 - c0;
    const int T = ((r1 - r0 + 1) >> 1) * ncols;
    int l_t = 0, l_r = r0, l_w = c0, l_base = 0;        // loader: next column, its upper row / column, its first slot
    // Addresses of the loader, kept incrementally (the first version multiplied a 64-bit pixel offset out for each of a
    // column's six blocks: ~15 scalar instructions per DMA in a loop whose one wave per SIMD has nobody to hide them behind):
    // per row pair the zero-page-relative origin of each of its four X rows and two Y rows and whether the row exists, per
    // column one offset that advances by a pixel.  A block outside the image - or a whole pair past the end of the segment,
    // or the lower Y row of a last, odd pair - is the zero page: `ok` masks the offset away, no branch.
    bw_u64 xrow[4], yrow[2], xco = 0, yco = 0;          // xco: offset of window column 2 of the loader's column
    bool xrok[4], yrok[2];
    auto set_pair = [&]() {
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const int xr = l_r - p.pb + kh;
        xrok[kh] = (l_r < r1) & ((unsigned)xr < (unsigned)p.Hx);
        xrow[kh] = (xu - zu) + (bw_u64)(unsigned)(xr * p.Hx) * xps;
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        yrok[rr] = l_r + rr < r1;
        yrow[rr] = (yu - zu) + (bw_u64)(unsigned)((l_r + rr) * p.Hy) * yps;
      }
      xco = (bw_u64)((long long)(c0 - p.pb + 2) * (long long)xps);
      yco = (bw_u64)(unsigned)c0 * yps;
    };
    set_pair();
    auto load_col = [&](int j) {                       // window column j (0..2) of the loader's column: four X rows
      const bool cok = (unsigned)(l_w - p.pb + j) < (unsigned)p.Hx;
      const bw_u64 co = xco - (bw_u64)(2 - j) * xps;
      unsigned char* dst = xwin + ((l_base + j) & (B2_NCS - 1)) * 1024;
#pragma unroll
      for (int kh = 0; kh < 4; ++kh) {
        const bw_u64 m = (bw_u64)0 - (bw_u64)(xrok[kh] & cok);
        __builtin_amdgcn_global_load_lds((bw_gptr_t)(zu + (m & (xrow[kh] + co)) + xoff_l), (bw_lptr_t)(dst + kh * XROW), 16, 0, 0);
      }
    };
    auto load_column = [&]() {
      if (l_w == c0) {                                  // (the counted waits rely on the number of these instructions)
        load_col(0);
        load_col(1);
      }
      load_col(2);
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const bw_u64 m = (bw_u64)0 - (bw_u64)yrok[rr];
        __builtin_amdgcn_global_load_lds((bw_gptr_t)(zu + (m & (yrow[rr] + yco)) + yoff_l),
                                         (bw_lptr_t)(ywin + (rr * B2_NYS + (l_t & (B2_NYS - 1))) * 1024), 16, 0, 0);
      }
      ++l_t;
      xco += xps;
      yco += yps;
      if (++l_w == c1) {
        l_w = c0;
        l_r += 2;
        l_base = (l_base + 3) & (B2_NCS - 1);
        set_pair();
      } else {
        l_base = (l_base + 1) & (B2_NCS - 1);
      }
    };
    int c_t = 0, c_w = 0, c_base = 0;                   // reader: next column to fetch from LDS (c_w counts from c0)
    // One step: column t + 1's operands go from LDS into the other register set while the eighteen MFMAs of column t run -
    // the nine of the upper pixel first, then the nine of the lower one, so that the two updates of an accumulator are nine
    // MFMAs apart - and column t + D comes in from memory in the shadow of the third MFMA.
    auto step = [&](BW2Ops& cur, BW2Ops& nxt) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B2_WAIT) : "memory");   // column t + 1 has landed (.. t+D-1 are issued)
      BW_PIN_ACC();
      const unsigned ya = yl_addr + (c_t & (B2_NYS - 1)) * 1024;
      unsigned xa[3];
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) xa[kw] = xl_addr + ((c_base + kw) & (B2_NCS - 1)) * 1024;
      __builtin_amdgcn_sched_barrier(0);
      nxt.b0[0] = tr_read<0>(ya);
      nxt.b1[0] = tr_read<YSECOND>(ya);
      nxt.b0[1] = tr_read<YROW>(ya);
      nxt.b1[1] = tr_read<YROW + YSECOND>(ya);
      if (YC16) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          cur.b0[rr] = yzero ? (bw_u32x2){0u, 0u} : cur.b0[rr];
          cur.b1[rr] = yzero ? (bw_u32x2){0u, 0u} : cur.b1[rr];
        }
      }
      if (XC16) {
#pragma unroll
        for (int q = 0; q < 12; ++q) {
          cur.a0[q] = xzero ? (bw_u32x2){0u, 0u} : cur.a0[q];
          cur.a1[q] = xzero ? (bw_u32x2){0u, 0u} : cur.a1[q];
        }
      }
      const bw_bf16x8 bu = tr_join(cur.b0[0], cur.b1[0]), bl = tr_join(cur.b0[1], cur.b1[1]);
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        const int t = i < 9 ? i : i - 9;                // tap
        const int q = i < 9 ? t : t + 3;                // X block: the lower pixel's rows are one further down
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_join(cur.a0[q], cur.a1[q]), i < 9 ? bu : bl, acc[t], 0, 0, 0);
        if (i < 12) {                                   // block i of the next column: window row i / 3, window column i % 3
          const int kh = i / 3, kw = i % 3;
          if (kh == 0) {
            nxt.a0[i] = tr_read<0>(xa[kw]);
            nxt.a1[i] = tr_read<XSECOND>(xa[kw]);
          } else if (kh == 1) {
            nxt.a0[i] = tr_read<XROW>(xa[kw]);
            nxt.a1[i] = tr_read<XROW + XSECOND>(xa[kw]);
          } else if (kh == 2) {
            nxt.a0[i] = tr_read<2 * XROW>(xa[kw]);
            nxt.a1[i] = tr_read<2 * XROW + XSECOND>(xa[kw]);
          } else {
            nxt.a0[i] = tr_read<3 * XROW>(xa[kw]);
            nxt.a1[i] = tr_read<3 * XROW + XSECOND>(xa[kw]);
          }
        }
        if (i == 2) load_column();
        __builtin_amdgcn_sched_barrier(0);
      }
      ++c_t;
      if (++c_w == ncols) {
        c_w = 0;
        c_base = (c_base + 3) & (B2_NCS - 1);
      } else {
        c_base = (c_base + 1) & (B2_NCS - 1);
      }
      tr_wait2(nxt);
    };
    BW2Ops A, B;
    for (int w = 0; w < B2_D; ++w) load_column();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(B2_WAIT) : "memory");
    {                                                  // operands of column 0
      const unsigned ya = yl_addr;
      A.b0[0] = tr_read<0>(ya);
      A.b1[0] = tr_read<YSECOND>(ya);
      A.b0[1] = tr_read<YROW>(ya);
      A.b1[1] = tr_read<YROW + YSECOND>(ya);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const unsigned xa = xl_addr + kw * 1024;
        A.a0[0 + kw] = tr_read<0>(xa);
        A.a1[0 + kw] = tr_read<XSECOND>(xa);
        A.a0[3 + kw] = tr_read<XROW>(xa);
        A.a1[3 + kw] = tr_read<XROW + XSECOND>(xa);
        A.a0[6 + kw] = tr_read<2 * XROW>(xa);
        A.a1[6 + kw] = tr_read<2 * XROW + XSECOND>(xa);
        A.a0[9 + kw] = tr_read<3 * XROW>(xa);
        A.a1[9 + kw] = tr_read<3 * XROW + XSECOND>(xa);
      }
      c_t = 1;
      c_w = ncols == 1 ? 0 : 1;
      c_base = ncols == 1 ? 3 : 1;
      tr_wait2(A);
    }
    for (int t = 0; t < T; t += 2) {
      step(A, B);
      if (t + 1 < T) step(B, A);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing zero-page loads, before the region is reused below
  }

  // ---- every wave parks its tile in its own window region; sums in wave order, one slab per workgroup ----
  {
    float* mine = reinterpret_cast<float*>(wl);
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) mine[(t * 16 + i) * 64 + lane] = acc[t][i];
  }
  __syncthreads();
  float* slab = p.part + (size_t)((sc4 * gm.nrseg + rseg) * gm.ncseg + cseg) * 9 * p.Cx * p.Cy;
  typedef float bw_f32x4 __attribute__((ext_vector_type(4)));
  const bw_f32x4* red0 = reinterpret_cast<const bw_f32x4*>(smem);
  constexpr int WSTRIDE = B2_WAVE_LDS / 16;
  for (int q = tid; q < 9 * 16 * 16; q += 256) {
    const int l4 = q & 15, ti = q >> 4;
    const int i = ti & 15, t = ti >> 4;
    const int cx = cx0 + (i & 3) + 8 * (i >> 2) + 4 * (l4 >> 3);
    const int cy = cy0 + (l4 & 7) * 4;
    const bw_f32x4 v = ((red0[q] + red0[WSTRIDE + q]) + red0[2 * WSTRIDE + q]) + red0[3 * WSTRIDE + q];
    if (cx < p.Cx && cy < p.Cy) *reinterpret_cast<bw_f32x4*>(slab + ((size_t)t * p.Cx + cx) * p.Cy + cy) = v;
  }
}

// stride-1 launches in the two-row form; returns 1 when the launch is not taken (the one-row form follows)
static int launch_bwgrad2(const BWgradParams& p, hipStream_t s, int* nsplit_out) {
  static const bool off = getenv("DV_BWGRAD_ONE_ROW") != nullptr;   // (A/B: the form until round 6)
  if (off || p.s != 1 || p.Hy < 4 || p.Hx != p.Hy) return 1;
  const bool xc16 = p.Cx == 16, yc16 = p.Cy == 16;
  BW2Geom gm;
  gm.ntx = (p.Cx + 31) / 32;
  gm.nty = (p.Cy + 31) / 32;
  gm.nsc4 = (p.NBp + 63) / 64;
  const size_t slab = (size_t)9 * p.Cx * p.Cy;
  const long budget = 24L << 20;
  long copies = std::max<long>(gm.nsc4, budget / (long)(slab * sizeof(float)));
  copies = std::min<long>(copies, (long)(p.part_capacity / slab));
  if (copies < gm.nsc4) return 1;
  // segments: row PAIRS first; a launch whose pairs alone leave the chip short of workgroups is also cut along the columns
  const long want_wgs = 256;
  const long per_seg = (long)gm.nsc4 * gm.ntx * gm.nty;
  const long nseg_want = std::max<long>(1, (want_wgs + per_seg - 1) / per_seg);
  const int npairs = (p.Hy + 1) / 2;
  const long seg_cap = std::max<long>(1, copies / gm.nsc4);
  int nrseg = (int)std::min<long>(std::min<long>(npairs, seg_cap), nseg_want);
  gm.rows_per = 2 * ((npairs + nrseg - 1) / nrseg);
  gm.nrseg = (p.Hy + gm.rows_per - 1) / gm.rows_per;
  // (never more workgroups than one round of the chip: a column segment is at least 16 columns long)
  int ncseg = (int)std::min<long>(std::min<long>(std::max<long>(1, seg_cap / gm.nrseg), std::max<long>(1, nseg_want / gm.nrseg)),
                                  std::max(1, p.Hy / 16));
  gm.cols_per = (p.Hy + ncseg - 1) / ncseg;
  gm.ncseg = (p.Hy + gm.cols_per - 1) / gm.cols_per;
  if (nsplit_out) *nsplit_out = gm.nrseg * gm.ncseg * gm.nsc4;
  const unsigned grid = (unsigned)((long)gm.nsc4 * gm.nrseg * gm.ncseg * gm.ntx * gm.nty);
  const size_t lds = (size_t)4 * B2_WAVE_LDS;
  {   // DV_EXP_SKIP_WGRAD >= 1 (MEASUREMENT, wrong gradients): geometry and slab bookkeeping as usual, no kernel
    static const bool exp_skip = DV_EXP_SWITCH("DV_EXP_SKIP_WGRAD") >= 1;
    if (exp_skip) return OK;
  }
#define BW2_LAUNCH(XC, YC)                                                                                      \
  do {                                                                                                          \
    static int attr = 0;                                                                                        \
    if (!attr) {                                                                                                \
      attr = hipFuncSetAttribute((const void*)bwgrad2_kernel<XC, YC>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                 (int)lds) == hipSuccess ? 1 : -1;                                              \
      if (attr < 0) (void)hipGetLastError();                                                                    \
    }                                                                                                           \
    if (attr < 0) return 1;               /* (160 KiB of LDS per workgroup refused: the one-row form) */        \
    hipLaunchKernelGGL((bwgrad2_kernel<XC, YC>), dim3(grid), dim3(256), lds, s, p, gm);                         \
  } while (0)
  if (xc16 && yc16) BW2_LAUNCH(true, true);
  else if (xc16) BW2_LAUNCH(true, false);
  else if (yc16) BW2_LAUNCH(false, true);
  else BW2_LAUNCH(false, false);
#undef BW2_LAUNCH
  DV_HIP(hipGetLastError());
  return OK;
}

int launch_bwgrad(const BWgradParams& p, hipStream_t s, int* nsplit_out) {
  const bool xc16 = p.Cx == 16, yc16 = p.Cy == 16;
  if ((!xc16 && p.Cx % 32) || (!yc16 && p.Cy % 32) || (p.NBp & 15) || p.s < 1 || p.s > 2) {
    set_error("bwgrad: channels must be 16 or multiples of 32 (Cx %d, Cy %d), stamps padded to 16 (%d), stride 1 or 2",
              p.Cx, p.Cy, p.NBp);
    return E_INVALID;
  }
  {
    const int r2 = launch_bwgrad2(p, s, nsplit_out);
    if (r2 <= 0) return r2;
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
  // prefetch distance D: pixel t + D is issued during step t, so the column ring holds the windows of pixels t+1 .. t+D:
  // (D-1)*s + (3-s)*ceil((D-1)/Hy) + 3 <= BW_NCS (a row start inside the span brings 3 columns instead of s)
  gm.D = p.s == 2 ? (p.Hy >= 3 ? 4 : 3) : (p.Hy >= 5 ? 6 : p.Hy == 4 ? 5 : 3);
#define BW_LAUNCH_W(XC, YC, S_, W)                                                                              \
  do {                                                                                                          \
    static bool attr = false;                                                                                   \
    if (!attr) {                                                                                                \
      DV_HIP(hipFuncSetAttribute((const void*)bwgrad_kernel<XC, YC, S_, W>,                                     \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
      attr = true;                                                                                              \
    }                                                                                                           \
    hipLaunchKernelGGL((bwgrad_kernel<XC, YC, S_, W>), dim3(grid), dim3(256), lds, s, p, gm);                   \
  } while (0)
#define BW_LAUNCH(XC, YC)                                                                                       \
  do {                                                                                                          \
    if (p.s == 2 && gm.D == 4) BW_LAUNCH_W(XC, YC, 2, 14);                                                      \
    else if (p.s == 2) BW_LAUNCH_W(XC, YC, 2, 7);                                                               \
    else if (gm.D == 6) BW_LAUNCH_W(XC, YC, 1, 16);                                                             \
    else if (gm.D == 5) BW_LAUNCH_W(XC, YC, 1, 12);                                                             \
    else BW_LAUNCH_W(XC, YC, 1, 4);                                                                             \
  } while (0)
  {   // DV_EXP_SKIP_WGRAD >= 1 (MEASUREMENT, wrong gradients): geometry and slab bookkeeping as usual, no kernel
    static const bool exp_skip = DV_EXP_SWITCH("DV_EXP_SKIP_WGRAD") >= 1;
    if (exp_skip) return OK;
  }
  if (xc16 && yc16) BW_LAUNCH(true, true);
  else if (xc16) BW_LAUNCH(true, false);
  else if (yc16) BW_LAUNCH(false, true);
  else BW_LAUNCH(false, false);
#undef BW_LAUNCH
#undef BW_LAUNCH_W
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
