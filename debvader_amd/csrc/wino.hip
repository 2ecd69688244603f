// Winograd F(2x2, 3x3) form of the stride-1 3x3 layers in fp32: the four stride-1 Conv2D layers of the encoder
// (model.py:81-83), the four stride-1 Conv2DTranspose layers of the decoder (model.py:128-134), the head conv
// (model.py:137) and the data gradients of all of them - two thirds of the network's multiply-adds (the head conv, 16
// output columns, stays with the strip kernel).
//
// A 3x3 stride-1 layer is  Y = A^T [ sum_c (G g_c G^T) . (B^T d_c B) ] A  per 4x4 input tile d / 2x2 output tile Y:
// 16 element-wise positions, each an independent [tiles x Cin] x [Cin x Cout] contraction, 4 multiply-adds per output
// pixel and channel pair instead of 9.  fp32 MFMA runs ON the vector lanes on gfx950 (157 TFLOP/s both ways), so the
// 2.25x fewer MFMA cycles are worth having - and every vector instruction of the transforms and of the address
// arithmetic costs MFMA time (measured: ~5.7 cycles each, whether or not it sits "in the shadow" of an MFMA; DESIGN.md
// 4a).  Everything is fused: a workgroup DMAs raw input patches and pre-transformed weight chunks (wino_weights_kernel:
// U = G g G^T, once per weight update) into LDS, transforms the 4x4 input tiles in registers (B^T d B on float4 channel
// quads), runs the 16 position GEMMs on v_mfma_f32_16x16x4_f32, applies the output transform to its accumulators and
// writes bias / PReLU outputs through a per-wave LDS staging tile.
//
// Three kernels: wino_conv_kernel (eight waves; wave = M block x 16 columns; used for 32 input channels),
// wino_conv4_kernel (four waves, one per SIMD; wave = M block x 32 columns; built around the vector-instruction count;
// >= 48 input channels) and wino_wgrad_kernel (weight gradient in the transform domain, F(3x3, 2x2)).
//
// Geometry: an M block is 4 x 4 tiles = 8 x 8 output pixels of one stamp (its input patch: 10 x 10 pixels); a
// workgroup owns 4 consecutive M blocks x 32 output channels and walks K in chunks of 16 input channels through a
// double-buffered LDS ring (patch chunk 4 x 6.25 KiB, weight chunk 32 KiB shared), persistent over (block group,
// column tile) items with the ring running across item boundaries.
// MFMA roles: A = transformed input V[tile][k], B = transformed weights U[k][n]; lane (l15, lg): A row l15 = tile
// (ty, tx) = (l15 >> 2, l15 & 3), k = lg; C rows 4 lg + r = tile (lg, r), column l15.
// Numerics: exact fp32 arithmetic in another association (sums of up to four inputs before the product, 0.5 factors
// in G); agreement with the direct kernels 2e-6 of a layer's largest output (dv_debug_gconv_check, tests/test_gpu_parity).
#include "common.h"
#include <algorithm>
#include <type_traits>
#include <stdlib.h>
#include <stdio.h>

namespace dv {

typedef const __attribute__((address_space(1))) void* wn_gptr_t;
typedef __attribute__((address_space(3))) void* wn_lptr_t;

namespace {
constexpr int WN_MB = 4;                              // M blocks (16 tiles = 8 x 8 output pixels each) per workgroup
constexpr int WN_WAVES = 8;                           // wave w: M block w & 3, column half w >> 2 (16 of the 32 columns)
constexpr int WN_THREADS = 64 * WN_WAVES;
constexpr int WN_PATCH_PIECES = 7;                    // 100 pixels x 4 quads = 400 16-byte slots -> 7 x 64
constexpr int WN_PATCH_FLOATS = WN_PATCH_PIECES * 256;
constexpr int WN_UCH_FLOATS = 16 * 32 * 16;           // [pos][n][k]
constexpr int WN_LDS_STAGE = 20;                      // floats per staged pixel row (16 columns + pad)
}  // namespace

__global__ __launch_bounds__(WN_THREADS, 2) void wino_conv_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                                            // [2][4 blocks][WN_PATCH_FLOATS]
  float* uch = smem + 2 * WN_MB * WN_PATCH_FLOATS;                // [2][WN_UCH_FLOATS]
  float* stage = uch + 2 * WN_UCH_FLOATS;                         // [8 waves][32][WN_LDS_STAGE]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mb = wave & 3, nh = wave >> 2;
  const int l15 = lane & 15, lg = lane >> 4;
  const int H = p.H, Cin = p.Cin, Cout = p.Cout, NC = p.NC;
  const int nb2 = p.nbh * p.nbh;

  // ---- item-invariant DMA addressing: the two waves of an M block share its patch pieces (4 + 3); slot e = 64 k + lane
  // -> (patch pixel L, channel quad) ----
  constexpr int NPP = 4;
  const int k0 = nh * 4;
  int prow[NPP], pcol[NPP], prel[NPP];
#pragma unroll
  for (int k = 0; k < NPP; ++k) {
    const int e = 64 * (k0 + k) + lane;
    const int L = e >> 2;
    const int pr = L / 10, pc = L - pr * 10;
    const int q = (e & 3) ^ (pr & 3);                      // LDS slot e & 3 of a pixel in patch row pr holds channel quad q
    prow[k] = (L < 100 && k0 + k < WN_PATCH_PIECES) ? pr - 1 : -100000;
    pcol[k] = pc - 1;
    prel[k] = ((pr - 1) * H + (pc - 1)) * Cin + q * 4;
  }

  const int item0 = blockIdx.x * p.items_per_wg;
  const int nitems = min(p.items, item0 + p.items_per_wg) - item0;
  if (nitems <= 0) return;
  const int qtotal = nitems * NC;

  // DMA of chunk q (item item0 + q / NC, K chunk q % NC) into ring buffer b: dma_setup works out the wave-uniform
  // parts once per chunk, dma_piece(k) issues piece k of this wave's eight (0-3: its share of the M block's patch,
  // 4-7: its share of the weight chunk).  The pieces are issued one at a time BETWEEN the MFMA groups of the previous
  // chunk: issued back to back in front of them they hold the wave (in-order issue, the memory pipeline accepts a
  // 1-KiB gather every ~150 cycles under load) while the matrix pipe idles.
  const float* dm_base = p.zero;
  const float* dm_usrc = p.zero;
  float *dm_dstp = patch, *dm_dstu = uch;
  bool dm_live = false;
  int dm_by = 0, dm_bx = 0;
  auto dma_setup = [&](int q, int b) {
    const int it = q / NC, c = q - it * NC;
    const int item = item0 + it;
    const int ct = item / p.groups, g = item - ct * p.groups;   // column tile outermost: one weight slice hot in every L2
    const int mblock = g * WN_MB + mb;
    dm_live = mblock < p.NB * nb2;
    const int n = dm_live ? mblock / nb2 : 0;
    const int rem = dm_live ? mblock - n * nb2 : 0;
    dm_by = rem / p.nbh;
    dm_bx = rem - dm_by * p.nbh;
    dm_base = p.X + ((size_t)(n * H + dm_by * 8) * H + dm_bx * 8) * Cin + c * 16;
    dm_dstp = patch + (b * WN_MB + mb) * WN_PATCH_FLOATS + k0 * 256;
    dm_usrc = p.Ut + ((size_t)(ct * NC + c) * WN_UCH_FLOATS) + wave * 1024 + lane * 4;
    dm_dstu = uch + b * WN_UCH_FLOATS + wave * 1024;
  };
  auto dma_piece = [&](int k) {                       // k is a compile-time constant at every call site
    if (k < NPP) {
      if (k0 + k < WN_PATCH_PIECES) {                 // wave-uniform
        const bool ok = dm_live && (unsigned)(dm_by * 8 + prow[k]) < (unsigned)H && (unsigned)(dm_bx * 8 + pcol[k]) < (unsigned)H;
        const float* src = ok ? dm_base + prel[k] : p.zero;
        __builtin_amdgcn_global_load_lds((wn_gptr_t)src, (wn_lptr_t)(dm_dstp + k * 256), 16, 0, 0);
      }
    } else {
      __builtin_amdgcn_global_load_lds((wn_gptr_t)(dm_usrc + (k - NPP) * 256), (wn_lptr_t)(dm_dstu + (k - NPP) * 256), 16, 0, 0);
    }
  };
  auto issue = [&](int q, int b) {
    dma_setup(q, b);
#pragma unroll
    for (int k = 0; k < 2 * NPP; ++k) dma_piece(k);
  };

  // ---- per-lane fragment addressing ----
  const int ty = l15 >> 2, tx = l15 & 3;
  // LDS images are swizzled so that the ds_read_b128 fragment reads spread over the banks (lane groups of
  // MI355X_MICROARCH.md, LDS): a pixel of patch row r keeps channel quad q in slot q ^ (r & 3) - tiles step by two
  // pixels, so 2-way is the floor for 64-byte pixel rows, and this reaches it (unswizzled: 4-way) -; weight row n keeps
  // quad q in slot q ^ {0,2,3,1}[(n >> 2) & 3] (conflict-free; unswizzled 2-way).  Both swizzles are applied on the
  // global side: by the DMA's per-lane source address and by wino_weights_kernel.
  int a_off[4];                                                      // float offset of d[i][0] in the block's patch
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = ((2 * ty + i) * 10 + 2 * tx) * 16 + ((lg ^ ((2 * ty + i) & 3)) << 2);
  const int b_off = (nh * 16 + l15) * 16 + ((lg ^ ((0x78 >> (2 * ((l15 >> 2) & 3))) & 3)) << 2);   // + pos * 32 * 16
  float* stg = stage + wave * (32 * WN_LDS_STAGE);

  // One barrier per chunk: every wave waits for its own DMA pieces of chunk q, meets the others (all pieces landed, all
  // reads of the other ring buffer done), issues the DMA of chunk q + 1 into that buffer and computes chunk q.  Raw
  // s_barrier with explicit counters: the epilogue's plain loads and stores share vmcnt with the DMA.
  // (Tried and measured slower on MI355X, kept out: a half-chunk stagger of waves 4-7 against their SIMD partners with
  // two barriers per chunk - 126 -> 147 us on the 256-channel layer; see DESIGN.md.)
  auto sync_point = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  f32x4 acc[16];
  issue(0, 0);
  int buf = 0;
  for (int q = 0; q < qtotal; ++q) {
    const int it = q / NC, c = q - it * NC;
    sync_point();
    const bool more = q + 1 < qtotal;
    if (more) dma_setup(q + 1, buf ^ 1);
    if (c == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const float* P = patch + (buf * WN_MB + mb) * WN_PATCH_FLOATS;
    const float* Uc = uch + buf * WN_UCH_FLOATS + b_off;
    // All LDS reads of the chunk go out first - the 16 weight fragments (one per position) and the 4 x 4 input tile - so
    // that the MFMA stream below never waits for one (left to itself hipcc sinks every fragment read to just in front
    // of its four MFMAs: read latency + a dependent accumulator chain per position).
    f32x4 bfr[16];                                    // (positions 8-15 are fetched behind the first MFMA groups)
#pragma unroll
    for (int pos = 0; pos < 8; ++pos) bfr[pos] = *reinterpret_cast<const f32x4*>(Uc + (pos * 32) * 16);
    f32x4 V[16];
    {
      // input transform V = B^T d B on the float4 channel quad of this lane's tile
      f32x4 d[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x4*>(P + a_off[i] + j * 16);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0][j] = d[0][j] - d[2][j];
        t[1][j] = d[1][j] + d[2][j];
        t[2][j] = d[2][j] - d[1][j];
        t[3][j] = d[1][j] - d[3][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        V[i * 4 + 0] = t[i][0] - t[i][2];
        V[i * 4 + 1] = t[i][1] + t[i][2];
        V[i * 4 + 2] = t[i][2] - t[i][1];
        V[i * 4 + 3] = t[i][1] - t[i][3];
      }
    }
    // epilogue geometry of this wave's block (used from the last chunk of an item on)
    bool e_on = false, e_live = false;
    int e_n = 0, e_by = 0, e_bx = 0, e_col = 0, e_col0 = 0;
    unsigned e_aoff[4] = {0, 0, 0, 0};                // offsets of the lane's four float4 in the [H][H][Cout] plane
    bool e_ok[4] = {false, false, false, false};
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, al4[4];
    // 16 position GEMMs, two accumulator chains interleaved (a dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles,
    // an independent one every 32)
    {
#pragma unroll
      for (int pos = 0; pos < 16; pos += 2) {
        if (pos == 2) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q2 = 8; q2 < 16; ++q2) bfr[q2] = *reinterpret_cast<const f32x4*>(Uc + (q2 * 32) * 16);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
          acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[pos][jj], bfr[pos][jj], acc[pos], 0, 0, 0);
          acc[pos + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[pos + 1][jj], bfr[pos + 1][jj], acc[pos + 1], 0, 0, 0);
        }
        if (pos == 8 && c == NC - 1) {
          __builtin_amdgcn_sched_barrier(0);
      const int item = item0 + it;
          const int ct = item / p.groups, g = item - ct * p.groups;
          const int mblock = g * WN_MB + mb;
          e_live = mblock < p.NB * nb2;
          e_n = e_live ? mblock / nb2 : 0;
          const int rem = e_live ? mblock - e_n * nb2 : 0;
          e_by = rem / p.nbh;
          e_bx = rem - e_by * p.nbh;
          e_col0 = ct * 32 + nh * 16;
          e_on = e_col0 < Cout;                           // wave-uniform
          e_col = e_col0 + (lane & 3) * 4;
#pragma unroll
          for (int i = 0; i < 4; ++i) {                   // i = 2 a + k: pass a (block rows 2 lg + a), half k
            const int sidx = (i & 1) * 16 + (lane >> 2);
            const int row = e_by * 8 + 2 * (sidx >> 3) + (i >> 1), cl = e_bx * 8 + (sidx & 7);
            e_ok[i] = e_live && e_on && row < H && cl < H;
            e_aoff[i] = e_ok[i] ? (unsigned)((row * H + cl) * Cout + e_col) : 0u;
          }
        }
        if (pos == 8 && c == NC - 1 && e_on) {
          // The epilogue's bias and PReLU slopes, fetched now (half of the fragment registers are free again) by
          // hand-written loads: a plain load's result makes hipcc wait vmcnt(0) at its first use while LDS-DMA is in
          // flight - all of the next chunk's DMA and the previous item's stores - which cost the 32- and 64-channel
          // layers a third of their time.  No DMA piece follows these loads (pos 8), hence vmcnt(0) in the epilogue waits for
          // DMA issued >= 4 MFMA groups earlier and for these loads only.
          __builtin_amdgcn_sched_barrier(0);
          if (p.epi >= 1) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bias4) : "v"(p.bias + e_col) : "memory");
          if (p.epi == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(al4[i]) : "v"(p.alpha + e_aoff[i]) : "memory");
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more && pos < 8) {
          // two DMA pieces of the next chunk behind each of the first four MFMA groups: spread out (issued back to back
          // they hold the wave), but early - a piece needs ~2500 cycles under load, and what is still in flight at the
          // next barrier is waited for by everybody
          __builtin_amdgcn_sched_barrier(0);
          dma_piece(pos);
          dma_piece(pos + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (c == NC - 1 && e_on) {
      // ---- output transform Y = A^T M A on the accumulators, then bias / PReLU / stores through the staging tile ----
      float y[2][4][2];                                 // [a][r][b]: pixel (2 lg + a, 2 r + b) of the 8 x 8 block
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float tt[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float m0 = acc[0 + j][r], m1 = acc[4 + j][r], m2 = acc[8 + j][r], m3 = acc[12 + j][r];
          tt[0][j] = m0 + m1 + m2;
          tt[1][j] = m1 - m2 - m3;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
          y[a][r][0] = tt[a][0] + tt[a][1] + tt[a][2];
          y[a][r][1] = tt[a][1] - tt[a][2] - tt[a][3];
        }
      }
      // the hand-issued bias / slope loads are the youngest vector-memory operations of this wave
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("" : "+v"(bias4), "+v"(al4[0]), "+v"(al4[1]), "+v"(al4[2]), "+v"(al4[3]));
      const int f4 = lane & 3;
      const size_t obase = (size_t)e_n * H * H * Cout;
#pragma unroll
      for (int a = 0; a < 2; ++a) {                     // two passes of 32 pixels: block rows 2 lg + a
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          stg[(lg * 8 + 2 * r) * WN_LDS_STAGE + l15] = y[a][r][0];
          stg[(lg * 8 + 2 * r + 1) * WN_LDS_STAGE + l15] = y[a][r][1];
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): wave-private region
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int i = 2 * a + k;
          if (!e_ok[i]) continue;
          const int sidx = k * 16 + (lane >> 2);
          f32x4 v = *reinterpret_cast<const f32x4*>(stg + sidx * WN_LDS_STAGE + f4 * 4);
          if (p.epi >= 1) v += bias4;
          if (p.U) *reinterpret_cast<f32x4*>(p.U + obase + e_aoff[i]) = v;
          if (p.epi == 2) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? v[e] : al4[i][e] * v[e];
            *reinterpret_cast<f32x4*>(p.A + obase + e_aoff[i]) = o;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
    buf ^= 1;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Second generation of the same kernel: ONE wave per SIMD that stays on the matrix pipe.
// The first kernel keeps the pipe ~0.42 busy: its per-chunk barrier puts all eight waves into the same phase, so for
// ~2000 of ~8000 cycles per chunk every wave reads and transforms and nobody issues MFMAs.  Here a workgroup has four
// waves (one per SIMD, up to 512 registers each); wave w owns M block w for ALL 32 columns (16 positions x 2 column
// halves = 128 accumulator registers), so the input transform is done once per block instead of once per column half,
// and everything that is not an MFMA is interleaved into the MFMA stream of the chunk BEFORE the one that needs it
// (sched_group_barrier: one MFMA, then an LDS read, a few vector instructions and a DMA piece in its shadow):
//   * the 4 x 4 input tile of chunk q + 1 is read from LDS during MFMA groups 0-3 of chunk q and transformed during
//     groups 1-7 - the last step writes column j of the result straight into the V registers whose last readers, the
//     MFMAs of groups 2 j and 2 j + 1, have just been issued -, the weight fragments of group g + 1 are read during group g;
//   * the DMA of the weights of chunk q + 1 and of the patch of chunk q + 2 (patches are wave-private) goes out three
//     pieces per group behind groups 0-4, a full chunk ahead of the barrier that waits for it;
//   * bias and PReLU slopes of an item's 8 x 8 x 32 output block arrive by LDS-DMA during its last chunk, so that the
//     epilogue holds no registers across the loop and needs no counted vmcnt.
// LDS: patch ring 2 x 4 x 6.25 KiB (wave-private blocks), weight ring 2 x 32 KiB, staging 4 x 2.5 KiB, slopes + bias
// 4 x 8.1 KiB = 156.5 KiB.
namespace {
constexpr int W4_WAVES = 4;
constexpr int W4_THREADS = 64 * W4_WAVES;
constexpr int W4_PATCH_FLOATS = 1600;                 // 100 pixels x 16 channels, no padding (piece 6: 16 lanes)
constexpr int W4_ALPHA_FLOATS = 64 * 32 + 32;         // [pixel of the 8 x 8 block][column of the 32] + bias[32]
// position order: the four positions of column j of the transform domain (rows 0-3) in two groups of two
__device__ __forceinline__ constexpr int w4_pos(int g, int pp) { return (g >> 1) + 4 * (2 * (g & 1) + pp); }
}  // namespace

// fp32 MFMA and the vector ALU are the same lanes on gfx950 (the fp32 matrix peak IS the vector peak): every vector
// instruction a wave issues is time its MFMA stream does not get.  Hence packed adds for the transforms and scalar-base
// DMA with execution masks instead of per-lane address selects.
typedef float w4_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 w4_add(f32x4 a, f32x4 b) {
  w4_f32x2 lo, hi;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
  return (f32x4){lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 w4_sub(f32x4 a, f32x4 b) {
  w4_f32x2 lo, hi;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
      : "=v"(lo) : "v"(__builtin_shufflevector(a, a, 0, 1)), "v"(__builtin_shufflevector(b, b, 0, 1)));
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]"
      : "=v"(hi) : "v"(__builtin_shufflevector(a, a, 2, 3)), "v"(__builtin_shufflevector(b, b, 2, 3)));
  return (f32x4){lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ w4_f32x2 w4_add2(w4_f32x2 a, w4_f32x2 b) {
  w4_f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ w4_f32x2 w4_sub2(w4_f32x2 a, w4_f32x2 b) {
  w4_f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// packed adds with operand-half selection: {a.x - b.x, a.y + b.x}, {b.x - a.y, a.y - b.y}, {a.x + a.y, a.x - a.y}
__device__ __forceinline__ w4_f32x2 w4_pk_h1(w4_f32x2 a, w4_f32x2 b) {
  w4_f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ w4_f32x2 w4_pk_h2(w4_f32x2 a, w4_f32x2 b) {
  w4_f32x2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ w4_f32x2 w4_pk_pm(w4_f32x2 a) {
  w4_f32x2 r;
  asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a));
  return r;
}
__global__ __launch_bounds__(W4_THREADS) void wino_conv4_kernel(const WinoParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                                            // [2][4 blocks][W4_PATCH_FLOATS]
  float* uch = smem + 2 * WN_MB * W4_PATCH_FLOATS;                // [2][WN_UCH_FLOATS]
  float* stage = uch + 2 * WN_UCH_FLOATS;                         // [4 waves][32][WN_LDS_STAGE]
  float* atile = stage + W4_WAVES * 32 * WN_LDS_STAGE;            // [4 waves][W4_ALPHA_FLOATS]

  const int tid = threadIdx.x, lane = tid & 63;
  const int mb = __builtin_amdgcn_readfirstlane(tid >> 6);        // wave = M block
  const int l15 = lane & 15, lg = lane >> 4;
  const int H = p.H, Cin = p.Cin, Cout = p.Cout, NC = p.NC;
  const int nb2 = p.nbh * p.nbh;

  // patch DMA: slot e = 64 k + lane -> (patch pixel L, channel quad), k = 0..6 (see wino_conv_kernel); poffb = byte offset
  // of the lane's 16 bytes from pixel (-1, -1) of the block, channel 0 of the chunk
  int prow[WN_PATCH_PIECES], pcol[WN_PATCH_PIECES];
  unsigned poffb[WN_PATCH_PIECES];
#pragma unroll
  for (int k = 0; k < WN_PATCH_PIECES; ++k) {
    const int e = 64 * k + lane;
    const int L = e >> 2;
    const int pr = L / 10, pc = L - pr * 10;
    const int q = (e & 3) ^ (pr & 3);
    prow[k] = L < 100 ? pr - 1 : -100000;
    pcol[k] = pc - 1;
    poffb[k] = (unsigned)(((pr * H + pc) * Cin + q * 4) * 4);
  }
  const unsigned lane16 = lane * 16;
  const unsigned aoffb = (unsigned)(((lane >> 3) * Cout + (lane & 7) * 4) * 4);   // slopes: pixel column, channel quad
  unsigned vnull = 0;
  asm volatile("" : "+v"(vnull));                      // (a register that holds 0: offset of the zero-page lanes)

  const int item0 = blockIdx.x * p.items_per_wg;
  const int nitems = min(p.items, item0 + p.items_per_wg) - item0;
  if (nitems <= 0) return;
  const int qtotal = nitems * NC;

  // ---- DMA streams: weights of chunk qw, patch of chunk qp (each advanced once per chunk) ----
  struct Geo { bool live; int n, by, bx, ct; };
  auto geometry = [&](int it) {
    Geo gq;
    const int item = item0 + it;
    gq.ct = item / p.groups;
    const int g = item - gq.ct * p.groups;                  // column tile outermost: one weight slice hot in every L2
    const int mblock = g * WN_MB + mb;
    gq.live = mblock < p.NB * nb2;
    gq.n = gq.live ? mblock / nb2 : 0;
    const int rem = gq.live ? mblock - gq.n * nb2 : 0;
    gq.by = rem / p.nbh;
    gq.bx = rem - gq.by * p.nbh;
    return gq;
  };
  // patch stream: origin (pixel (-1, -1) of the block, channel 0) and lane masks of the current item
  const char* p_org = reinterpret_cast<const char*>(p.zero);
  unsigned long long m_dma[WN_PATCH_PIECES];
  auto patch_item = [&](int it) {
    const Geo gq = geometry(it);
    p_org = reinterpret_cast<const char*>(p.X) +
            ((long long)(gq.n * H + gq.by * 8 - 1) * H + (gq.bx * 8 - 1)) * (long long)Cin * 4;
#pragma unroll
    for (int k = 0; k < WN_PATCH_PIECES; ++k) {
      const bool ok = gq.live && (unsigned)(gq.by * 8 + prow[k]) < (unsigned)H && (unsigned)(gq.bx * 8 + pcol[k]) < (unsigned)H;
      const bool in = k < WN_PATCH_PIECES - 1 || lane < 16;     // slots 384-399: the block's buffer ends there
      m_dma[k] = __builtin_amdgcn_ballot_w64(ok && in);
    }
  };
  const unsigned patch_lds = (unsigned)(size_t)(wn_lptr_t)(patch + mb * W4_PATCH_FLOATS);   // + buffer * 4 * 6400
  const unsigned uch_lds = (unsigned)(size_t)(wn_lptr_t)(uch + mb * 2048);                   // + buffer * 32768
  const char* ps_base = p_org;                        // origin + chunk * 64 bytes
  unsigned ps_lds = patch_lds;
  auto piece_p = [&](int k) {
    if (k < WN_PATCH_PIECES - 1) dv_dma_masked<false>(ps_lds + k * 1024, poffb[k], ps_base, m_dma[k], vnull, p.zero);
    else dv_dma_masked<true>(ps_lds + k * 1024, poffb[k], ps_base, m_dma[k], vnull, p.zero);
  };
  const char* ws_base = reinterpret_cast<const char*>(p.Ut);   // this wave's 8 KiB of the chunk
  unsigned ws_lds = uch_lds;
  auto piece_w = [&](int k) { dv_dma(ws_lds + k * 1024, lane16, ws_base + k * 1024); };

  // ---- fragment addressing (swizzles: see wino_conv_kernel) ----
  const int ty = l15 >> 2, tx = l15 & 3;
  int a_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) a_off[i] = ((2 * ty + i) * 10 + 2 * tx) * 16 + ((lg ^ ((2 * ty + i) & 3)) << 2);
  const int b_off0 = l15 * 16 + ((lg ^ ((0x78 >> (2 * ((l15 >> 2) & 3))) & 3)) << 2);       // column half 1: + 256
  float* stg = stage + mb * (32 * WN_LDS_STAGE);
  float* alp = atile + mb * W4_ALPHA_FLOATS;
  const unsigned alp_lds = (unsigned)(size_t)(wn_lptr_t)alp;

  auto load_tile = [&](const float* P, f32x4 (&d)[16], int i) {          // row i of the 4 x 4 tile
#pragma unroll
    for (int j = 0; j < 4; ++j) d[i * 4 + j] = *reinterpret_cast<const f32x4*>(P + a_off[i] + j * 16);
  };
  auto tf_row = [&](f32x4 (&d)[16], int i) {                              // d B, row i (in place)
    const f32x4 t0 = d[i * 4], t1 = d[i * 4 + 1], t2 = d[i * 4 + 2], t3 = d[i * 4 + 3];
    d[i * 4] = w4_sub(t0, t2);
    d[i * 4 + 1] = w4_add(t1, t2);
    d[i * 4 + 2] = w4_sub(t2, t1);
    d[i * 4 + 3] = w4_sub(t1, t3);
  };
  f32x4 V[16];
  auto tf_col = [&](const f32x4 (&d)[16], int j) {                        // column j of V = B^T (d B)
    const f32x4 d0 = d[j], d1 = d[4 + j], d2 = d[8 + j], d3 = d[12 + j];
    V[j] = w4_sub(d0, d2);
    V[4 + j] = w4_add(d1, d2);
    V[8 + j] = w4_sub(d2, d1);
    V[12 + j] = w4_sub(d1, d3);
  };

  // ---- prologue: weights and patch of chunk 0, patch of chunk 1 (NC >= 2: same item); transform chunk 0 ----
  const char* w_base = reinterpret_cast<const char*>(p.Ut + (size_t)(item0 / p.groups) * NC * WN_UCH_FLOATS + mb * 2048);
  ws_base = w_base;
  ws_lds = uch_lds;
#pragma unroll
  for (int k = 0; k < 8; ++k) piece_w(k);
  patch_item(0);
  ps_base = p_org;
  ps_lds = patch_lds;
#pragma unroll
  for (int k = 0; k < WN_PATCH_PIECES; ++k) piece_p(k);
  ps_base = p_org + 64;
  ps_lds = patch_lds + WN_MB * W4_PATCH_FLOATS * 4;
#pragma unroll
  for (int k = 0; k < WN_PATCH_PIECES; ++k) piece_p(k);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    f32x4 d0[16];
    const float* P0 = patch + mb * W4_PATCH_FLOATS;
#pragma unroll
    for (int i = 0; i < 4; ++i) load_tile(P0, d0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) tf_row(d0, i);
#pragma unroll
    for (int j = 0; j < 4; ++j) tf_col(d0, j);
  }

  f32x4 acc[16][2];
  const int f4 = lane & 3;
  // stream positions: the weights of chunk 0 and the patches of chunks 0 and 1 are on their way
  int w_it = 0, w_c = 0, p_it = 0, p_c = 1;
  int it = 0, c = 0;                                   // item / K chunk of chunk q
  for (int q = 0; q < qtotal; ++q) {
    // all DMA this wave issued during the previous chunk has landed (weights of chunk q, patch of chunk q + 1), and so have
    // the stores of an epilogue in between; behind the barrier everybody's share of the weights is visible and nobody
    // reads the other weight buffer any more
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const int wb = q & 1;
    const float* Uc = uch + wb * WN_UCH_FLOATS + b_off0;
    const float* Pn = patch + ((wb ^ 1) * WN_MB + mb) * W4_PATCH_FLOATS;     // patch of chunk q + 1
    const bool last = c == NC - 1;
    // DMA targets of this chunk: weights of chunk q + 1, patch of chunk q + 2 (the two streams advance by one chunk per
    // chunk; the divisions of an item's geometry are paid once per item).  Past the end of the stream the last chunk is
    // fetched again (into buffers nobody reads any more): no branches around the DMA.
    if (q + 1 < qtotal) {
      if (++w_c == NC) { w_c = 0; ++w_it; }
      if (w_c == 0)
        w_base = reinterpret_cast<const char*>(p.Ut + (size_t)((item0 + w_it) / p.groups) * NC * WN_UCH_FLOATS + mb * 2048);
    }
    ws_base = w_base + (size_t)w_c * (WN_UCH_FLOATS * 4);
    ws_lds = uch_lds + (wb ^ 1) * (WN_UCH_FLOATS * 4);
    if (q + 2 < qtotal) {
      if (++p_c == NC) { p_c = 0; ++p_it; }
      if (p_c == 0) patch_item(p_it);
    }
    ps_base = p_org + p_c * 64;
    ps_lds = patch_lds + wb * (WN_MB * W4_PATCH_FLOATS * 4);
    Geo ge = {false, 0, 0, 0, 0};
    if (last) {
      // bias and slopes of this item's output block -> LDS: piece k = 8 pixels (block row k) x 32 columns, then the bias.
      // Scalar row base + constant lane offset; lanes outside the image / beyond Cout and rows below the image are left
      // out (nothing reads their slots).
      ge = geometry(it);
      const int col0 = ge.ct * 32;
      if (p.epi == 2) {
        const unsigned long long am = __builtin_amdgcn_ballot_w64(ge.live && ge.bx * 8 + (lane >> 3) < H && col0 + (lane & 7) * 4 < Cout);
        const char* arow = reinterpret_cast<const char*>(p.alpha + ((size_t)(ge.by * 8) * H + ge.bx * 8) * Cout + col0);
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (ge.by * 8 + k < H) dv_dma_exec(alp_lds + k * 1024, aoffb, arow + (size_t)k * H * Cout * 4, am);
      }
      if (p.epi >= 1) {
        const unsigned long long bm = __builtin_amdgcn_ballot_w64(lane < 8 && col0 + lane * 4 < Cout);
        dv_dma_exec(alp_lds + 8192, lane16, reinterpret_cast<const char*>(p.bias + col0), bm);
      }
    }
    f32x4 bq[2][4];                                    // weight fragments of two groups: [g & 1][2 pp + nh]
    auto load_b = [&](int g) {
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
          bq[g & 1][2 * pp + nh] = *reinterpret_cast<const f32x4*>(Uc + w4_pos(g, pp) * 512 + nh * 256);
    };
    f32x4 d[16];
    load_b(0);
    // FIRST: K chunk 0 of an item - the accumulators start from the constant 0 (no zeroing pass)
    auto groups = [&](auto first_t) {
    constexpr bool FIRST = decltype(first_t)::value;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      __builtin_amdgcn_sched_barrier(0);
      if (g + 1 < 8) load_b(g + 1);
      if (g < 4) load_tile(Pn, d, g);                  // row g of the next chunk's 4 x 4 tile
      // 15 DMA pieces, three per group: the patch first (it comes from HBM), then the weights (L2 hits).  Measured: where in
      // the chunk they are issued makes no difference (they have landed when the chunk ends); each costs its issue slot.
      if (g < 5) {
#pragma unroll
        for (int k = 3 * g; k < 3 * g + 3; ++k) {
          if (k < 7) piece_p(k);
          else piece_w(k - 7);
        }
      }
      // (the empty asm statements pin each step to its group: the results are only read by the next iteration, and LLVM
      // would otherwise sink the whole transform into the loop latch, behind the last MFMA)
      if (g >= 1 && g <= 4) {
        tf_row(d, g - 1);
#pragma unroll
        for (int k = 0; k < 4; ++k) asm volatile("" : "+v"(d[(g - 1) * 4 + k]));
      }
      if (g == 5 || g == 6) {                          // (last readers of columns 0 / 1 / 2: groups 1 / 3 / 5)
#pragma unroll
        for (int j = (g == 5 ? 0 : 2); j < (g == 5 ? 2 : 3); ++j) {
          tf_col(d, j);
#pragma unroll
          for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(V[4 * i + j]));
        }
      }
#pragma unroll
      for (int jj = 0; jj < 4; ++jj)
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
          for (int nh = 0; nh < 2; ++nh)
            acc[w4_pos(g, pp)][nh] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                V[w4_pos(g, pp)][jj], bq[g & 1][2 * pp + nh][jj],
                (FIRST && jj == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[w4_pos(g, pp)][nh], 0, 0, 0);
      // spread the rest between the MFMAs (per MFMA: an LDS read, up to three vector instructions, a DMA piece)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      }
    }
    };
    if (c == 0) groups(std::integral_constant<bool, true>{});
    else groups(std::integral_constant<bool, false>{});
    __builtin_amdgcn_sched_barrier(0);
    tf_col(d, 3);
    if (last) {
      // ---- epilogue: output transform Y = A^T M A on the accumulators, bias / PReLU / stores through the staging tile ----
      const int col0 = ge.ct * 32;
      const size_t obase = (size_t)ge.n * H * H * Cout;
      unsigned e_aoff[4];
      bool e_ok[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {                    // i = 2 a + k: pass a (block rows 2 lg' + a), half k
        const int sidx = (i & 1) * 16 + (lane >> 2);
        const int row = ge.by * 8 + 2 * (sidx >> 3) + (i >> 1), cl = ge.bx * 8 + (sidx & 7);
        e_ok[i] = ge.live && row < H && cl < H;
        e_aoff[i] = e_ok[i] ? (unsigned)((row * H + cl) * Cout + col0 + f4 * 4) : 0u;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's bias / slope pieces (wave-private tile)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        if (col0 + nh * 16 >= Cout) continue;           // wave-uniform
        float y[2][4][2];                               // [a][r][b]: pixel (2 lg + a, 2 r + b) of the 8 x 8 block
#pragma unroll
        for (int rp = 0; rp < 2; ++rp) {                // tiles r = 2 rp, 2 rp + 1 as one packed pair
          w4_f32x2 tt[2][4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const w4_f32x2 m0 = {acc[j][nh][2 * rp], acc[j][nh][2 * rp + 1]}, m1 = {acc[4 + j][nh][2 * rp], acc[4 + j][nh][2 * rp + 1]},
                           m2 = {acc[8 + j][nh][2 * rp], acc[8 + j][nh][2 * rp + 1]},
                           m3 = {acc[12 + j][nh][2 * rp], acc[12 + j][nh][2 * rp + 1]};
            tt[0][j] = w4_add2(w4_add2(m0, m1), m2);
            tt[1][j] = w4_sub2(w4_sub2(m1, m2), m3);
          }
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            const w4_f32x2 y0 = w4_add2(w4_add2(tt[a][0], tt[a][1]), tt[a][2]);
            const w4_f32x2 y1 = w4_sub2(w4_sub2(tt[a][1], tt[a][2]), tt[a][3]);
            y[a][2 * rp][0] = y0.x;
            y[a][2 * rp + 1][0] = y0.y;
            y[a][2 * rp][1] = y1.x;
            y[a][2 * rp + 1][1] = y1.y;
          }
        }
        f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
        if (p.epi >= 1) bias4 = *reinterpret_cast<const f32x4*>(alp + 2048 + nh * 16 + f4 * 4);
#pragma unroll
        for (int a = 0; a < 2; ++a) {                   // two passes of 32 pixels: block rows 2 lg + a
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            stg[(lg * 8 + 2 * r) * WN_LDS_STAGE + l15] = y[a][r][0];
            stg[(lg * 8 + 2 * r + 1) * WN_LDS_STAGE + l15] = y[a][r][1];
          }
          __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): wave-private region
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int i = 2 * a + k;
            if (!e_ok[i]) continue;
            const int sidx = k * 16 + (lane >> 2);
            f32x4 v = *reinterpret_cast<const f32x4*>(stg + sidx * WN_LDS_STAGE + f4 * 4);
            v += bias4;
            if (p.U) *reinterpret_cast<f32x4*>(p.U + obase + e_aoff[i] + nh * 16) = v;
            if (p.epi == 2) {
              const int px = (2 * (sidx >> 3) + a) * 8 + (sidx & 7);
              const f32x4 al = *reinterpret_cast<const f32x4*>(alp + px * 32 + nh * 16 + f4 * 4);
              f32x4 o;
#pragma unroll
              for (int e = 0; e < 4; ++e) o[e] = v[e] > 0.f ? v[e] : al[e] * v[e];
              *reinterpret_cast<f32x4*>(p.A + obase + e_aoff[i] + nh * 16) = o;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    if (++c == NC) { c = 0; ++it; }
  }
}

// ---- weight transform U = G g G^T, laid out [column tile][K chunk][pos][n 32][k 16] ----------------------------
// one thread per (n, k) of the padded [nct * 32][NC * 16] matrix; grid.y = descriptor
__global__ __launch_bounds__(256) void wino_weights_kernel(const WinoWDesc* __restrict__ descs) {
  const WinoWDesc d = descs[blockIdx.y];
  const int NC = d.Cin / 16, nct = (d.Cout + 31) / 32;
  const long total = (long)nct * NC * 512;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    const int kk = (int)(e & 15), nn = (int)((e >> 4) & 31);
    const long blk = e >> 9;                            // ct * NC + kc
    const int kc = (int)(blk % NC), ct = (int)(blk / NC);
    const int k = kc * 16 + kk, n = ct * 32 + nn;
    float g[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int wt = d.wtmap[r * 3 + s];
        g[r][s] = n < d.Cout ? (d.nmajor ? d.W[((size_t)wt * d.Cout + n) * d.Cin + k] : d.W[((size_t)wt * d.Cin + k) * d.Cout + n])
                             : 0.f;
      }
    // rows: G g
    float t[4][3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
      t[0][s] = g[0][s];
      t[1][s] = 0.5f * (g[0][s] + g[1][s] + g[2][s]);
      t[2][s] = 0.5f * (g[0][s] - g[1][s] + g[2][s]);
      t[3][s] = g[2][s];
    }
    float* out = d.Ut + (size_t)blk * (16 * 512) + nn * 16 + ((((kk >> 2) ^ ((0x78 >> (2 * ((nn >> 2) & 3))) & 3)) << 2) | (kk & 3));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[(i * 4 + 0) * 512] = t[i][0];
      out[(i * 4 + 1) * 512] = 0.5f * (t[i][0] + t[i][1] + t[i][2]);
      out[(i * 4 + 2) * 512] = 0.5f * (t[i][0] - t[i][1] + t[i][2]);
      out[(i * 4 + 3) * 512] = t[i][2];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a stride-1 3x3 layer in the Winograd domain, F(3x3, 2x2):
//     dW = G^T [ sum over 2x2 output tiles  (B^T d B) . (A dy A^T) ] G
// d = 4x4 tile of the gathered operand X (pad 1), dy = 2x2 tile of the dense operand Y: 16 positions, each a
// [Cx x tiles] x [tiles x Cy] contraction over ALL tiles of the batch - a GEMM with a tiny output (16 x Cx x Cy) and a
// very long K, 4 multiply-adds per pixel and channel pair instead of 9.
// A workgroup owns a 64 x 64 (cx, cy) output tile for all 16 positions and a contiguous range of 8 x 8-pixel blocks
// (the M blocks of wino_conv_kernel); wave w owns the 32 x 32 quarter (w >> 1, w & 1): 16 x 2 x 2 accumulator blocks
// (256 registers, one wave per SIMD).  Per block the 10 x 10 x 64 patch of X and the 8 x 8 x 64 tile of Y arrive by
// LDS-DMA into a two-deep ring (41 KiB a stage: 5 bytes per MFMA cycle - the contraction re-uses every operand 32
// times from registers).  MFMA roles (v_mfma_f32_16x16x4_f32): rows = 16 channels of X, columns = 16 channels of Y,
// k = the four tiles of one tile row of the block; lane (l15, lg) transforms tile lg of its channel l15 itself (scalar
// B^T d B and A dy A^T from 16 + 4 ds_read_b32) - no transposes, no transformed tensors in memory.
// Each workgroup writes its partial [16][64][64] into slab s of `part`; wino_wgrad_finish_kernel sums the slabs in a fixed
// order and applies G^T . G.  Out-of-image tiles contribute zeros (zero page on the DMA side), so there is no masking.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int WG_XP = 25;                       // 1-KiB DMA pieces of the X patch: 100 pixels x 256 B
constexpr int WG_YP = 16;                       // of the Y tile: 64 pixels x 256 B
constexpr int WG_STAGE = (WG_XP + WG_YP) * 256; // floats per ring stage
}  // namespace

__global__ __launch_bounds__(256, 1) void wino_wgrad_kernel(const WinoWgradParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [2][WG_STAGE]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, lg = lane >> 4;
  const int H = p.H, Cx = p.Cx, Cy = p.Cy;
  const int nb2 = p.nbh * p.nbh;
  // workgroup -> (output tile, split)
  const int tile = blockIdx.x / p.S, split = blockIdx.x - tile * p.S;
  const int kt = tile / p.nnt, nt = tile - kt * p.nnt;
  const int b_begin = split * p.bps, b_end = min(p.nblocks, b_begin + p.bps);
  const int kw = wave >> 1, nw = wave & 1;       // this wave's 32 x 32 quarter

  // fp32 MFMA and the vector ALU are the same lanes (see wino_conv4_kernel): the first version of this loop spent as long
  // on its 356 vector instructions per 64 MFMAs - 270 of them integer address arithmetic - as on the MFMAs.  Now every
  // address is a constant: LDS reads are a lane base + an immediate, a DMA piece is a scalar block origin + a lane offset
  // fixed for the whole kernel, out-of-image lanes are switched to the zero page by an execution mask computed once per
  // block, and the two 16-channel blocks of a wave go through the transforms as packed pairs.
  //
  // DMA of a block into a ring stage: 25 + 16 pieces of 1 KiB, this wave's are k = wave + 4 i, i = 0..10 (k < 41);
  // slot e = 64 k + lane -> (pixel L = e >> 4, 16-byte piece e & 15).  The piece index is XOR-ed with 4 * ((L >> 1) & 1) so
  // that the two tiles a 32-lane read group touches (pixels two apart) sit in different halves of the 32 banks.
  constexpr int NPW = 11;
  unsigned doff[NPW];                              // byte offset from the block's X origin (pixel (-1, -1), channel kt * 64) / Y origin
  int drc[NPW];                                    // pixel relative to the block's first pixel: (row + 1) << 8 | (col + 1)
#pragma unroll
  for (int i = 0; i < NPW; ++i) {
    const int k = wave + 4 * i;
    if (k < WG_XP) {
      const int L = 4 * k + (lane >> 4);
      const int pr = L / 10, pc = L - pr * 10;
      drc[i] = (pr << 8) | pc;
      doff[i] = (unsigned)(((pr * H + pc) * Cx + (((lane & 15) ^ (((L >> 1) & 1) << 2)) << 2)) * 4);
    } else {
      const int L = (k - WG_XP) * 4 + (lane >> 4);
      drc[i] = (((L >> 3) + 1) << 8) | ((L & 7) + 1);
      doff[i] = (unsigned)((((L >> 3) * H + (L & 7)) * Cy + (((lane & 15) ^ (((L >> 1) & 1) << 2)) << 2)) * 4);
    }
  }
  unsigned vnull = 0;
  asm volatile("" : "+v"(vnull));
  const unsigned smem_lds = (unsigned)(size_t)(wn_lptr_t)smem;
  // block being fetched: scalar origins and the lane masks of this wave's pieces
  const char* d_xorg = reinterpret_cast<const char*>(p.zero);
  const char* d_yorg = d_xorg;
  unsigned long long d_mask[NPW];
  unsigned d_lds = smem_lds;
  auto dma_begin = [&](int blk, int stg) {
    const int n = blk / nb2;
    const int rem = blk - n * nb2;
    const int by = rem / p.nbh, bx = rem - by * p.nbh;
    d_xorg = reinterpret_cast<const char*>(p.X) + (((long long)(n * H + by * 8 - 1) * H + (bx * 8 - 1)) * Cx + kt * 64) * 4;
    d_yorg = reinterpret_cast<const char*>(p.Y) + (((long long)(n * H + by * 8) * H + bx * 8) * Cy + nt * 64) * 4;
    d_lds = smem_lds + stg * (WG_STAGE * 4);
#pragma unroll
    for (int i = 0; i < NPW; ++i)
      d_mask[i] = __builtin_amdgcn_ballot_w64((unsigned)(by * 8 - 1 + (drc[i] >> 8)) < (unsigned)H &&
                                              (unsigned)(bx * 8 - 1 + (drc[i] & 255)) < (unsigned)H);
  };
  auto dma_piece = [&](int i) {                        // i is a compile-time constant at every call site
    const int k = wave + 4 * i;
    if (k >= WG_XP + WG_YP) return;                    // wave-uniform (i = 10: wave 0 only)
    dv_dma_masked<false>(d_lds + k * 1024, doff[i], k < WG_XP ? d_xorg : d_yorg, d_mask[i], vnull, p.zero);
  };

  f32x4 acc[16][2][2];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[i][a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // LDS read bases (float index in a stage; the pixel / tile-row part of an address is an immediate).  X: pixel
  // L = (2 ty + i) * 10 + 2 lg + j, swizzle bit ((L >> 1) & 1) = (i + (j >> 1) + lg) & 1 - two bases per channel block;
  // Y: L = (2 ty + a) * 8 + 2 lg + b, swizzle bit lg & 1 - one base per channel block.
  int xb[2][2], yb[2];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
    for (int c = 0; c < 2; ++c) xb[kb][c] = 2 * lg * 64 + (((kw * 2 + kb) * 16 + l15) ^ (((c ^ lg) & 1) << 4));
    yb[kb] = WG_XP * 256 + 2 * lg * 64 + (((nw * 2 + kb) * 16 + l15) ^ ((lg & 1) << 4));
  }
  // raw operands of one tile quad (tile row ty of the block; this lane: tile tx = lg, channel l15 of each 16-block),
  // the two channel blocks of a wave as one packed pair
  auto load_raw = [&](const float* S, int ty, w4_f32x2 (&xr)[16], w4_f32x2 (&yr)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = (i + (j >> 1)) & 1, off = ((2 * ty + i) * 10 + j) * 64;
        xr[i * 4 + j] = (w4_f32x2){S[xb[0][c] + off], S[xb[1][c] + off]};
      }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int off = ((2 * ty + a) * 8 + b) * 64;
        yr[a * 2 + b] = (w4_f32x2){S[yb[0] + off], S[yb[1] + off]};
      }
  };
  // transforms of one quad's raw operands: V = B^T d B, M = A dy A^T (packed over the two channel blocks)
  auto transform = [&](const w4_f32x2 (&xr)[16], const w4_f32x2 (&yr)[4], w4_f32x2 (&V)[16], w4_f32x2 (&M)[16]) {
    w4_f32x2 t[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      t[0][j] = w4_sub2(xr[0 * 4 + j], xr[2 * 4 + j]);
      t[1][j] = w4_add2(xr[1 * 4 + j], xr[2 * 4 + j]);
      t[2][j] = w4_sub2(xr[2 * 4 + j], xr[1 * 4 + j]);
      t[3][j] = w4_sub2(xr[1 * 4 + j], xr[3 * 4 + j]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      V[i * 4 + 0] = w4_sub2(t[i][0], t[i][2]);
      V[i * 4 + 1] = w4_add2(t[i][1], t[i][2]);
      V[i * 4 + 2] = w4_sub2(t[i][2], t[i][1]);
      V[i * 4 + 3] = w4_sub2(t[i][1], t[i][3]);
    }
    // rows of A dy: [y0; y0 + y1; y0 - y1; -y1], then the same on the columns
    const w4_f32x2 zero2 = {0.f, 0.f};
    w4_f32x2 r[4][2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const w4_f32x2 y0 = yr[0 * 2 + b], y1 = yr[1 * 2 + b];
      r[0][b] = y0;
      r[1][b] = w4_add2(y0, y1);
      r[2][b] = w4_sub2(y0, y1);
      r[3][b] = w4_sub2(zero2, y1);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      M[i * 4 + 0] = r[i][0];
      M[i * 4 + 1] = w4_add2(r[i][0], r[i][1]);
      M[i * 4 + 2] = w4_sub2(r[i][0], r[i][1]);
      M[i * 4 + 3] = w4_sub2(zero2, r[i][1]);
    }
  };

  // Software pipeline, written out (one wave per SIMD: nothing else hides an LDS round trip): per tile quad
  //   wait for its raw operands -> transform -> issue the NEXT quad's 40 reads -> 64 MFMAs,
  // with the DMA pieces of the next block issued between the MFMA groups of this one.
  if (b_begin < b_end) {
    dma_begin(b_begin, 0);
#pragma unroll
    for (int i = 0; i < NPW; ++i) dma_piece(i);
  }
  int st = 0;
  for (int blk = b_begin; blk < b_end; ++blk) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const float* S = smem + st * WG_STAGE;
    // the block after the last is the last once more (into the stage nobody reads): no branches around the DMA
    dma_begin(min(blk + 1, b_end - 1), st ^ 1);
    w4_f32x2 xr[16], yr[4];
    load_raw(S, 0, xr, yr);
#pragma unroll
    for (int ty = 0; ty < 4; ++ty) {
      w4_f32x2 V[16], M[16];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      transform(xr, yr, V, M);
      __builtin_amdgcn_sched_barrier(0);
      if (ty + 1 < 4) load_raw(S, ty + 1, xr, yr);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pos = 0; pos < 16; ++pos) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[pos][kb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(V[pos][kb], M[pos][nb], acc[pos][kb][nb], 0, 0, 0);
        if ((pos & 3) == 3 && ty * 4 + (pos >> 2) < NPW) {   // DMA of the next block: one piece behind every 16th MFMA
          __builtin_amdgcn_sched_barrier(0);
          dma_piece(ty * 4 + (pos >> 2));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    st ^= 1;
  }
  // ---- partial slab: part[split][pos][cx][cy] ----
  float* out = p.part + (size_t)split * 16 * Cx * Cy;
#pragma unroll
  for (int pos = 0; pos < 16; ++pos)
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int cx = kt * 64 + (kw * 2 + kb) * 16 + 4 * lg + r, cy = nt * 64 + (nw * 2 + nb) * 16 + l15;
          out[((size_t)pos * Cx + cx) * Cy + cy] = acc[pos][kb][nb][r];
        }
}

// out[(r * 3 + s) * Cx + cx][cy] = (G^T (sum_split part[split]) G)[r][s]; one thread per (cx, cy), fixed summation order
__global__ __launch_bounds__(256) void wino_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                                int S, int Cx, int Cy) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  const long n = (long)Cx * Cy;
  if (e >= n) return;
  float u[16];
#pragma unroll
  for (int pos = 0; pos < 16; ++pos) u[pos] = 0.f;
  for (int sp = 0; sp < S; ++sp)
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) u[pos] += part[((size_t)sp * 16 + pos) * n + e];
  // G^T = [[1, .5, .5, 0], [0, .5, -.5, 0], [0, .5, .5, 1]]
  float t[3][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = u[0 * 4 + j] + 0.5f * (u[1 * 4 + j] + u[2 * 4 + j]);
    t[1][j] = 0.5f * (u[1 * 4 + j] - u[2 * 4 + j]);
    t[2][j] = 0.5f * (u[1 * 4 + j] + u[2 * 4 + j]) + u[3 * 4 + j];
  }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    out[(size_t)(r * 3 + 0) * n + e] = t[r][0] + 0.5f * (t[r][1] + t[r][2]);
    out[(size_t)(r * 3 + 1) * n + e] = 0.5f * (t[r][1] - t[r][2]);
    out[(size_t)(r * 3 + 2) * n + e] = 0.5f * (t[r][1] + t[r][2]) + t[r][3];
  }
}

// first stage of the slab sum when a launch has many splits (few output tiles, e.g. 256 slabs for a 64 x 64 layer): group
// g sums its splits [g * per, (g + 1) * per) in order into tmp[g]; the finish kernel then sums the (at most 16) groups
__global__ __launch_bounds__(256) void wino_wgrad_presum_kernel(const float* __restrict__ part, float* __restrict__ tmp,
                                                                int S, int per, long slab4) {
  const long f = (long)blockIdx.x * 256 + threadIdx.x;
  if (f >= slab4) return;
  const int g = blockIdx.y;
  const int s0 = g * per, s1 = min(S, s0 + per);
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int sp = s0; sp < s1; ++sp) a += reinterpret_cast<const f32x4*>(part)[(size_t)sp * slab4 + f];
  reinterpret_cast<f32x4*>(tmp)[(size_t)g * slab4 + f] = a;
}

bool wino_wgrad_supported(int NB, int H, int Cx, int Cy) {
  return Cx >= 64 && Cy >= 64 && Cx % 64 == 0 && Cy % 64 == 0 && H >= 5 && H <= 1024 && NB >= 1 &&
         (size_t)H * H * (size_t)std::max(Cx, Cy) < ((size_t)1 << 31);
}

// floats of partial slabs a launch writes (the caller provides them): splits x 16 x Cx x Cy
size_t wino_wgrad_part_floats(int NB, int H, int Cx, int Cy, int* splits_out) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  const int nbh = (H + 7) / 8;
  const long nblocks = (long)NB * nbh * nbh;
  const int tiles = (Cx / 64) * (Cy / 64);
  // Workgroups of a launch: HALF the CUs.  Alone the kernel is fastest with one workgroup per CU; in the training step it
  // runs on the weight-gradient stream beside the main stream's persistent one-workgroup-per-CU Winograd conv launches,
  // and with 128 workgroups the overlapped step takes 4.68 ms against 4.75 with 256 (192: 4.70, 96: 4.87, 64: 5.13; same
  // box, alternating runs).
  long S = std::max(1, cus / 2 / tiles);
#ifdef DV_DEBUG_EXPORTS
  {   // development build: DV_EXP_WINOW_WGS=n workgroups per launch instead of half the CUs (a tuning sweep, same results up
      // to the order of the slab sums)
    static const int wgs = DV_EXP_SWITCH("DV_EXP_WINOW_WGS");
    if (wgs > 0) S = std::max(1, wgs / tiles);
  }
#endif
  S = std::min(S, nblocks);
  if (splits_out) *splits_out = (int)S;
  return (size_t)(S + (S >= 64 ? 16 : 0)) * 16 * Cx * Cy;    // + the 16 group sums of the two-stage reduction
}

int launch_wino_wgrad(WinoWgradParams p, float* out, hipStream_t s_kernel) {
  if (!wino_wgrad_supported(p.NB, p.H, p.Cx, p.Cy) || !p.zero || !p.part) return 1;
  int S = 1;
  const size_t need = wino_wgrad_part_floats(p.NB, p.H, p.Cx, p.Cy, &S);
  if (need > p.part_capacity) return 1;
  p.nbh = (p.H + 7) / 8;
  p.nkt = p.Cx / 64;
  p.nnt = p.Cy / 64;
  p.nblocks = p.NB * p.nbh * p.nbh;
  p.S = S;
  p.bps = (p.nblocks + S - 1) / S;
  const size_t smem = (size_t)2 * WG_STAGE * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(wino_wgrad_kernel, dim3(p.nkt * p.nnt * S), dim3(256), smem, s_kernel, p);
  DV_HIP(hipGetLastError());
  (void)out;
  return OK;
}

int launch_wino_wgrad_finish(const float* part, float* out, int S, int Cx, int Cy, hipStream_t s) {
  const long n = (long)Cx * Cy;
  if (S >= 64) {                                      // (with 32 splits the one-stage sum is faster: measured)
    const int per = (S + 15) / 16, G = (S + per - 1) / per;
    const long slab4 = 16 * n / 4;
    float* tmp = const_cast<float*>(part) + (size_t)S * 16 * n;
    hipLaunchKernelGGL(wino_wgrad_presum_kernel, dim3((unsigned)((slab4 + 255) / 256), G), dim3(256), 0, s, part, tmp, S, per, slab4);
    part = tmp;
    S = G;
  }
  hipLaunchKernelGGL(wino_wgrad_finish_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, out, S, Cx, Cy);
  DV_HIP(hipGetLastError());
  return OK;
}

size_t wino_weight_floats(int Cin, int Cout) { return (size_t)((Cout + 31) / 32) * (Cin / 16) * WN_UCH_FLOATS; }

bool wino_supported(int NB, int H, int Cin, int Cout) {
  return Cin >= 16 && Cin % 16 == 0 && Cout >= 16 && Cout % 16 == 0 && H >= 5 && H <= 1024 && NB >= 1 &&
         (size_t)H * H * (size_t)std::max(Cin, Cout) < ((size_t)1 << 31);
}

int launch_wino_weights(const WinoWDesc* descs_dev, const WinoWDesc* descs_host, int n, hipStream_t s) {
  if (n <= 0) return OK;
  long most = 0;
  for (int i = 0; i < n; ++i) most = std::max(most, (long)wino_weight_floats(descs_host[i].Cin, descs_host[i].Cout) / 16);
  const int gx = (int)std::min<long>(256, (most + 255) / 256);
  hipLaunchKernelGGL(wino_weights_kernel, dim3(gx, n), dim3(256), 0, s, descs_dev);
  DV_HIP(hipGetLastError());
  return OK;
}

static int g_wino_variant = 2;                         // 1: wino_conv_kernel (eight waves), 2: wino_conv4_kernel
void debug_set_wino_variant(int v) { g_wino_variant = v; }

int launch_wino_conv(WinoParams p, hipStream_t s) {
  if (!wino_supported(p.NB, p.H, p.Cin, p.Cout) || !p.zero || !p.Ut || p.epi < 0 || p.epi > 2) return 1;
  if (p.epi == 2 && (!p.alpha || !p.A)) return 1;
  p.nbh = (p.H + 7) / 8;
  p.nct = (p.Cout + 31) / 32;
  p.NC = p.Cin / 16;
  const long blocks = (long)p.NB * p.nbh * p.nbh;
  p.groups = (int)((blocks + WN_MB - 1) / WN_MB);
  p.items = p.groups * p.nct;
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
    cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  }
  p.items_per_wg = (p.items + cus - 1) / cus;
  const int grid = (p.items + p.items_per_wg - 1) / p.items_per_wg;
  // the four-wave kernel wins where an item has >= 3 K chunks (measured per layer, tools/wino_check.py bench: 8-20 % on the
  // 64- to 256-channel layers); with two chunks per item (32 input channels) its per-item work is not amortised and the
  // eight-wave kernel is 1-4 % ahead
  if (g_wino_variant == 2 && p.NC >= 3) {
    const size_t smem4 = ((size_t)2 * WN_MB * W4_PATCH_FLOATS + 2 * WN_UCH_FLOATS + (size_t)W4_WAVES * 32 * WN_LDS_STAGE +
                          (size_t)W4_WAVES * W4_ALPHA_FLOATS) * sizeof(float);
    static bool attr4_set = false;
    if (!attr4_set) {
      DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)smem4));
      attr4_set = true;
    }
    hipLaunchKernelGGL(wino_conv4_kernel, dim3(grid), dim3(W4_THREADS), smem4, s, p);
    DV_HIP(hipGetLastError());
    return OK;
  }
  const size_t smem = ((size_t)2 * WN_MB * WN_PATCH_FLOATS + 2 * WN_UCH_FLOATS + (size_t)WN_WAVES * 32 * WN_LDS_STAGE) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    DV_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)smem));
    attr_set = true;
  }
  hipLaunchKernelGGL(wino_conv_kernel, dim3(grid), dim3(WN_THREADS), smem, s, p);
  DV_HIP(hipGetLastError());
  return OK;
}

}  // namespace dv
