// bf16 kernel family of the debvader_amd engine (gfx950 / MI355X only): BASELINE configs[2].
//
// Activations of the conv / conv-transpose stacks (model.py:79-98,112-137) are stored in bf16, STAMP-INNER:
// a tensor is [H][W][NBp][C] (NBp = stamps per step padded to a multiple of 16), so that "the same pixel of 16
// consecutive stamps" is one contiguous [16][C] block.  Every contraction then has a tap validity that is uniform
// over a 16-row MFMA block (no row tables, no zero-padding MACs), one 1-KiB LDS-DMA instruction moves one
// (16 stamps x 32 channels) operand block, and the batch reductions of the PReLU backward (d(alpha) over stamps)
// are sums over the rows of an accumulator tile.  MFMA: v_mfma_f32_16x16x32_bf16, fp32 accumulation.
// fp32 stay: master weights / Adam, the dense trunk + sampler (SURVEY hard part 4), the head conv output and the
// relu / crop / Normal-NLL head.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace dv {

enum { BEPI_RAW32 = 0, BEPI_FWD = 1, BEPI_RAWBF = 2, BEPI_BWD = 3, BEPI_HEAD = 4 };
// BEPI_HEAD (row-strip form of the 16-column head conv only): the head of the network - relu / crop / sigma floor / Normal NLL
// and its gradient, bf_head_kernel's arithmetic - runs in the conv's epilogue on the fp32 tile the wave has just produced:
// the fp32 head tensor (67 MB at 256 stamps) is neither written nor read back, one launch less at the end of every forward
struct BHeadFuse {
  const float* y;      // dataset labels [*,H,H,nb]
  const int* idx;
  int first;
  void* dt;            // bf16 [Hd*Hd][NBp][16] gradient wrt the head conv's output (zero outside the crop / pad stamps)
  float* part;         // [workgroups][2] partial sums (NLL, squared error)
  int NB, H, nb, crop0;
  float sigma_floor, gscale;
  int mse_sample;
  unsigned mse_stream;
  unsigned long long mse_seed;
};

// Gather-GEMM over stamp-inner tensors: out[pix][b][n] = sum_{tap,c} X[src(pix,tap)][b][c] * W[n][tap*Cin + c]
//   form 0 (Conv2D forward, Conv2DTranspose data gradient): source pixel = out*s + k - pb
//   form 1 (Conv2DTranspose forward, Conv2D data gradient): out + pb = s*source + k
struct BConvParams {
  const void* X;       // bf16 [Hin*Hin][NBp][Cin]
  const void* W;       // bf16 [Cout][Kpad], k = tap*Cin + c, zero padded to Kpad (multiple of 32)
  const void* zero;    // >= 1 KiB of zeros (source of out-of-image operand blocks)
  void* U;             // bf16 [Hout*Hout][NBp][Cout]: FWD pre-activation, RAWBF raw sums, BWD d(pre-activation)
  void* A;             // bf16, FWD: PReLU output
  float* Uf;           // fp32 [Hout*Hout][NBp][Cout] (RAW32, + bias)
  const float* bias;   // [Cout] (FWD, RAW32) or null
  const float* alpha;  // [Hout*Hout][Cout] (FWD, BWD)
  const void* Uin;     // bf16 (BWD): pre-activation of the layer whose output gradient this launch produces
  float* dal_part;     // (BWD, may be null) d(alpha) partials [nparts][Hout*Hout][Cout], nparts = NBp/64
  float* db_part;      // (BWD, may be null) d(bias)  partials, same shape
  int Hin, Hout, Cin, Cout, NBp;
  int form, s, pb;
  int Kpad;
  int epi;
  int ksz;             // kernel size 1 .. 5 (0: 3); taps t = kh * ksz + kw, k = t * Cin + c
  BHeadFuse hd;        // BEPI_HEAD only
#ifdef DV_DEBUG_EXPORTS   // (development library only: every translation unit that sees this struct is built both ways)
  int exp;             // measurement switch (DV_EXP_BCONV; results are wrong): 1 = the K loop stops after one step,
                       // 2 = launch_bconv returns without launching, 3 = K loop as usual but no epilogue stores
#endif
};
int launch_bconv(const BConvParams& p, hipStream_t s);
// > 0: a launch with these parameters takes the row-strip form that carries BEPI_HEAD, with that many workgroups (= rows of
// hd.part); 0: it would not (another tile form, another stamp padding) and the head has to run as a kernel of its own
long bconv_head_tiles(const BConvParams& p);
// BWD epilogue is usable (a wave's four 16-stamp groups share one pixel)
static inline bool bconv_bwd_fusable(int NBp) { return (NBp & 63) == 0; }

// Weight gradient over stamp-inner tensors:
//   out[tap][cx][cy] = sum_{gy pixel, stamp} X[gy*s + tap - pb][stamp][cx] * Y[gy][stamp][cy]
// written as nsplit partial slabs [nsplit][9][Cx][Cy] (fp32) that reduce_partials() sums in a fixed order.
struct BWgradParams {
  const void* X;       // bf16 [Hx*Hx][NBp][Cx]
  const void* Y;       // bf16 [Hy*Hy][NBp][Cy]
  const void* zero;
  float* part;         // [nsplit][9*Cx*Cy]
  size_t part_capacity;
  int Hx, Cx, Hy, Cy, NBp;
  int s, pb;
};
int launch_bwgrad(const BWgradParams& p, hipStream_t s, int* nsplit_out);

// ---- dense trunk on the bf16 matrix cores (btrunk.hip) ----------------------------------------------------------
// C[m][n] = sum_k A[m][k] * B[n][k], m = stamp: Flatten -> PReLU -> Dense (model.py:94-98), Dense -> PReLU -> Reshape
// (model.py:116-119) and their data gradients, with the layout changes and PReLU passes of the two seams between the
// conv stacks and the sampler folded into the operand loads and the epilogues.
enum { BGA_ROWS_F32 = 0, BGA_STAMP = 1, BGA_STAMP_PRELU = 2 };
enum { BGE_SLAB = 0, BGE_STAMP_BIAS_PRELU = 1, BGE_STAMP_GATE2 = 2 };
struct BGemmParams {
  const void* A;         // ROWS_F32: fp32 [Mreal][lda] (rounded to bf16 on load; rows >= Mreal and columns >= Kreal read as 0)
                         // STAMP / STAMP_PRELU: bf16 stamp-inner [P][NBp][C], k = p * C + c (C a multiple of 64)
  const void* B;         // bf16 [N][ldb], k contiguous; rows past the real n count and columns past the real k count are zero
  int amode, epi;
  int M, Mreal;          // M = NBp rows are computed, Mreal = NB exist (pad stamps come out as zeros)
  int N, K, Kreal;       // N a multiple of 32, K of 64
  int lda, ldb;
  int NBp, C;            // stamp-inner geometry of A
  const float* a_alpha;  // STAMP_PRELU: [K] slopes applied on load
  // SLAB: fp32 slab[s][M][ldc], s < nslab: the K range in nslab * wg_ksplit runs, the wg_ksplit runs of one workgroup
  // summed through LDS in wave order; the consumer adds the slabs in order
  float* slab;
  long slab_stride;
  int ldc, nslab, wg_ksplit;
  // STAMP_*: column n = p * Co + c of row m goes to bf16 [(p * NBp + m) * Co + c]
  int Co;
  const float* bias;     // BIAS_PRELU: [N]
  const float* alpha;    // BIAS_PRELU: [N]
  void* U;               // BIAS_PRELU: pre-activation (may be null)
  void* Aout;            // BIAS_PRELU: PReLU output
  // GATE2 (C = d(flatten PReLU output)): du = C * (a7 > 0 ? 1 : alpha_flat) * (u7 > 0 ? 1 : alpha7)
  const void* a7;        // bf16 stamp-inner: activation of the last encoder conv (input of the flatten PReLU)
  const void* u7;        // its pre-activation
  const float* alpha_flat;
  const float* alpha7;
  void* dU;              // bf16 stamp-inner d(pre-activation)
  float* part_dal_flat;  // [ceil(M / 32)][N] partial column sums of C * min(a7, 0) (all three null: no gradients wanted)
  float* part_dal7;      // ... of C * gate_flat * min(u7, 0)
  float* part_db;        // ... of du
};
int launch_bgemm(const BGemmParams& p, hipStream_t s);
// G[i][j] = sum_m X[m][i] * Y[m][j] (dense kernel gradients, train.py:27-37's backward): fp32, written once, no slabs
struct BGemmTnParams {
  const void* X;         // xmode ROWS_F32: fp32 [Mreal][ldx]; STAMP / STAMP_PRELU: bf16 [P][NBp][Cx], i = p * Cx + c
  const void* Y;
  int xmode, ymode;
  int NBp, Mreal;
  int I, Ireal, J, Jreal;   // I, J multiples of 32; rows i >= Ireal / columns j >= Jreal are neither read nor written
  int ldx, ldy, Cx, Cy;
  const float* x_alpha;  // STAMP_PRELU: [I]
  float* G;              // [Ireal][ldg]
  int ldg;
};
int launch_bgemm_tn(const BGemmTnParams& p, hipStream_t s);
// Backward of the sampler side of the trunk in ONE launch (one workgroup per stamp), model.py:112-115 and :43-58 in reverse:
//   d(hidden activation) = sum of the K-split slabs of the data gradient of Dense(560 -> flat)   [bgemm SLAB]
//   d(hidden pre-activation) = that * PReLU gate(u_h)                  -> duh   (+ rows of d(alpha_h) terms -> dalh)
//   d(z') = d(hidden pre-activation) . W0^T                             (560 -> latent_dim, wave butterflies)
//   d(z)  = d(z') * PReLU gate(z)                                       -> dz    (+ rows of d(alpha_in) terms -> dalin)
//   d(t)  = backward of z = mu + L eps and of the KL regulariser        -> dt    (sampler_bwd_kernel's arithmetic)
// until round 6: split-K finish, PReLU backward, narrow Dense, PReLU backward, sampler backward - five launches of the
// main stream's chain.  The batch sums (d(bias), d(alpha)) are column sums of the row outputs: launch_bt_colsums.
struct BMidBwdParams {
  const float* slab;     // [nslab][*][lds]
  long slab_stride;
  int nslab, lds;
  const float* uh;       // [NB][hid] pre-activation of the hidden Dense
  const float* alpha_h;  // [hid]
  const float* W0;       // [>= d][hid] kernel of Dense(latent -> hid), row = latent index
  const float* z;        // [NB][ldz]
  const float* alpha_in; // [d]
  const float* eps;      // [NB][ldz]
  const float* t;        // [NB][ldt]
  float* duh;            // [NB][hid]
  float* dalh;           // [NB][hid] or null
  float* dz;             // [NB][ldz]
  float* dalin;          // [NB][ldz] or null
  float* dt;             // [NB][ldt] (pad columns zero)
  int NB, hid, d, ldt, ldz;
  float diag_shift, kls;
};
int launch_bt_mid_bwd(const BMidBwdParams& p, hipStream_t s);
// up to four column sums in one launch: out[c] = sum_b x[b * ld + c], c < n (fp64 accumulation, fixed order)
struct BColsums {
  const float* x[4];
  float* out[4];
  int ld[4], n[4];
  int count, NB;
};
int launch_bt_colsums(const BColsums& c, hipStream_t s);
// out[b][i] = (bias ? bias[i] : 0) + sum_s slab[s][b][i] for i < n, 0 for n <= i < ldo
int launch_bt_finish_rows(const float* slab, int nslab, long slab_stride, int lds, const float* bias, float* out, int NB,
                          int n, int ldo, hipStream_t s);

// ---- pointwise kernels of the family (bf16_point.hip) -------------------------------------------------------
// dataset rows (fp32 [*, HW, C], row = idx ? idx[b] : first + b) -> normalised input, bf16 [HW][NBp][16]:
// channels 0..C-1 = (x - mean) * inv_std (bnstate[2C..4C)), channel C = 1, rest 0; stamps >= NB are zero
int launch_bf_input(const float* x, const int* idx, int first, int NB, int NBp, int HW, int C, const float* bnstate,
                    void* xh, hipStream_t s);
// bf16 [P][NBp][C] -> fp32 [NB][P*C]
int launch_bf_to_rows(const void* src, float* dst, int NB, int NBp, int P, int C, hipStream_t s);
// fp32 [NB][P*C] -> bf16 [P][NBp][C] (stamps >= NB zero)
int launch_bf_from_rows(const float* src, void* dst, int NB, int NBp, int P, int C, hipStream_t s);
// PReLU backward over a stamp-inner tensor: du = da * (u > 0 ? 1 : alpha[pix][c]) (in place allowed);
// dalpha[pix][c] = sum_b da * min(u, 0) written directly (may be null); dbias partial rows [P][C] (may be null)
int launch_bf_prelu_bwd(const void* da, const void* u, const float* alpha, void* du, float* dalpha, float* db_rows,
                        int NBp, int P, int C, hipStream_t s);
// column sums of a bf16 [rows][C] tensor into partial rows [nrows_out][C] (fp32)
int launch_bf_colsum(const void* x, long rows, int C, float* part, int* nrows_out, hipStream_t s);

struct BHeadParams {
  const float* tpre;   // fp32 [Hd*Hd][NBp][cw] head conv output before relu
  const float* y;      // dataset labels [*,H,H,nb] (null: no loss)
  const int* idx;
  int first;
  void* dt;            // bf16 [Hd*Hd][NBp][cw] gradient wrt tpre (null: none); zero outside the crop / pad rows
  float* loc;          // fp32 [NB,H,H,nb] or null
  float* scale;
  float* part;         // [nblocks][2]
  int NB, NBp, Hd, H, nb, crop0;
  int cw;              // columns of tpre / dt: 16 (1 .. 7 bands) or 32 (8 .. 15 bands); 2 * nb <= cw
  float sigma_floor, gscale;
  int mse_sample;      // see HeadParams (common.h)
  unsigned mse_stream;
  unsigned long long mse_seed;
};
int launch_bf_head(const BHeadParams& p, hipStream_t s, int* nblocks_out);

// Batched fixed-order reductions of the fused PReLU-backward partials of a whole backward pass (two launches instead of
// two per layer): entry i sums nparts slabs of n floats, out[e] = sum_p src[p*n + e]; entries with cols > 0 are then
// column-summed, final[c] = sum_r out[r*cols + c] (d(bias): out is a scratch [pixels][cols] image).
#define DV_BF_MAX_RED 64   /* 8 L - 2 entries per backward pass: covers DV_MAX_LEVELS = 8; bf_dgrad_prelu flushes when full */
struct BRedEntry {
  const float* src;
  float* out;
  float* final_out;    // null: out is the result
  int nparts, n, cols;
};
struct BRedBatch {
  BRedEntry e[DV_BF_MAX_RED];
  int count;
};
int launch_bf_reduce_batch(const BRedBatch& b, hipStream_t s);

// one weight matrix of the family, produced from the fp32 master tensor by bf_cast_weights
struct BCastDesc {
  const float* src;    // master tensor [taps][A][B] (taps, then two channel axes), or null (descriptor unused)
  void* dst;           // bf16 [N][Kpad]
  int A, B;            // master axes after the tap axis
  int n_is_b;          // 1: n = B index, c = A index (dst[n][tap*A + c] = src[tap][c][n]); 0: n = A, c = B
  int taps;            // kernel taps (ksz * ksz; 0: 9)
  int N, Cin, Kpad;    // N rows written (>= real n count: extra rows zero), Cin = c extent in dst (>= real: zero)
  // first conv with the input BatchNorm folded in: c < nbands scaled by gamma[c], c == nbands = sum_c w*beta
  const float* gamma;
  const float* beta;
  int nbands;
};
int launch_bf_cast_weights(const BCastDesc* descs_dev, const BCastDesc* descs_host, int n, hipStream_t s);

}  // namespace dv
