// Shared declarations for the debvader_amd HIP engine (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace dv {

// status codes (mirrored in include/debvader_hip.h)
enum : int {
  OK = 0,
  E_INVALID = -1,
  E_HIP = -2,
  E_NOMEM = -3,
  E_RCCL = -4,
  E_STATE = -5,
  E_NODEVICE = -6,
};

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define DV_HIP(call)                                                     \
  do {                                                                   \
    hipError_t e__ = (call);                                             \
    if (e__ != hipSuccess) return dv::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

#define DV_TRY(call)            \
  do {                          \
    int s__ = (call);           \
    if (s__ != dv::OK) return s__; \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// MEASUREMENT switches (DV_EXP_*, DV_TIME_ENQUEUE) leave work out of a step to price it: they give WRONG results, so they
// exist only in the development library (libdebvader_hip_debug.so, -DDV_DEBUG_EXPORTS), which announces every one that is
// set on stderr.  In the product library DV_EXP_SWITCH(...) is the constant 0: the environment is not read, the names do
// not appear in the binary (tests/test_abi_and_host.py checks `strings`), and the branches fold away.
#ifdef DV_DEBUG_EXPORTS
int exp_switch(const char* name);        // engine.hip: atoi of the variable (unset or "0": off), one warning per name
#define DV_EXP_SWITCH(name) ::dv::exp_switch(name)
#else
#define DV_EXP_SWITCH(name) 0
#endif

#ifdef __HIPCC__
// Philox4x32-10 and the standard normal the engine derives from it (pairs (0,1), (2,3) of a block are one
// Box-Muller draw): shared by the sampler and the sampled-"mse" metric of the head kernels
__device__ __forceinline__ void dv_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                                 unsigned k1, unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float dv_philox_normal(unsigned row, unsigned col, unsigned stream, unsigned long long seed) {
  unsigned r[4];
  dv_philox4x32_10(row, col >> 2, stream, 0u, (unsigned)seed, (unsigned)(seed >> 32), r);
  const int a = col & 3;
  const float u1 = ((float)r[a & ~1] + 1.0f) * 2.3283064365386963e-10f;
  const float u2 = ((float)r[(a & ~1) + 1] + 1.0f) * 2.3283064365386963e-10f;
  const float rad = sqrtf(-2.0f * logf(fminf(u1, 1.0f)));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  return (a & 1) ? rad * sn : rad * cs;
}
#endif

// ---------------------------------------------------------------------------------------------
// Gather-GEMM ("gconv"): Y[m, n] = sum_k A[m, k] * B[k, n]  with
//   m  -> (stamp nb, class-grid row i, class-grid col j),   M = NB*Hc*Wc
//   k  -> (tap t, input channel c),                        K = ntaps*Cin
//   A[m,k] = X[nb, i*sin + dh[t], j*sin + dw[t], c]   (0 outside the image)
//   B[k,n] = W[wt[t]][c][n]  (k-major)   or   W[wt[t]][n][c]  (n-major)
//   output pixel = (i*sout + ph, j*sout + pw) of an [NB,Hout,Wout,Cout] tensor
// Conv2D forward, Conv2DTranspose forward (stride-2 as 4 parity classes), their data gradients and
// the Dense layers are all instances of this one contraction.
// ---------------------------------------------------------------------------------------------
// Tap table of the general kernels (gconv.hip, wgrad.hip): any kernel size up to 5 x 5 (model.py:81-91,121-134 take
// kernels[i] freely).  Entry t = (dh + 8) | (dw + 8) << 4 | weight tap index << 8.
#define DV_MAX_TAPS 25
struct TapTab {
  int n;
  unsigned t[DV_MAX_TAPS];
};

struct GConvParams {
  const float* X;
  const float* W;
  float* U;            // pre-activation / raw output (may be null)
  float* A;            // post-activation output (epi 2) (may be null)
  const float* bias;   // [Cout] or null
  const float* alpha;  // [Hout*Wout*Cout] or null
  int NB, Hin, Win, Cin;
  int Hout, Wout, Cout;
  int Hc, Wc;
  int sin, sout, ph, pw;
  int ntaps;
  TapTab xt;                   // the taps: (dh, dw) source offset and weight tap index of each
  int M, K;
  int w_nmajor;
  int epi;             // 0 raw, 1 +bias, 2 +bias then PReLU (alpha)
  int cin_shift;       // log2(Cin) if power of two else -1
  int dbg;             // 0 normal; >0 timing ablations (wrong results), see gconv.hip
};

int launch_gconv(const GConvParams& p, hipStream_t s);
void debug_set_gconv_tile(int code);  // -1: automatic; code = tile + 100*ablation
int debug_mfma_peak(float* out, int blocks, int iters, int nacc, int randomize, hipStream_t s);

// Second-generation gather-GEMM (gconv2.hip): same contraction, Cin % 32 == 0, all output-parity classes of
// one layer in a single launch.
struct GClass2 {
  int Hc, Wc, M, ph, pw, ntaps, tile0;
  unsigned long long tapcode, wtcode;
};
struct GConv2Params {
  const float* X;
  const float* W;
  float* U;
  float* A;
  const float* bias;
  const float* alpha;
  int NB, Hin, Win, Cin;
  int Hout, Wout, Cout;
  int sin, sout;
  int nclass;
  GClass2 cls[4];
  int w_nmajor;
  int epi;
  int ksplit;          // >1: grid.y K slices, raw partial slabs [ksplit][M][N] written to U
  int batch_major;     // tiles walk (pixel, stamp) instead of (stamp, pixel)
  const float* Uin;    // epi 3: pre-activation of the layer whose output gradient this launch produces
  float* dal_part;     // epi 3: d(alpha) partial slots [slots][alpha_elems] (may be null: no parameter gradients)
  float* db_part;      // epi 3: d(bias) partial rows [m tiles * WGM][Cout] (may be null)
  long alpha_elems;    // Hout*Wout*Cout
  int dbg;             // timing build: phase stamps
  float* dbg_out;
  int prio;            // ablation switch: 4 = one-chunk prefetch with the loads in front of the MFMA block
};
void debug_set_gconv2_dbg(int v, float* out);
void debug_set_gconv2_prio(int v);
int launch_splitk_finish(const float* slabs, int ksplit, long total, int N, const float* bias, const float* alpha,
                         long alpha_per_stamp, float* U, float* A, hipStream_t s);
int launch_gconv2(const GConv2Params& p, hipStream_t s);
// tile geometry the dispatcher will pick (for sizing the epi-3 partial buffers): BM rows, WGM waves along M
void gconv2_tile_geometry(const GConv2Params& p, int* bm, int* wgm, long* mtiles);
void debug_set_gconv2_tile(int code);

// Scene compositing (scene.hip): host float64 buffers in, host float64 buffers out
// device-resident gather + float32 cast (dv_infer_cutouts): starts_dev points at the first cutout of the chunk
int launch_scene_extract_f32(const double* field_dev, int F, int nb, const int* starts_dev, long count, int cs,
                             float* out_dev, hipStream_t s);
// compositing of one inference chunk on the device (dv_infer_cutouts_composite): mean / stddev / residual fields += the
// chunk's stamps at integer placements, in object order; per-stamp centre MSE against the field's own cutout
int launch_scene_composite_chunk(double* mean_f, double* std_f, double* res_f, int F, int nb, const float* loc,
                                 const float* scale, const int* places_dev, int n, int cs, hipStream_t s);
int launch_scene_center_mse(const double* field_dev, int F, int nb, const int* starts_dev, const float* loc, int n, int cs,
                            double* out_dev, hipStream_t s);
int scene_extract(const double* field_h, int F, int nb, const int32_t* starts_h, int N, int cs, double* out_h,
                  hipStream_t s);
int scene_composite(double* field_h, int F, int nb, const double* stamps_h, const double* pos_h, int N, int cs,
                    double sign, hipStream_t s);

// Strip form of the stride-1 3x3 gather-GEMM for the 32-channel high-resolution layers (gconv_strip.hip)
struct GStripParams {
  const float* X;
  const float* W;
  float* U;
  float* A;
  const float* bias;
  const float* alpha;
  const float* zero;   // >= 16 bytes of zeros (source of the out-of-image patch slots)
  int NB, H, Wd, Cin, Cout;
  unsigned long long tapcode, wtcode;   // nine taps: (dh, dw) and weight index, as in GConvParams
  int epi;             // 0 raw, 1 +bias, 2 +bias then PReLU
  int R, patch_floats, strips_per_stamp, nstrips, strips_per_wg;   // filled by the launcher
};
int launch_gconv_strip(GStripParams p, bool nmajor, hipStream_t s);   // 1 = not taken (use gconv2)
int launch_gconv_strip8(GStripParams p, hipStream_t s);               // first layer (Cin 8 -> 32, k-major); 1 = not taken

// Winograd F(2x2, 3x3) form of the stride-1 3x3 layers (wino.hip): X [NB,H,H,Cin] -> [NB,H,H,Cout], pad 1
struct WinoParams {
  const float* X;
  const float* Ut;     // transformed weights [column tile][K chunk][16 pos][32 n][16 k] (wino_weights_kernel)
  float* U;
  float* A;
  const float* bias;
  const float* alpha;
  const float* zero;   // >= 16 bytes of zeros (source of the out-of-image patch slots)
  int NB, H, Cin, Cout;
  int epi;             // 0 raw, 1 +bias, 2 +bias then PReLU
  int nbh, nct, NC, groups, items, items_per_wg;   // filled by the launcher
};
struct WinoWDesc {
  const float* W;      // nine taps, [wt][k][n] or (nmajor) [wt][n][k]
  float* Ut;
  int Cin, Cout, nmajor;
  int wtmap[9];        // weight tap index of the input offset (r - 1, s - 1), r * 3 + s
};
// weight gradient of a stride-1 3x3 layer in the Winograd domain (wino.hip): out[9][Cx][Cy] = sum X (gathered, pad 1) x Y
struct WinoWgradParams {
  const float* X;      // [NB,H,H,Cx]
  const float* Y;      // [NB,H,H,Cy]
  float* part;         // partial slabs [S][16][Cx][Cy]
  size_t part_capacity;
  const float* zero;
  int NB, H, Cx, Cy;
  int nbh, nkt, nnt, nblocks, S, bps;   // filled by the launcher
};
bool wino_wgrad_supported(int NB, int H, int Cx, int Cy);
size_t wino_wgrad_part_floats(int NB, int H, int Cx, int Cy, int* splits_out);
int launch_wino_wgrad(WinoWgradParams p, float* out, hipStream_t s);   // 1 = not taken
int launch_wino_wgrad_finish(const float* part, float* out, int S, int Cx, int Cy, hipStream_t s);
bool wino_supported(int NB, int H, int Cin, int Cout);
size_t wino_weight_floats(int Cin, int Cout);
int launch_wino_weights(const WinoWDesc* descs_dev, const WinoWDesc* descs_host, int n, hipStream_t s);
int launch_wino_conv(WinoParams p, hipStream_t s);   // 1 = not taken

// Stride-2 data-gradient form with the four parity classes fused per workgroup (gconv_s2.hip).
// Class c = 2*[row parity has two taps] + [column parity has two taps]; neighbour e = 2*[dh == x] + [dw == x].
struct GConvS2Params {
  const float* X;
  const float* W;      // n-major taps: W[wt][n][k]
  float* U;
  float* A;
  const float* bias;
  const float* alpha;
  int NB, Hin, Win, Cin;
  int Hout, Wout, Cout;
  int Hc, Wc, M;       // base grid (ceil(Hout/2)) and NB*Hc*Wc
  int epi;             // 0 raw, 1 +bias, 2 +bias then PReLU
  int ndh[4], ndw[4];  // source offset of neighbour e
  int cph[4], cpw[4];  // output parity of class c
  int wt[4][4];        // weight tap index of (neighbour e, class c); unused pairs 0
  unsigned* dbg_out;   // per-workgroup timeline stamps (tuning aid) or null
};
int launch_gconv_s2(const GConvS2Params& p, hipStream_t s);
void debug_set_gconv_s2_tile(int code);
void debug_set_wino_variant(int v);   // 1: eight-wave Winograd conv kernel, 2: four-wave pipelined one (default)
void debug_set_gconv_s2_dbg(unsigned* out);

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[(t, cx), cy] = sum_p Xg[p, t][cx] * dY[p][cy], split over pixel ranges into
// partial slabs [nsplit][9*Cx][Cy] that reduce_partials() sums in a fixed order (deterministic).
// ---------------------------------------------------------------------------------------------
struct WGradParams {
  const float* X;     // gathered side [NB,Hx,Wx,Cx]
  const float* Y;     // dense side    [NB,Hy,Wy,Cy]
  float* part;        // [nsplit][rows_total][Cy]
  int NB, Hx, Wx, Cx;
  int Hy, Wy, Cy;
  int Hc, Wc, sx, sy, ph, pw;
  int ntaps;
  TapTab xt;          // the taps (dh, dw, weight tap index)
  int P;              // NB*Hc*Wc
  int rows_total;     // ntaps*Cx (slab rows; launch covers rows wt*Cx..)
  int nsplit;
  int pchunk;         // pixels per split (multiple of 32)
  // filled by launch_wgrad: exact division of a pixel index by Hc*Wc and by Wc as multiply-high + shifts (the gather
  // decodes its pixel once per chunk and thread; a 32-bit division is ~35 vector instructions, this is 5 - and on
  // gfx950 every vector instruction is fp32-MFMA time)
  unsigned div_hw_m, div_hw_s1, div_hw_s2, div_w_m, div_w_s1, div_w_s2;
};
int launch_wgrad(const WGradParams& p, hipStream_t s);
// Dense kernel gradient G[i][j] = sum_b X[b][i] * Y[b][j] (fp32 rows X [NB][ldx], Y [NB][ldy]; Dense layers of the trunk,
// model.py:96-98,114-117): one pass, written once, no slabs (wgrad.hip, round 6)
int launch_dense_wgrad_tn(const float* X, int ldx, const float* Y, int ldy, int NB, int I, int J, float* G, int ldg,
                          hipStream_t s);

// Strip form for the high-resolution few-channel layers (wgrad_strip.hip)
struct WStripParams {
  const float* X;
  const float* Y;
  float* part;
  size_t part_capacity;   // floats available in `part`
  int NB, Hx, Wx, Hy, Wy;
  int pb;
  const float* zero;      // >= 16 bytes of zeros in device memory (source of out-of-image LDS-DMA pieces)
  int R, XR, xs_floats, ys_floats, strips_per_stamp, nstrips, strips_per_wg;   // filled by the launcher
  int dbg;                // timing ablations (wrong results), bit flags: 1 no refill DMA, 2 no MFMA, 4 phase stamps, 8 MFMA only
  // first-layer form only (Cx = 8): fuse the PReLU backward of the layer's output.  Y is then d(activation).
  const float* U;         // pre-activation [NB,Hy,Wy,32]; null: Y already is d(pre-activation)
  const float* alpha;     // PReLU slopes [Hy,Wy,32]
  float* dal_part;        // d(alpha) partials [groups][Hy*Wy*32]
  float* db_part;         // d(bias) partials [workgroups][32]
  size_t dal_capacity, db_capacity;   // floats available in dal_part / db_part
  long alpha_elems;       // Hy*Wy*32
  int groups;             // filled by the launcher
  int* groups_out;        // host: number of d(alpha) partial slabs written
};
bool wgrad_strip8_fusable(int Hy, int Wy);
void debug_set_strip(int v);
bool wgrad_strip_supported(int Cx, int Cy, int sx, int ntaps);
int launch_wgrad_strip(const WStripParams& p, int Cx, int Cy, int sx, hipStream_t s, int* nsplit_out);

// out[(r/Cpad)*Creal + r%Cpad][c] = scale * sum_s part[s][r][c]   for r%Cpad < Creal
// the slab reductions of a whole backward pass in one launch: entry i sums nsplit slabs of slab4 float4 elements
// ([rows][ncols4] each) into out, rows compacted from cpad to creal channels like launch_reduce_partials
#define DV_WRED_MAX 24
struct WRedEntry {
  const float* part;
  float* out;
  int nsplit, slab4, ncols4, cpad, creal;
};
struct WRedBatch {
  WRedEntry e[DV_WRED_MAX];
  int count;
};
int launch_reduce_partials_batch(const WRedBatch& b, hipStream_t s);
int launch_reduce_partials(const float* part, float* out, int nsplit, long slab_elems, int ncols, int cpad,
                           int creal, hipStream_t s);

// pointwise / reduction kernels -----------------------------------------------------------------
struct HeadParams {
  const float* tpre;   // [NB,Hd,Hd,2*nb] head conv output before relu
  const float* y;      // dataset labels [*,H,H,nb] (null: no loss)
  const int* idx;      // per-stamp dataset row (null: first + b)
  int first;
  float* dt;           // [NB,Hd,Hd,2*nb] gradient wrt tpre (null: none)
  float* loc;          // [NB,H,H,nb] or null
  float* scale;        // [NB,H,H,nb] or null
  float* part;         // [nblocks][2] partial sums (nll, squared error)
  int NB, Hd, H, nb, crop0;
  int ld;              // channels per pixel row of tpre / dt (2*nb, or padded to 16)
  float sigma_floor;
  float gscale;        // 1/(Bglobal*H*H*nb)
  // Keras' "mse" metric of the reference compares the labels with a SAMPLE of the output distribution (model.py:158
  // convert_to_tensor_fn = sample, train.py:128): with mse_sample the squared error is taken against
  // loc + sigma * eps, eps(stamp b, element e) = Philox4x32-10(counter (b, e / 4, mse_stream, 0), key mse_seed), e the
  // element index inside the [H,H,nb] stamp (oracle: vae_oracle.philox_normal(seed, stream, B, H*H*nb))
  int mse_sample;
  int mse_row0;        // index of this launch's first stamp inside the rank's batch (forward lanes)
  unsigned mse_stream;
  unsigned long long mse_seed;
};
#define DV_MSE_STREAM 0x4D534500u   /* + rank */
int launch_head(const HeadParams& p, hipStream_t s, int* nblocks_out);

// input BatchNorm: the per-band state / sum rows are DV_BN_MAXC wide (bands 1 .. 15: the folded first conv reads bands + 1
// channels of an 8- or 16-channel row)
constexpr int DV_BN_MAXC = 16;
int launch_bn_stats(const float* x, const int* idx, int first, int NB, int HW, int C, float* part, int* nblocks,
                    hipStream_t s);
// out[c] = scale * sum_r part[r*ld + c] for c < ncols (ld = row stride, 0: ncols)
int launch_reduce_rows_f64(const float* part, int nrows, int ncols, float* out, float scale, hipStream_t s,
                           int ld = 0);
// bnstate: [0..C) scale, [C..2C) shift, [2C..3C) mean, [3C..4C) inv_std
int launch_bn_finalize(const float* sums, float count, int C, const float* gamma, const float* beta,
                       float* moving_mean, float* moving_var, float eps, float momentum, int unbiased,
                       int training, int update_moving, float* bnstate, hipStream_t s);
int launch_bn_apply(const float* x, const int* idx, int first, int NB, int HW, int C, int Cpad, const float* bnstate,
                    float* xn, hipStream_t s);
int launch_prelu_fwd(const float* u, const float* alpha, float* a, long NB, int E, hipStream_t s);
// da -> du in place; dalpha partials [nsplit][E]; dbias partials (mode by HW): see pointwise.hip
int launch_prelu_bwd(float* da, const float* u, const float* alpha, int NB, int E, int C, int nsplit,
                     float* dalpha_part, float* dbias_part, int* dbias_rows, hipStream_t s);
int launch_colsum(const float* x, long rows, int C, float* part, int* nrows_part, hipStream_t s);
// out[b][n] = sum_k x[b][k] * W[n][k], N <= 64 (one wave per row)
int launch_dense_narrow(const float* x, const float* W, float* out, int NB, int K, int N, hipStream_t s);


struct SamplerParams {
  const float* t;      // [NB, d + d(d+1)/2]
  float* eps;          // [NB, d] (read; written first when gen != 0)
  float* z;            // [NB, d]
  float* kl;           // [NB]
  float* stddev;       // [NB, d] or null
  int NB, d;
  int ldt, ldz;        // row strides of t and of eps / z / stddev (>= d + d(d+1)/2 and >= d; pad columns are written as zeros)
  float diag_shift;
  int gen;             // 1: generate eps with Philox(seed, stream, row0 + b)
  unsigned long long seed;
  unsigned stream;
  unsigned row0;
  int rep_nb;          // > 0: row r is Monte-Carlo sample r / rep_nb of stamp r % rep_nb (seed + sample, t row of the stamp)
  const unsigned long long* seed_ptr;   // non-null: the seed is read from device memory (replayed hipGraphs)
  // nslab > 0 (bf16 engine, btrunk.hip): the encoder Dense arrives as K-split partial sums; the row is
  // t[b][i] = tbias[i] + sum_s slab[s][b][i] (added in order), formed here and also written to `t_out` (same strides as
  // t, pad columns zero) for the backward pass and the API - the finish launch of the split product is this kernel
  const float* slab;   // [nslab][*][lds]
  const float* tbias;  // [d + d(d+1)/2]
  float* t_out;
  long slab_stride;
  int nslab, lds;
  // non-null: the PReLU in front of the decoder's first Dense (model.py:113) is applied here as well,
  // ain[b][i] = z > 0 ? z : alpha_in[i] * z (row stride ldz, pad columns zero) - one launch less on the forward chain
  const float* alpha_in;
  float* ain;
};
int launch_sampler_fwd(const SamplerParams& p, hipStream_t s);
// ldt: row stride of t and dt, ldz: of eps, z and dz (pad columns of dt are written as zeros)
int launch_sampler_bwd(const float* t, const float* eps, const float* z, const float* dz, float* dt, int NB, int d,
                       int ldt, int ldz, float diag_shift, float kls, hipStream_t s);

int launch_adam(float* w, float* m, float* v, const float* g, long n, float lr_t, float b1, float b2, float eps,
                hipStream_t s);
int launch_pad_w1(const float* w, const float* gamma, const float* beta, float* wp, int taps, int cin, int cpad,
                  int cout, hipStream_t s);
int launch_bn_conv0_grads(const float* G, const float* w, const float* gamma, const float* beta, float* dW,
                          float* dgamma, float* dbeta, int taps, int cin, int cpad, int cout, hipStream_t s);
int launch_fill(float* p, long n, float v, hipStream_t s);
int launch_normalise(float* x, long n, bool inverse, hipStream_t s);
int launch_welford_update(const float* x, float* mean, float* m2, long n, int k, hipStream_t s);
int launch_welford_finish(float* m2, long n, int count, hipStream_t s);
// x holds `reps` consecutive blocks of n elements (samples k0 .. k0+reps-1 of the same n statistics), folded in order
int launch_welford_update_multi(const float* x, float* mean, float* m2, long n, int reps, int k0, hipStream_t s);
int launch_pad_cols(const float* src, float* dst, int rows, int nsrc, int ndst, hipStream_t s);
int launch_take_cols(const float* src, float* dst, int rows, int nsrc, int ndst, hipStream_t s);
int launch_gather_rows(const float* src, const int* idx, int first, int NB, long row_elems, float* dst,
                       hipStream_t s);


// ---- scalar-base LDS-DMA (device code only) -------------------------------------------------------------------
// fp32 MFMA and the vector ALU are the same lanes on gfx950 (the fp32 matrix peak IS the vector peak), so every vector
// instruction a wave issues - including the address selects in front of a gather - is time its MFMA stream does not get.
// These helpers issue one 1-KiB global_load_lds piece whose address is a SCALAR base plus a 32-bit lane offset that can be
// a kernel-long constant; lanes that must read zeros (outside the image) are switched by an execution mask, which costs one
// compare per piece instead of a 64-bit select.  All lanes of the wave must be active at the call.
#if defined(__HIPCC__)
// one 1-KiB LDS-DMA piece, address = scalar base + 32-bit lane offset; the lanes outside `m_dma` (of the first 16 lanes
// when LAST16: the piece that ends a buffer) get 16 bytes of the zero page instead
__device__ __forceinline__ void dv_dma_exec(unsigned lds, unsigned voff, const void* sbase, unsigned long long mask);
template <bool LAST16>
__device__ __forceinline__ void dv_dma_masked(unsigned lds, unsigned voff, const void* sbase, unsigned long long m_dma,
                                              unsigned vnull, const void* zero) {
  dv_dma_exec(lds, voff, sbase, m_dma);
  const unsigned long long m_zero = LAST16 ? (~m_dma & 0xffffull) : ~m_dma;
  if (m_zero) dv_dma_exec(lds, vnull, zero, m_zero);   // (scalar branch: interior pieces issue one instruction)
}
// one LDS-DMA piece for the lanes of `mask` only (the other lanes' slots keep what they hold)
__device__ __forceinline__ void dv_dma_exec(unsigned lds, unsigned voff, const void* sbase, unsigned long long mask) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_mov_b64 exec, %1\n\t"
      "global_load_lds_dwordx4 %2, %3\n\t"
      "s_mov_b64 exec, -1"
      :
      : "s"(lds), "s"(mask), "v"(voff), "s"(sbase)
      : "memory");
}
__device__ __forceinline__ void dv_dma(unsigned lds, unsigned voff, const void* sbase) {
  asm volatile(
      "s_mov_b32 m0, %0\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2"
      :
      : "s"(lds), "v"(voff), "s"(sbase)
      : "memory");
}
#endif

}  // namespace dv
