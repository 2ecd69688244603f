/* Development entry points of libdebvader_hip_debug.so (NOT of the product library libdebvader_hip.so): kernel micro-benchmarks and cross-checks that tools/ (layer_bench.py,
 * timeline.py, one_layer.py, peak.py, s2_check.py, clock_probe.py) call.  NOT part of the drop-in boundary
 * (include/debvader_hip.h): nothing in debvader_amd/ uses them. */
#ifndef DEBVADER_HIP_DEBUG_H
#define DEBVADER_HIP_DEBUG_H
#include "debvader_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: these declarations are its surface */
#endif

/* kernel micro-benchmarks on random data (tuning aid; average ms per call over `iters`).
 * gconv: source [NB,Hs,Hs,Cs] -> target [NB,Ht,Ht,Ct]; dgrad_form selects the parity-class form; tile -1 = automatic */
int dv_debug_gconv(dv_ctx* ctx, int32_t NB, int32_t Hs, int32_t Cs, int32_t Ht, int32_t Ct, int32_t stride,
                   int32_t pad_before, int32_t dgrad_form, int32_t nmajor, int32_t epi, int32_t single_tap,
                   int32_t tile, int32_t iters, float* ms_out);
/* runs one layer through the specialised kernel the dispatcher picks (strip form, fused stride-2 form) and through the
 * general gather-GEMM on the same pseudo-random operands: out2 = {max |difference|, max |reference|} over U and A */
int dv_debug_gconv_check(dv_ctx* ctx, int32_t NB, int32_t Hs, int32_t Cs, int32_t Ht, int32_t Ct, int32_t stride,
                         int32_t pad_before, int32_t dgrad_form, int32_t nmajor, int32_t epi, float* out2);
/* issue-rate probe of v_mfma_f32_16x16x4_f32 (no memory traffic), nacc = 16 or 36 independent accumulators,
 * trivial or pseudo-random operands: out3 = {TFLOP/s, in-kernel clock MHz, shader cycles per MFMA} */
int dv_debug_mfma_peak(dv_ctx* ctx, int32_t blocks, int32_t iters, int32_t nacc, int32_t randomize, float* out3);
/* process-wide: the fp32 engine routes every conv / conv-transpose / weight-gradient launch through the general
 * gather-GEMM and tiled weight-gradient kernels (no strip or fused stride-2 forms): the same arithmetic in another
 * summation order.  The control of tests/test_gpu_bf16.py (two fp32 summation orders against the bf16 engine). */
int dv_debug_general_kernels(int32_t on);
/* process-wide: on = 0 sends the stride-1 3x3 layers of the fp32 engine through the direct kernels (strip form /
 * gather-GEMM) instead of the Winograd F(2x2, 3x3) kernel (wino.hip); on = 1 restores the default */
int dv_debug_winograd(int32_t on);
/* weight gradient of a stride-1 3x3 layer X [NB,H,H,Cx] x Y [NB,H,H,Cy] through the Winograd-domain kernel and through the
 * direct kernels on the same pseudo-random operands: out2 = {max |difference|, max |reference|} */
int dv_debug_wgrad_check(dv_ctx* ctx, int32_t NB, int32_t H, int32_t Cx, int32_t Cy, float* out2);
/* process-wide, read when a model is created: the fp32 engine's PReLU backward inside the data-gradient epilogue
 * (batch-major tiles) instead of the separate pass.  Measured slower at every batch size; kept for the A/B and for the
 * summation-order test of tests/test_gpu_api.py. */
int dv_debug_fuse_prelu_bwd(int32_t on);
int dv_debug_wgrad(dv_ctx* ctx, int32_t NB, int32_t Hx, int32_t Cx, int32_t Hy, int32_t Cy, int32_t sx,
                   int32_t pad_before, int32_t single_tap, int32_t iters, float* ms_out);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DEBVADER_HIP_DEBUG_H */
