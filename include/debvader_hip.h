/* debvader_hip.h — C-ABI of libdebvader_hip.so, the MI355X (gfx950) engine behind debvader's
 * create_model_vae / train_* / deblend() call surface.
 *
 * The reference (astrodeepnet/debvader) has no FFI for this path: its boundary is the Python call
 * surface that bottoms out in Keras/TFP (`net.fit`, `net(x)`, `net.compile`, `net.load_weights`).
 * Each entry point below names the reference call it replaces (paths relative to the reference repo).
 * The Python shim in debvader_amd/ binds these with ctypes (see INTEGRATION.md for the stub).
 *
 * Conventions: every call returns an int status (0 = DV_OK, <0 = DV_E_*); dv_last_error() holds the
 * message of the last failure on the calling thread.  Nothing throws or exits across this boundary.
 * The caller owns every host buffer (C-contiguous float32 / int32); the library owns all device memory.
 * A dv_model is not thread-safe.  One process drives one GPU; ranks are joined through RCCL.
 */
#ifndef DEBVADER_HIP_H
#define DEBVADER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: these declarations are its surface */
#endif

#define DV_OK 0
#define DV_E_INVALID (-1)
#define DV_E_HIP (-2)
#define DV_E_NOMEM (-3)
#define DV_E_RCCL (-4)
#define DV_E_STATE (-5)
#define DV_E_NODEVICE (-6)

#define DV_MAX_LEVELS 8
#define DV_UNIQUE_ID_BYTES 128

typedef struct dv_ctx dv_ctx;
typedef struct dv_model dv_model;

/* Architecture + numerics of create_model_vae(input_shape, latent_dim, filters, kernels)
 * (src/debvader/model/model.py:164-218; fixed values used by train_deblender: training/train.py:104-107). */
typedef struct dv_config {
  int32_t height, width, bands;      /* input_shape (59,59,6); square stamps, 1 .. 15 bands (train.py:86 nb_of_bands; the bf16
                                        engine 1 .. 7) */
  int32_t latent_dim;                /* 32; any value 1 .. 64 (model.py:164) */
  int32_t n_levels;                  /* len(filters) */
  int32_t filters[DV_MAX_LEVELS];    /* [32,64,128,256]; multiples of 4 (bf16 engine: of 16) */
  int32_t kernels[DV_MAX_LEVELS];    /* [3,3,3,3]; 1 .. 5 per level (model.py:81-91,121-134), both engines */
  int32_t max_batch;                 /* stamps per device step (workspace capacity) */
  float kl_weight;                   /* KLDivergenceRegularizer weight, model.py:213 (0.01) */
  int32_t kl_multiplicity;           /* times Keras adds the activity loss (2) */
  float bn_eps, bn_momentum;         /* Keras BatchNormalization defaults, model.py:79 (1e-3, 0.99) */
  int32_t bn_moving_var_unbiased;    /* fused-BN moving variance uses the Bessel-corrected batch variance (1) */
  float sigma_floor;                 /* model.py:156 (1e-4) */
  float diag_shift;                  /* model.py:49 (1e-5) */
  int32_t dtype;                     /* DV_DTYPE_F32 (0, the reference's precision) or DV_DTYPE_BF16 (1): bf16 storage and
                                        bf16 MFMA operands for the conv / conv-transpose stacks (model.py:79-98,112-137),
                                        fp32 accumulation, fp32 master weights / Adam / dense trunk / sampler / head */
  int32_t infer_graph;               /* 0 (default) / 1: dv_infer / dv_infer_f64 calls of fewer than 64 stamps in one chunk with
                                        engine-drawn noise replay a captured hipGraph of their forward pass (the ~45 launches of
                                        BASELINE configs[4]'s "hipGraph-captured decode", deblender.py:18) instead of launching
                                        them one by one: first call of a size eager, second captured, then replayed; graphs are
                                        dropped when a parameter changes.  Same bits either way.  Off by default: on MI355X a
                                        one-stamp forward is bound by the duration of its kernels, not by submission */
} dv_config;
#define DV_DTYPE_F32 0
#define DV_DTYPE_BF16 1

/* scalars written by the step functions */
enum { DV_S_LOSS = 0, DV_S_NLL_MEAN = 1, DV_S_KL_REG = 2, DV_S_MSE = 3, DV_N_SCALARS = 4 };

int dv_version(void);
/* 0: the product library (this header is its whole exported surface; it reads no measurement or rehearsal switch);
 * 1: the DEVELOPMENT build (same sources under -DDV_DEBUG_EXPORTS: include/debvader_hip_debug.h, the wrong-result
 * measurement switches DV_EXP_*, the one-GPU rehearsal hook DV_DEBUG_FAKE_PEERS).  The Python package honours its own
 * rehearsal variable (DV_DEBUG_SAME_GPU) only when this returns 1.  No counterpart in the reference (tooling). */
int dv_build_kind(void);
/* CRC-32C of `n` bytes continuing from `crc` (0 to start): the checksum of TensorFlow tensor-bundle checkpoints,
 * which load_weights / ModelCheckpoint read and write (model.py:262-266, train.py:49-75).  Host only. */
uint32_t dv_crc32c(uint32_t crc, const void* data, size_t n);
int dv_last_error(char* buf, size_t n);
int dv_config_default(dv_config* cfg);

/* ---- architecture queries: host only, no GPU needed (replace net.summary(), train.py:118) ---- */
int dv_arch_counts(const dv_config* cfg, int32_t* n_tensors, int64_t* n_encoder, int64_t* n_decoder,
                   int64_t* n_trainable);
int dv_arch_describe(const dv_config* cfg, int32_t i, char* name, size_t name_len, int64_t shape[4], int32_t* ndim,
                     int32_t* trainable);
/* flat layout of the gradient buffer the data-parallel step all-reduces: tensor i occupies
 * [off, off+count) floats; the buckets are, in the order they are reduced, [n_enc_train, n_train) (decoder, queued
 * when the decoder backward is done), [split, n_enc_train) (deep half of the encoder, queued mid-backward) and
 * [0, split) (shallow half, at the end).  out = {split, n_enc_train, n_train, n_total} */
int dv_arch_buckets(const dv_config* cfg, int64_t out[4]);
int dv_arch_offset(const dv_config* cfg, int32_t i, int64_t* off, int64_t* count);
/* forward multiply-accumulates per stamp (padding taps counted), for roofline accounting */
int dv_arch_macs(const dv_config* cfg, int64_t* encoder_macs, int64_t* decoder_macs);

/* ---- context: one per process / GPU -------------------------------------------------------- */
int dv_device_count(int32_t* n);
/* PCI bus id of visible device `device` (>= 16 bytes), without creating a context: the ranks of a job compare
 * (host, bus id) BEFORE they build the communicator, so that two ranks mapped onto one GPU fail fast with a clear message
 * instead of inside ncclCommInitRank (debvader_amd.parallel.make_context).  DV_E_NODEVICE when the index is not visible. */
int dv_device_bus_id(int32_t device, char* bus_id, size_t bus_len);
int dv_comm_unique_id(void* out_id /* DV_UNIQUE_ID_BYTES */);
/* world == 1: id may be NULL.  world > 1: every rank passes rank 0's id (exchanged by the host). */
int dv_ctx_create(int32_t device, int32_t rank, int32_t world, const void* unique_id, dv_ctx** out);
/* Destroys the models still alive on the context first (their handles become invalid), then the communicator, events
 * and streams.  Once the process is inside exit() (the HIP runtime's own exit handlers may have run) both destroy calls
 * only release host memory. */
int dv_ctx_destroy(dv_ctx* ctx);
int dv_ctx_sync(dv_ctx* ctx);
/* sum `n` floats over ranks in place (host buffer); used by the host loop for History scalars */
int dv_ctx_allreduce_host(dv_ctx* ctx, float* buf, int32_t n);

/* What the context's communicator really spans (the multi-rank bench prints it so that a scaling record can be checked):
 * comm_ranks = ncclCommCount (0 without a communicator), comm_rank = this rank inside it, device = HIP device index,
 * bus_id = its PCI bus id (>= 16 bytes), rehearsal = 1 when DV_DEBUG_FAKE_PEERS gave this rank a one-rank communicator
 * although world > 1 (a launch rehearsal on one GPU: nothing is summed across ranks).  Any pointer may be NULL. */
int dv_ctx_comm_info(dv_ctx* ctx, int32_t* comm_ranks, int32_t* comm_rank, int32_t* device, char* bus_id, size_t bus_len,
                     int32_t* rehearsal);
/* Timing of the collectives of the data-parallel step: while enabled, every all-reduce on the comm stream
 * and every wait of the main stream for one is bracketed by timed HIP events.  dv_comm_prof_read synchronises, returns
 * the number of collectives and their summed duration (comm_ms) and the number of main-stream waits and what they cost
 * (exposed_ms: communication NOT hidden behind the backward pass) since the last read, and resets the counters.  The event
 * records perturb the step a little: use a separate pass, not the timed region. */
int dv_comm_prof_enable(dv_ctx* ctx, int32_t on);
int dv_comm_prof_read(dv_ctx* ctx, int64_t* n_collectives, double* comm_ms, int64_t* n_waits, double* exposed_ms);

/* ---- model --------------------------------------------------------------------------------- */
/* replaces create_model_vae (model.py:164); weights start at Keras defaults (Glorot-uniform kernels,
 * zero biases/alphas, BN gamma=1): dv_model_init draws them from the engine's own Philox stream */
int dv_model_create(dv_ctx* ctx, const dv_config* cfg, dv_model** out);
int dv_model_destroy(dv_model* m);
int dv_model_init(dv_model* m, uint64_t seed);
/* tensor i in TF-checkpoint order (dv_arch_describe): replaces net.get_weights / net.load_weights (model.py:266) */
int dv_model_get_param(dv_model* m, int32_t i, float* host, size_t nbytes);
int dv_model_set_param(dv_model* m, int32_t i, const float* host, size_t nbytes);
int dv_model_get_grad(dv_model* m, int32_t i, float* host, size_t nbytes);
/* optimizer slots of tensor i (which = 0: m, 1: v); replaces the checkpoint's .OPTIMIZER_SLOT entries */
int dv_model_get_slot(dv_model* m, int32_t i, int32_t which, float* host, size_t nbytes);
int dv_model_set_slot(dv_model* m, int32_t i, int32_t which, const float* host, size_t nbytes);
/* decoder.trainable = False (train.py:175, model.py:252) takes effect at the next optimizer reset */
int dv_model_set_trainable(dv_model* m, int32_t encoder_trainable, int32_t decoder_trainable);
/* net.compile(optimizer=legacy.Adam(lr)) (train.py:125-130,178-183): fresh slots, iteration 0 */
int dv_optimizer_reset(dv_model* m, float lr, float beta1, float beta2, float eps);
int dv_optimizer_get_iter(dv_model* m, int64_t* iter);
int dv_optimizer_set_iter(dv_model* m, int64_t iter);

/* ---- data resident in HBM ------------------------------------------------------------------ */
/* slot 0/1 (train / validation): copies n stamps x[n,H,W,C], y[n,H,W,C] to the device once per fit() */
int dv_data_upload(dv_model* m, int32_t slot, const float* x, const float* y, int64_t n);
int dv_data_free(dv_model* m, int32_t slot);

/* ---- steps: replace one Keras train_function / test_function call inside net.fit (train.py:27-37) ---- */
/* Batch = rows idx[0..B) of `slot` (idx == NULL: rows first..first+B).  eps == NULL: the engine draws
 * eps ~ N(0,I) from Philox(seed); else eps[B,latent] is used (parity tests).  global_batch = sum of B over ranks
 * (0: B).  out[DV_N_SCALARS] are the GLOBAL loss, nll mean, kl regulariser and mse (against the predicted mean, or
 * against a sample of the output distribution - Keras' metric - while dv_model_set_mse_sample is on). */
int dv_train_step(dv_model* m, int32_t slot, const int32_t* idx, int64_t first, int32_t B, int32_t global_batch,
                  const float* eps, uint64_t seed, float* out);
/* forward + losses in inference mode (moving BN statistics), no update: Keras validation step */
int dv_eval_step(dv_model* m, int32_t slot, const int32_t* idx, int64_t first, int32_t B, int32_t global_batch,
                 const float* eps, uint64_t seed, float* out);
/* gradients only (training-mode forward + backward, no Adam, no moving-stat update): parity tests */
int dv_grad_step(dv_model* m, int32_t slot, const int32_t* idx, int64_t first, int32_t B, int32_t global_batch,
                 const float* eps, uint64_t seed, float* out);
/* dv_train_step with a deferred result: the step is queued under `ticket` (0..3) and dv_step_result(ticket) later
 * waits for it and returns its scalars, so the host loop of Model.fit (train.py:27-37) can queue the next batch before
 * it reads the previous loss.  The index array is copied before the call returns.  eps is engine-generated. */
int dv_train_step_async(dv_model* m, int32_t slot, const int32_t* idx, int64_t first, int32_t B, int32_t global_batch,
                        uint64_t seed, int32_t ticket);
int dv_step_result(dv_model* m, int32_t ticket, float* out_scalars /* DV_N_SCALARS */);
/* queue K back-to-back training steps on consecutive batches of `slot` without host round trips
 * (bench.py timed region); scalars of the last step are returned */
int dv_train_steps(dv_model* m, int32_t slot, int64_t first, int32_t B, int32_t global_batch, int32_t steps,
                   uint64_t seed, float* out);

/* ---- inference: replaces net(x) in deblend() (deblend_cutout/deblender.py:18,24) ----------- */
/* normalise=True of deblend() (deblender.py:14-22, normalize/normalize.py:3-7): while set, dv_infer / dv_infer_f64 /
 * dv_infer_mc apply tanh(arcsinh(x)) to the staged stamps on the GPU and the inverse, sinh(arctanh(.)), to the
 * predicted mean (the scale stays in normalised units) */
int dv_model_set_normalise(dv_model* m, int32_t on);

/* Keras' "mse" metric of the reference (train.py:128 metrics=["mse", ...]) compares the labels with a SAMPLE of the output
 * distribution (model.py:158 convert_to_tensor_fn = sample), not with its mean.  While set, DV_S_MSE of the step
 * functions is mean((y - (loc + sigma * eps))^2) with eps drawn from the engine's Philox stream (seed of the step, stream
 * 0x4D534500 + rank, counter (stamp, element / 4)); off (default at this level): the squared error against the mean.
 * The Python surface (debvader_amd.model) switches it on when compile(metrics=[..."mse"...]) asks for the Keras metric. */
int dv_model_set_mse_sample(dv_model* m, int32_t on);

/* Gradient / train steps also write the output distribution (loc, scale) of their forward pass, for
 * dv_model_get_activation("loc" / "scale") - what the parity tests compare with the oracle.  Off by default: the
 * train step of the reference (train.py:27) has no reader for them (42 MB of stores per 256-stamp step). */
int dv_model_set_keep_outputs(dv_model* m, int32_t on);
/* x[N,H,W,C] host.  Outputs (any may be NULL): loc/scale [N,H,W,C] = distribution mean / stddev;
 * mu [N,latent], zstd [N,latent] = z.mean()/z.stddev(); z [N,latent] = the sample fed to the decoder. */
int dv_infer(dv_model* m, const float* x, int64_t N, const float* eps, uint64_t seed, float* loc, float* scale,
             float* mu, float* zstd, float* z);
/* same, for float64 stamps (the reference's numpy default): the float32 cast of deblender.py:18 happens while the
 * library stages the array */
int dv_infer_f64(dv_model* m, const double* x, int64_t N, const float* eps, uint64_t seed, float* loc, float* scale,
             float* mu, float* zstd, float* z);
/* deblend() on cutouts of a field without the host round trip: out = net(float32(field[x:x+H, y:y+H, :])) for every
 * start (x, y), H = the network's stamp size.  Replaces the pair extract_cutouts(field, ...) -> deblend(net, cutouts)
 * of DeblendField.deblend_field (deblend/field_deblender.py:260-274 with extract/extraction.py:4-43 and
 * deblend_cutout/deblender.py:18): the float64 field is uploaded once, each chunk's cutouts are gathered and cast on the
 * GPU straight into the network's input buffer, and only mean / stddev travel back.  Results are bit-identical to
 * dv_infer_f64 on the cutouts dv_scene_extract returns (same cast, same kernels, same noise numbering).  Every window
 * must lie inside the field (DV_E_INVALID otherwise); engine-drawn noise only. */
int dv_infer_cutouts(dv_model* m, const double* field, int32_t F, int32_t nb, const int32_t* starts, int64_t N,
                     uint64_t seed, float* loc, float* scale, float* mu, float* zstd, float* z);

/* The same call for a caller that also needs the cutouts themselves: DeblendField.deblend_field's recarray carries
 * `cutout_images` (float64, field_deblender.py:360) and its quality cut compares them with the predicted means (:323-327).
 * `cutouts` [N][H][H][nb] receives field[x:x+H, y:y+H, :] for every start - exact copies of host data, so the library
 * assembles them on the host (the reference's numpy slice assignment, extraction.py:26-32, over the pipeline's copy threads)
 * while the GPU runs the forward passes; they never cross the host link.  loc / scale as in dv_infer_cutouts.  Replaces the
 * reference's sequence extract_cutouts -> deblend (field_deblender.py:260-274) without the float64 D2H -> host cast -> H2D
 * loop that sequence implies on a GPU. */
int dv_infer_cutouts_keep(dv_model* m, const double* field, int32_t F, int32_t nb, const int32_t* starts, int64_t N,
                          uint64_t seed, float* loc, float* scale, double* cutouts);

/* The same, streaming: instead of filling N-stamp result arrays (167 KB per stamp - a million cutouts do not belong on one
 * host) the library hands every finished chunk to `consumer(user, first, count, mean, stddev)`, stamps
 * [first, first + count) in input order, `mean` / `stddev` pointing into the pinned transfer ring (valid until the
 * consumer returns; nothing is copied on the host).  The consumer runs on the calling thread while the GPU works on the
 * next chunks; a non-zero return value stops the call (DV_E_STATE).  Chunks are max_batch stamps (BASELINE configs[4]:
 * 8192). */
typedef int (*dv_chunk_fn)(void* user, int64_t first, int32_t count, const float* mean, const float* stddev);
int dv_infer_cutouts_stream(dv_model* m, const double* field, int32_t F, int32_t nb, const int32_t* starts, int64_t N,
                            uint64_t seed, dv_chunk_fn consumer, void* user);

/* The same forward passes with the consumer that FOLLOWS in the reference fused in, so that no stamp crosses the host link:
 * DeblendField.deblend_field + get_predicted_field + get_residual_field (deblend/field_deblender.py:219-383, :99-189,
 * :46-97) for integer positions.  For every cutout i (window start starts[i], as extract_cutouts computes it) the
 * network's mean and stddev stamps are added on the GPU, in object order, into
 *   mean_field   += stamp placed with its top-left corner at places[i] = (row, col)   (predicted_mean_field)
 *   stddev_field += the stddev stamp at the same place                                  (predicted_stddev_field)
 *   residual_field (optional) = field - the same mean stamps                            (get_residual_field)
 * places[i] is int((F - cs) / 2) + the galaxy's distance to the field centre, as the reference pads and shifts
 * (field_deblender.py:70,130-160); parts of a stamp that leave the field are dropped (scipy.ndimage.shift, mode
 * "constant").  mse_center (optional, [N]) receives each stamp's centre-10x10 MSE against its cutout (:323-327), the input
 * of the reference's quality cut.  Only the F x F x bands float64 fields (and N doubles) travel back.  Sums are in float64
 * and in object order: bit-identical to dv_scene_composite on the stamps dv_infer_cutouts returns for the same seed. */
int dv_infer_cutouts_composite(dv_model* m, const double* field, int32_t F, int32_t nb, const int32_t* starts,
                               const int32_t* places, int64_t N, uint64_t seed, double* mean_field, double* stddev_field,
                               double* residual_field, double* mse_center);

/* Monte-Carlo epistemic uncertainty: encode each stamp once, decode it `nsamples` times with fresh eps, return the
 * mean and the standard deviation (ddof 0) of the predicted means over the samples.  Replaces the per-object loop
 * `np.std(deblend(net, [stamp]*100)[0], axis=0)` of deblend/field_deblender.py:303-313. */
int dv_infer_mc(dv_model* m, const float* x, int64_t N, int32_t nsamples, uint64_t seed, float* mean_out,
                float* std_out);
/* encoder(x) -> t[N, latent + latent(latent+1)/2]  (model.py:61-100) */
int dv_encode(dv_model* m, const float* x, int64_t N, float* t);
/* decoder(z) -> loc, scale  (model.py:103-161) */
int dv_decode(dv_model* m, const float* z, int64_t N, float* loc, float* scale);

/* ---- scene compositing around the network ------------------------------
 * Host buffers are float64 like the reference's numpy arrays; the library stages them through the GPU.
 * dv_scene_extract: out[i] = field[starts[i][0] : +cs, starts[i][1] : +cs, :] for a field [F][F][nb]
 * (extract/extraction.py:4-43; every window must lie inside the field, else DV_E_INVALID).
 * dv_scene_composite: field += sign * sum_i shift(pad(stamps[i]), pos[i]) in object order, where pad() centres the
 * cs x cs stamp in a zero F x F image and shift() is scipy.ndimage.shift with default arguments (order-3 spline,
 * mode "constant"): deblend/field_deblender.py:46-97 (sign -1, the residual field) and :99-189 (sign +1, the
 * predicted mean / stddev / epistemic fields).  pos[i] = {row shift, column shift}. */
int dv_scene_extract(dv_ctx* ctx, const double* field, int32_t F, int32_t nb, const int32_t* starts, int32_t N,
                     int32_t cs, double* out);
int dv_scene_composite(dv_ctx* ctx, double* field, int32_t F, int32_t nb, const double* stamps, const double* pos,
                       int32_t N, int32_t cs, double sign);

/* ---- introspection for tests and bench ----------------------------------------------------- */
/* copy a named activation of the last step to host: "t","z","kl","eps","loc","scale","head_pre" */
int dv_model_get_activation(dv_model* m, const char* name, float* host, size_t nbytes);
/* HIP-event timing of kernel classes on the engine stream: class 0 gconv, 1 wgrad, 2 everything else */
int dv_prof_enable(dv_model* m, int32_t on);
int dv_prof_read(dv_model* m, int32_t klass, int64_t* launches, double* total_ms);
int dv_prof_reset(dv_model* m);
/* the same timing per MFMA kernel family (fam = 0 .. until DV_E_INVALID): `name` is the kernel name rocprofv3 prints
 * (without template arguments), flops the algorithmic FLOPs of the timed launches (padding taps counted),
 * executed_flops what the matrix pipe executes for them (a Winograd kernel: 16 multiplies per 2 x 2 tile and channel pair
 * instead of 36, over blocks / column tiles padded to its geometry; direct kernels: the algorithmic count),
 * algorithmic_bytes the HBM bytes of the launches with every operand read once and every result written once (0 for
 * families that do not report them).  Any out pointer may be NULL. */
int dv_prof_read_family(dv_model* m, int32_t fam, char* name, size_t name_len, int64_t* launches, double* total_ms,
                        double* flops, double* executed_flops, double* algorithmic_bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* DEBVADER_HIP_H */
