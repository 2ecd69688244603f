// bf16 pipeline of the engine (included by engine.hip inside namespace dv): the conv / conv-transpose stacks of
// create_encoder / create_decoder (model.py:79-98,112-137) on the bf16 kernel family of bf16.h, the dense trunk,
// the sampler and the head arithmetic on the fp32 kernels the f32 engine uses.  Seams: the encoder's last
// activation leaves the stamp-inner bf16 layout as fp32 rows (input of the flatten PReLU, model.py:94-95), the
// decoder's Reshape (model.py:119) enters it; the backward pass crosses the same two seams in reverse.
// Forward on the main stream; in the backward pass the conv and dense weight gradients and their reductions run on the
// weight-gradient / reduction streams beside the data-gradient chain.

static int bf_pad32(int k) { return (k + 31) & ~31; }

// ---- allocation + weight-matrix descriptors (dv_model_create) -----------------------------------------------------
static int bf_alloc(dv_model* m) {
  const Arch& A = m->A;
  BfState& bf = m->bf;
  const size_t Bp = ((size_t)m->Bc + 15) & ~(size_t)15;
  for (int i = 0; i < A.L; ++i) {
    const int f = A.cfg.filters[i];
    if (f % 16) {
      set_error("bf16 engine: filters must be multiples of 16 (filters[%d]=%d)", i, f);
      return E_INVALID;
    }
  }
  if ((A.dec_out * A.dec_out) & 15) {
    // (only a one-level net on stamps whose half size is odd: the head kernel walks the decoder output 16 pixels at a time)
    set_error("bf16 engine: the decoder output (%d x %d pixels) must be a multiple of 16 pixels", A.dec_out, A.dec_out);
    return E_INVALID;
  }
  if (A.cfg.filters[A.L - 1] & 7 || (A.C2p != 16 && A.C2p != 32) || A.C > 15) {
    set_error("bf16 engine: unsupported head / trunk geometry");
    return E_INVALID;
  }
  const int HC = A.C2p;            // columns of the head tensors: 16 for 1 .. 7 bands, 32 for 8 .. 15 (train.py:86,104-107)
  auto balloc = [&](void** p, size_t bytes) -> int {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, std::max<size_t>(bytes, 64));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __FILE__, __LINE__);
    m->allocs.push_back(q);
    *p = q;
    return OK;
  };
  const size_t HW = (size_t)A.H * A.H;
  DV_TRY(balloc(&bf.xh, HW * Bp * 16 * 2));
  DV_TRY(balloc(&bf.xh_alt, HW * Bp * 16 * 2));
  // zero page: as long as ONE pixel's [Bp][C] block of the widest conv layer, so that the row-strip kernel can read an
  // out-of-image block at its usual lane offsets from a uniform base (bconv_row_kernel); >= 1 KiB for the DMA kernels
  size_t zero_bytes = 1024;
  for (int i = 0; i < A.L; ++i) zero_bytes = std::max(zero_bytes, (size_t)Bp * A.cfg.filters[i] * 2);
  DV_TRY(balloc(&bf.zero, zero_bytes));
  DV_HIP(hipMemsetAsync(bf.zero, 0, zero_bytes, m->ctx->stream));
  size_t max_e = (size_t)A.dec_out * A.dec_out * HC;
  bf.enc_u.resize(2 * A.L); bf.enc_a.resize(2 * A.L); bf.dec_u.resize(2 * A.L); bf.dec_a.resize(2 * A.L);
  bf.enc_w.resize(2 * A.L); bf.dec_w.resize(2 * A.L);
  for (int j = 0; j < 2 * A.L; ++j) {
    int hin, cin, hout, cout, s;
    A.enc_layer(j, &hin, &cin, &hout, &cout, &s);
    const size_t e = (size_t)hout * hout * cout;
    max_e = std::max(max_e, e);
    DV_TRY(balloc(&bf.enc_u[j], e * Bp * 2));
    DV_TRY(balloc(&bf.enc_a[j], e * Bp * 2));
    A.dec_layer(j, &hin, &cin, &hout, &cout, &s);
    const size_t e2 = (size_t)hout * hout * cout;
    max_e = std::max(max_e, e2);
    max_e = std::max(max_e, (size_t)hin * hin * cin);
    DV_TRY(balloc(&bf.dec_u[j], e2 * Bp * 2));
    DV_TRY(balloc(&bf.dec_a[j], e2 * Bp * 2));
  }
  const size_t r = (size_t)A.w0 * A.w0 * A.cfg.filters[A.L - 1];
  DV_TRY(balloc(&bf.dec_in, r * Bp * 2));
  DV_TRY(balloc((void**)&bf.tpre32, (size_t)A.dec_out * A.dec_out * Bp * HC * 4));
  DV_TRY(balloc(&bf.dt, (size_t)A.dec_out * A.dec_out * Bp * HC * 2));
  DV_TRY(balloc((void**)&bf.flat_in, (size_t)m->Bc * A.flat * 4));
  {
    const size_t w[5] = {r, (size_t)A.dec_hidden, (size_t)A.dp, (size_t)A.twp, (size_t)A.flat};
    for (int k = 0; k < 5; ++k) DV_TRY(balloc((void**)&bf.trunk[k], (size_t)m->Bc * w[k] * 4));
  }
  bf.gpool.resize(4 * A.L + 4);
  for (auto& g : bf.gpool) DV_TRY(balloc(&g, max_e * Bp * 2));
  // slab pool of one backward pass: ~9.4 MB per launch with the launcher's 256-workgroup rule (bwgrad.hip), 17 launches
  // for the 59-pixel net; a pass that needs more flushes its reductions early (bf_flush_wred) and starts over
  bf.slab_elems = (size_t)64 << 20;
  for (auto& sp : A.specs)          // a launch needs one slab per 64-stamp chunk at least: room for two such launches
    if (sp.ndim == 4) bf.slab_elems = std::max(bf.slab_elems, sp.count * 2 * std::max<size_t>(4, (Bp + 63) / 64));
  bf.slab_tail = (size_t)8 << 20;       // BF_MAIN_SLOTS regions (the launches a pass may queue on the main stream)
  DV_TRY(balloc((void**)&bf.slab, (bf.slab_elems + bf.slab_tail) * 4));

  // bf16 weight matrices and the descriptors the cast kernel walks
  float* P = m->P;
  auto add = [&](const float* src, int Aax, int Bax, int n_is_b, int N, int Cin, int ksz, void** dst, int* Kout) -> int {
    BCastDesc d;
    memset(&d, 0, sizeof d);
    d.src = src; d.A = Aax; d.B = Bax; d.n_is_b = n_is_b; d.N = N; d.Cin = Cin; d.taps = ksz * ksz;
    d.Kpad = bf_pad32(ksz * ksz * Cin);
    DV_TRY(balloc(dst, (size_t)N * d.Kpad * 2));
    d.dst = *dst;
    *Kout = d.Kpad;
    bf.descs.push_back(d);
    return OK;
  };
  for (int j = 0; j < 2 * A.L; ++j) {
    int hin, cin, hout, cout, s;
    A.enc_layer(j, &hin, &cin, &hout, &cout, &s);
    const float* k = P + A.specs[A.enc_k(j)].off;           // HWIO [9][cin][cout]
    if (j == 0) {
      DV_TRY(add(k, cin, cout, 1, cout, 16, A.enc_ksz(j), &bf.enc_w[j].f, &bf.enc_w[j].Kf));
      BCastDesc& d = bf.descs.back();
      d.gamma = P + A.specs[0].off;
      d.beta = P + A.specs[1].off;
      d.nbands = A.C;
    } else {
      DV_TRY(add(k, cin, cout, 1, cout, cin, A.enc_ksz(j), &bf.enc_w[j].f, &bf.enc_w[j].Kf));
      DV_TRY(add(k, cin, cout, 0, cin, cout, A.enc_ksz(j), &bf.enc_w[j].d, &bf.enc_w[j].Kd));
    }
    A.dec_layer(j, &hin, &cin, &hout, &cout, &s);
    const float* kt = P + A.specs[A.dec_k(j)].off;           // (kh,kw,cout,cin)
    DV_TRY(add(kt, cout, cin, 0, cout, cin, A.dec_ksz(j), &bf.dec_w[j].f, &bf.dec_w[j].Kf));
    DV_TRY(add(kt, cout, cin, 1, cin, cout, A.dec_ksz(j), &bf.dec_w[j].d, &bf.dec_w[j].Kd));
  }
  {
    const int f0 = A.cfg.filters[0];
    const float* kh = P + A.specs[A.head_k()].off;           // HWIO [9][f0][2C]
    DV_TRY(add(kh, f0, 2 * A.C, 1, HC, f0, 3, &bf.head_w.f, &bf.head_w.Kf));
    DV_TRY(add(kh, f0, 2 * A.C, 0, f0, HC, 3, &bf.head_w.d, &bf.head_w.Kd));
  }
  {
    // dense trunk on the bf16 matrix cores (btrunk.hip): two forms of each large Dense kernel, the K-split slabs, the
    // trunk output's pre-activation in the stamp-inner layout
    static const bool trunk_off = getenv("DV_BF_TRUNK") != nullptr && atoi(getenv("DV_BF_TRUNK")) == 0;
    const int fl = A.cfg.filters[A.L - 1];
    bf.FL = A.flat;
    bf.TWn = bf_pad32(A.tw);
    bf.TWk = (A.tw + 63) & ~63;
    bf.HIDn = bf_pad32(A.dec_hidden);
    bf.HIDk = (A.dec_hidden + 63) & ~63;
    bf.trunk_mfma = !trunk_off && fl % 64 == 0 && A.flat == A.w0 * A.w0 * fl && A.enc_sizes[A.L] == A.w0 &&
                    A.dec_hidden % 4 == 0 && A.twp % 4 == 0;
    if (bf.trunk_mfma) {
      const float* wenc = P + A.specs[A.enc_dk()].off;         // [flat][tw]
      const float* w1 = P + A.specs[A.D0 + 4].off;              // [560][flat]
      int kk = 0;
      DV_TRY(add(wenc, A.flat, A.tw, 1, bf.TWn, A.flat, 1, &bf.wenc_f, &kk));     // [n = tw][k = flat]
      DV_TRY(add(wenc, A.flat, A.tw, 0, A.flat, bf.TWk, 1, &bf.wenc_d, &kk));     // [n = flat][k = tw]
      DV_TRY(add(w1, A.dec_hidden, A.flat, 1, A.flat, bf.HIDk, 1, &bf.w1_f, &kk));    // [n = flat][k = 560]
      DV_TRY(add(w1, A.dec_hidden, A.flat, 0, bf.HIDn, A.flat, 1, &bf.w1_d, &kk));    // [n = 560][k = flat]
      DV_TRY(balloc((void**)&bf.tslab, (size_t)8 * Bp * std::max(bf.TWn, bf.HIDn) * 4));
      DV_TRY(balloc(&bf.dec_ur, r * Bp * 2));
    }
  }
  {
    bool any = false;
    for (int i = 0; i < A.L; ++i) any = any || A.cfg.kernels[i] != 3 || !(A.cfg.filters[i] == 16 || A.cfg.filters[i] % 32 == 0);
    if (any) {                     // operands of the fp32 weight-gradient kernels (bf_wgrad_f32)
      max_e = std::max(max_e, (size_t)A.H * A.H * 16);
      DV_TRY(balloc((void**)&bf.wx32, max_e * (size_t)m->Bc * 4));
      DV_TRY(balloc((void**)&bf.wy32, max_e * (size_t)m->Bc * 4));
    }
  }
  {   // descriptors sorted by gradient bucket (see BfState::dirty_mask)
    const size_t split = enc_bucket_split(A);
    auto bucket = [&](const BCastDesc& d) { const size_t off = (size_t)(d.src - P); return off < split ? 0 : off < A.n_enc_train ? 1 : 2; };
    std::stable_sort(bf.descs.begin(), bf.descs.end(), [&](const BCastDesc& a, const BCastDesc& b) { return bucket(a) < bucket(b); });
    int n = 0;
    for (int b = 0; b < 3; ++b) {
      bf.desc_off[b] = n;
      while (n < (int)bf.descs.size() && bucket(bf.descs[n]) == b) ++n;
    }
    bf.desc_off[3] = n;
  }
  DV_TRY(balloc((void**)&bf.descs_dev, bf.descs.size() * sizeof(BCastDesc)));
  DV_HIP(hipMemcpyAsync(bf.descs_dev, bf.descs.data(), bf.descs.size() * sizeof(BCastDesc), hipMemcpyHostToDevice,
                        m->ctx->stream));
  DV_HIP(hipStreamSynchronize(m->ctx->stream));
  bf.on = true;
  bf.dirty_mask = 7;
  return OK;
}

static int bf_cast_buckets(dv_model* m, unsigned mask, hipStream_t s) {
  BfState& bf = m->bf;
  ProfScope ps(m, 2, s);
  for (int b = 0; b < 3; ++b) {
    if (!(mask & (1u << b))) continue;
    int b1 = b;
    while (b1 + 1 < 3 && (mask & (1u << (b1 + 1)))) ++b1;          // neighbouring buckets in one launch
    const int lo = bf.desc_off[b], hi = bf.desc_off[b1 + 1];
    if (hi > lo) DV_TRY(launch_bf_cast_weights(bf.descs_dev + lo, bf.descs.data() + lo, hi - lo, s));
    b = b1;
  }
  return OK;
}

static int bf_refresh_weights(dv_model* m, hipStream_t s) {
  BfState& bf = m->bf;
  if (!bf.dirty_mask) return OK;
  DV_TRY(bf_cast_buckets(m, bf.dirty_mask, s));
  bf.dirty_mask = 0;
  return OK;
}

// the bf16 matrices of a bucket that early Adam has just updated on `st` (the comm stream), re-cast behind the update:
// no kernel of this step reads them any more (the same argument that lets Adam run there), and the next step's forward
// is ordered behind the comm stream
static int bf_cast_bucket_early(dv_model* m, int b, hipStream_t st) {
  DV_TRY(bf_cast_buckets(m, 1u << b, st));
  m->bf.early_cast |= 1u << b;
  return OK;
}

static int bf_conv(dv_model* m, const void* X, const void* W, int Kpad, int form, int Hin, int Cin, int Hout, int Cout,
                   int s, int pb, int epi, void* U, void* Aout, float* Uf, const float* bias, const float* alpha,
                   const void* Uin, float* dalp, float* dbp, int ksz = 3) {
  BConvParams p;
  memset(&p, 0, sizeof p);
  p.X = X; p.W = W; p.zero = m->bf.zero; p.U = U; p.A = Aout; p.Uf = Uf; p.bias = bias; p.alpha = alpha; p.Uin = Uin;
  p.dal_part = dalp; p.db_part = dbp;
  p.Hin = Hin; p.Hout = Hout; p.Cin = Cin; p.Cout = Cout; p.NBp = m->bf.NBp;
  p.form = form; p.s = s; p.pb = pb; p.Kpad = Kpad; p.epi = epi; p.ksz = ksz;
  if (epi == BEPI_HEAD) p.hd = m->bf.hfuse;
  // algorithmic FLOPs: form 0 meets nine taps per output pixel, form 1 nine per source pixel; the folded
  // first conv and the padded head count their real channels
  const BfState& bf = m->bf;
  const double cin_alg = (W == bf.enc_w[0].f) ? m->A.C : (W == bf.head_w.d ? 2 * m->A.C : Cin);
  const double cout_alg = (W == bf.head_w.f) ? 2 * m->A.C : Cout;
  const double px = form == 0 ? (double)Hout * Hout : (double)Hin * Hin;
  ProfScope ps(m, 0, nullptr, PF_BCONV, 2.0 * bf.NBp * px * (double)(ksz * ksz) * cin_alg * cout_alg);
  return launch_bconv(p, fwd_stream(m));
}

// ---- forward ------------------------------------------------------------------------------------------------------
// K split of a trunk product: the products are latency-bound (a wave keeps three 64-wide K steps in flight), so K is cut
// until a wave has about three steps - sixteen runs for K = 4096: four per workgroup (summed through LDS), four slabs
static void bf_trunk_ksplit(int tiles, int K, int* wg_ksplit, int* nslab) {
  (void)tiles;
  const int ksteps = K >> 6;
  int total = 1;
  while (total < 16 && ksteps > 3 * total) total *= 2;
  *wg_ksplit = std::min(total, 4);
  *nslab = total / *wg_ksplit;
}

// t = bias + sum of the slabs the encoder Dense left behind, for consumers of m->t that do not run the sampler
static int bf_finish_t(dv_model* m, int NB, hipStream_t s) {
  BfState& bf = m->bf;
  if (bf.t_nslab <= 0) return OK;
  const Arch& A = m->A;
  ProfScope ps(m, 2, s);
  DV_TRY(launch_bt_finish_rows(bf.tslab, bf.t_nslab, (long)bf.NBp * bf.TWn, bf.TWn, m->P + A.specs[A.enc_db()].off, m->t, NB,
                               A.tw, A.twp, s));
  bf.t_nslab = 0;
  return OK;
}

static int bf_encoder_forward(dv_model* m, const float* xsrc, const int* idx, int first, int NB, bool keep_u, bool defer_t) {
  const Arch& A = m->A;
  BfState& bf = m->bf;
  hipStream_t s = fwd_stream(m);
  float* P = m->P;
  bf.NBp = (NB + 15) & ~15;
  DV_TRY(bf_refresh_weights(m, s));
  if (bf.in_pre) {
    std::swap(bf.xh, bf.xh_alt);           // normalised ahead on the comm stream (bn_prefetch); bn_prepare has waited for it
    bf.in_pre = false;
  } else {
    ProfScope ps(m, 2);
    DV_TRY(launch_bf_input(xsrc, idx, first, NB, bf.NBp, A.H * A.H, A.C, m->bnstate, bf.xh, s));
  }
  if (m->bnpre_go_pending) {               // bnstate, the sums and the other input buffer may be overwritten from here on
    DV_HIP(hipEventRecord(m->ev_bnpre_go, s));
    m->bnpre_go_pending = false;
  }
  const void* in = bf.xh;
  for (int j = 0; j < 2 * A.L; ++j) {
    int hin, cin, hout, cout, st;
    A.enc_layer(j, &hin, &cin, &hout, &cout, &st);
    const int pb = same_pad_before(hin, A.enc_ksz(j), st, nullptr);
    DV_TRY(bf_conv(m, in, bf.enc_w[j].f, bf.enc_w[j].Kf, 0, hin, j == 0 ? 16 : cin, hout, cout, st, pb,
                   exp_epi(keep_u, hout) == 1 ? BEPI_RAWBF : BEPI_FWD,   // (DV_EXP_NO_A: measurement only, see engine.hip)
                   keep_u ? bf.enc_u[j] : nullptr, bf.enc_a[j], nullptr, P + A.specs[A.enc_b(j)].off,
                   P + A.specs[A.enc_al(j)].off, nullptr, nullptr, nullptr, A.enc_ksz(j)));
    in = bf.enc_a[j];
  }
  const int sl = A.enc_sizes[A.L], fl = A.cfg.filters[A.L - 1];
  if (bf.trunk_mfma) {
    // Flatten -> PReLU -> Dense (model.py:94-98) as ONE product over the stamp-inner activation of the last conv: the
    // flatten PReLU is applied where the A fragments are loaded, K = flat is split, and the slabs + bias are added by
    // the sampler launch that follows (defer_t) or by bf_finish_t
    BGemmParams g;
    memset(&g, 0, sizeof g);
    g.A = in; g.B = bf.wenc_f; g.amode = BGA_STAMP_PRELU; g.epi = BGE_SLAB;
    g.M = bf.NBp; g.Mreal = NB; g.N = bf.TWn; g.K = bf.FL; g.Kreal = bf.FL; g.ldb = bf.FL;
    g.NBp = bf.NBp; g.C = fl; g.a_alpha = P + A.specs[A.enc_flat_al()].off;
    g.slab = bf.tslab; g.slab_stride = (long)bf.NBp * bf.TWn; g.ldc = bf.TWn;
    bf_trunk_ksplit(((bf.NBp + 31) / 32) * (bf.TWn / 32), bf.FL, &g.wg_ksplit, &g.nslab);
    {
      ProfScope ps(m, 0, nullptr, PF_BTRUNK, 2.0 * bf.NBp * (double)A.flat * A.tw);
      DV_TRY(launch_bgemm(g, s));
    }
    bf.t_nslab = g.nslab;
    return defer_t ? OK : bf_finish_t(m, NB, s);
  }
  {
    ProfScope ps(m, 2);
    if (!exp_skip_small()) DV_TRY(launch_bf_to_rows(in, bf.flat_in, NB, bf.NBp, sl * sl, fl, s));
    if (!exp_skip_small()) DV_TRY(launch_prelu_fwd(bf.flat_in, P + A.specs[A.enc_flat_al()].off, m->flat_a, NB, A.flat, s));
  }
  return gconv_fprop(m, m->flat_a, enc_dense_w(m), false, enc_dense_b(m), nullptr, m->t,
                     nullptr, 1, NB, 1, A.flat, 1, A.twp, 1, 0, true);
}

static int bf_decoder_forward(dv_model* m, int NB, bool keep_u) {
  const Arch& A = m->A;
  BfState& bf = m->bf;
  hipStream_t s = fwd_stream(m);
  float* P = m->P;
  bf.NBp = (NB + 15) & ~15;
  DV_TRY(bf_refresh_weights(m, s));
  {
    ProfScope ps(m, 2);
    if (!exp_skip_small() && !m->ain_done) DV_TRY(launch_prelu_fwd(m->z, P + A.specs[A.D0].off, m->dec_ain, NB, A.dp, s));
    m->ain_done = false;
  }
  DV_TRY(gconv_fprop(m, m->dec_ain, dec_dense0_w(m), false, P + A.specs[A.D0 + 2].off,
                     P + A.specs[A.D0 + 3].off, keep_u ? m->dec_uh : nullptr, m->dec_ah, 2, NB, 1, A.dp, 1, A.dec_hidden,
                     1, 0, true));
  const int fl = A.cfg.filters[A.L - 1];
  const int r = A.w0 * A.w0 * fl;
  if (bf.trunk_mfma) {
    // Dense(560 -> w*w*f) -> PReLU -> Reshape (model.py:116-119): the fp32 rows of the hidden layer are rounded to bf16
    // where they are loaded, bias and PReLU run in the epilogue, which writes the stamp-inner tensors the first
    // Conv2DTranspose reads (and, in training, the pre-activation its PReLU backward needs)
    BGemmParams g;
    memset(&g, 0, sizeof g);
    g.A = m->dec_ah; g.B = bf.w1_f; g.amode = BGA_ROWS_F32; g.epi = BGE_STAMP_BIAS_PRELU;
    g.M = bf.NBp; g.Mreal = NB; g.N = bf.FL; g.K = bf.HIDk; g.Kreal = A.dec_hidden; g.lda = A.dec_hidden; g.ldb = bf.HIDk;
    g.NBp = bf.NBp; g.nslab = 1; g.wg_ksplit = 4; g.Co = fl;      // (K = 576: three steps per wave, summed through LDS)
    g.bias = P + A.specs[A.D0 + 5].off; g.alpha = P + A.specs[A.D0 + 6].off;
    g.U = keep_u ? bf.dec_ur : nullptr; g.Aout = bf.dec_in;
    ProfScope ps(m, 0, nullptr, PF_BTRUNK, 2.0 * bf.NBp * (double)A.dec_hidden * r);
    DV_TRY(launch_bgemm(g, s));
  } else {
    DV_TRY(gconv_fprop(m, m->dec_ah, P + A.specs[A.D0 + 4].off, false, P + A.specs[A.D0 + 5].off,
                       P + A.specs[A.D0 + 6].off, keep_u ? m->dec_ur : nullptr, m->dec_ar, 2, NB, 1, A.dec_hidden, 1, r, 1,
                       0, true));
    ProfScope ps(m, 2);
    if (!exp_skip_small()) DV_TRY(launch_bf_from_rows(m->dec_ar, bf.dec_in, NB, bf.NBp, A.w0 * A.w0, fl, s));
  }
  const void* in = bf.dec_in;
  for (int j = 0; j < 2 * A.L; ++j) {
    int hin, cin, hout, cout, st;
    A.dec_layer(j, &hin, &cin, &hout, &cout, &st);
    const int pb = same_pad_before(hout, A.dec_ksz(j), st, nullptr);
    DV_TRY(bf_conv(m, in, bf.dec_w[j].f, bf.dec_w[j].Kf, 1, hin, cin, hout, cout, st, pb,
                   exp_epi(keep_u, hout) == 1 ? BEPI_RAWBF : BEPI_FWD,
                   keep_u ? bf.dec_u[j] : nullptr, bf.dec_a[j], nullptr, P + A.specs[A.dec_b(j)].off,
                   P + A.specs[A.dec_al(j)].off, nullptr, nullptr, nullptr, A.dec_ksz(j)));
    in = bf.dec_a[j];
  }
  bf.hfuse_tiles = 0;
  if (bf.hfuse_req) {
    bf.hfuse_req = false;
    BConvParams q;
    memset(&q, 0, sizeof q);
    q.Hin = q.Hout = A.dec_out; q.Cin = A.cfg.filters[0]; q.Cout = A.C2p; q.NBp = bf.NBp; q.s = 1; q.pb = 1; q.ksz = 3;
    const long tiles = bconv_head_tiles(q);
    if (tiles > 0 && (size_t)tiles * 2 + 64 <= m->ws3_elems) {
      bf.hfuse_tiles = tiles;
      bf.tpre_stale = true;
      return bf_conv(m, in, bf.head_w.f, bf.head_w.Kf, 0, A.dec_out, A.cfg.filters[0], A.dec_out, A.C2p, 1, 1, BEPI_HEAD,
                     nullptr, nullptr, nullptr, m->bhp, nullptr, nullptr, nullptr, nullptr);
    }
  }
  bf.tpre_stale = false;
  return bf_conv(m, in, bf.head_w.f, bf.head_w.Kf, 0, A.dec_out, A.cfg.filters[0], A.dec_out, A.C2p, 1, 1, BEPI_RAW32,
                 nullptr, nullptr, bf.tpre32, m->bhp, nullptr, nullptr, nullptr, nullptr);
}

// The head in the head conv's epilogue: a training / gradient step that keeps no outputs (loc / scale are only stored on
// request) on a batch padded to 64 stamps with a 16-column head (1 .. 7 bands), whole batch in one lane.  Everything else -
// inference, kept outputs, 8 .. 15 bands, odd paddings - takes bf_head_kernel as before.  DV_BF_HEAD_FUSED=0: never.
static void bf_head_fuse_request(dv_model* m, const float* ysrc, const int* idx, int first, int NB, int Bg, bool want_grad,
                                 bool want_out, int part_block0) {
  static const bool off = getenv("DV_BF_HEAD_FUSED") != nullptr && atoi(getenv("DV_BF_HEAD_FUSED")) == 0;
  const Arch& A = m->A;
  BfState& bf = m->bf;
  bf.hfuse_req = false;
  if (off || !ysrc || !want_grad || want_out || part_block0 != 0 || A.C2p != 16 || m->prof_on || exp_skip_tail()) return;
  BHeadFuse& h = bf.hfuse;
  memset(&h, 0, sizeof h);
  h.y = ysrc; h.idx = idx; h.first = first; h.dt = bf.dt;
  h.part = m->defer_loss_sums ? m->ws_head : m->ws3;
  h.NB = NB; h.H = A.H; h.nb = A.C; h.crop0 = A.crop0;
  h.sigma_floor = A.cfg.sigma_floor;
  h.gscale = (float)(1.0 / ((double)Bg * A.H * A.H * A.C));
  h.mse_sample = m->mse_sample ? 1 : 0;
  h.mse_stream = DV_MSE_STREAM + (unsigned)m->ctx->rank;
  h.mse_seed = m->cur_seed;
  bf.hfuse_req = true;
}

static int bf_head_lane(dv_model* m, const float* ysrc, const int* idx, int first, int NB, int Bg, bool want_grad,
                        bool want_out, int part_block0, int* nblk) {
  const Arch& A = m->A;
  BfState& bf = m->bf;
  if (bf.hfuse_tiles > 0) {                  // the head conv's epilogue has done it (bf_decoder_forward)
    if (nblk) *nblk = (int)bf.hfuse_tiles;
    bf.hfuse_tiles = 0;
    return OK;
  }
  BHeadParams hp;
  memset(&hp, 0, sizeof hp);
  hp.tpre = bf.tpre32;
  hp.y = ysrc;
  hp.idx = idx;
  hp.first = first;
  hp.dt = want_grad ? bf.dt : nullptr;
  hp.loc = want_out ? m->loc : nullptr;
  hp.scale = want_out ? m->scale : nullptr;
  hp.part = (m->defer_loss_sums ? m->ws_head : m->ws3) + (size_t)part_block0 * 2;
  hp.NB = NB;
  hp.NBp = bf.NBp;
  hp.Hd = A.dec_out;
  hp.H = A.H;
  hp.nb = A.C;
  hp.cw = A.C2p;
  hp.crop0 = A.crop0;
  hp.sigma_floor = A.cfg.sigma_floor;
  hp.gscale = (float)(1.0 / ((double)Bg * A.H * A.H * A.C));
  hp.mse_sample = m->mse_sample ? 1 : 0;
  hp.mse_stream = DV_MSE_STREAM + (unsigned)m->ctx->rank;
  hp.mse_seed = m->cur_seed;
  const long blocks = ((long)A.dec_out * A.dec_out * bf.NBp + 255) / 256;
  if ((size_t)(part_block0 + blocks) * 2 > m->ws3_elems) {
    set_error("head workspace too small");
    return E_STATE;
  }
  ProfScope ps(m, 2);
  return launch_bf_head(hp, fwd_stream(m), nblk);
}

// ---- backward -----------------------------------------------------------------------------------------------------
// Weight gradients are queued on the aux stream (beside the data-gradient chain of the main stream) once the main
// stream has produced their operands; under the profiler's class timing everything stays on the main stream.
static hipStream_t bf_wstream(dv_model* m) {
  return (m->overlap_wgrad && !m->prof_on && m->ctx->aux_stream) ? m->ctx->aux_stream : m->ctx->stream;
}
// queues the reductions registered so far (one launch) on the weight-gradient stream and empties the slab pool
// rs: the stream the sums run on.  The weight-gradient stream itself (default): the pool is rewound behind them.  The
// REDUCTION stream (bucket boundaries, round 6): the caller has ordered it behind the launches registered so far; the pool
// is NOT rewound - the later launches of the pass keep filling it - so that the sums do not sit in the weight-gradient
// stream's queue, which is the stream the END of a pass waits for (two boundaries x ~45 us of sums per pass)
static int bf_flush_wred(dv_model* m, hipStream_t rs = nullptr) {
  BfState& bf = m->bf;
  hipStream_t wst = bf_wstream(m);
  if (!rs) rs = wst;
  if (rs == wst && bf.red_pending) {
    // the pool is about to be rewound: sums that an earlier boundary queued on the reduction stream must have read it
    DV_HIP(hipEventRecord(m->ctx->ev_red, m->ctx->red_stream));
    DV_HIP(hipStreamWaitEvent(wst, m->ctx->ev_red, 0));
    bf.red_pending = false;
  }
  if (bf.wred.count > 0) {
    ProfScope ps(m, 2, rs);
    DV_TRY(launch_reduce_partials_batch(bf.wred, rs));
  }
  bf.wred.count = 0;
  if (rs == wst) bf.slab_off = 0;
  else bf.red_pending = true;
  return OK;
}

// the stream a bucket boundary's sums go to: the reduction stream when the pass has one, ordered here behind everything
// the weight-gradient stream and the main stream have queued (slabs; the fused epilogues' partials); else the
// weight-gradient stream, which the caller has already ordered behind the main stream
static bool bf_boundary_on_red(bool trunk_red) {
  static const bool off = getenv("DV_BF_BOUNDARY_ON_WGRAD_STREAM") != nullptr;      // (A/B: the form until round 5)
  return trunk_red && !off;
}
static hipStream_t bf_boundary_stream(dv_model* m, hipStream_t ws, hipStream_t s, bool trunk_red, int* status) {
  *status = OK;
  if (!bf_boundary_on_red(trunk_red)) return ws;
  dv_ctx* cx = m->ctx;
  if (hipEventRecord(cx->ev_wred, ws) != hipSuccess || hipStreamWaitEvent(cx->red_stream, cx->ev_wred, 0) != hipSuccess ||
      hipStreamWaitEvent(cx->red_stream, cx->ev_ready, 0) != hipSuccess) {
    *status = E_HIP;
    return ws;
  }
  return cx->red_stream;
}

constexpr int BF_MAIN_SLOTS = 2;   // slab regions behind the pool for the launches queued on the main stream (layers 0, 1)
// weight gradient of one layer: partial slabs into this launch's region of the pool; the fixed-order sum over the slabs
// is registered and runs with all the others of the pass in one launch (17 five-microsecond reductions otherwise sit
// between the weight-gradient kernels of the aux stream, which is the tail of the step)
// on_main: queue the kernel on the main stream (the last launch of a pass: the main stream has nothing left to do while
// the weight-gradient stream works off its backlog); the caller orders the slab reduction behind it
static int bf_wgrad(dv_model* m, const void* X, int Hx, int Cx, const void* Y, int Hy, int Cy, int s, int pb, float* out,
                    int cpad, int creal, bool on_main = false, int main_slot = 0) {
  BfState& bf = m->bf;
  const size_t slab = (size_t)9 * Cx * Cy;
  if (bf.wred.count >= DV_WRED_MAX || bf.slab_off + 4 * ((size_t)(bf.NBp + 63) / 64) * slab > bf.slab_elems) {
    // (rare: a pass without bucket boundaries on a deep net, or a full pool.)  The batch may hold a launch that was queued
    // on the main stream: the reduction, on the weight-gradient stream, has to wait for it
    hipStream_t wst = bf_wstream(m);
    if (wst != m->ctx->stream) {
      DV_HIP(hipEventRecord(m->ctx->ev_ready, m->ctx->stream));
      DV_HIP(hipStreamWaitEvent(wst, m->ctx->ev_ready, 0));
    }
    DV_TRY(bf_flush_wred(m));
  }
  BWgradParams p;
  memset(&p, 0, sizeof p);
  p.X = X; p.Y = Y; p.zero = bf.zero; p.part = bf.slab + bf.slab_off; p.part_capacity = bf.slab_elems - bf.slab_off;
  if (on_main) {                 // its own region: the pool may still be read by a reduction the aux stream has not run yet
    p.part = bf.slab + bf.slab_elems + (size_t)main_slot * (bf.slab_tail / BF_MAIN_SLOTS);
    p.part_capacity = bf.slab_tail / BF_MAIN_SLOTS;
  }
  p.Hx = Hx; p.Cx = Cx; p.Hy = Hy; p.Cy = Cy; p.NBp = bf.NBp; p.s = s; p.pb = pb;
  int ns = 0;
  hipStream_t st = on_main ? m->ctx->stream : bf_wstream(m);
  static const bool exp_no_events = DV_EXP_SWITCH("DV_EXP_NO_WGRAD_EVENTS") != 0;   // MEASUREMENT only (races: use with DV_EXP_SKIP_WGRAD)
  if (st != m->ctx->stream && !exp_no_events) {
    if (!bf.head_marked) DV_HIP(hipEventRecord(m->ctx->ev_ready, m->ctx->stream));   // (else recorded at the head of the pass)
    DV_HIP(hipStreamWaitEvent(st, m->ctx->ev_ready, 0));
  }
  bf.head_marked = false;
  {
    const double cx_alg = X == bf.xh ? m->A.C : Cx, cy_alg = out == m->Ghs ? 2 * m->A.C : Cy;
    ProfScope ps(m, 1, st, PF_BWGRAD, 2.0 * bf.NBp * (double)Hy * Hy * 9.0 * cx_alg * cy_alg);
    DV_TRY(launch_bwgrad(p, st, &ns));
  }
  if ((slab & 3) || (Cy & 3)) {
    set_error("bf_wgrad: slab sizes must be multiples of 4 floats");
    return E_INVALID;
  }
  WRedEntry& e = bf.wred.e[bf.wred.count++];
  e.part = p.part; e.out = out; e.nsplit = ns; e.slab4 = (int)(slab / 4); e.ncols4 = Cy / 4; e.cpad = cpad; e.creal = creal;
  if (!on_main) bf.slab_off += (((size_t)ns * slab + 63) / 64) * 64;
  return OK;
}

// Weight gradient of a layer whose kernel size is not 3 (model.py:81-91 takes any kernels[i]): the bf16 weight-gradient
// kernel keeps a 32 x 32 x NINE-tap accumulator per wave, so these layers take the fp32 engine's table-driven kernel on fp32
// copies of the two bf16 operands (exact: every bf16 value is an fp32 value, products and sums in fp32 as on the bf16 matrix
// cores).  Everything runs on the MAIN stream - copies, kernel, slab sum - after the weight-gradient and reduction streams
// have been joined (their launches rotate through the same slab workspace): functional, not tuned, like the fp32 engine's
// own k != 3 path.
static bool bf_wgrad_takes(int c) { return c == 16 || c % 32 == 0; }   // channel counts of bwgrad_kernel's 32-wide tiles
static int bf_wgrad_f32(dv_model* m, const void* X, int Hx, int Cx, const void* Y, int Hy, int Cy, int NB, int s, int pb,
                        float* out, int cpad, int creal, int ksz) {
  BfState& bf = m->bf;
  dv_ctx* cx = m->ctx;
  hipStream_t st = cx->stream;
  if (!bf.wx32 || !bf.wy32) {
    set_error("bf16 engine: no fp32 operand buffers for a %d x %d weight gradient", ksz, ksz);
    return E_STATE;
  }
  hipStream_t ws = bf_wstream(m);
  if (ws != st) {
    DV_HIP(hipEventRecord(cx->ev_join, ws));
    DV_HIP(hipStreamWaitEvent(st, cx->ev_join, 0));
    if (cx->red_stream) {
      DV_HIP(hipEventRecord(cx->ev_red, cx->red_stream));
      DV_HIP(hipStreamWaitEvent(st, cx->ev_red, 0));
    }
  }
  {
    ProfScope ps(m, 2, st);
    DV_TRY(launch_bf_to_rows(X, bf.wx32, NB, bf.NBp, Hx * Hx, Cx, st));
    DV_TRY(launch_bf_to_rows(Y, bf.wy32, NB, bf.NBp, Hy * Hy, Cy, st));
  }
  hipStream_t keep = m->wstream;
  m->wstream = st;                                   // kernel and slab sum on the main stream, the whole of ws1
  const int r = wgrad(m, bf.wx32, Hx, Cx, bf.wy32, Hy, Cy, NB, s, pb, false, out, cpad, creal, nullptr, true, ksz);
  m->wstream = keep;
  return r;
}

// data gradient into `out` with the PReLU backward of the target layer (pre-activation u, slopes / bias specs) applied:
// fused into the epilogue when the stamp padding allows it, else a separate pass
// dense_bias: the bias of the target layer has one entry per (pixel, channel) - the decoder trunk's Dense -> Reshape
// (model.py:116-119) - instead of one per channel: its gradient is the stamp sum itself, not summed over the pixels
static int bf_dgrad_prelu(dv_model* m, const void* X, const void* W, int Kpad, int form, int Hin, int Cin, int Hout,
                          int Cout, int s, int pb, void* out, const void* u, int alpha_spec, int bias_spec,
                          bool want_grads, int ksz = 3, bool dense_bias = false) {
  const Arch& A = m->A;
  BfState& bf = m->bf;
  hipStream_t st = m->ctx->stream;
  const long P = (long)Hout * Hout, E = P * Cout;
  const float* alpha = m->P + A.specs[alpha_spec].off;
  if (bconv_bwd_fusable(bf.NBp)) {
    const int nparts = bf.NBp >> 6;
    float *dal = nullptr, *db = nullptr;
    if (want_grads) {
      // partial slabs + the [pixels][Cout] image the d(bias) column sum reads
      const size_t img = std::max<size_t>((size_t)E, (size_t)64 * Cout);    // partial rows of the d(bias) column sum
      if (bf.red.count + 2 > DV_BF_MAX_RED) {
        // the batch is full (deep nets without a bucket flush in between): sum what has been registered so far.  The
        // partials were written by main-stream kernels, so the reduction is ordered behind them on the same stream.
        ProfScope ps(m, 2, st);
        DV_TRY(launch_bf_reduce_batch(bf.red, st));
        bf.red.count = 0;
      }
      if (m->arena_off + (size_t)2 * nparts * E + img > m->arena_elems) {
        set_error("gradient-partial arena exhausted");
        return E_STATE;
      }
      dal = m->arena + m->arena_off;
      db = dal + (size_t)nparts * E;
      float* dbimg = db + (size_t)nparts * E;
      m->arena_off += (size_t)2 * nparts * E + img;
      BRedEntry& a = bf.red.e[bf.red.count++];
      a.src = dal; a.out = m->G + A.specs[alpha_spec].off; a.final_out = nullptr; a.nparts = nparts; a.n = (int)E; a.cols = 0;
      BRedEntry& b = bf.red.e[bf.red.count++];
      b.src = db; b.nparts = nparts; b.n = (int)E;
      if (dense_bias) { b.out = m->G + A.specs[bias_spec].off; b.final_out = nullptr; b.cols = 0; }
      else { b.out = dbimg; b.final_out = m->G + A.specs[bias_spec].off; b.cols = Cout; }
    }
    return bf_conv(m, X, W, Kpad, form, Hin, Cin, Hout, Cout, s, pb, BEPI_BWD, out, nullptr, nullptr, nullptr, alpha, u, dal,
                   db, ksz);
  }
  DV_TRY(bf_conv(m, X, W, Kpad, form, Hin, Cin, Hout, Cout, s, pb, BEPI_RAWBF, out, nullptr, nullptr, nullptr, nullptr,
                 nullptr, nullptr, nullptr, ksz));
  float* dbr = nullptr;
  if (want_grads) {
    if (m->arena_off + (size_t)E > m->arena_elems) {
      set_error("gradient-partial arena exhausted");
      return E_STATE;
    }
    dbr = m->arena + m->arena_off;
    m->arena_off += (size_t)E;
  }
  ProfScope ps(m, 2, st);
  DV_TRY(launch_bf_prelu_bwd(out, u, alpha, out, want_grads ? m->G + A.specs[alpha_spec].off : nullptr, dbr, bf.NBp,
                             (int)P, Cout, st));
  if (want_grads) {
    if (dense_bias) DV_TRY(launch_reduce_rows_f64(dbr, 1, (int)E, m->G + A.specs[bias_spec].off, 1.0f, st));
    else DV_TRY(launch_reduce_rows_f64(dbr, (int)P, Cout, m->G + A.specs[bias_spec].off, 1.0f, st));
  }
  return OK;
}

// a dense kernel gradient on the weight-gradient stream (btrunk.hip): ordered behind the main stream's product of its
// operands, written once, no slabs - the bucket boundaries that follow join the weight-gradient stream
static int bf_trunk_wgrad(dv_model* m, const BGemmTnParams& t, double flops) {
  hipStream_t st = bf_wstream(m);
  if (st != m->ctx->stream) {
    DV_HIP(hipEventRecord(m->ctx->ev_ready, m->ctx->stream));
    DV_HIP(hipStreamWaitEvent(st, m->ctx->ev_ready, 0));
  }
  ProfScope ps(m, 1, st, PF_BTRUNK, flops);
  return launch_bgemm_tn(t, st);
}

static int bf_backward(dv_model* m, int NB, int Bg) {
  const Arch& A = m->A;
  BfState& bf = m->bf;
  hipStream_t s = m->ctx->stream;
  float* P = m->P;
  float* G = m->G;
  const bool dg = m->dec_trainable;
  m->wstream = s;
  m->arena_off = 0;
  bf.red.count = 0;
  bf.wred.count = 0;
  bf.slab_off = 0;
  bf.red_pending = false;          // (the previous pass ended with the main stream joined to the reduction stream)
  m->ws_count = 0;
  m->main_marked = false;
  const int Hd = A.dec_out, f0 = A.cfg.filters[0], C2 = 2 * A.C;
  size_t gnext = 0;
  auto next_buf = [&]() -> void* { return bf.gpool[gnext++ % bf.gpool.size()]; };
  void* cur = next_buf();
  void* oth = nullptr;
  bf.du_enc.assign(2 * A.L, nullptr);
  bf.du_dec.assign(2 * A.L, nullptr);
  hipStream_t ws = bf_wstream(m);
  bool head_cols_taken = false, dec_bucket_done = false;
  size_t enc_reduced = A.n_enc_train;                    // [enc_reduced, n_enc_train) all-reduced inside this pass
  const bool trunk_red = ws != s && m->arena_reduce && m->ctx->red_stream != nullptr;   // sums on the reduction stream
  // ---- what the forward pass left for the reduction stream: the loss sums, and with them the head's bias column sums ----
  // (both used to sit on the main stream between the head kernel and the first data gradient: four launches of 5 - 10 us
  // in front of the chain the whole pass hangs on, and in front of the record the first weight gradient waits for)
  const bool sums_on_red = m->loss_pending;
  if (sums_on_red) {
    if (!trunk_red) {
      set_error("deferred loss sums without a reduction stream");
      return E_STATE;
    }
    dv_ctx* cx = m->ctx;
    hipStream_t rs = cx->red_stream;
    DV_HIP(hipEventRecord(cx->ev_ready, s));               // behind the head kernel; the head's weight gradient shares it
    DV_HIP(hipStreamWaitEvent(rs, cx->ev_ready, 0));
    bf.head_marked = true;
    {
      ProfScope ps(m, 2, rs);
      DV_TRY(launch_reduce_rows_f64(m->ws_head, m->loss_blocks, 2, m->scal, 1.0f, rs));
      DV_TRY(launch_reduce_rows_f64(m->kl, m->loss_NB, 1, m->scal + 2, 1.0f, rs));
    }
    if (cx->comm) {                                        // (allreduce_small's place in enqueue_step, one stream over)
      DV_HIP(hipEventRecord(cx->ev_small, rs));
      DV_HIP(hipStreamWaitEvent(cx->comm_stream, cx->ev_small, 0));
      DV_TRY(comm_allreduce(cx, m->scal, 4));
    }
    m->loss_pending = false;
  }
  // ---- head conv ----
#ifdef DV_DEBUG_EXPORTS
  static const int exp_defer = DV_EXP_SWITCH("DV_EXP_DEFER_WGRAD");     // MEASUREMENT only, see backward() in engine.hip
#else
  [[maybe_unused]] constexpr int exp_defer = 0;
#endif
  if (dg) {
    if (bf_wgrad_takes(f0)) {
#ifdef DV_DEBUG_EXPORTS
      if (exp_defer > 0 && ws != s) {
        const void* xa = bf.dec_a[2 * A.L - 1];
        const void* ya = bf.dt;
        const int c2p = A.C2p;
        m->exp_deferred.push_back([=]() -> int { return bf_wgrad(m, xa, Hd, f0, ya, Hd, c2p, 1, 1, m->Ghs, f0, f0); });
      } else
#endif
        DV_TRY(bf_wgrad(m, bf.dec_a[2 * A.L - 1], Hd, f0, bf.dt, Hd, A.C2p, 1, 1, m->Ghs, f0, f0));   // (columns taken at the end)
    } else DV_TRY(bf_wgrad_f32(m, bf.dec_a[2 * A.L - 1], Hd, f0, bf.dt, Hd, A.C2p, NB, 1, 1, m->Ghs, f0, f0, 3));
    hipStream_t bs = sums_on_red ? m->ctx->red_stream : s;
    float* part = sums_on_red ? m->ws_head : m->ws3;       // (same stream as the loss sums above: ordered behind their reads)
    ProfScope ps(m, 2, bs);
    int nr = 0;
    if (!(exp_skip_tail() & 2)) {
      DV_TRY(launch_bf_colsum(bf.dt, (long)Hd * Hd * bf.NBp, A.C2p, part, &nr, bs));
      DV_TRY(launch_reduce_rows_f64(part, nr, C2, G + A.specs[A.head_b()].off, 1.0f, bs, A.C2p));
    }
  }
  bf.head_marked = false;
  {
    const int jl = 2 * A.L - 1;
    DV_TRY(bf_dgrad_prelu(m, bf.dt, bf.head_w.d, bf.head_w.Kd, 1, Hd, A.C2p, Hd, f0, 1, 1, cur, bf.dec_u[jl], A.dec_al(jl),
                          A.dec_b(jl), dg));
  }
  // ---- decoder conv-transpose stack: cur = d(pre-activation) of layer j ----
  for (int j = 2 * A.L - 1; j >= 0; --j) {
    int hin, cin, hout, cout, st;
    A.dec_layer(j, &hin, &cin, &hout, &cout, &st);
    const int ksz = A.dec_ksz(j);
    const int pb = same_pad_before(hout, ksz, st, nullptr);
    const void* xin = j == 0 ? bf.dec_in : bf.dec_a[j - 1];
    bf.du_dec[j] = cur;
    const bool wg_bf = ksz == 3 && bf_wgrad_takes(cout) && bf_wgrad_takes(cin);
#ifdef DV_DEBUG_EXPORTS
    if (dg && wg_bf && exp_defer > 0 && j >= 2 * A.L - exp_defer && ws != s) {
      const void* dy = cur;
      float* gk = G + A.specs[A.dec_k(j)].off;
      m->exp_deferred.push_back([=]() -> int { return bf_wgrad(m, dy, hout, cout, xin, hin, cin, st, pb, gk, cout, cout); });
    } else
#endif
    if (dg && wg_bf) DV_TRY(bf_wgrad(m, cur, hout, cout, xin, hin, cin, st, pb, G + A.specs[A.dec_k(j)].off, cout, cout));
    if (dg && !wg_bf) DV_TRY(bf_wgrad_f32(m, cur, hout, cout, xin, hin, cin, NB, st, pb, G + A.specs[A.dec_k(j)].off, cout, cout, ksz));
    oth = next_buf();
    if (j > 0) {
      DV_TRY(bf_dgrad_prelu(m, cur, bf.dec_w[j].d, bf.dec_w[j].Kd, 0, hout, cout, hin, cin, st, pb, oth, bf.dec_u[j - 1],
                            A.dec_al(j - 1), A.dec_b(j - 1), dg, ksz));
    } else if (bf.trunk_mfma) {
      // the PReLU behind the decoder trunk's Dense (model.py:117-118) is the "layer below" of the first transposed conv:
      // its pre-activation sits in the stamp-inner layout (bf_decoder_forward), so its backward runs in this epilogue
      // like every other layer's, and d(pre-activation) arrives where the two dense products below read it
      DV_TRY(bf_dgrad_prelu(m, cur, bf.dec_w[j].d, bf.dec_w[j].Kd, 0, hout, cout, hin, cin, st, pb, oth, bf.dec_ur,
                            A.D0 + 6, A.D0 + 5, dg, ksz, /*dense_bias=*/true));
    } else {
      DV_TRY(bf_conv(m, cur, bf.dec_w[j].d, bf.dec_w[j].Kd, 0, hout, cout, hin, cin, st, pb, BEPI_RAWBF, oth, nullptr,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ksz));
    }
    cur = oth;
  }
  bf.d_dec_in = cur;
  // ---- dense trunk of the decoder, sampler, encoder dense: fp32 rows in bf.trunk[] ----
  const int fl = A.cfg.filters[A.L - 1];
  const int r = A.w0 * A.w0 * fl;
  // One buffer per stage: the dense weight gradients read these rows from the weight-gradient stream (wgrad() with
  // m->wstream = ws: kernel on the aux stream, slab sum and the d(alpha) / d(bias) sums of prelu_bwd on the reduction
  // stream), so the main stream only carries the chain prelu_bwd -> data gradient and never rewrites a row buffer.
  float* const tr0 = bf.trunk[0];   // d(decoder trunk output)   [NB, r]
  float* const tr1 = bf.trunk[1];   // d(hidden)                 [NB, dec_hidden]
  float* const tr2 = bf.trunk[2];   // d(z)                      [NB, d]
  float* const tr3 = bf.trunk[3];   // d(t)                      [NB, tw]
  float* const tr4 = bf.trunk[4];   // d(flatten)                [NB, flat]
  m->wstream = ws;
  if (bf.trunk_mfma) {
    // cur = d(pre-activation of the trunk's output), bf16 stamp-inner (the epilogue above).  Its two consumers read it
    // as it lies: the kernel gradient of Dense(560 -> flat) on the weight-gradient stream, and the data gradient - K =
    // flat, split, slabs added by one small launch - on the main stream
    if (dg) {
      BGemmTnParams t;
      memset(&t, 0, sizeof t);
      t.X = m->dec_ah; t.xmode = BGA_ROWS_F32; t.ldx = A.dec_hidden; t.I = bf.HIDn; t.Ireal = A.dec_hidden;
      t.Y = cur; t.ymode = BGA_STAMP; t.Cy = fl; t.J = bf.FL; t.Jreal = bf.FL;
      t.NBp = bf.NBp; t.Mreal = NB; t.G = G + A.specs[A.D0 + 4].off; t.ldg = r;
      DV_TRY(bf_trunk_wgrad(m, t, 2.0 * bf.NBp * (double)A.dec_hidden * r));
    }
    BGemmParams g;
    memset(&g, 0, sizeof g);
    g.A = cur; g.B = bf.w1_d; g.amode = BGA_STAMP; g.epi = BGE_SLAB;
    g.M = bf.NBp; g.Mreal = NB; g.N = bf.HIDn; g.K = bf.FL; g.Kreal = bf.FL; g.ldb = bf.FL; g.NBp = bf.NBp; g.C = fl;
    g.slab = bf.tslab; g.slab_stride = (long)bf.NBp * bf.HIDn; g.ldc = bf.HIDn;
    bf_trunk_ksplit(((bf.NBp + 31) / 32) * (bf.HIDn / 32), bf.FL, &g.wg_ksplit, &g.nslab);
    {
      ProfScope ps(m, 0, s, PF_BTRUNK, 2.0 * bf.NBp * (double)A.dec_hidden * r);
      DV_TRY(launch_bgemm(g, s));
    }
    // ... and everything between that product and d(t) in one launch: slab sum, PReLU backward of the hidden layer,
    // Dense(latent -> 560) data gradient, PReLU backward of z, sampler backward (bt_mid_bwd_kernel).  The four batch sums
    // it leaves as rows (d(bias) / d(alpha) of the hidden layer, d(alpha) of z's PReLU, d(bias) of the encoder Dense) are
    // column sums on the reduction stream, off the chain
    hipStream_t rs = (m->arena_reduce && ws != s) ? m->ctx->red_stream : s;
    float *dalh = nullptr, *dalin = nullptr;
    if (dg) {
      const size_t need = (size_t)NB * A.dec_hidden + (size_t)NB * A.dp;
      if (m->arena_off + need > m->arena_elems) {
        set_error("gradient-partial arena exhausted");
        return E_STATE;
      }
      dalh = m->arena + m->arena_off;
      dalin = dalh + (size_t)NB * A.dec_hidden;
      m->arena_off += (need + 3) & ~(size_t)3;
    }
    const float kls_mid = (float)((double)A.cfg.kl_multiplicity * A.cfg.kl_weight / ((double)Bg * (double)Bg));
    {
      BMidBwdParams q;
      memset(&q, 0, sizeof q);
      q.slab = bf.tslab; q.slab_stride = g.slab_stride; q.nslab = g.nslab; q.lds = bf.HIDn;
      q.uh = m->dec_uh; q.alpha_h = P + A.specs[A.D0 + 3].off; q.W0 = dec_dense0_w(m);
      q.z = m->z; q.alpha_in = P + A.specs[A.D0].off; q.eps = m->eps; q.t = m->t;
      q.duh = tr1; q.dalh = dalh; q.dz = tr2; q.dalin = dalin; q.dt = tr3;
      q.NB = NB; q.hid = A.dec_hidden; q.d = A.d; q.ldt = A.twp; q.ldz = A.dp;
      q.diag_shift = A.cfg.diag_shift; q.kls = kls_mid;
      ProfScope ps(m, 2, s);
      DV_TRY(launch_bt_mid_bwd(q, s));
    }
    m->main_marked = false;
    if (rs != s) {
      DV_HIP(hipEventRecord(m->ctx->ev_ready, s));
      DV_HIP(hipStreamWaitEvent(rs, m->ctx->ev_ready, 0));
      m->main_marked = true;             // the weight gradient of Dense(latent -> 560) below waits on the same record
    }
    {
      BColsums c;
      memset(&c, 0, sizeof c);
      c.NB = NB;
      auto add = [&](const float* x, int ld, int n, int spec) {
        c.x[c.count] = x; c.ld[c.count] = ld; c.n[c.count] = n; c.out[c.count] = G + A.specs[spec].off; ++c.count;
      };
      add(tr3, A.twp, A.tw, A.enc_db());
      if (dg) {
        add(tr1, A.dec_hidden, A.dec_hidden, A.D0 + 2);
        add(dalh, A.dec_hidden, A.dec_hidden, A.D0 + 3);
        add(dalin, A.dp, A.d, A.D0);
      }
      ProfScope ps(m, 2, rs);
      DV_TRY(launch_bt_colsums(c, rs));
    }
    if (dg) {
      DV_TRY(wgrad(m, m->dec_ain, 1, A.dp, tr1, 1, A.dec_hidden, NB, 1, 0, true, dec_dense0_g(m), 1, 1));
      if (m->G0p) DV_TRY(take_padded_grad(m, m->G0p, G + A.specs[A.D0 + 1].off, 1, A.dp * A.dec_hidden, A.d * A.dec_hidden));
    }
  } else {
    {
      ProfScope ps(m, 2, s);
      if (!exp_skip_small()) DV_TRY(launch_bf_to_rows(cur, tr0, NB, bf.NBp, A.w0 * A.w0, fl, s));
    }
    DV_TRY(prelu_bwd(m, tr0, m->dec_ur, A.D0 + 6, A.D0 + 5, NB, r, r, dg));
    if (dg) DV_TRY(wgrad(m, m->dec_ah, 1, A.dec_hidden, tr0, 1, r, NB, 1, 0, true, G + A.specs[A.D0 + 4].off, 1, 1));
    DV_TRY(gconv_fprop(m, tr0, P + A.specs[A.D0 + 4].off, true, nullptr, nullptr, tr1, nullptr, 0, NB, 1, r, 1,
                       A.dec_hidden, 1, 0, true));
  }
  if (!bf.trunk_mfma) {
    DV_TRY(prelu_bwd(m, tr1, m->dec_uh, A.D0 + 3, A.D0 + 2, NB, A.dec_hidden, A.dec_hidden, dg));
    if (dg) {
      DV_TRY(wgrad(m, m->dec_ain, 1, A.dp, tr1, 1, A.dec_hidden, NB, 1, 0, true, dec_dense0_g(m), 1, 1));
      if (m->G0p) DV_TRY(take_padded_grad(m, m->G0p, G + A.specs[A.D0 + 1].off, 1, A.dp * A.dec_hidden, A.d * A.dec_hidden));
    }
    DV_TRY(gconv_fprop(m, tr1, dec_dense0_w(m), true, nullptr, nullptr, tr2, nullptr, 0, NB, 1, A.dec_hidden, 1,
                       A.dp, 1, 0, true));
    DV_TRY(prelu_bwd(m, tr2, m->z, A.D0, -1, NB, A.dp, A.dp, dg));
  }
  // Every decoder gradient has been queued and no later kernel of the step reads a decoder parameter: finish the
  // decoder's reductions now (slab sums and d(alpha) / d(bias) partials, one launch each, on the weight-gradient
  // stream), all-reduce the bucket on the comm stream while the encoder backward runs and - early_adam - update it there
  // (18.3 of the 33.3 MB; SURVEY 8(e): buckets in reverse layer order).
  {
    dv_ctx* cx = m->ctx;
    const bool ovl = ws != s;
    const bool early = m->early_adam && ovl;
    if ((cx->comm || early) && dg && A.n_train > A.n_enc_train) {
      if (ovl) {                                         // the partials of the fused epilogues come from the main stream
        DV_HIP(hipEventRecord(cx->ev_ready, s));
        if (!bf_boundary_on_red(trunk_red)) DV_HIP(hipStreamWaitEvent(ws, cx->ev_ready, 0));   // (else the reduction stream waits)
      }
      int bst = OK;
      hipStream_t rs = bf_boundary_stream(m, ws, s, trunk_red, &bst);
      DV_TRY(bst);
      DV_TRY(bf_flush_wred(m, rs));
      {
        ProfScope ps(m, 2, rs);
        DV_TRY(launch_take_cols(m->Ghs, G + A.specs[A.head_k()].off, 9 * f0, A.C2p, C2, rs));
        DV_TRY(launch_bf_reduce_batch(bf.red, rs));
      }
      bf.red.count = 0;
      head_cols_taken = true;
      DV_HIP(hipEventRecord(cx->ev_dec, s));             // the main stream has finished reading decoder parameters
      DV_HIP(hipStreamWaitEvent(cx->comm_stream, cx->ev_dec, 0));
      if (ovl) {
        DV_HIP(hipEventRecord(cx->ev_join, ws));
        DV_HIP(hipStreamWaitEvent(cx->comm_stream, cx->ev_join, 0));
      }
      if (trunk_red) {                                   // the decoder trunk's slab / d(alpha) / d(bias) sums
        DV_HIP(hipEventRecord(cx->ev_red, cx->red_stream));
        DV_HIP(hipStreamWaitEvent(cx->comm_stream, cx->ev_red, 0));
      }
      if (cx->comm)
        DV_TRY(comm_allreduce(cx, G + A.n_enc_train, A.n_train - A.n_enc_train));
      if (early && m->opt_dec) {
        DV_TRY(adam_range(m, A.n_enc_train, A.n_train, cx->comm_stream));
        DV_TRY(refresh_head_pad(m, cx->comm_stream));
        DV_TRY(bf_cast_bucket_early(m, 2, cx->comm_stream));
        m->adam_done_from = A.n_enc_train;
      }
      dec_bucket_done = true;
    }
  }
  const float kls = (float)((double)A.cfg.kl_multiplicity * A.cfg.kl_weight / ((double)Bg * (double)Bg));
  if (!bf.trunk_mfma) {
    {
      ProfScope ps(m, 2);
      DV_TRY(launch_sampler_bwd(m->t, m->eps, m->z, tr2, tr3, NB, A.d, A.twp, A.dp, A.cfg.diag_shift, kls, s));
    }
    m->main_marked = false;   // (the record prelu_bwd left behind predates the sampler: the dense weight gradient below needs its own)
    DV_TRY(bias_grad_colsum(m, tr3, NB, A.twp, A.tw, A.enc_db()));
  }
  if (bf.trunk_mfma) {
    const int jl = 2 * A.L - 1;
    // kernel gradient of the encoder Dense on the weight-gradient stream: X = PReLU(flatten(activation of the last conv)),
    // read from the stamp-inner tensor with the slope applied on load, Y = d(t) rows
    {
      BGemmTnParams t;
      memset(&t, 0, sizeof t);
      t.X = bf.enc_a[jl]; t.xmode = BGA_STAMP_PRELU; t.Cx = fl; t.I = bf.FL; t.Ireal = bf.FL;
      t.x_alpha = P + A.specs[A.enc_flat_al()].off;
      t.Y = tr3; t.ymode = BGA_ROWS_F32; t.ldy = A.twp; t.J = bf.TWn; t.Jreal = A.tw;
      t.NBp = bf.NBp; t.Mreal = NB; t.G = G + A.specs[A.enc_dk()].off; t.ldg = A.tw;
      DV_TRY(bf_trunk_wgrad(m, t, 2.0 * bf.NBp * (double)A.flat * A.tw));
    }
    // data gradient of the encoder Dense with BOTH PReLU backward passes of the seam in its epilogue (the flatten PReLU,
    // model.py:95, and the last conv's, :92): d(pre-activation) of that conv leaves in the stamp-inner layout, the
    // column sums of d(alpha) of either and of d(bias) as one partial row per 32 stamps (summed with the other fused
    // epilogues' partials at the next bucket boundary)
    const int MT = (bf.NBp + 31) / 32;
    const size_t E = (size_t)bf.FL;
    if (bf.red.count + 3 > DV_BF_MAX_RED) {
      ProfScope ps(m, 2, s);
      DV_TRY(launch_bf_reduce_batch(bf.red, s));
      bf.red.count = 0;
    }
    if (m->arena_off + (size_t)3 * MT * E + std::max<size_t>(E, (size_t)64 * fl) > m->arena_elems) {
      set_error("gradient-partial arena exhausted");
      return E_STATE;
    }
    float* pflat = m->arena + m->arena_off;
    float* p7 = pflat + (size_t)MT * E;
    float* pdb = p7 + (size_t)MT * E;
    float* dbimg = pdb + (size_t)MT * E;
    m->arena_off += (size_t)3 * MT * E + std::max<size_t>(E, (size_t)64 * fl);
    {
      BRedEntry& e0 = bf.red.e[bf.red.count++];
      e0.src = pflat; e0.out = G + A.specs[A.enc_flat_al()].off; e0.final_out = nullptr; e0.nparts = MT; e0.n = (int)E; e0.cols = 0;
      BRedEntry& e1 = bf.red.e[bf.red.count++];
      e1.src = p7; e1.out = G + A.specs[A.enc_al(jl)].off; e1.final_out = nullptr; e1.nparts = MT; e1.n = (int)E; e1.cols = 0;
      BRedEntry& e2 = bf.red.e[bf.red.count++];
      e2.src = pdb; e2.out = dbimg; e2.final_out = G + A.specs[A.enc_b(jl)].off; e2.nparts = MT; e2.n = (int)E; e2.cols = fl;
    }
    cur = next_buf();
    BGemmParams g;
    memset(&g, 0, sizeof g);
    g.A = tr3; g.B = bf.wenc_d; g.amode = BGA_ROWS_F32; g.epi = BGE_STAMP_GATE2;
    g.M = bf.NBp; g.Mreal = NB; g.N = bf.FL; g.K = bf.TWk; g.Kreal = A.twp; g.lda = A.twp; g.ldb = bf.TWk;
    g.NBp = bf.NBp; g.nslab = 1; g.wg_ksplit = 4; g.Co = fl;
    g.a7 = bf.enc_a[jl]; g.u7 = bf.enc_u[jl];
    g.alpha_flat = P + A.specs[A.enc_flat_al()].off; g.alpha7 = P + A.specs[A.enc_al(jl)].off;
    g.dU = cur; g.part_dal_flat = pflat; g.part_dal7 = p7; g.part_db = pdb;
    ProfScope ps(m, 0, s, PF_BTRUNK, 2.0 * bf.NBp * (double)A.flat * A.tw);
    DV_TRY(launch_bgemm(g, s));
    m->wstream = s;
  } else {
  DV_TRY(wgrad(m, m->flat_a, 1, A.flat, tr3, 1, A.twp, NB, 1, 0, true, enc_dense_g(m), 1, 1));
  if (m->Gdp) DV_TRY(take_padded_grad(m, m->Gdp, G + A.specs[A.enc_dk()].off, A.flat, A.twp, A.tw));
  DV_TRY(gconv_fprop(m, tr3, enc_dense_w(m), true, nullptr, nullptr, tr4, nullptr, 0, NB, 1, A.twp, 1, A.flat,
                     1, 0, true));
  DV_TRY(prelu_bwd(m, tr4, bf.flat_in, A.enc_flat_al(), -1, NB, A.flat, A.flat, true));
  m->wstream = s;
  float* const c32 = tr4;
  // ---- back into the stamp-inner layout: d(activation) of the last encoder conv, then its PReLU backward ----
  {
    const int jl = 2 * A.L - 1;
    const int sl = A.enc_sizes[A.L];
    const long Pn = (long)sl * sl, E = Pn * fl;
    cur = next_buf();
    if (m->arena_off + (size_t)E > m->arena_elems) {
      set_error("gradient-partial arena exhausted");
      return E_STATE;
    }
    float* dbr = m->arena + m->arena_off;
    m->arena_off += (size_t)E;
    ProfScope ps(m, 2, s);
    if (!exp_skip_small()) DV_TRY(launch_bf_from_rows(c32, cur, NB, bf.NBp, (int)Pn, fl, s));
    DV_TRY(launch_bf_prelu_bwd(cur, bf.enc_u[jl], P + A.specs[A.enc_al(jl)].off, cur, G + A.specs[A.enc_al(jl)].off, dbr,
                               bf.NBp, (int)Pn, fl, s));
    DV_TRY(launch_reduce_rows_f64(dbr, (int)Pn, fl, G + A.specs[A.enc_b(jl)].off, 1.0f, s));
  }
  }
  // ---- encoder conv stack: cur = d(pre-activation) of layer j ----
  // The tail of a pass is the weight-gradient stream's backlog: when the main stream has queued its last data gradient
  // the aux stream still owes the launches of the last three or four layers.  Layer 1's launch is therefore held back and
  // queued on the MAIN stream behind that data gradient, next to layer 0's (2.110 -> 2.093 ms; layers 2 and 3 as well:
  // no further gain, four alternating same-box runs each).
  const int want_defer = 1;
  bool deferred[BF_MAIN_SLOTS] = {false, false};
  bool any_deferred = false, tail_on_main = false;
  for (int j = 2 * A.L - 1; j >= 0; --j) {
    int hin, cin, hout, cout, st;
    A.enc_layer(j, &hin, &cin, &hout, &cout, &st);
    const int ksz = A.enc_ksz(j);
    const int pb = same_pad_before(hin, ksz, st, nullptr);
    bf.du_enc[j] = cur;
    if (j == 0) {
      // first conv with the folded input BatchNorm: the gradient w.r.t. the 16-channel folded kernel (channels
      // 0..C-1 = bands, C = the constant one) yields d(kernel), d(gamma), d(beta); no data gradient
      // last launch of the pass: on the (otherwise idle) main stream, slabs in the pool's tail region
      const bool wg0 = ksz == 3 && bf_wgrad_takes(cout);
      const bool lm = !wg0 || (ws != s && (size_t)((bf.NBp + 63) / 64) * 9 * 16 * cout <= bf.slab_tail / BF_MAIN_SLOTS);
      // The tail of the pass (round 6).  The d(alpha) / d(bias) partials of the shallow layers are complete once the last
      // data gradient has been queued: their sums go to the reduction stream NOW, beside the last weight gradients, instead
      // of behind them.  And what follows the last weight gradients - slab sums, the first conv's gradients - runs on the
      // MAIN stream behind its join with the weight-gradient stream, where the optimizer continues, instead of on the
      // weight-gradient stream with a second hand-over behind it.  DV_BF_TAIL_ON_WGRAD_STREAM=1: the form until round 6.
      static const bool old_tail = getenv("DV_BF_TAIL_ON_WGRAD_STREAM") != nullptr;
      tail_on_main = trunk_red && lm && wg0 && !old_tail;
      if (tail_on_main && bf.red.count > 0) {
        dv_ctx* cx = m->ctx;
        DV_HIP(hipEventRecord(cx->ev_ready, s));
        DV_HIP(hipStreamWaitEvent(cx->red_stream, cx->ev_ready, 0));
        ProfScope ps(m, 2, cx->red_stream);
        DV_TRY(launch_bf_reduce_batch(bf.red, cx->red_stream));
        bf.red.count = 0;
      }
      // (the folded kernel's gradient has A.C0p input channels: 8 for 1 .. 7 bands, all 16 for 8 .. 15)
      if (wg0) DV_TRY(bf_wgrad(m, bf.xh, hin, 16, cur, hout, cout, st, pb, m->G0s, 16, A.C0p, lm));
      else DV_TRY(bf_wgrad_f32(m, bf.xh, hin, 16, cur, hout, cout, NB, st, pb, m->G0s, 16, A.C0p, ksz));
      for (int jd = 1; jd < BF_MAIN_SLOTS; ++jd) {       // the launches held back behind the last data gradient
        if (!deferred[jd]) continue;
        int h1, c1, ho1, co1, s1;
        A.enc_layer(jd, &h1, &c1, &ho1, &co1, &s1);
        DV_TRY(bf_wgrad(m, bf.enc_a[jd - 1], h1, c1, bf.du_enc[jd], ho1, co1, s1, same_pad_before(h1, 3, s1, nullptr),
                        G + A.specs[A.enc_k(jd)].off, c1, c1, true, jd));     // (only 3 x 3 layers are deferred)
      }
      hipStream_t ts = ws;                               // the stream the tail runs on
      if (tail_on_main) {                                // the main stream joins the weight-gradient stream here
        DV_HIP(hipEventRecord(m->ctx->ev_join, ws));
        DV_HIP(hipStreamWaitEvent(s, m->ctx->ev_join, 0));
        ts = s;
      } else if (lm || any_deferred) {                   // their slabs are summed on the weight-gradient stream
        DV_HIP(hipEventRecord(m->ctx->ev_ready, s));
        DV_HIP(hipStreamWaitEvent(ws, m->ctx->ev_ready, 0));
      }
      // every slab reduction of the pass, then what reads the two scratch gradients (padded head kernel, folded first conv)
      if (exp_skip_tail() & 1) break;
      DV_TRY(bf_flush_wred(m, ts));
      ProfScope ps(m, 2, ts);
      if (dg && !head_cols_taken) DV_TRY(launch_take_cols(m->Ghs, G + A.specs[A.head_k()].off, 9 * f0, A.C2p, C2, ts));
      DV_TRY(launch_bn_conv0_grads(m->G0s, P + A.specs[A.enc_k(0)].off, P + A.specs[0].off, P + A.specs[1].off,
                                   G + A.specs[A.enc_k(0)].off, G + A.specs[0].off, G + A.specs[1].off, ksz * ksz, A.C, A.C0p, cout,
                                   ts));
      break;
    }
    if (ksz != 3 || !bf_wgrad_takes(cin) || !bf_wgrad_takes(cout)) {
      DV_TRY(bf_wgrad_f32(m, bf.enc_a[j - 1], hin, cin, cur, hout, cout, NB, st, pb, G + A.specs[A.enc_k(j)].off, cin, cin, ksz));
    } else if (j <= want_defer && ws != s && j < BF_MAIN_SLOTS && j < A.L &&
        (size_t)((bf.NBp + 63) / 64) * 9 * cin * cout <= bf.slab_tail / BF_MAIN_SLOTS) {
      deferred[j] = any_deferred = true;                 // queued on the main stream behind the last data gradient
    } else {
      DV_TRY(bf_wgrad(m, bf.enc_a[j - 1], hin, cin, cur, hout, cout, st, pb, G + A.specs[A.enc_k(j)].off, cin, cin));
    }
    oth = next_buf();
    DV_TRY(bf_dgrad_prelu(m, cur, bf.enc_w[j].d, bf.enc_w[j].Kd, 1, hout, cout, hin, cin, st, pb, oth, bf.enc_u[j - 1],
                          A.enc_al(j - 1), A.enc_b(j - 1), true, ksz));
    cur = oth;
    if (j == A.L && A.L >= 2) {
      // Middle bucket, as in the fp32 backward: the deep half of the encoder (conv L .. conv 2L-1, their PReLUs, the
      // flatten PReLU, the dense layer: 13.8 of 15 MB) is final once this layer's weight gradient has been queued, and
      // the data-gradient kernel just queued was the last reader of those layers.  Its reductions are flushed, the
      // bucket all-reduced on the comm stream and (early Adam) updated there while the shallow half is differentiated.
      dv_ctx* cx = m->ctx;
      const bool ovl = ws != s;
      const bool early = m->early_adam && ovl;
      const size_t split = enc_bucket_split(A);
      if ((cx->comm || early) && split < A.n_enc_train) {
        if (ovl) {
          DV_HIP(hipEventRecord(cx->ev_ready, s));
          if (!bf_boundary_on_red(trunk_red)) DV_HIP(hipStreamWaitEvent(ws, cx->ev_ready, 0));
        }
        int bst = OK;
        hipStream_t rs = bf_boundary_stream(m, ws, s, trunk_red, &bst);
        DV_TRY(bst);
        DV_TRY(bf_flush_wred(m, rs));
        {
          ProfScope ps(m, 2, rs);
          DV_TRY(launch_bf_reduce_batch(bf.red, rs));
        }
        bf.red.count = 0;
        DV_HIP(hipEventRecord(cx->ev_mid, ws));
        DV_HIP(hipStreamWaitEvent(cx->comm_stream, cx->ev_mid, 0));
        if (trunk_red) {                                 // encoder dense slab sum, flatten d(alpha)
          DV_HIP(hipEventRecord(cx->ev_red, cx->red_stream));
          DV_HIP(hipStreamWaitEvent(cx->comm_stream, cx->ev_red, 0));
        }
        DV_HIP(hipEventRecord(cx->ev_dec, s));           // the main stream has finished reading those parameters
        DV_HIP(hipStreamWaitEvent(cx->comm_stream, cx->ev_dec, 0));
        if (cx->comm) {
          DV_TRY(comm_allreduce(cx, G + split, A.n_enc_train - split));
          enc_reduced = split;
        }
        if (early && m->opt_enc) {
          DV_TRY(adam_range(m, split, A.n_enc_train, cx->comm_stream));
          DV_TRY(bf_cast_bucket_early(m, 1, cx->comm_stream));
          m->adam_done_from = std::min(m->adam_done_from, split);
        }
      }
    }
  }
  if (!(exp_skip_tail() & 1)) {
    ProfScope ps(m, 2, s);
    DV_TRY(launch_bf_reduce_batch(bf.red, s));           // d(alpha) / d(bias) of every fused epilogue of this pass
  }
  if (ws != s && !tail_on_main) {                        // join: every parameter gradient is final past this point
    DV_HIP(hipEventRecord(m->ctx->ev_join, ws));
    DV_HIP(hipStreamWaitEvent(s, m->ctx->ev_join, 0));
  }
  if (trunk_red) {
    DV_HIP(hipEventRecord(m->ctx->ev_red, m->ctx->red_stream));
    DV_HIP(hipStreamWaitEvent(s, m->ctx->ev_red, 0));
  }
  // data parallelism: the decoder bucket went out above; the encoder bucket [0, n_enc_train) is all-reduced by
  // enqueue_step behind this pass (with a frozen decoder it is the only one)
  (void)dec_bucket_done;
  m->enc_reduced_from = enc_reduced;
  return OK;
}
