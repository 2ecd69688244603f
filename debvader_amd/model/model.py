"""create_model_vae / load_deblender on the MI355X engine.

Drop-in for src/debvader/model/model.py:61-271 of the reference: same function names, argument
meaning and return tuples, with Keras-like objects whose arithmetic runs in libdebvader_hip.so.
The four returned models share one set of weights, as the Keras sub-models do.
"""
from __future__ import annotations

import os
import time
from typing import Dict, List, Optional, Sequence

import numpy as np

from debvader_amd import engine as E
from debvader_amd.distributions import MultivariateNormalTriL, Normal, Tensor
from debvader_amd.training.metrics import vae_loss


class Adam:
    """Stand-in for tf.optimizers.legacy.Adam(learning_rate=...) (train.py:126); defaults as in TF."""

    def __init__(self, learning_rate=1e-3, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon


class History:
    def __init__(self):
        self.history: Dict[str, List[float]] = {}
        self.epoch: List[int] = []


def _optimizer_hparams(opt):
    if opt is None:
        return 1e-4, 0.9, 0.999, 1e-7
    if isinstance(opt, str):
        if opt.lower() != "adam":
            raise ValueError("only Adam is implemented (the reference compiles with legacy Adam, train.py:126)")
        return 1e-3, 0.9, 0.999, 1e-7
    get = lambda n, d: float(getattr(opt, n, d)) if not callable(getattr(opt, n, d)) else d
    return get("learning_rate", 1e-3), get("beta_1", 0.9), get("beta_2", 0.999), get("epsilon", 1e-7)


class _SubModel:
    """encoder / decoder handle: carries the Keras `trainable` flag (train.py:175, model.py:252)."""

    def __init__(self, core: "_Core", which: str):
        self._core, self._which, self.trainable = core, which, True

    def count_params(self):
        c = E.arch_counts(self._core.cfg)
        return c["encoder"] if self._which == "encoder" else c["decoder"]


class _Core:
    """State shared by net / encoder / decoder / z."""

    def __init__(self, input_shape, latent_dim, filters, kernels, max_batch, ctx, seed=None, **cfg_over):
        self.input_shape = tuple(int(v) for v in input_shape)
        self.latent_dim = int(latent_dim)
        self.cfg = E.make_config(self.input_shape, latent_dim, tuple(filters), tuple(kernels),
                                 max_batch=max_batch, **cfg_over)
        self.engine = E.Engine(self.cfg, ctx)
        self.ctx = self.engine.ctx
        self.seed_counter = int.from_bytes(os.urandom(4), "little")
        self.shuffle_base = 0          # create_model_vae(seed=...) makes shuffling (and the initial weights) reproducible
        self.upload_keys = {}          # data slot -> identity of the arrays resident in HBM (VAENet._upload_cached)
        self.compiled = False
        # Keras draws fresh Glorot weights for every model; the engine's dv_model_create always starts from seed 0, so
        # draw one here (seed=None: from the OS), identical on every rank (rank 0's value travels through the engine's
        # own all-reduce in exact 16-bit pieces), and keep it for reproducibility (net.init_seed)
        if seed is None:
            seed = int.from_bytes(os.urandom(6), "little")
        seed = int(seed) & ((1 << 48) - 1)
        if self.ctx.world > 1:
            parts = [float((seed >> (16 * k)) & 0xFFFF) if self.ctx.rank == 0 else 0.0 for k in range(3)]
            parts = self.ctx.allreduce(parts)
            seed = sum(int(round(float(v))) << (16 * k) for k, v in enumerate(parts))
        self.init_seed = seed
        self.shuffle_base = seed & 0xFFFFFFFF
        self.engine.init(seed)

    def next_seed(self):
        self.seed_counter += 1
        return self.seed_counter


class Encoder(_SubModel):
    """Model(input, Dense(params_size)) of create_encoder (model.py:61-100)."""

    def __call__(self, x, training=False):
        return Tensor(self._core.engine.encode(np.asarray(x, dtype=np.float32)))

    predict = __call__


class Decoder(_SubModel):
    """Model(z, Normal) of create_decoder (model.py:103-161)."""

    def __call__(self, z, training=False):
        z = np.asarray(z.numpy() if hasattr(z, "numpy") else z, dtype=np.float32)
        loc, scale = self._core.engine.decode(z)
        return Normal(loc, scale)


class LatentModel:
    """Model(x, MultivariateNormalTriL) — the 4th return value of create_model_vae (model.py:218)."""

    def __init__(self, core):
        self._core = core

    def __call__(self, x, training=False):
        t = self._core.engine.encode(np.asarray(x, dtype=np.float32))
        return MultivariateNormalTriL(t, self._core.latent_dim, self._core.cfg.diag_shift)


class VAENet:
    """`net` of create_model_vae: compile / fit / __call__ / summary / load_weights / losses."""

    def __init__(self, core: _Core, encoder: Encoder, decoder: Decoder):
        self._core, self.encoder, self.decoder = core, encoder, decoder
        self.losses: List[float] = []
        self.init_seed = core.init_seed
        self.metrics_names: List[str] = ["loss"]
        self._metrics: List = []
        self.stop_training = False
        self.history = None

    # -- Keras surface -----------------------------------------------------------------------
    def compile(self, optimizer=None, loss=None, metrics=None, mse_of_mean=False, **kwargs):
        """net.compile(optimizer=legacy.Adam(1e-4), loss=vae_loss, metrics=["mse", kl_metric], ...)
        (train.py:125-130,178-183; model.py:255-259).  Fresh Adam slots; honours `trainable` flags.
        "mse" is Keras' metric on the model OUTPUT, which for this net is a sample of the Normal (model.py:158) - the
        value that drives the reference's val_mse ModelCheckpoint (train.py:54-62).  mse_of_mean=True (engine
        extension) reports the squared error of the predicted mean instead."""
        if loss is not None and loss is not vae_loss and getattr(loss, "__name__", "") != "vae_loss":
            raise NotImplementedError("the engine fuses the reference's vae_loss (Normal NLL); other losses are "
                                      "not implemented")
        lr, b1, b2, eps = _optimizer_hparams(optimizer)
        eng = self._core.engine
        eng.set_trainable(bool(self.encoder.trainable), bool(self.decoder.trainable))
        eng.optimizer_reset(lr, b1, b2, eps)
        self.optimizer = Adam(lr, b1, b2, eps)
        self._metrics = list(metrics or [])
        eng.set_mse_sample(not mse_of_mean)
        self._core.compiled = True

    def summary(self, print_fn=print):
        cfg = self._core.cfg
        c = E.arch_counts(cfg)
        specs = self._core.engine.specs
        enc_tr = sum(int(np.prod(s)) for n, s, t in specs if t and n.startswith("enc/")) if self.encoder.trainable else 0
        dec_tr = sum(int(np.prod(s)) for n, s, t in specs if t and n.startswith("dec/")) if self.decoder.trainable else 0
        total = c["encoder"] + c["decoder"]
        print_fn('Model: "model"')
        print_fn(f" input                    (None, {cfg.height}, {cfg.width}, {cfg.bands})")
        print_fn(f" encoder (Functional)     (None, {self._core.engine.tw})          {c['encoder']}")
        print_fn(f" multivariate_normal_tri_l ((None, {cfg.latent_dim}), (None, {cfg.latent_dim}))   0")
        print_fn(f" decoder (Functional)     (None, {cfg.height}, {cfg.width}, {cfg.bands})   {c['decoder']}")
        print_fn(f"Total params: {total}")
        print_fn(f"Trainable params: {enc_tr + dec_tr}")
        print_fn(f"Non-trainable params: {total - enc_tr - dec_tr}")

    def count_params(self):
        c = E.arch_counts(self._core.cfg)
        return c["encoder"] + c["decoder"]

    def __call__(self, x, training=False):
        """net(x): one stochastic forward pass in inference mode (deblender.py:18)."""
        x = np.asarray(x.numpy() if hasattr(x, "numpy") else x)
        if x.dtype != np.float64:
            x = x.astype(np.float32, copy=False)
        r = self._core.engine.infer(x, seed=self._core.next_seed(), want=("loc", "scale"))
        return Normal(r["loc"], r["scale"])

    def predict(self, x, batch_size=None, verbose=0):
        return self(x).sample().numpy()

    # -- weights -----------------------------------------------------------------------------
    def get_weights(self):
        return [self._core.engine.get_param(i) for i in range(len(self._core.engine.specs))]

    def set_weights(self, weights):
        for i, w in enumerate(weights):
            self._core.engine.set_param(i, w)

    def save_weights(self, filepath, overwrite=True, save_format=None):
        """Keras `Model.save_weights`.  By default writes a TensorFlow tensor-bundle checkpoint (`<filepath>.index`,
        `<filepath>.data-00000-of-00001`, `checkpoint`) with the Keras object graph, i.e. what
        ModelCheckpoint(save_weights_only=True) leaves on disk in the reference (train.py:54-71) and what its
        `net.load_weights(tf.train.latest_checkpoint(dir))` restores (model.py:262-266) - variables, Adam slots and
        step counter included.  A path ending in `.npz` (or save_format="npz") writes a single numpy archive."""
        eng = self._core.engine
        if self._core.ctx.rank != 0:
            return            # data-parallel replicas hold identical weights: rank 0 writes, the others must not race it
        if not overwrite and (os.path.exists(filepath + ".index") or os.path.exists(filepath)):
            raise FileExistsError(filepath)
        if save_format in ("h5", "hdf5", "keras") or filepath.endswith((".h5", ".hdf5", ".keras")):
            raise NotImplementedError("HDF5 / .keras weight files are not supported; use the TensorFlow checkpoint format")
        if not (filepath.endswith(".npz") or save_format == "npz"):
            from debvader_amd.model import tf_checkpoint

            opt = getattr(self, "optimizer", None)
            hyper = None
            if opt is not None:
                hyper = {"learning_rate": opt.learning_rate, "beta_1": opt.beta_1, "beta_2": opt.beta_2, "decay": 0.0}
            tf_checkpoint.save_from_engine(eng, filepath, optimizer=hyper)
            return
        path = filepath if filepath.endswith(".npz") else filepath + ".npz"
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        blob = {"__iter__": np.array(eng.iterations, dtype=np.int64)}
        for i, (name, _, tr) in enumerate(eng.specs):
            blob["p/" + name] = eng.get_param(i)
            if tr:
                blob["m/" + name] = eng.get_slot(i, 0)
                blob["v/" + name] = eng.get_slot(i, 1)
        from debvader_amd.model import tf_checkpoint
        import io

        buf = io.BytesIO()
        np.savez(buf, **blob)
        tf_checkpoint.atomic_write(path, buf.getvalue())
        tf_checkpoint.atomic_write(os.path.join(os.path.dirname(os.path.abspath(path)), "checkpoint"),
                                   f'model_checkpoint_path: "{os.path.basename(path)}"\n'.encode())

    def load_weights(self, filepath):
        """Loads a checkpoint written by save_weights (.npz) or a TensorFlow tensor-bundle checkpoint written by
        the reference (`<prefix>.index` + `.data-*` shards; debvader_amd/model/tf_checkpoint.py)."""
        if filepath is None:
            raise FileNotFoundError("no checkpoint found (latest_checkpoint returned None)")
        if os.path.exists(filepath + ".index"):
            # a TensorFlow tensor-bundle: written by the reference (ModelCheckpoint / net.save_weights) or by save_weights
            from debvader_amd.model import tf_checkpoint

            tf_checkpoint.load_into_engine(self._core.engine, filepath, load_slots=True)
            return self
        path = filepath if filepath.endswith(".npz") else filepath + ".npz"
        if not os.path.exists(path):
            raise FileNotFoundError(f"neither {filepath}.index nor {path} exists")
        eng = self._core.engine
        with np.load(path) as z:
            for i, (name, shape, tr) in enumerate(eng.specs):
                eng.set_param(i, z["p/" + name])
                if tr and ("m/" + name) in z:
                    eng.set_slot(i, 0, z["m/" + name])
                    eng.set_slot(i, 1, z["v/" + name])
            if "__iter__" in z:
                eng.iterations = int(z["__iter__"])
        return self

    # -- training ----------------------------------------------------------------------------
    def _metric_values(self, scal):
        out = {"loss": scal["loss"]}
        for mt in self._metrics:
            if isinstance(mt, str):
                if mt in ("mse", "mean_squared_error"):
                    out["mse"] = scal["mse"]
                else:
                    raise NotImplementedError(f"metric {mt!r} is not implemented")
            elif callable(mt):
                out[getattr(mt, "__name__", "metric")] = float(mt(None, None))
        return out

    def _set_losses(self, scal):
        # net.losses: one activity-regulariser scalar per application; their sum is kl_metric
        k = max(1, int(self._core.cfg.kl_multiplicity))
        self.losses = [scal["kl_reg"] / k] * k

    def fit(self, x=None, y=None, batch_size=None, epochs=1, verbose=1, callbacks=None, shuffle=True,
            validation_data=None, validation_steps=None, initial_epoch=0, reuse_device_data=False, **kwargs):
        """Keras Model.fit for in-memory arrays (train.py:27-37): per-epoch shuffle, last partial batch used,
        validation in inference mode, returns History with loss / mse / <metric fn names> / val_*.

        Several ranks (debvader_amd.parallel): `batch_size` is the GLOBAL batch; rank r runs the r-th contiguous slice
        of every global batch and gradients are all-reduced inside the engine.  Every rank passes the same arrays and
        keeps only ITS rows in HBM: the rows the unshuffled batch sequence gives it (`_home_rows`), 1/world of the set.
        Shuffling then permutes each rank's rows among themselves, so a global batch is the union of one random piece
        per rank - every stamp once per epoch, batches stratified over the ranks' shards instead of drawn from one
        global permutation; with one rank it is Keras' global shuffle, and shuffle=False reproduces the sequential
        batches exactly for any number of ranks.

        reuse_device_data: Keras re-reads the arrays on every call, and so does this method (the upload is PCIe-bound,
        faster than any content hash of the arrays would be).  reuse_device_data=True skips the upload when the same
        array OBJECTS (identity, shape, batch geometry) are still resident from the previous fit() - the caller's
        promise that their contents have not changed in between."""
        if not self._core.compiled:
            raise RuntimeError("You must compile your model before training/testing. Use `model.compile(...)`.")
        core, eng = self._core, self._core.engine
        x_in, y_in = x, y
        x = np.asarray(x)
        y = np.asarray(y)
        n = x.shape[0]
        batch_size = int(batch_size or 32)
        rank, world = core.ctx.rank, core.ctx.world
        from debvader_amd.parallel import shard_range, shard_sizes

        if shard_range(batch_size, 0, world)[1] > eng.max_batch:
            raise ValueError(f"batch_size {batch_size} over {world} rank(s) exceeds the engine's max_batch "
                             f"{eng.max_batch}; pass max_batch= to create_model_vae")
        nv = 0
        xv = yv = None
        if validation_data is not None:
            xv = np.asarray(validation_data[0])
            yv = np.asarray(validation_data[1])
            nv = xv.shape[0]
        # Every batch of the epoch - the ragged last ones included - must give every rank at least one stamp: a rank
        # with an empty shard would skip its collectives while the others have queued theirs.  All ranks see the same
        # sizes, so all of them raise here, before anything is queued.
        val_steps = 0
        if nv:
            val_steps = int(validation_steps) if validation_steps else -(-nv // batch_size)
            val_steps = min(val_steps, -(-nv // batch_size))
        if world > 1:
            ragged = [n % batch_size] if n % batch_size else []
            if n >= batch_size:
                ragged.append(batch_size)
            ragged += [min(batch_size, nv - s * batch_size) for s in range(val_steps)]
            small = [b for b in ragged if min(shard_sizes(b, world)) == 0]
            if small:
                raise ValueError(f"a batch of {small[0]} stamp(s) cannot be split over {world} ranks: choose a batch size "
                                 f"(and data-set sizes modulo the batch size) of at least {world}")
        # this rank's rows of the training set, per step; resident rows are addressed by their position in `home`
        home, pieces = self._home_rows(n, batch_size, rank, world)
        self._upload(0, x_in, x, y_in, y, home, (n, batch_size, rank, world), reuse_device_data)
        vpieces = []
        if nv:
            vhome, vpieces = self._home_rows(min(nv, val_steps * batch_size), batch_size, rank, world)
            self._upload(1, validation_data[0], xv, validation_data[1], yv, vhome, (nv, batch_size, val_steps, rank, world),
                         reuse_device_data)
        hist = History()
        cbs = list(callbacks or [])
        for cb in cbs:
            if hasattr(cb, "set_model"):
                cb.set_model(self)
        # Keras reshuffles from the global generator, so every fit() call sees new permutations; here the seed is
        # (model seed, number of fit calls so far): fresh per call, identical on every rank, reproducible per model
        self._fit_calls = getattr(self, "_fit_calls", 0) + 1
        rng = np.random.default_rng(kwargs["shuffle_seed"] if "shuffle_seed" in kwargs
                                    else [0x5EED, core.shuffle_base, self._fit_calls])
        all_sizes = [self._home_rows(n, batch_size, r, world)[0].size for r in range(world)] if world > 1 else [n]
        self.stop_training = False
        for epoch in range(initial_epoch, epochs):
            t0 = time.time()
            # one permutation per rank from the SHARED generator (its state stays identical on all ranks); this rank
            # applies its own to its resident rows
            order = None
            for r, sz in enumerate(all_sizes):
                perm = rng.permutation(sz) if shuffle else np.arange(sz)
                if r == rank:
                    order = perm
            sums: Dict[str, float] = {}
            seen = 0
            # Steps are queued two ahead of the one whose loss the host is reading (deferred results), so the GPU
            # does not idle while Python slices the next batch; metrics are accumulated in step order as before.
            pending = []
            ticket = 0

            def collect(item):
                nonlocal seen
                tk, count = item
                scal = eng.step_result(tk)
                self._set_losses(scal)
                for k, v in self._metric_values(scal).items():
                    sums[k] = sums.get(k, 0.0) + v * count
                seen += count

            try:
                for (c0, c1, gb) in pieces:
                    eng.train_step_async(ticket, 0, idx=order[c0:c1].astype(np.int32), global_batch=gb,
                                         seed=core.next_seed())
                    pending.append((ticket, gb))
                    ticket = (ticket + 1) % 4
                    if len(pending) > 2:
                        collect(pending.pop(0))
                while pending:
                    collect(pending.pop(0))
            finally:
                # an exception (or Ctrl-C) between queuing and collecting must not leave tickets occupied: the next
                # fit() would otherwise fail with "ticket has an uncollected result" until the model is recreated
                for tk, _ in pending:
                    try:
                        eng.step_result(tk)
                    except Exception:
                        pass
            logs = {k: v / seen for k, v in sums.items()}
            if nv:
                vs: Dict[str, float] = {}
                vseen = 0
                for (c0, c1, gb) in vpieces:
                    scal = eng.eval_step(1, first=c0, B=c1 - c0, global_batch=gb, seed=core.next_seed())
                    self._set_losses(scal)
                    for k, v in self._metric_values(scal).items():
                        vs[k] = vs.get(k, 0.0) + v * gb
                    vseen += gb
                if vseen:
                    logs.update({"val_" + k: v / vseen for k, v in vs.items()})
            hist.epoch.append(epoch)
            for k, v in logs.items():
                hist.history.setdefault(k, []).append(float(v))
            if verbose and rank == 0:
                msg = " - ".join(f"{k}: {v:.4f}" for k, v in logs.items())
                print(f"Epoch {epoch + 1}/{epochs} - {time.time() - t0:.1f}s - {msg}")
            # Callbacks run on every rank (the logs are global, so their decisions - stop_training, "improved" - agree);
            # what they WRITE goes to disk from rank 0 only (VAENet.save_weights), and the ranks meet again behind it.
            for cb in cbs:
                if hasattr(cb, "on_epoch_end"):
                    cb.on_epoch_end(epoch, logs)
            if world > 1 and cbs:
                core.ctx.allreduce([0.0])
            if self.stop_training:
                break
        self.history = hist
        return hist

    @staticmethod
    def _home_rows(n, batch_size, rank, world):
        """Rows of an n-stamp set that `rank` owns, and its slice of every step.  Global batch k of the unshuffled
        sequence is rows [k*batch_size, ...) and rank r takes its r-th contiguous slice (contiguous equal
        shards of the global batch by index); the union over k is what the rank keeps resident - a partition of the
        set over the ranks, sizes equal up to one stamp per step.
        returns (home, pieces): global row numbers in resident order; per step (begin, end, global_batch) with begin /
        end positions in `home`."""
        from debvader_amd.parallel import shard_range

        rows, pieces, c = [], [], 0
        for b0 in range(0, n, batch_size):
            gb = min(batch_size, n - b0)
            lo, hi = shard_range(gb, rank, world)
            rows.append(np.arange(b0 + lo, b0 + hi, dtype=np.int64))
            pieces.append((c, c + hi - lo, gb))
            c += hi - lo
        return (np.concatenate(rows) if rows else np.zeros(0, np.int64)), pieces

    def _upload(self, slot, x_orig, x, y_orig, y, home, geometry, reuse):
        """dv_data_upload of this rank's rows (float32, as Keras casts them in fit).  With reuse=True a later fit() on the
        same array objects and the same batch geometry keeps the resident copy (see fit's reuse_device_data)."""
        key = (id(x_orig), id(y_orig), np.shape(x), np.shape(y), geometry)
        cache = self._core.upload_keys
        if reuse and cache.get(slot) == key:
            return
        cache.pop(slot, None)
        whole = home.size == np.shape(x)[0]                          # one rank: no gather, the cast is the only copy
        xs = np.asarray(x if whole else x[home], dtype=np.float32)
        ys = np.asarray(y if whole else y[home], dtype=np.float32)
        self._core.engine.upload(slot, xs, ys)
        cache[slot] = key

    def evaluate(self, x, y, batch_size=32, verbose=0):
        eng, core = self._core.engine, self._core
        core.upload_keys.pop(1, None)        # slot 1 no longer holds fit()'s validation set (reuse_device_data)
        n = eng.upload(1, x, y)
        tot: Dict[str, float] = {}
        for b0 in range(0, n, batch_size):
            gb = min(batch_size, n - b0)
            scal = eng.eval_step(1, first=b0, B=gb, seed=core.next_seed())
            self._set_losses(scal)
            for k, v in self._metric_values(scal).items():
                tot[k] = tot.get(k, 0.0) + v * gb
        return {k: v / n for k, v in tot.items()}


def _build(input_shape, latent_dim, filters, kernels, for_onnx, max_batch, ctx, cfg_over):
    cfg_over = dict(cfg_over)
    seed = cfg_over.pop("seed", None)
    if isinstance(cfg_over.get("dtype"), str):                 # "float32" / "bf16" as well as the DV_DTYPE_* codes
        names = {"f32": 0, "fp32": 0, "float32": 0, "bf16": 1, "bfloat16": 1}
        if cfg_over["dtype"].lower() not in names:
            raise ValueError(f"dtype {cfg_over['dtype']!r}: the engine stores activations as float32 or bf16")
        cfg_over["dtype"] = names[cfg_over["dtype"].lower()]
    if for_onnx:
        raise NotImplementedError("for_onnx=True builds a TF graph for tf2onnx export (model.py:151-152,203-204); "
                                  "ONNX export is outside this engine's scope")
    core = _Core(input_shape, latent_dim, filters, kernels, max_batch, ctx, seed=seed, **cfg_over)
    enc, dec = Encoder(core, "encoder"), Decoder(core, "decoder")
    if core.cfg.height != int(np.ceil(core.cfg.height / 2 ** len(filters))) * 2 ** len(filters):
        print("in cropping")                                  # model.py:142
    return core, enc, dec


def create_encoder(input_shape, latent_dim, filters, kernels, conv_activation=None, dense_activation=None,
                   max_batch=256, ctx=None, **cfg_over):
    """Encoder-only model (model.py:61-100).  Activations are ignored, as in the reference."""
    core, enc, _ = _build(input_shape, latent_dim, filters, kernels, False, max_batch, ctx, cfg_over)
    return enc


def create_decoder(input_shape, latent_dim, filters, kernels, conv_activation=None, dense_activation=None,
                   for_onnx=False, max_batch=256, ctx=None, **cfg_over):
    """Decoder-only model (model.py:103-161)."""
    core, _, dec = _build(input_shape, latent_dim, filters, kernels, for_onnx, max_batch, ctx, cfg_over)
    return dec


def create_model_vae(input_shape, latent_dim, filters, kernels, conv_activation=None, dense_activation=None,
                     for_onnx=False, max_batch=256, ctx=None, **cfg_over):
    """Create the VAE model (model.py:164-218).

    parameters:
        input_shape: shape of input tensor
        latent_dim: size of the latent space
        filters: filters used for the convolutional layers
        kernels: kernels used for the convolutional layers
        conv_activation, dense_activation: accepted and ignored (the reference passes None down, model.py:187-197)
        max_batch: per-GPU stamps per step the engine allocates workspaces for (engine-specific)
        seed: initial-weight (and shuffle) seed; None draws one from the OS, like Keras' fresh initialisation per model
              (net.init_seed holds the value used)
        dtype (keyword): "float32" (default) or "bf16": bf16 activations / MFMA operands (BASELINE configs[2])
        ctx: debvader_amd.engine.Context (GPU / rank); default: GPU 0, single rank
    returns (net, encoder, decoder, z)
    """
    core, enc, dec = _build(input_shape, latent_dim, filters, kernels, for_onnx, max_batch, ctx, cfg_over)
    net = VAENet(core, enc, dec)
    return net, enc, dec, LatentModel(core)


def latest_checkpoint(directory):
    """tf.train.latest_checkpoint: reads `<dir>/checkpoint` (same text format as TensorFlow's)."""
    f = os.path.join(directory, "checkpoint")
    if not os.path.exists(f):
        return None
    with open(f) as fh:
        for line in fh:
            if line.startswith("model_checkpoint_path:"):
                return os.path.join(directory, line.split(":", 1)[1].strip().strip('"'))
    return None


def weights_dir(survey):
    """data/weights/<survey> (model.py:262-263).  DEBVADER_WEIGHTS may point at another `weights` directory,
    e.g. the reference package's src/debvader/data/weights with its TensorFlow checkpoints."""
    root = os.environ.get("DEBVADER_WEIGHTS") or os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "weights")
    return os.path.join(root, str(survey))


def load_deblender(survey, input_shape, latent_dim, filters, kernels, return_encoder_decoder_z=False,
                   for_onnx=False, max_batch=256, ctx=None):
    """load weights trained for a particular dataset (model.py:221-271).

    The reference ships only `dc2`, and its tensor shard is missing from the repository
    (.MISSING_LARGE_BLOBS); this function loads checkpoints written by this engine under
    debvader_amd/data/weights/<survey>/ (see VAENet.save_weights).
    """
    net, encoder, decoder, z = create_model_vae(input_shape, latent_dim, filters, kernels, for_onnx=for_onnx,
                                                max_batch=max_batch, ctx=ctx)
    decoder.trainable = False                                       # model.py:252
    net.compile(optimizer=Adam(learning_rate=1e-4), loss=vae_loss)   # model.py:255-259
    loading_path = weights_dir(survey)
    print(loading_path)
    latest = latest_checkpoint(loading_path)
    if latest is None:
        raise FileNotFoundError(f"no checkpoint under {loading_path}")
    net.load_weights(latest)
    if return_encoder_decoder_z:
        return net, encoder, decoder, z
    return net
