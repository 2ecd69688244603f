"""Training entry points of the drop-in API, driven by the HIP engine.

API parity with the reference's training module (`/root/reference/src/debvader/training/train.py`):
`train_network` (:11-39), `define_callbacks` (:42-75) and `train_deblender` (:78-205) keep their names, positional
arguments, keyword defaults and return values.  Everything below them is different: `net.fit` queues fused HIP train
steps, the checkpoints are written by `debvader_amd.model.tf_checkpoint`, and the two training stages share their code
(`_compile`, `_fit_stage`).
"""
import os

import numpy as np

from debvader_amd.model import model
from debvader_amd.training.callbacks import ModelCheckpoint
from debvader_amd.training.metrics import vae_loss

# architecture of the reference's deblender (train.py:104-107): the only one it ever trains
_LATENT = 32
_FILTERS = (32, 64, 128, 256)
_KERNELS = (3, 3, 3, 3)
_STAMP = 59
_LEARNING_RATE = 1e-4            # legacy Adam, train.py:126
_BAND_MESSAGE = ("The number of bands in the data does not correspond to the number of filters in the network. "
                 "Correct this before starting again.")


def train_network(net, epochs, training_data, validation_data, batch_size, callbacks=None, verbose=1):
    """Fit `net` for `epochs` epochs and return the History.

    `training_data` / `validation_data` are (inputs, labels) pairs - tuples or arrays of shape (2, N, H, W, bands).
    Batches are reshuffled every epoch; validation runs on whole batches only, as the reference asks Keras to
    (`validation_steps = len(validation inputs) // batch_size`).
    """
    x, y = training_data[0], training_data[1]
    xv, yv = validation_data[0], validation_data[1]
    whole_validation_batches = len(xv) // batch_size
    print("\nStart the training")
    return net.fit(x, y, batch_size=batch_size, epochs=epochs, shuffle=True, verbose=verbose, callbacks=callbacks,
                   validation_data=(xv, yv), validation_steps=whole_validation_batches)


def define_callbacks(vae_or_deblender, survey_name):
    """Two best-only weight checkpoints under `<weights dir>/<survey>/<vae|deblender>/{val_mse,val_loss}/`, written
    in TensorFlow checkpoint format with the reference's file name (so its `load_weights` finds them)."""
    root = os.path.join(model.weights_dir(survey_name), str(vae_or_deblender))
    return [
        ModelCheckpoint(filepath=os.path.join(root, monitored, "weights_noisy_v4.ckpt"), monitor=monitored,
                        mode="min", save_best_only=True, save_weights_only=True, save_freq="epoch", verbose=1)
        for monitored in ("val_mse", "val_loss")
    ]


def _check_bands(stamps, nb_of_bands, channel_last):
    """The reference's data-format guard (train.py:132-142).  Its first test parses as
    `not (channel_last & (shape[2] != nb))`, which never fires for channel-last data of the notebook's
    (2, N, 59, 59, bands) layout and always fires for channel_last=False; the second compares the last axis.
    Same outcomes here: channel-first data is refused, channel-last data must end in `nb_of_bands`."""
    if channel_last and np.asarray(stamps).shape[-1] == nb_of_bands:
        return
    print(_BAND_MESSAGE)
    raise ValueError(_BAND_MESSAGE)


def _compile(net, kl_metric):
    """Fresh legacy-Adam state, current `trainable` flags, the reference's loss and metrics (train.py:125-130)."""
    net.compile(optimizer=model.Adam(learning_rate=_LEARNING_RATE), loss=vae_loss, metrics=["mse", kl_metric],
                experimental_run_tf_function=False)


def _fit_stage(net, label, folder, survey_name, epochs, train, valid, batch_size, with_callbacks, verbose):
    cbs = define_callbacks(folder, survey_name) if with_callbacks else None
    history = train_network(net, epochs, train, valid, batch_size, cbs, verbose)
    print(f"\nTraining of {label} done.")
    return history


def train_deblender(survey_name, from_survey, epochs, training_data_vae, validation_data_vae,
                    training_data_deblender, validation_data_deblender, nb_of_bands=6, channel_last=True,
                    batch_size=5, with_callbacks=False, verbose=2, max_batch=None, ctx=None):
    """Train a deblender for `survey_name` in the reference's two stages and return
    `(history of stage 1, history of stage 2, net)`.

    Stage 1 trains the whole VAE on isolated galaxies (`*_data_vae`); stage 2 freezes the decoder, starts Adam
    afresh and trains the encoder on blended scenes (`*_data_deblender`).  `from_survey` names a survey whose latest
    checkpoint initialises the weights (None: Keras-default initialisation).  `with_callbacks` saves best-only
    checkpoints.  `max_batch` (workspace capacity per GPU, default `batch_size`) and `ctx` (GPU / rank context) are
    engine-specific additions.
    """
    net, encoder, decoder, z = model.create_model_vae(
        (_STAMP, _STAMP, nb_of_bands), _LATENT, list(_FILTERS), list(_KERNELS), conv_activation=None,
        dense_activation=None, max_batch=max_batch or max(int(batch_size), 1), ctx=ctx)

    def kl_metric(y_true, y_pred):
        # the KL regulariser's activity losses of the last step, as the reference displays them (train.py:121-122)
        return sum(net.losses)

    print("VAE model")
    net.summary()
    _compile(net, kl_metric)            # before the weights are loaded, as in the reference: a checkpoint's Adam slots
    _check_bands(training_data_vae[0], nb_of_bands, channel_last)           # and step counter survive the compile
    if from_survey is not None:
        start_dir = model.weights_dir(from_survey)
        print(start_dir)
        net.load_weights(model.latest_checkpoint(start_dir))
    hist_vae = _fit_stage(net, "VAE", "vae", survey_name, epochs, training_data_vae, validation_data_vae, batch_size,
                          with_callbacks, verbose)

    decoder.trainable = False           # train.py:175-183: freeze, re-compile (fresh Adam), train the encoder only
    _compile(net, kl_metric)
    print("\n\nDeblender model")
    net.summary()
    hist_deblender = _fit_stage(net, "Deblender", "deblender", survey_name, epochs, training_data_deblender,
                                validation_data_deblender, batch_size, with_callbacks, verbose)
    return hist_vae, hist_deblender, net
