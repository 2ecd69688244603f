"""Two-stage training driver on the MI355X engine.

Drop-in for src/debvader/training/train.py:11-205 of the reference (same names, arguments and
return values).  `net.fit` runs the fused HIP train step instead of a Keras train_function.
"""
import os

from debvader_amd.model import model
from debvader_amd.training.callbacks import ModelCheckpoint
from debvader_amd.training.metrics import vae_loss


def train_network(net, epochs, training_data, validation_data, batch_size, callbacks=None, verbose=1):
    """
    train a network on data for a fixed number of epochs (train.py:11-39)
    parameters:
        net: network to train
        epochs: number of epochs
        training_data: training data under the format of numpy arrays (inputs, labels)
        validation_data: validation data under the format of numpy arrays (inputs, labels)
        batch_size: size of batch for training
        callbacks: callbacks wanted for the training
        verbose: display of training (1:yes, 2: no)
    """
    print("\nStart the training")
    hist = net.fit(
        training_data[0],
        training_data[1],
        epochs=epochs,
        batch_size=batch_size,
        verbose=verbose,
        shuffle=True,
        validation_data=(validation_data[0], validation_data[1]),
        validation_steps=int(len(validation_data[0]) / batch_size),
        callbacks=callbacks,
    )
    return hist


def define_callbacks(vae_or_deblender, survey_name):
    """
    Define callbacks for a network to train (train.py:42-75): two best-only weight checkpoints,
    one monitoring val_mse and one monitoring val_loss.
    """
    saving_path = os.path.join(model.weights_dir(survey_name), str(vae_or_deblender), "")
    checkpointer_val_mse = ModelCheckpoint(
        filepath=saving_path + "val_mse/weights_noisy_v4.ckpt", monitor="val_mse", verbose=1,
        save_best_only=True, save_weights_only=True, mode="min", save_freq="epoch")
    checkpointer_val_loss = ModelCheckpoint(
        filepath=saving_path + "val_loss/weights_noisy_v4.ckpt", monitor="val_loss", verbose=1,
        save_best_only=True, save_weights_only=True, mode="min", save_freq="epoch")
    return [checkpointer_val_mse, checkpointer_val_loss]


def train_deblender(survey_name, from_survey, epochs, training_data_vae, validation_data_vae,
                    training_data_deblender, validation_data_deblender, nb_of_bands=6, channel_last=True,
                    batch_size=5, with_callbacks=False, verbose=2, max_batch=None, ctx=None):
    """
    function to train a network for a new survey (train.py:78-205)
    survey_name: name of the survey
    from_survey: survey whose saved weights initialise the network (None: Keras-default initialisation)
    epochs: number of epochs of training
    training_data_{}: numpy arrays (inputs, labels) for the vae or the deblender
    validation_data_{}: numpy arrays (inputs, labels) for the vae or the deblender
    batch_size: size of batch for training
    with_callbacks: save best-only checkpoints during training
    verbose: display of training (1:yes, 2: no)
    max_batch, ctx: engine-specific (workspace capacity per GPU; GPU/rank context)
    returns (hist_vae, hist_deblender, net)
    """
    # The architecture is fixed (train.py:104-107).
    input_shape = (59, 59, nb_of_bands)
    latent_dim = 32
    filters = [32, 64, 128, 256]
    kernels = [3, 3, 3, 3]

    net, encoder, decoder, z = model.create_model_vae(
        input_shape, latent_dim, filters, kernels, conv_activation=None, dense_activation=None,
        max_batch=max_batch or max(int(batch_size), 1), ctx=ctx)
    print("VAE model")
    net.summary()

    # Custom metric to display the KL divergence during training (train.py:121-122)
    def kl_metric(y_true, y_pred):
        return sum(net.losses)

    net.compile(optimizer=model.Adam(learning_rate=1e-4), loss=vae_loss, metrics=["mse", kl_metric],
                experimental_run_tf_function=False)

    # Data-format check (train.py:132-142).  With the notebook's (2, N, 59, 59, bands) arrays the reference's
    # first test, `not channel_last & (shape[2] != nb)`, parses as not(channel_last & ...): it never fires for
    # channel-last stamps and always fires for channel_last=False; the second test compares the last axis.
    import numpy as np

    last_axis = np.asarray(training_data_vae[0]).shape[-1]
    if not channel_last:
        print("The number of bands in the data does not correspond to the number of filters in the network. "
              "Correct this before starting again.")
        raise ValueError
    if channel_last and last_axis != nb_of_bands:
        print("The number of bands in the data does not correspond to the number of filters in the network. "
              "Correct this before starting again.")
        raise ValueError

    if from_survey is not None:
        path_output = model.weights_dir(from_survey)
        print(path_output)
        latest = model.latest_checkpoint(path_output)
        net.load_weights(latest)

    callbacks = define_callbacks("vae", survey_name) if with_callbacks else None
    hist_vae = train_network(net, epochs, training_data_vae, validation_data_vae, batch_size, callbacks, verbose)
    print("\nTraining of VAE done.")

    # Set the decoder as non-trainable (train.py:175) and re-compile: fresh Adam state, encoder-only updates
    decoder.trainable = False
    net.compile(optimizer=model.Adam(learning_rate=1e-4), loss=vae_loss, metrics=["mse", kl_metric],
                experimental_run_tf_function=False)
    print("\n\nDeblender model")
    net.summary()

    callbacks = define_callbacks("deblender", survey_name) if with_callbacks else None
    hist_deblender = train_network(net, epochs, training_data_deblender, validation_data_deblender, batch_size,
                                   callbacks, verbose)
    print("\nTraining of Deblender done.")

    return hist_vae, hist_deblender, net
