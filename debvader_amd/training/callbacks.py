"""ModelCheckpoint stand-in for the two callbacks define_callbacks builds (reference train.py:54-71)."""
import numpy as np


class ModelCheckpoint:
    def __init__(self, filepath, monitor="val_loss", verbose=0, save_best_only=False, save_weights_only=True,
                 mode="min", save_freq="epoch"):
        if mode not in ("min", "max"):
            raise ValueError("mode must be 'min' or 'max'")
        self.filepath, self.monitor, self.verbose = filepath, monitor, verbose
        self.save_best_only, self.mode = save_best_only, mode
        self.best = np.inf if mode == "min" else -np.inf
        self.model = None

    def set_model(self, model):
        self.model = model

    def on_epoch_end(self, epoch, logs=None):
        logs = logs or {}
        cur = logs.get(self.monitor)
        if self.save_best_only:
            if cur is None:
                return
            better = cur < self.best if self.mode == "min" else cur > self.best
            if not better:
                return
            if self.verbose:
                print(f"\nEpoch {epoch + 1}: {self.monitor} improved from {self.best:.5f} to {cur:.5f}, "
                      f"saving model to {self.filepath}")
            self.best = cur
        self.model.save_weights(self.filepath)
