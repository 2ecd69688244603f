"""Throughput of the drop-in `net.fit` loop (reference train.py:27-37) against the queued engine steps bench.py times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd.model import model
from debvader_amd.training.metrics import vae_loss
from debvader_amd.data import synthetic_stamps

B, NB = 256, 40
x, y = synthetic_stamps(1024, seed=1)
x = np.tile(x, (B * NB // 1024, 1, 1, 1)); y = np.tile(y, (B * NB // 1024, 1, 1, 1))
net, _, _, _ = model.create_model_vae((59, 59, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=B)
net.compile(optimizer=model.Adam(learning_rate=1e-4), loss=vae_loss, metrics=["mse"])
net.fit(x[:4 * B], y[:4 * B], epochs=1, batch_size=B, verbose=0)          # warm-up (also uploads nothing persistent)
t0 = time.perf_counter(); h = net.fit(x, y, epochs=1, batch_size=B, verbose=0); dt = time.perf_counter() - t0
print(f"net.fit: {NB} steps of {B} in {dt*1e3:.1f} ms = {B*NB/dt:.0f} stamps/s (includes the one-off upload of {x.nbytes*2/1e9:.2f} GB); loss {h.history['loss'][-1]:.4f}")
t0 = time.perf_counter(); h = net.fit(x, y, epochs=2, batch_size=B, verbose=0); dt = time.perf_counter() - t0
print(f"net.fit 2 epochs: {2*B*NB/dt:.0f} stamps/s")
t0 = time.perf_counter(); h = net.fit(x, y, epochs=8, batch_size=B, verbose=0); dt = time.perf_counter() - t0
print(f"net.fit 8 epochs: {8*B*NB/dt:.0f} stamps/s")
