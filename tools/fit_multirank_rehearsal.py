"""Rehearsal of a multi-rank `net.fit` with the REAL engine on a one-GPU box (see parallel.make_context:
DV_DEBUG_SAME_GPU=1 DV_DEBUG_FAKE_PEERS=1): the global batch is split over the ranks, every rank keeps only its rows
resident, callbacks / checkpoints run on rank 0, validation runs every epoch.  Gradients are NOT summed across the fake
peers, so the ranks' weights drift apart - the point is that the whole multi-rank host path runs and ends cleanly.
  DEBVADER_AMD_LIB=$PWD/debvader_amd/lib/libdebvader_hip_debug.so DV_DEBUG_SAME_GPU=1 DV_DEBUG_FAKE_PEERS=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \\
      --master-addr 127.0.0.1 --master-port 29521 tools/fit_multirank_rehearsal.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from debvader_amd import parallel
from debvader_amd.model import model
from debvader_amd.training.metrics import vae_loss
from debvader_amd.training.callbacks import ModelCheckpoint
from debvader_amd.data import synthetic_stamps

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
ctx = parallel.make_context(rank, world)
B = 128                                   # global batch; 64 per rank with two ranks
x, y = synthetic_stamps(4 * B, seed=3)
xv, yv = synthetic_stamps(B, seed=4)
net, _, _, _ = model.create_model_vae((59, 59, 6), 32, [32, 64, 128, 256], [3, 3, 3, 3], max_batch=B // world, ctx=ctx)
net.compile(optimizer=model.Adam(learning_rate=1e-4), loss=vae_loss, metrics=["mse"])
# inference sharded by contiguous index ranges (configs[4]), gathered on rank 0 through the rendezvous group (each rank
# draws its own latent noise - the Philox stream id is the rank - so the pieces are not comparable bit for bit with a
# one-rank call; shapes, finiteness and the order of the pieces are)
from debvader_amd.deblend_cutout.deblender import deblend_sharded
ms, ss = deblend_sharded(net, xv)                    # dist defaults to the context's HostGroup
if rank == 0:
    good = ms.shape == xv.shape and ss.shape == xv.shape and np.isfinite(ms).all() and (ss > 0).all()
    print("deblend_sharded over", world, "ranks gathered on rank 0:", ms.shape, "finite:", bool(good), flush=True)
    if not good:
        sys.exit(1)
else:
    assert ms is None and ss is None
tmp = tempfile.mkdtemp() if rank == 0 else None
cbs = [ModelCheckpoint(os.path.join(tmp or "/nonexistent", "w", "weights"), save_weights_only=True)]   # (only rank 0 writes)
h = net.fit(x, y, epochs=2, batch_size=B, validation_data=(xv, yv), callbacks=cbs, verbose=0)
if rank == 0:
    files = sorted(os.listdir(os.path.join(tmp, "w")))
    ok = all(np.isfinite(v).all() for v in h.history.values()) and any(f.endswith(".index") for f in files)
    print("fit rehearsal:", {k: [round(float(v), 4) for v in vs] for k, vs in h.history.items()}, "checkpoint files:", files,
          "OK" if ok else "FAILED", flush=True)
    if not ok:
        sys.exit(1)
