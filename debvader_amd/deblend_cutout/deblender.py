"""Batched inference entry point (reference: src/debvader/deblend_cutout/deblender.py:6-24)."""
import numpy as np


def deblend(net, images, normalise=False):
    """Deblend stamps with the network.

    parameters:
        net: network returned by create_model_vae / load_deblender
        images: array (N, size, size, bands), any float dtype (cast to float32 as the reference does)
        normalise: apply tanh(arcsinh(.)) to the inputs and undo it on the predicted mean.
            The reference's normalise=True branch cannot run (it applies numpy arctanh to a
            distribution object, deblender.py:20-24); the evident intent is implemented instead.
    returns (mean ndarray (N,size,size,bands), distribution)
    """
    images = np.asarray(images)
    if not normalise:
        out = net(images)  # one stochastic forward pass, BN in inference mode; the float32 cast happens in the engine
        return out.mean().numpy(), out
    eng = net._core.engine
    eng.set_normalise(True)                        # both transforms run on the GPU around the forward pass
    try:
        out = net(images)
    finally:
        eng.set_normalise(False)
    return out.mean().numpy(), out                 # mean in flux units, stddev in normalised units

def deblend_field_cutouts(net, field, starts, normalise=False, on_chunk=None):
    """deblend(net, cutouts) for cutouts[i] = field[starts[i,0]:+size, starts[i,1]:+size, :] without materialising the
    cutouts on the host: what DeblendField.deblend_field does with extract_cutouts followed by deblend
    (deblend/field_deblender.py:260-274, extract/extraction.py:4-43, deblender.py:18) as ONE engine call - the float64
    field goes to the GPU once, every chunk's cutouts are gathered and cast to float32 there, only mean and stddev come
    back.  Same numbers, bit for bit, as deblend(net, field cutouts) with the same noise seed.

    parameters:
        field: (F, F, bands) float array;  starts: (N, 2) integer top-left corners, windows inside the field
        on_chunk: None, or a function (first, mean, stddev) that consumes the results chunk by chunk (max_batch stamps,
            float32 views of the transfer buffers, valid during the call) instead of two N-stamp arrays - BASELINE
            configs[4]'s million cutouts are 167 GB of mean and stddev; then the function returns None
    returns (mean ndarray (N,size,size,bands), distribution)
    """
    from debvader_amd.distributions import Normal

    eng, core = net._core.engine, net._core
    eng.set_normalise(bool(normalise))
    try:
        if on_chunk is not None:
            eng.infer_cutouts_stream(field, starts, on_chunk, seed=core.next_seed())
            return None
        r = eng.infer_cutouts(field, starts, seed=core.next_seed(), want=("loc", "scale"))
    finally:
        eng.set_normalise(False)
    out = Normal(r["loc"], r["scale"])
    return out.mean().numpy(), out


def deblend_sharded(net, images, normalise=False, dist=None, gather=True, rank=None, world=None):
    """deblend() over the GPUs of one node (BASELINE configs[4]): the reference calls the network ONCE on all N stamps
    (deblender.py:18, from field_deblender.py:265-274); stamps are independent, so rank r runs deblend() on the
    contiguous index range parallel.shard_range(N, r, world) of `images` and no collective touches the data path.

    parameters:
        net, images, normalise: as deblend().  Every rank passes the same `images` (or at least its own range of it).
        dist: how the pieces reach rank 0 - a parallel.HostGroup (default: the group of the network's context, i.e. the
              torch-free rendezvous the ranks already met in) or a torch.distributed-like object with
              gather_object(obj, object_gather_list, dst) (gloo)
        gather: True - rank 0 returns the full (mean, stddev) arrays in input order, other ranks (None, None);
                False - every rank returns its own shard (mean, stddev) and (begin, end); nothing is exchanged
                (the 1M-cutout case: 83 KB of output per stamp does not belong on one host)
        rank, world: default to the context the network was created on
    """
    from debvader_amd.parallel import shard_range

    ctx = getattr(getattr(net, "_core", None), "ctx", None)
    rank = ctx.rank if rank is None else rank
    world = ctx.world if world is None else world
    images = np.asarray(images)
    lo, hi = shard_range(images.shape[0], rank, world)
    if hi > lo:
        mean, out = deblend(net, images[lo:hi], normalise)
        std = out.stddev().numpy()
    else:                                              # fewer stamps than ranks
        mean = np.empty((0,) + images.shape[1:], np.float32)
        std = np.empty((0,) + images.shape[1:], np.float32)
    if not gather:
        return mean, std, (lo, hi)
    if world == 1:
        return mean, std
    if dist is None:
        dist = getattr(ctx, "group", None)
    if dist is None:
        raise ValueError("gather=True with more than one rank needs the context's HostGroup or a torch.distributed-like "
                         "`dist` (gather_object)")
    if hasattr(dist, "allgather") and not hasattr(dist, "get_backend"):
        pieces = dist.gather_object((lo, hi, mean, std), dst=0)          # parallel.HostGroup: returns the list on dst
    else:
        pieces = [None] * world if rank == 0 else None
        dist.gather_object((lo, hi, mean, std), pieces, dst=0)           # torch.distributed form: fills the list
    if rank != 0:
        return None, None
    full_m = np.empty((images.shape[0],) + mean.shape[1:], np.float32)
    full_s = np.empty_like(full_m)
    for b, e, m_, s_ in pieces:
        full_m[b:e] = m_
        full_s[b:e] = s_
    return full_m, full_s


def deblend_epistemic(net, images, n_samples=100, normalise=False):
    """Epistemic-uncertainty estimate of the reference's field deblender (deblend/field_deblender.py:303-313:
    `np.std(deblend(net, [stamp] * 100)[0], axis=0)` in a Python loop over objects) as one engine call:
    every stamp is encoded once and decoded `n_samples` times with fresh latent samples on the GPU.

    returns (mean over samples of the predicted mean, std over samples), each (N, size, size, bands)
    """
    images = np.asarray(images)
    eng = net._core.engine
    eng.set_normalise(bool(normalise))
    try:
        mean, std = eng.infer_mc(images.astype(np.float32), n_samples, seed=net._core.next_seed())
    finally:
        eng.set_normalise(False)
    return mean, std
