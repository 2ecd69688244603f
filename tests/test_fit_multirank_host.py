"""Host logic of net.fit() with several ranks (the global batch is split into contiguous per-rank shards, NLL
mean / KL use the GLOBAL batch, no data-path collective on the host), on CPU with a recording stand-in for the engine:
what every rank queues must partition every global batch; every rank uploads ONLY its rows (1/world of the set) and
addresses them by resident position; a batch some rank would get no stamp of is refused on EVERY rank before anything is
queued (a rank that skipped its step would leave the others waiting in their all-reduce); checkpoints are written by
rank 0 only; the shuffle is fresh per fit() call; the arrays are re-read on every fit() (Keras semantics) unless the caller
opts into reuse_device_data, and evaluate() never leaves fit() a stale validation set."""
import types

import numpy as np
import pytest

from debvader_amd.model import model as M


class _Ctx:
    def __init__(self, rank, world):
        self.rank, self.world, self.barriers = rank, world, 0

    def allreduce(self, v):
        self.barriers += 1
        return list(v)


class _Engine:
    def __init__(self, max_batch):
        self.max_batch = max_batch
        self.train, self.evals, self.uploads, self.saved = [], [], 0, 0
        self.tickets = {}
        self.specs = []
        self.resident = {}

    def upload(self, slot, x, y):
        self.uploads += 1
        x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)        # (the real Engine.upload casts like this)
        assert x.shape == y.shape
        self.resident[slot] = (np.array(x), np.array(y))
        return x.shape[0]

    def rows(self, slot, idx):
        """global row numbers of resident positions `idx` (the test data encode their row number, see _xy)"""
        return (self.resident[slot][0][np.asarray(idx)][:, 0, 0, 0] // 4).astype(np.int64)

    def train_step_async(self, ticket, slot, idx=None, first=0, B=None, global_batch=None, seed=0):
        assert ticket not in self.tickets, "ticket reused before its result was collected"
        self.tickets[ticket] = True
        self.train.append((np.array(idx), int(global_batch)))

    def step_result(self, ticket):
        del self.tickets[ticket]
        return {"loss": 1.0, "nll_mean": 0.9, "kl_reg": 0.1, "mse": 0.5}

    def eval_step(self, slot, idx=None, first=0, B=None, global_batch=None, eps=None, seed=0):
        assert B > 0, "a rank was asked to evaluate an empty shard"
        self.evals.append((int(first), int(B), int(global_batch or B)))
        return {"loss": 2.0, "nll_mean": 1.9, "kl_reg": 0.1, "mse": 0.7}


def _net(rank, world, max_batch=64):
    core = types.SimpleNamespace(
        engine=_Engine(max_batch), ctx=_Ctx(rank, world), compiled=True, shuffle_base=1234, upload_keys={},
        init_seed=7, seed_counter=100 * (rank + 1), cfg=types.SimpleNamespace(kl_multiplicity=2))
    core.next_seed = lambda: 1
    net = M.VAENet(core, M.Encoder(core, "encoder"), M.Decoder(core, "decoder"))
    net._metrics = ["mse"]
    return net, core


def _xy(n):
    x = np.arange(n * 4, dtype=np.float32).reshape(n, 2, 2, 1)
    return x, x + 1


def test_per_rank_shards_partition_every_global_batch_and_validation_step():
    n, nv, batch, world = 52, 11, 16, 3
    x, y = _xy(n)
    xv, yv = _xy(nv)
    per_rank = []
    for r in range(world):
        net, core = _net(r, world)
        hist = net.fit(x, y, batch_size=batch, epochs=2, verbose=0, shuffle=True, validation_data=(xv, yv))
        assert sorted(hist.history) == ["loss", "mse", "val_loss", "val_mse"] and len(hist.history["loss"]) == 2
        assert not core.engine.tickets                              # every queued step was collected
        per_rank.append(core.engine)
    steps = len(per_rank[0].train)
    assert steps == 2 * 4                                            # 52 / 16: three full batches and one of 4, per epoch
    for s in range(steps):
        gb = per_rank[0].train[s][1]
        assert all(e.train[s][1] == gb for e in per_rank)            # same global batch on every rank
        merged = np.concatenate([e.rows(0, e.train[s][0]) for e in per_rank])
        assert merged.size == gb and np.unique(merged).size == gb    # disjoint, complete
        sizes = [e.train[s][0].size for e in per_rank]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    # every rank keeps only its rows resident: a partition of the set, 1/world each (up to one stamp per step)
    res = [np.sort(e.rows(0, np.arange(e.resident[0][0].shape[0]))) for e in per_rank]
    assert sorted(np.concatenate(res).tolist()) == list(range(n))
    assert [r.size for r in res] == [20, 16, 16]                     # 3 x (6, 5, 5) + (2, 1, 1)
    assert np.array_equal(per_rank[0].resident[0][1], per_rank[0].resident[0][0] + 1)     # labels travel with their rows
    # each epoch is a permutation of the data set, and the two epochs differ
    for ep in range(2):
        seen = np.concatenate([np.concatenate([e.rows(0, e.train[ep * 4 + s][0]) for e in per_rank]) for s in range(4)])
        assert sorted(seen.tolist()) == list(range(n))
    assert not np.array_equal(per_rank[0].train[0][0], per_rank[0].train[4][0])
    # validation: one step of 11 stamps, contiguous shards 4 + 4 + 3, each rank holding only its own
    v = [e.evals[0] for e in per_rank]
    assert [b for _, b, _ in v] == [4, 4, 3] and [f for f, _, _ in v] == [0, 0, 0] and all(g == 11 for _, _, g in v)
    assert [e.rows(1, np.arange(b)).tolist() for e, (_, b, _) in zip(per_rank, v)] == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10]]


def test_unshuffled_multi_rank_batches_are_the_sequential_keras_batches():
    n, batch, world = 21, 8, 2
    x, y = _xy(n)
    engines = []
    for r in range(world):
        net, core = _net(r, world)
        net.fit(x, y, batch_size=batch, epochs=1, verbose=0, shuffle=False)
        engines.append(core.engine)
    for s, b0 in enumerate(range(0, n, batch)):
        merged = np.concatenate([e.rows(0, e.train[s][0]) for e in engines])
        assert merged.tolist() == list(range(b0, min(n, b0 + batch)))   # rank order = index order inside the batch


def test_a_batch_that_leaves_a_rank_empty_is_refused_on_every_rank_before_anything_is_queued():
    x, y = _xy(33)                                                   # 33 % 16 = 1 stamp in the last batch
    for r in range(4):
        net, core = _net(r, 4)
        with pytest.raises(ValueError, match="cannot be split over 4 ranks"):
            net.fit(x, y, batch_size=16, verbose=0)
        assert core.engine.train == [] and core.engine.uploads == 0
    xv, yv = _xy(18)                                                 # validation: 16 + 2 stamps over 4 ranks
    for r in range(4):
        net, core = _net(r, 4)
        with pytest.raises(ValueError, match="cannot be split"):
            net.fit(*_xy(32), batch_size=16, verbose=0, validation_data=(xv, yv))
        assert core.engine.train == []
    net, core = _net(0, 1)                                           # a single rank takes any ragged batch
    net.fit(x, y, batch_size=16, verbose=0)
    assert [g for _, g in core.engine.train] == [16, 16, 1]


def test_fresh_shuffle_per_fit_call_cached_upload_and_rank0_only_checkpoints(tmp_path, monkeypatch):
    x, y = _xy(24)
    net, core = _net(0, 2)
    net.fit(x, y, batch_size=8, verbose=0)
    first = [i.copy() for i, _ in core.engine.train]
    core.engine.train.clear()
    net.fit(x, y, batch_size=8, verbose=0)
    assert core.engine.uploads == 2                                  # Keras re-reads the arrays on every fit()
    assert any(not np.array_equal(a, b) for a, (b, _) in zip(first, core.engine.train))
    net.fit(x, y, batch_size=8, verbose=0, reuse_device_data=True)   # the caller vouches for the resident copy:
    net.fit(x, y, batch_size=8, verbose=0, reuse_device_data=True)   # same objects, same geometry -> no upload
    assert core.engine.uploads == 2
    net.fit(x, y, batch_size=6, verbose=0, reuse_device_data=True)   # another batch size = other rows per rank
    assert core.engine.uploads == 3
    net.fit(x.copy(), y, batch_size=6, verbose=0, reuse_device_data=True)   # another array object
    assert core.engine.uploads == 4
    core.engine.train.clear()
    net2, core2 = _net(0, 2)
    net2.fit(x, y, batch_size=8, verbose=0)
    other, ocore = _net(1, 2)                                        # rank 1 of the same model: the other half of each batch
    other.fit(x, y, batch_size=8, verbose=0)
    merged = np.concatenate([core2.engine.rows(0, core2.engine.train[0][0]), ocore.engine.rows(0, ocore.engine.train[0][0])])
    assert np.unique(merged).size == 8
    # checkpoints: only rank 0 touches the disk, every rank passes the rendezvous behind the callbacks
    writes = []
    monkeypatch.setattr(M.VAENet, "get_weights", lambda self: writes.append(self._core.ctx.rank) or [])

    class Saver:
        def set_model(self, m):
            self.m = m

        def on_epoch_end(self, epoch, logs):
            self.m.save_weights(str(tmp_path / f"r{self.m._core.ctx.rank}" / "w.npz"))

    for net_, core_ in ((net, core), (other, ocore)):
        before = core_.ctx.barriers
        try:
            net_.fit(x, y, batch_size=8, verbose=0, callbacks=[Saver()])
        except Exception:
            assert core_.ctx.rank == 0                               # the stand-in engine has no weights to write
        else:
            assert core_.ctx.barriers == before + 1
    assert not (tmp_path / "r1").exists()


def test_an_in_place_edit_between_two_fits_reaches_the_engine():
    """the round-2 cache keyed the resident copy on id() and a sparse probe, so a few edited stamps
    trained on stale data.  fit() now uploads on every call."""
    x, y = _xy(16)
    net, core = _net(0, 1)
    net.fit(x, y, batch_size=8, verbose=0)
    assert np.array_equal(core.engine.resident[0][0], x)
    x[5, 1, 1, 0] = -123.0                                            # one pixel of one stamp, same array object
    y[9] = 77.0
    net.fit(x, y, batch_size=8, verbose=0)
    assert core.engine.resident[0][0][5, 1, 1, 0] == -123.0 and (core.engine.resident[0][1][9] == 77.0).all()


def test_evaluate_never_leaves_fit_a_stale_validation_set():
    """fit(val) -> evaluate(test) -> fit(same val arrays) validated on the TEST set left in slot 1."""
    x, y = _xy(16)
    xv, yv = _xy(8)
    xt, yt = _xy(12)
    xt = xt + 1000
    net, core = _net(0, 1)
    net.fit(x, y, batch_size=8, verbose=0, validation_data=(xv, yv), reuse_device_data=True)
    assert np.array_equal(core.engine.resident[1][0], xv)
    net.evaluate(xt, yt, batch_size=4)
    assert np.array_equal(core.engine.resident[1][0], xt)
    net.fit(x, y, batch_size=8, verbose=0, validation_data=(xv, yv), reuse_device_data=True)
    assert np.array_equal(core.engine.resident[1][0], xv)            # the cache entry of slot 1 was dropped by evaluate()
    assert core.engine.evals[-1][:2] == (0, 8)
