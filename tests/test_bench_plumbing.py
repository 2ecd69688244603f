"""bench.py's N > 1 plumbing on CPU: launched exactly as the driver launches it (torch.distributed.run, one rank per GPU)
with a stub engine in place of the HIP library.  Checked: rank 0's RCCL id reaches every rank, the barriers bracket the
timed region, the elapsed time is the MAX over ranks, ONE JSON line comes out, on rank 0's stdout only, `value` is the
whole-job aggregate, the rank processes exit 0 on their own (no os._exit) and never import torch (the launcher is the
only torch process: the ranks meet through debvader_amd.parallel.HostGroup).  A second launch without torch.distributed.run
(two plain processes with RANK / WORLD_SIZE / MASTER_* set, MASTER_PORT free) covers HostGroup's direct-port mode."""
import glob
import json
import os
import socket
import subprocess
import sys

import pytest

pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_bench_launch_with_a_stub_engine(tmp_path):
    log = str(tmp_path / "stub")
    env = dict(os.environ, DV_BENCH_STUB_ENGINE="tests.stub_engine", DV_STUB_LOG=log, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    steps, warmup, world = 4, 2, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps),
           "--warmup", str(warmup)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout                         # one JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 256 * world and d["config"]["parallelism"] == "dp2"
    # the slow rank (rank 1: 25 ms per step) sets the time: MAX over ranks, not rank 0's own 5 ms per step
    assert d["ms_per_step"] >= 24.0, d["ms_per_step"]
    assert abs(d["value"] - 256 * world * steps / (d["ms_per_step"] * 1e-3 * steps)) <= 1e-6 * d["value"]
    logs = {}
    for f in glob.glob(log + ".*"):
        logs[int(f.rsplit(".", 1)[1])] = [json.loads(ln) for ln in open(f)]
    assert sorted(logs) == [0, 1]
    uids = {rk: next(e["uid"] for e in ev if e["event"] == "ctx") for rk, ev in logs.items()}
    assert uids[0] == uids[1] and len(uids[0]) == 256         # rank 0's 128 bytes on both ranks
    for rk, ev in logs.items():
        ts = [e for e in ev if e["event"] == "train_steps"]
        assert [t["steps"] for t in ts] == [warmup, steps]
        assert all(t["global_batch"] == 256 * world and t["B"] == 256 for t in ts)
        assert sum(e["event"] == "sync" for e in ev) >= 3      # before the timed region, inside it and at its end
        assert not any(e.get("torch_loaded") for e in ev), "torch was imported into a rank process"
        assert ev[-1]["event"] == "close"                      # the context was closed, the process left normally


def test_two_plain_processes_meet_on_master_port(tmp_path):
    """The same bench without a launcher: RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT by hand, the port free, so rank 0
    listens on it directly."""
    log = str(tmp_path / "stub")
    port, world, steps = _free_port(), 2, 3
    procs = []
    for rank in range(world):
        env = dict(os.environ, DV_BENCH_STUB_ENGINE="tests.stub_engine", DV_STUB_LOG=log, PYTHONPATH=ROOT, RANK=str(rank),
                   LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps),
                                       "--warmup", "1"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                                      cwd=ROOT))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], [o[1][-1500:] for o in outs]
    lines0 = [ln for ln in outs[0][0].splitlines() if ln.strip().startswith("{")]
    assert len(lines0) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.strip().startswith("{")]
    d = json.loads(lines0[0])
    assert d["n_gpus"] == world and d["ms_per_step"] >= 24.0
