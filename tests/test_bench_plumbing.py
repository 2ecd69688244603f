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


@pytest.mark.parametrize("world", [2, 8])
def test_multi_rank_bench_launch_with_a_stub_engine(tmp_path, world):
    """world 2, and world 8 - the node the driver's scaling run uses: rendezvous of eight ranks beside
    torchrun's store, eight communicator reports gathered into `multi_rank`, `verified` true, MAX over eight ranks."""
    log = str(tmp_path / "stub")
    env = dict(os.environ, DV_BENCH_STUB_ENGINE="tests.stub_engine", DV_STUB_LOG=log, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
    steps, warmup = 4, 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", str(steps),
           "--warmup", str(warmup)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout                         # one JSON line, from rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 256 * world and d["config"]["parallelism"] == f"dp{world}"
    # the slowest rank (rank world-1: 5 + 20 ms per rank index and step) sets the time: MAX over ranks, not rank 0's own 5 ms
    assert d["ms_per_step"] >= 4.0 + 20.0 * (world - 1), d["ms_per_step"]
    assert abs(d["value"] - 256 * world * steps / (d["ms_per_step"] * 1e-3 * steps)) <= 1e-6 * d["value"]
    # the N > 1 line is self-verifying: what RCCL reports for the communicator on EVERY rank, which
    # device each rank drives, the communication of a step and its exposed part, and a whole-step fraction
    mr = d["multi_rank"]
    assert mr["world"] == world and mr["rccl_ranks"] == [world] * world and mr["rccl_rank_ids"] == list(range(world))
    assert [v["rank"] for v in mr["devices"]] == list(range(world)) and mr["distinct_devices"] == world
    assert len({v["pci_bus_id"] for v in mr["devices"]}) == world and mr["verified"] is True and mr["rehearsal"] is False
    assert mr["collectives_per_step"] == 5 and abs(mr["comm_ms_per_step"] - 0.4) < 1e-9
    assert abs(mr["exposed_comm_ms_per_step"] - 0.05 * world) < 1e-9          # MAX over ranks (rank r: 0.05 * (r + 1))
    assert d["roofline"]["whole_step_frac"] > 0 and d["roofline"]["bound"] == "mfma" and "rehearsal" not in d
    logs = {}
    for f in glob.glob(log + ".*"):
        logs[int(f.rsplit(".", 1)[1])] = [json.loads(ln) for ln in open(f)]
    assert sorted(logs) == list(range(world))
    uids = {rk: next(e["uid"] for e in ev if e["event"] == "ctx") for rk, ev in logs.items()}
    assert len(set(uids.values())) == 1 and len(uids[0]) == 256         # rank 0's 128 bytes on every rank
    for rk, ev in logs.items():
        ts = [e for e in ev if e["event"] == "train_steps"]
        assert [t["steps"] for t in ts] == [warmup, steps, steps]       # warm-up, the timed region, the comm-timing pass
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


def test_bench_config_flag_selects_the_baseline_configuration(tmp_path):
    """--config 3 (128 px, 64 per GPU) and --config 2 (bf16) under the driver's launch line: the line names the
    configuration, the per-GPU batch and the stamp geometry it ran."""
    for conf, want_batch, want in ((3, 64, "128x128x6"), (2, 256, "59x59x6")):
        log = str(tmp_path / f"stub{conf}")
        env = dict(os.environ, DV_BENCH_STUB_ENGINE="tests.stub_engine", DV_STUB_LOG=log, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
               "--config", str(conf)]
        r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
        assert f"configs[{conf}]" in d["config"]["workload"] and d["config"]["per_gpu_batch"] == want_batch
        assert d["config"]["global_batch"] == 2 * want_batch and want in d["metric"]
        assert d["dtype"] == ("bf16" if conf == 2 else "f32") and d["multi_rank"]["verified"]
