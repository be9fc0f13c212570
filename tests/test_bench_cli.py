"""bench.py's own multi-GPU entry (`python bench.py --gpus N` without a launcher around it)."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_more_gpus_than_the_machine_has_is_a_clean_refusal():
    """The supervisor never touches a GPU: asked for more ranks than there are devices it says so and exits with status 2
    (nothing is spawned).  On a box with >= 2 GPUs this request would run; here (CPU container / 1-GPU box) it must refuse."""
    import torch

    if torch.cuda.device_count() >= 64:
        pytest.skip("a machine with 64 GPUs")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "64", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "--gpus 64" in r.stderr and "GPU(s)" in r.stderr
    assert r.stdout.strip() == ""  # the contract is ONE JSON line, or none


def test_rank_environment_is_what_torch_distributed_run_would_set():
    """spawn_ranks() builds each child's environment by hand; the rank side (bench.main) reads RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT - the variables the driver's `python -m torch.distributed.run` sets.  Source-level check that
    both spellings stay in sync."""
    src = (ROOT / "bench.py").read_text()
    for var in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        assert f"{var}=" in src or f'"{var}"' in src, var
    assert 'MASTER_ADDR="127.0.0.1"' in src


_FAKE_WORKER = r'''
import json, os, sys, time
rank = int(os.environ["RANK"])
assert os.environ.get("MPPO_BENCH_WORKER") == "1"
mode = os.environ["FAKE_MODE"]
transport = os.environ.get("MPPO_ALLREDUCE", "peer")
if mode == "rank1_fails_with_peer" and transport != "rccl":
    if rank == 1:
        sys.exit(3)          # this rank dies at once ...
    time.sleep(600)          # ... the others would hang in a collective: their supervisors must kill them
if mode == "rank1_hangs_after_selftest":
    # every rank starts, connects and passes the self-test; with the peer transport rank 1 then hangs in its first update (and rank 0
    # waits for its gradient).  Nothing fails fast: only the time limit counted from "all ranks started" can end the attempt.
    from pathlib import Path
    Path(os.environ["MPPO_BENCH_MILESTONE_DIR"], f"attempt{os.environ['MPPO_BENCH_ATTEMPT']}.started.rank{rank}").write_text("started")
    if transport != "rccl":
        time.sleep(600)
if mode == "always_fails":
    sys.exit(4)
if rank == 0:
    print("RCCL banner on stdout")
    print(json.dumps({"metric": "fake", "transport": transport, "graph_comm": os.environ.get("MPPO_GRAPH_COMM", ""), "port": os.environ["MASTER_PORT"],
                      "agent_store": os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "")}))
'''


def _run_supervised(tmp_path, mode, world=2, timeout=120, **extra):
    import os

    fake = tmp_path / "fake_worker.py"
    fake.write_text(_FAKE_WORKER)
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT="29999", FAKE_MODE=mode,
                   MPPO_BENCH_WORKER_SCRIPT=str(fake), MPPO_BENCH_STATUS_DIR=str(tmp_path), MPPO_BENCH_JOB=f"test_{mode}", **{"MPPO_BENCH_RANK_TIMEOUT": "60", **extra})
        env.pop("MPPO_BENCH_WORKER", None)
        env.pop("MPPO_GRAPH_COMM", None)
        env.pop("MPPO_ALLREDUCE", None)
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", str(world), "--steps", "1", "--warmup", "0"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=timeout) for p in procs]
    return [p.returncode for p in procs], outs


def test_rank_supervisor_forwards_rank0_json(tmp_path):
    """Under a launcher (`torch.distributed.run ... bench.py --gpus N`) every rank process supervises ONE child that measures;
    rank 0 forwards exactly the JSON line."""
    import json

    rcs, outs = _run_supervised(tmp_path, "ok")
    assert rcs == [0, 0], (rcs, outs)
    lines = outs[0][0].strip().splitlines()
    assert len(lines) == 1 and json.loads(lines[0])["transport"] == "peer"
    assert outs[1][0].strip() == ""


def test_rank_supervisor_retries_with_rccl_when_a_rank_fails(tmp_path):
    """One rank dying with the default transport (the peer-to-peer exchange inside the hipGraph): every supervisor kills its own
    child (the survivors would wait for the dead rank's gradient until their time limit) and all ranks repeat with eager RCCL
    all-reduces (MPPO_ALLREDUCE=rccl, MPPO_GRAPH_COMM=0) on a fresh rendezvous port."""
    import json

    rcs, outs = _run_supervised(tmp_path, "rank1_fails_with_peer")
    assert rcs == [0, 0], (rcs, outs)
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["transport"] == "rccl" and d["graph_comm"] == "0" and d["port"] != "29999" and d["agent_store"] == "False"
    assert "repeating with eager RCCL all-reduces" in outs[0][1]


def test_rank_supervisor_reports_failure(tmp_path):
    rcs, outs = _run_supervised(tmp_path, "always_fails")
    assert all(rc == 1 for rc in rcs), (rcs, outs)
    assert outs[0][0].strip() == ""


def test_rank_supervisor_repeats_on_rccl_when_a_rank_hangs_after_the_self_test(tmp_path):
    """The case a fast failure does not cover: every rank starts and the connect-time self-test passes, then rank 1 hangs inside its
    first update (the first 8-GPU contact of the fused peer form is where that could happen) and rank 0 waits for it.  The attempt's
    limit counts from the moment all ranks have reported `started`; when it expires every supervisor kills exactly its own child and
    the RCCL repeat produces the JSON line - well inside the budget the limits are chosen for (bench._attempt_limits)."""
    import json
    import time

    t0 = time.monotonic()
    rcs, outs = _run_supervised(tmp_path, "rank1_hangs_after_selftest", MPPO_BENCH_RANK_TIMEOUT="4", MPPO_BENCH_START_TIMEOUT="30", MPPO_BENCH_RETRY_TIMEOUT="30")
    took = time.monotonic() - t0
    assert rcs == [0, 0], (rcs, outs)
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["transport"] == "rccl" and d["graph_comm"] == "0"
    err = outs[0][1] + outs[1][1]  # (whichever supervisor's clock ran out first says so; the other one may learn it from that one's status file)
    assert "did not finish within 4 s of having started" in err and "repeating with eager RCCL all-reduces" in outs[0][1]
    assert took < 45, took  # 4 s limit + teardown + the repeat; nowhere near start + peer + rccl


def test_default_limits_fit_the_drivers_time_box():
    """A first attempt that hangs to its limit, the teardown and a healthy RCCL repeat must fit the driver's 600 s command limit with room
    to spare: start + peer + verdict wait + retry rendezvous + (a warm start + a healthy run, generously 120 s) <= 540 s."""
    import importlib
    import os
    import sys

    sys.path.insert(0, str(ROOT))
    bench = importlib.import_module("bench")
    saved = {k: os.environ.pop(k, None) for k in ("MPPO_BENCH_RANK_TIMEOUT", "MPPO_BENCH_RETRY_TIMEOUT", "MPPO_BENCH_START_TIMEOUT")}
    try:
        lim = bench._attempt_limits()
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
    assert lim["start"] + lim["peer"] + 20 + 30 + 120 <= 540
    assert lim["rccl"] >= 2 * lim["peer"]
