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
