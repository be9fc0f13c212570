"""Launches the two hot kernels in isolation (for rocprofv3 --pmc runs): the hidden-layer GEMM of one minibatch and env_step."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat  # noqa: E402
from minppo_amd.config import load_config_from_cli  # noqa: E402
from minppo_amd.train import Trainer  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = load_config_from_cli(["stompy_pro", "training.num_envs=4096", *sys.argv[3:]])  # further arguments: config overrides (training.mlp_dtype=bf16)
tr = Trainer(cfg, use_graph=False)
tr.reset()
tr.update()
tr._sync()
s = tr.stream
if what in ("all", "gemm"):
    mb, H = 1280, 256
    a = torch.randn(2, mb, H, device="cuda"); w = torch.randn(2, H, H, device="cuda") * 0.06; b = torch.zeros(2, H, device="cuda"); c = torch.empty(2, mb, H, device="cuda")
    descs = (nat.GemmDesc * 2)()
    for i in range(2):
        descs[i] = nat.GemmDesc(a[i].data_ptr(), w[i].data_ptr(), c[i].data_ptr(), b[i].data_ptr(), 0, 0, 0, mb, H, H, H, H, H, 0, 1 if i == 0 else 2)
    for _ in range(reps):
        tr.lib.gemm_batch(descs, 2, 0, 1, 0, 0, s.cuda_stream)
    s.synchronize()
if what in ("all", "env"):
    for _ in range(max(1, reps // 4)):
        tr.rollout()
    s.synchronize()
if what in ("all", "learn"):
    tr.learn()
    s.synchronize()
tr.close()
