"""Per-update wall time of the first updates of a fresh process (does the engine's first fraction of a second run slower than its
steady state - clock ramp, first graph replays - and how long does that last?).  Usage: python tools/ramp_probe.py [n_updates] [--no-graph]"""
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd.config import load_config_from_cli
from minppo_amd.train import Trainer

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 60
cfg = load_config_from_cli(["stompy_pro", "training.num_envs=4096"])
tr = Trainer(cfg, device="cuda:0", use_graph="--no-graph" not in sys.argv)
tr.init_comm()
tr.reset()
ts = []
torch.cuda.synchronize()
t_start = time.perf_counter()
for k in range(n):
    t0 = time.perf_counter()
    tr.update()
    tr.stream.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("first updates, ms each (synchronised after every update):")
print(" ".join("%.2f" % t for t in ts))
print("elapsed %.1f ms; last-10 mean %.3f ms" % ((time.perf_counter() - t_start) * 1e3, sum(ts[-10:]) / 10))
# the same without a synchronisation per update
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(20):
    tr.update()
tr.stream.synchronize()
print("20 updates back to back: %.3f ms each" % ((time.perf_counter() - t0) * 1e3 / 20))
tr.close()
