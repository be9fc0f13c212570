"""Where does fused_mlp_kernel's time go: phase breakdown of one wave (wave 0 of row tile 40, actor network; s_memtime stamps in
shader-clock ticks) from a -DMPPO_FUSED_TIMERS build (tools/build_variant.sh ftimers k_fused.hip -DMPPO_FUSED_TIMERS).
usage: python tools/fused_phases.py tools/_variants/libminppo_ftimers.so [config overrides, e.g. training.mlp_dtype=bf16]"""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
nat.HIP_LIB_PATH = Path(sys.argv[1]).resolve()
from minppo_amd.config import load_config_from_cli
from minppo_amd.train import Trainer
cfg = load_config_from_cli(["stompy_pro", "training.num_envs=4096", *sys.argv[2:]])
tr = Trainer(cfg, use_graph=False)
tr.reset(); tr.update(); tr._sync()
dll = C.CDLL(str(nat.HIP_LIB_PATH))
# interval i ends at stamp FT(i + 1) of csrc/k_fused.hip
names = {0: "entry -> arguments; weights / biases / head weights requested (bf16 row pass: x, W1, biases)", 1: "index -> rows chain (index wait, dependent loads issued; bf16 row pass: head weights / loss scalars requested)", 2: "x tile to LDS + barrier", 3: "L1 GEMM (this wave)",
         4: "L1 epilogue + stores + barrier (waits for the SIMD's second wave)", 5: "L2 GEMM (this wave)", 6: "L2 epilogue + stores + barrier", 7: "(two stamps in a row)", 8: "head GEMM + partials + barrier", 9: "head sums + loss + dOut",
         10: "barrier + loss partials", 11: "dZ2 + stores + barrier", 12: "dZ1 GEMM", 13: "dZ1 epilogue + stores"}
acc = {}
for k in range(10):
    tr.learn(); tr._sync()
    t = (C.c_ulonglong * 88)(); assert dll.mppo_debug_fused_timers(t) == 0
    t = list(t)
    for i in range(14):
        acc.setdefault(i, []).append(float(t[i + 1] - t[i]))
    if k == 9:
        print("per wave, ticks after wave 0 entered the kernel: entry", [int(t[56 + w] - t[0]) for w in range(8)], "| loads issued", [int(t[64 + w] - t[0]) for w in range(8)],
              "| x tile written, at the barrier", [int(t[72 + w] - t[0]) for w in range(8)], "| past the barrier", [int(t[80 + w] - t[0]) for w in range(8)])
        for layer in (0, 1):
            print(f'layer {layer + 1} GEMM per wave (start, end) in ticks after wave 0 entered the layer loop:', [(int(t[24 + 16 * layer + w] - t[3]), int(t[24 + 16 * layer + 8 + w] - t[3])) for w in range(8)])
tot = sum(float(np.median(v)) for v in acc.values())
for i in sorted(acc):
    v = float(np.median(acc[i]))
    print(f"{names[i]:46s} {v:8.0f} ticks {100 * v / tot:5.1f} %")
print(f"total {tot:.0f} ticks")
tr.close()
