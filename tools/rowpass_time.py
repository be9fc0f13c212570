"""fused_mlp_kernel time per launch (bench.py's rowpass probe) for a given build of the library (argv[1])."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
if len(sys.argv) > 1:
    nat.HIP_LIB_PATH = Path(sys.argv[1]).resolve()
import bench
from minppo_amd.config import load_config_from_cli
from minppo_amd.train import Trainer
cfg = load_config_from_cli(["stompy_pro", "training.num_envs=4096", "training.mlp_dtype=" + (sys.argv[2] if len(sys.argv) > 2 else "f32")])
tr = Trainer(cfg, use_graph=False)
tr.reset(); tr.rollout(); tr._sync()
for _ in range(3):
    sec, flops, desc = bench.rowpass_probe(tr)
    print(f"{nat.HIP_LIB_PATH.name}: {sec * 1e6:.2f} us per launch")
tr.close()
