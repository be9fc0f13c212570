"""bench.py against another build of the library: python tools/bench_with_lib.py <lib.so> [bench args...]
(with --gpus N the rank processes are started through this file too, so every rank loads the same library)"""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
if os.environ.get("MPPO_BENCH_WITH_LIB"):  # a rank started by the supervisor below
    nat.HIP_LIB_PATH = Path(os.environ["MPPO_BENCH_WITH_LIB"])
    sys.argv = ["bench.py"] + sys.argv[1:]
else:
    nat.HIP_LIB_PATH = Path(sys.argv[1]).resolve()
    os.environ["MPPO_BENCH_WITH_LIB"] = str(nat.HIP_LIB_PATH)
    os.environ["MPPO_BENCH_WORKER_SCRIPT"] = str(Path(__file__).resolve())
    sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
