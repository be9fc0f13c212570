"""bench.py against another build of the library: python tools/bench_with_lib.py <lib.so> [bench args...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat
nat.HIP_LIB_PATH = Path(sys.argv[1]).resolve()
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
