"""tools/peer_debug.py <world> [config overrides...] - `world` rank processes on cuda:0 through the peer-to-peer exchange (tests/dist_worker.py's
hip mode), with the exchange headers dumped at the end (MPPO_PEER_DEBUG=1).  Environment variables of the engine pass through."""
import os
import socket
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
world = int(sys.argv[1])
ovr = sys.argv[2:] or ["training.num_envs=256", "training.num_minibatches=4", "training.update_epochs=2", "training.total_timesteps=100000000"]
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
tmp = tempfile.mkdtemp()
env = dict(os.environ, MPPO_TEST_BACKEND="hip", MPPO_TEST_GRAPH=os.environ.get("MPPO_TEST_GRAPH", "1"), MPPO_ALLREDUCE="peer", MPPO_TEST_HOST_DRIVEN="0", MPPO_PEER_DEBUG="1",
           HSA_ENABLE_IPC_MODE_LEGACY="0")
procs = [subprocess.Popen([sys.executable, str(ROOT / "tests" / "dist_worker.py"), str(r), str(world), str(port), os.environ.get("UPDATES", "3"), f"{tmp}/r{r}.npz", *ovr],
                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env) for r in range(world)]
rc = 0
for r, p in enumerate(procs):
    try:
        out = p.communicate(timeout=120)[0]
    except subprocess.TimeoutExpired:
        p.kill(); out = p.communicate()[0]
    lines = [l for l in out.splitlines() if not any(k in l for k in ("socket.cpp", "Gloo", "amdgpu.ids"))]
    print(f"---- rank {r}: exit {p.returncode}")
    print("\n".join(l[:2000] for l in lines[-14:]))
    rc |= p.returncode != 0
sys.exit(rc)
