"""Fixed cost vs per-MFMA cost of the forward GEMM: hipGraph replay of 100 launches (no host in the loop), K swept."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat  # noqa: E402

lib = nat.load()
mb, H = 1280, 256
s = torch.cuda.Stream()
for nets in (2, 1):
    for K in (32, 64, 128, 256, 512, 1024, 2048):
        a = torch.randn(nets, mb, K, device="cuda"); w = torch.randn(nets, K, H, device="cuda") * 0.05
        b = torch.zeros(nets, H, device="cuda"); c = torch.empty(nets, mb, H, device="cuda")
        descs = (nat.GemmDesc * nets)(*[nat.GemmDesc(a[i].data_ptr(), w[i].data_ptr(), c[i].data_ptr(), b[i].data_ptr(), 0, 0, 0, mb, H, K, K, H, H, 0, 2) for i in range(nets)])
        for _ in range(3):
            lib.gemm_batch(descs, nets, 0, 1, 0, 0, s.cuda_stream)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(100):
                lib.gemm_batch(descs, nets, 0, 1, 0, 0, s.cuda_stream)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 300
        print(f"nets {nets} K {K:5d}: {us:7.2f} us/launch   MFMA-chain floor {K / 2 * 0.029:6.2f} us   blocks {nets * 20 * 4}")
