"""On-GPU microbenchmark of the batched GEMM variants at the minibatch shapes (HIP events, 200 launches each).
   MPPO_GEMM_IMPL=ddd|lll|... selects direct / LDS per variant."""
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from minppo_amd import _native as nat  # noqa: E402

lib = nat.load()
dev = "cuda"
mb, H, O, OP, A, B = 1280, 256, 225, 228, 10, 40960
s = torch.cuda.Stream()
obs = torch.randn(B, OP, device=dev); obs[:, O:] = 0
idx = torch.randperm(B, device=dev)[:mb].to(torch.int32)
W1 = torch.randn(2, O, H, device=dev) * 0.1; W2 = torch.randn(2, H, H, device=dev) * 0.1
bias = torch.zeros(2, H, device=dev)
h1 = torch.randn(2, mb, H, device=dev); h2 = torch.randn(2, mb, H, device=dev); dz = torch.randn(2, mb, H, device=dev); dz1 = torch.randn(2, mb, H, device=dev)
dout = torch.randn(mb, 16, device=dev)
P = 250136
slabs = torch.zeros(3, P, device=dev)


def D(*a):
    return nat.GemmDesc(*a)


def timeit(name, descs, n, variant, ksplit, stride, reps=200):
    arr = (nat.GemmDesc * n)(*descs)
    for _ in range(10):
        lib.gemm_batch(arr, n, variant, ksplit, stride, 0, s.cuda_stream)
    s.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(reps):
        lib.gemm_batch(arr, n, variant, ksplit, stride, 0, s.cuda_stream)
    e1.record(s)
    s.synchronize()
    print(f"{name:28s} {e0.elapsed_time(e1) * 1e3 / reps:8.2f} us/launch")


p = lambda t: t.data_ptr()
print("impl", os.environ.get("MPPO_GEMM_IMPL", "default"))
timeit("fwd L1 (gather, K=225)", [D(p(obs), p(W1[i]), p(h1[i]), p(bias[i]), 0, p(idx), 0, mb, H, O, OP, H, H, 0, 1 if i == 0 else 2) for i in range(2)], 2, 0, 1, 0)
timeit("fwd L2 (K=256)", [D(p(h1[i]), p(W2[i]), p(h2[i]), p(bias[i]), 0, 0, 0, mb, H, H, H, H, H, 0, 1 if i == 0 else 2) for i in range(2)], 2, 0, 1, 0)
timeit("fwd L2 one net", [D(p(h1[0]), p(W2[0]), p(h2[0]), p(bias[0]), 0, 0, 0, mb, H, H, H, H, H, 0, 1)], 1, 0, 1, 0)
timeit("bwd dZ1", [D(p(dz[i]), p(W2[i]), p(dz1[i]), 0, p(h1[i]), 0, 0, mb, H, H, H, H, H, H, 1 if i == 0 else 2) for i in range(2)], 2, 1, 1, 0)
offs = [0, 100000, 200000]
dw = []
for i in range(2):
    base = i * 125000
    dw += [D(p(h2[i]), p(dout) + (0 if i == 0 else 48), p(slabs) + 4 * (base + 0), 0, 0, 0, p(slabs) + 4 * (base + 3000), H, A if i == 0 else 1, mb, H, 16, A if i == 0 else 1, 0, 0),
           D(p(h1[i]), p(dz[i]), p(slabs) + 4 * (base + 4000), 0, 0, 0, p(slabs) + 4 * (base + 70000), H, H, mb, H, H, H, 0, 0),
           D(p(obs), p(dz1[i]), p(slabs) + 4 * (base + 60000 + 11000), 0, 0, p(idx), p(slabs) + 4 * (base + 124000), O, H, mb, OP, H, H, 0, 0)]
timeit("dW (6 problems, ksplit 3)", dw, 6, 2, 3, P)
timeit("dW W2 only (2 problems)", [dw[1], dw[4]], 2, 2, 3, P)
timeit("dW W1 only (gather)", [dw[2], dw[5]], 2, 2, 3, P)
