#!/usr/bin/env python3
"""tools/kernel_resources.py <source.hip> [name filter] [extra hipcc flags...] - registers, spills, scratch and occupancy of every kernel in one csrc/ source
file, from hipcc's -Rpass-analysis=kernel-resource-usage remarks (cross-compiles for gfx950; no GPU needed)."""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
stem = Path(src).name
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=" + ("off" if stem == "k_physics.hip" else "fast"), f"-I{ROOT}/minppo_amd/csrc", f"-I{ROOT}/include",
         "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage", "-c", str(ROOT / "minppo_amd/csrc" / stem), "-o", "/dev/null", *extra]
r = subprocess.run(["/opt/rocm/bin/hipcc", *flags], capture_output=True, text=True)
if r.returncode != 0:
    sys.exit(r.stderr[-3000:])
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        cur = d.split("(")[0].replace("void mppo::", "")
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for k, d in rows.items():
    if flt and flt not in k:
        continue
    print(f"{k[:110]:110s} VGPR {d.get('VGPRs','?'):>3} AGPR {d.get('AGPRs','?'):>3} spillV {d.get('VGPRs Spill','?'):>3} spillS {d.get('SGPRs Spill','?'):>3} scratch {d.get('ScratchSize [bytes/lane]','?'):>4} occ {d.get('Occupancy [waves/SIMD]','?')}")
