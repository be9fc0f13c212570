"""Writes `<summary>.meta.json` next to a profile summary: which tree it was taken on.

    python tools/profile_meta.py <summary file> [<summary file> ...] --command "<what produced it>"

`kernel_sources_sha` is bench.kernel_sources_sha() of the tree the measurement ran from (the GPU box's snapshot: no .git there);
`git_head` is what tools/gpurun.sh recorded in tools/steps/git_head.txt before the snapshot was pushed (or `git rev-parse HEAD` when
run inside the repository).  bench.py quotes a committed summary in its `roofline` object only if the hash equals the hash of the
tree bench.py itself runs from (bench.committed_summary)."""
import argparse
import json
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def git_head() -> str:
    f = ROOT / "tools" / "steps" / "git_head.txt"
    if (ROOT / ".git").exists():
        r = subprocess.run(["git", "-C", str(ROOT), "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True)
        d = subprocess.run(["git", "-C", str(ROOT), "status", "--porcelain", "--", "minppo_amd", "include", "bench.py"], capture_output=True, text=True)
        if r.returncode == 0:
            return r.stdout.strip() + ("+dirty" if d.stdout.strip() else "")
    return f.read_text().strip() if f.exists() else "unknown"


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="+")
    ap.add_argument("--command", default="")
    a = ap.parse_args()
    meta = {"kernel_sources_sha": bench.kernel_sources_sha(), "git_head": git_head(), "command": a.command, "taken": time.strftime("%Y-%m-%d %H:%M:%S")}
    for f in a.files:
        p = Path(f)
        if not p.exists():
            print(f"profile_meta: {f} does not exist, skipped", file=sys.stderr)
            continue
        p.with_name(p.name + ".meta.json").write_text(json.dumps(meta, indent=1) + "\n")
        print(f"profile_meta: {p.name} <- kernel sources {meta['kernel_sources_sha']}, commit {meta['git_head']}")


if __name__ == "__main__":
    main()
