#!/bin/bash
# tools/gpurun.sh <gpurun arguments...>: records the commit the snapshot is taken from (tools/steps/git_head.txt travels with the
# snapshot, .git does not; tools/profile_meta.py stamps every profile summary with it) and then calls gpurun with the same arguments.
cd "$(dirname "$0")/.." || exit 1
mkdir -p tools/steps
H=$(git rev-parse --short=12 HEAD)
if [ -n "$(git status --porcelain -- minppo_amd include bench.py)" ]; then H="$H+dirty"; fi
echo "$H" > tools/steps/git_head.txt
exec /usr/local/graft/bin/gpurun "$@"
