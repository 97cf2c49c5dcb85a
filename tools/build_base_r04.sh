#!/bin/bash
# tools/libzigma_base_r04.so = the library as it stood at the end of round 4 (commit 309314b), for the same-process A/Bs of tools/r05_scan_ab.py
# and the ZIGMA_AMD_LIB=... forward A/Bs of tools/r05_call*.sh.  Built from a temporary worktree; *.so files are git-ignored but travel with gpurun.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
git -C "$R" worktree add -f "$T" 309314b -q
(cd "$T" && python -m zigma_amd.build > /dev/null)
cp "$T/zigma_amd/lib/libzigma_hip.so" "$R/tools/libzigma_base_r04.so"
git -C "$R" worktree remove --force "$T"
echo "built $R/tools/libzigma_base_r04.so"
