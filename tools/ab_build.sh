#!/bin/bash
# A/B builds for kernel experiments: libfpt_hip_a.so = git HEAD, libfpt_hip.so = working tree.
# Then on the GPU box: bash tools/ab_head.sh   (AB_CFGS="3 2 4" to choose bench configs)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
rm -rf /tmp/ab_a && mkdir -p /tmp/ab_a
git -C "$ROOT" archive HEAD footprint_tools_amd/csrc include | tar -x -C /tmp/ab_a
make -s -C /tmp/ab_a/footprint_tools_amd/csrc -j4 OUT="$ROOT/footprint_tools_amd/libfpt_hip_a.so"
make -s -C "$ROOT/footprint_tools_amd/csrc" -j4
ls -la "$ROOT"/footprint_tools_amd/libfpt_hip*.so
