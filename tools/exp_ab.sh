#!/bin/bash
# Run ON THE GPU BOX: bash tools/exp_ab.sh [pattern]  -- A/B of two prebuilt libraries (tools/_ab/A.so, B.so; untracked) on
# the same box, alternating, per-dispatch durations of one bench step filtered by the pattern
PAT=${1:-wgrad_core}
for v in A B A B; do
  cp tools/_ab/$v.so zeronotesamba_amd/libzns_hip.so
  echo "== $v"
  bash tools/exp_trace.sh ab_$v "$PAT" | awk '{print $5, $6, substr($0, index($0,$9), 70)}' 
done
