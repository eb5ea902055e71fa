#!/bin/bash
# A/B timing of library builds on the GPU box: bash tools/ab.sh "<args after the lib>" lib1.so lib2.so ...
# (two interleaved rounds, one process per measurement)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
ARGS=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    timeout -k 10 120 python tools/time_variant.py "$lib" $ARGS || exit 1
  done
done
