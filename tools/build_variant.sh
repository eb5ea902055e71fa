#!/bin/bash
# Builds an experiment variant of the library: bash tools/build_variant.sh <name> [extra hipcc flags...]  -> tools/bin/<name>.so
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$R/tools/bin"
hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -munsafe-fp-atomics -I"$R/include" "$@" -o "$R/tools/bin/$NAME.so" "$R/instaorder_amd/csrc/msda_hip.hip"
