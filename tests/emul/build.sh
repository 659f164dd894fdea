#!/bin/sh
# TEST INFRASTRUCTURE ONLY: builds the one-lane host emulation of the walk kernel's source (see walk_emul.cpp)
set -e
here=$(cd "$(dirname "$0")" && pwd)
out=${1:-$here/walk_emul}
g++ -O1 -g -std=c++17 $EMUL_FLAGS -Wall -Wno-unknown-pragmas -Wno-unused-function -Wno-unused-variable -Wno-maybe-uninitialized -I"$here/shim" -I"$here/../../re2-modification_amd/csrc" -I"$here/../../include" \
    -o "$out" "$here/walk_emul.cpp" "$here/../../re2-modification_amd/csrc/walk_tables.cpp" "$here/../../re2-modification_amd/csrc/image_host.cpp"
