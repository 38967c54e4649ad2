#!/bin/bash
# Build laser_amd/libsgclip_alt.so from the kernel sources of another git revision (default HEAD), for A/B runs
# against the working tree's library inside one GPU call (SGCLIP_LIB selects the library, see tools/ab_lib.sh).
set -e
REV=${1:-HEAD}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/laser_amd/csrc" "$TMP/include"
for f in $(git -C "$ROOT" ls-tree --name-only "$REV" laser_amd/csrc/ include/); do
    git -C "$ROOT" show "$REV:$f" > "$TMP/$f"
done
make -C "$TMP/laser_amd/csrc" -j8 LIB="$ROOT/laser_amd/libsgclip_alt.so" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
rm -rf "$TMP"
ls -la "$ROOT/laser_amd/libsgclip_alt.so"
