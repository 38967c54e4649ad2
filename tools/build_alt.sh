#!/bin/bash
# Build another copy of the kernel library for A/B runs inside one GPU call (SGCLIP_LIB selects the library, see
# tools/ab_lib.sh):   tools/build_alt.sh [git-rev|WORK] [output name] [extra hipcc flags]
#   tools/build_alt.sh HEAD                       -> laser_amd/libsgclip_alt.so from the committed sources
#   tools/build_alt.sh WORK abl1 -DATTN_ABLATE=1  -> laser_amd/libsgclip_abl1.so from the working tree with a define
set -e
REV=${1:-HEAD}
NAME=${2:-alt}
EXTRA=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
mkdir -p "$TMP/laser_amd/csrc" "$TMP/include"
if [ "$REV" = "WORK" ]; then
    cp "$ROOT"/laser_amd/csrc/*.hip "$ROOT"/laser_amd/csrc/*.h "$ROOT"/laser_amd/csrc/Makefile "$TMP/laser_amd/csrc/"
    cp "$ROOT"/include/*.h "$TMP/include/"
else
    for f in $(git -C "$ROOT" ls-tree --name-only "$REV" laser_amd/csrc/ include/); do
        git -C "$ROOT" show "$REV:$f" > "$TMP/$f"
    done
fi
make -C "$TMP/laser_amd/csrc" -j8 LIB="$ROOT/laser_amd/libsgclip_$NAME.so" HIPCC="/opt/rocm/bin/hipcc $EXTRA" > "$TMP/build.log" 2>&1 || { tail -20 "$TMP/build.log"; exit 1; }
rm -rf "$TMP"
ls -la "$ROOT/laser_amd/libsgclip_$NAME.so"
