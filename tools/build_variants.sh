#!/bin/bash
# Builds experiment variants of libsvx.so side by side (svim_amd/variants/libsvx_<name>.so; select one with SVX_LIB=<path>).
#   tools/build_variants.sh name1="-DFLAG ..." name2="" ...      current tree with extra compiler flags
#   tools/build_variants.sh @<git-rev>=name                       a committed revision (built in a scratch worktree)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/svim_amd/variants
mkdir -p "$OUT"
for spec in "$@"; do
  if [[ $spec == @* ]]; then
    rev=${spec#@}; rev=${rev%%=*}; name=${spec#*=}
    wt=$(mktemp -d /tmp/svx_wt.XXXXXX)
    git -C "$ROOT" worktree add -f --detach "$wt" "$rev" >/dev/null 2>&1
    make -s -j8 -C "$wt/svim_amd/csrc" >/dev/null 2>&1
    cp "$wt/svim_amd/libsvx.so" "$OUT/libsvx_$name.so"
    git -C "$ROOT" worktree remove --force "$wt"
  else
    name=${spec%%=*}; flags=${spec#*=}
    bd=$(mktemp -d /tmp/svx_var.XXXXXX)
    cp "$ROOT"/svim_amd/csrc/*.hip "$ROOT"/svim_amd/csrc/*.hpp "$ROOT"/svim_amd/csrc/*.cpp "$ROOT"/svim_amd/csrc/Makefile "$bd"/
    mkdir -p "$bd/../../include_tmp"
    sed -i "s#\.\./\.\./include/svx\.h#$ROOT/include/svx.h#g" "$bd"/Makefile "$bd"/*.hip "$bd"/*.hpp "$bd"/*.cpp
    make -s -j8 -C "$bd" EXTRA="$flags" OUT="$OUT/libsvx_$name.so" > "$OUT/build_$name.log" 2>&1 || { echo "build of $name FAILED: $OUT/build_$name.log"; tail -5 "$OUT/build_$name.log"; }
    rm -rf "$bd"
  fi
  echo "built $OUT/libsvx_$name.so"
done
