#!/bin/bash
# Builds a variant of libpt_hip.so with extra preprocessor definitions, without touching the objects of
# the normal build: tools/build_variant.sh <name> "<defs>"  ->  path-tracing_amd/variants/libpt_<name>.so
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; DEFS=$2
B=/tmp/pt_variant_$NAME
rm -rf "$B"; mkdir -p "$B/path-tracing_amd" "$R/path-tracing_amd/variants"
cp -r "$R/path-tracing_amd/csrc" "$R/path-tracing_amd/Makefile" "$B/path-tracing_amd/"
cp -r "$R/include" "$B/"
rm -f "$B"/path-tracing_amd/csrc/*.o
make -C "$B/path-tracing_amd" -j8 libpt_hip.so EXTRA_DEFS="$DEFS" > "$B/build.log" 2>&1 || { tail -20 "$B/build.log"; exit 1; }
cp "$B/path-tracing_amd/libpt_hip.so" "$R/path-tracing_amd/variants/libpt_$NAME.so"
echo "built path-tracing_amd/variants/libpt_$NAME.so"
