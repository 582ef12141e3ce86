#!/bin/bash
# Regenerates tests/golden/*_from_node.bin: the outputs of the reference's own STL formatter
# (prepare_data/data_formatter.js, run with node; only its two hard-coded path constants are
# rewritten) for the two small STL files kept under assets/objects/ and for awkward.stl beside this script.  Needs /root/reference.
set -e
REF=${REF:-/root/reference}
OUT=$(cd "$(dirname "$0")" && pwd)
W=$(mktemp -d)
mkdir -p "$W/prepare_data" "$W/objects"
cp "$OUT/awkward.stl" "$W/objects/"     # hand-written: ragged spacing, CRLF, non-numbers, f32 range ends
for m in small_knight small_teapot awkward; do
  [ -f "$W/objects/$m.stl" ] || cp "$REF/objects/$m.stl" "$W/objects/"
  sed -e "s#../objects/teapot.stl#../objects/$m.stl#" -e "s#../objects/teapot.bin#../objects/$m.bin#" \
      "$REF/prepare_data/data_formatter.js" > "$W/prepare_data/fmt.js"
  (cd "$W/prepare_data" && node fmt.js > /dev/null)
  cp "$W/objects/$m.bin" "$OUT/${m}_from_node.bin"
done
rm -rf "$W"
