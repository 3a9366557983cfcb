#!/bin/bash
# Builds the reference's OWN two native extensions (tools/bindings/interpolate/src/{module,interpolate}.cpp
# via pybind11, tools/bindings/utils/makePoints.pyx via cython) straight from the sources where they lie
# under /root/reference, into oracle/_ref/RAiDER/.  TEST INFRASTRUCTURE: used to pin the oracle
# (tests) and, optionally, as the "reference" CPU baseline for interp3d.  Nothing is copied into the
# repo; oracle/_ref/ is git-ignored (but travels to the GPU box via gpurun).
# pybind11, cython and numpy are present in this image; no stand-in headers are written.
set -euo pipefail
REF=${RAIDER_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref/RAiDER"
if [ ! -d "$REF/tools/bindings" ]; then
  echo "build_ref: $REF not present - keeping prebuilt oracle/_ref" >&2
  exit 0
fi
mkdir -p "$OUT" "$HERE/_ref/build"
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
PBINC=$(python3 -c "import pybind11; print(pybind11.get_include())")
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
SUFFIX=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
SRC="$REF/tools/bindings/interpolate/src"
if [ ! -f "$OUT/interpolate$SUFFIX" ] || [ "$SRC/interpolate.cpp" -nt "$OUT/interpolate$SUFFIX" ]; then
  g++ -O3 -std=c++17 -shared -fPIC -pthread -w -I"$PYINC" -I"$PBINC" -I"$SRC" \
      "$SRC/module.cpp" "$SRC/interpolate.cpp" -o "$OUT/interpolate$SUFFIX"
fi
PYX="$REF/tools/bindings/utils/makePoints.pyx"
if [ ! -f "$OUT/makePoints$SUFFIX" ] || [ "$PYX" -nt "$OUT/makePoints$SUFFIX" ]; then
  # cython writes its generated C next to -o target (build dir), never into the reference tree
  cython -3 "$PYX" -o "$HERE/_ref/build/makePoints.c"
  gcc -O3 -shared -fPIC -w -I"$PYINC" -I"$NPINC" "$HERE/_ref/build/makePoints.c" -o "$OUT/makePoints$SUFFIX"
fi
echo "build_ref: ok -> $OUT"
