#!/bin/bash
# Build recipe for oracle/_ref/ -- the reference's own native module compiled
# from the source where it lies (/root/reference, read-only).  Outputs go only
# into oracle/_ref/ (git-ignored).  The intermediate generated C file is
# removed after compilation so that only the binary remains.
# Build container only: on the GPU box /root/reference does not exist and this
# script exits 0 without doing anything.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC=/root/reference/tt_sketch/drm/fast_lazy_gaussian.pyx
[ -f "$SRC" ] || { echo "reference not present; skipping oracle/_ref build"; exit 0; }
mkdir -p "$HERE/_ref"
EXT=$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
OUT="$HERE/_ref/fast_lazy_gaussian$EXT"
if [ -f "$OUT" ] && [ "$OUT" -nt "$SRC" ]; then exit 0; fi
NPINC=$(python3 -c "import numpy; print(numpy.get_include())")
PYINC=$(python3 -c "import sysconfig; print(sysconfig.get_paths()['include'])")
cython -3 "$SRC" -o "$HERE/_ref/fast_lazy_gaussian.c"
gcc -O2 -shared -fPIC -fopenmp -w -I"$NPINC" -I"$PYINC" \
    "$HERE/_ref/fast_lazy_gaussian.c" -o "$OUT"
rm -f "$HERE/_ref/fast_lazy_gaussian.c"
echo "built $OUT"
