#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY.  Builds the reference's own `successive` NCC Cython module from the
# sources where they lie under /root/reference into oracle/_ref/ (git-ignored, never committed).
# It is used by oracle/make_ref_vectors.py to generate golden vectors in THIS container only;
# nothing under /root/reference (source, bytecode or otherwise) travels to the GPU box.
#
# Only PyMaSC/core/successive/ncc.pyx is built: it needs cython + numpy, both present.
# NOT built: core/bitarray/* (links the absent noporpoise/BitArray library) and
# core/successive/mscc.pyx, reader/bigwig.pyx, core/mappability.pyx (import pyBigWig, absent).
set -euo pipefail
REF=/root/reference
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF/PyMaSC" ] || { echo "no /root/reference here: skipping oracle/_ref build"; exit 0; }
mkdir -p "$OUT"
PYINC=$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])')
NPINC=$(python3 -c 'import numpy; print(numpy.get_include())')
EXT=$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))')
python3 -m cython -3 -I "$REF" -o "$OUT/ncc.c" "$REF/PyMaSC/core/successive/ncc.pyx"
gcc -O2 -fPIC -shared -w -DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION -I"$PYINC" -I"$NPINC" \
    -o "$OUT/ncc$EXT" "$OUT/ncc.c"
rm -f "$OUT/ncc.c"
echo "built $OUT/ncc$EXT"
