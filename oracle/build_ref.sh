#!/bin/bash
# Build the real reference hot path from its own sources, in place, into oracle/_ref/.
# TEST INFRASTRUCTURE ONLY. Needs /root/reference (absent on the GPU box: there the prebuilt
# oracle/_ref/libfoldcomp_ref.so that travelled with the snapshot is used).
# Flags mirror the reference's Release build (CMakeLists.txt: -O3 -DNDEBUG, C++17, x86-64 baseline,
# i.e. no -march: the binary contains no FMA), see SURVEY.md §4 [probe].
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
REF="${FOLDCOMP_REFERENCE:-/root/reference}"
OUT="$HERE/_ref"
if [ ! -d "$REF/src" ]; then
  echo "build_ref.sh: $REF/src not present; keeping prebuilt $OUT (if any)" >&2
  exit 0
fi
mkdir -p "$OUT"
# the codec (8 files) plus the rows either side of it: the gemmi-based structure reader (header-only gemmi as vendored under
# lib/, zlib from the system) and the database container
SRCS="amino_acid atom_coordinate discretizer foldcomp nerf sidechain torsion_angle utility structure_reader database_reader database_writer"
OBJS=""
for s in $SRCS; do
  g++ -O3 -DNDEBUG -std=c++17 -fPIC -D_USE_MATH_DEFINES=1 -w -I"$REF/src" -I"$REF/lib" \
      -c "$REF/src/$s.cpp" -o "$OUT/$s.o" &
  OBJS="$OBJS $OUT/$s.o"
done
g++ -O3 -DNDEBUG -std=c++17 -fPIC -fopenmp -D_USE_MATH_DEFINES=1 -w -I"$REF/src" -I"$REF/lib" \
    -c "$HERE/ref_shim.cpp" -o "$OUT/ref_shim.o" &
wait
g++ -shared -fopenmp -o "$OUT/libfoldcomp_ref.so" $OBJS "$OUT/ref_shim.o" -lz
rm -f "$OUT"/*.o
echo "built $OUT/libfoldcomp_ref.so"
# the reference's own command line (src/main.cpp over the same sources + its vendored microtar; the definitions its CMakeLists.txt
# gives the executable target): the checker of the drivers' container handling -- tar archives in and out, output naming -- in
# tests/test_tar_vs_reference.py
COBJS=""
for s in $SRCS main; do
  g++ -O3 -DNDEBUG -std=c++17 -fopenmp -DOPENMP -DFOLDCOMP_EXECUTABLE -D_USE_MATH_DEFINES=1 -w -I"$REF/src" -I"$REF/lib" -I"$REF/lib/microtar" \
      -c "$REF/src/$s.cpp" -o "$OUT/cli_$s.o" &
  COBJS="$COBJS $OUT/cli_$s.o"
done
gcc -O2 -w -c "$REF/lib/microtar/microtar.c" -o "$OUT/cli_microtar.o" &
wait
g++ -fopenmp -o "$OUT/foldcomp_ref" $COBJS "$OUT/cli_microtar.o" -lz
rm -f "$OUT"/*.o
echo "built $OUT/foldcomp_ref"
# the reference's own Python module (foldcomp/foldcomp.cxx over the same sources, against this interpreter's headers): the checker of
# the Python surface in tests/test_api_vs_reference_module.py. It is called `foldcomp` like the drop-in, so it lives in a directory of
# its own that only the checker's subprocess puts on its path.
PYINC="$(python3 -c 'import sysconfig; print(sysconfig.get_paths()["include"])' 2>/dev/null || true)"
PYEXT="$(python3 -c 'import sysconfig; print(sysconfig.get_config_var("EXT_SUFFIX"))' 2>/dev/null || true)"
if [ -n "$PYINC" ] && [ -f "$PYINC/Python.h" ] && [ -f "$REF/foldcomp/foldcomp.cxx" ]; then
  POBJS=""
  for s in $SRCS; do
    g++ -O3 -DNDEBUG -std=c++17 -fPIC -D_USE_MATH_DEFINES=1 -w -I"$REF/src" -I"$REF/lib" -c "$REF/src/$s.cpp" -o "$OUT/py_$s.o" &
    POBJS="$POBJS $OUT/py_$s.o"
  done
  g++ -O3 -DNDEBUG -std=c++17 -fPIC -D_USE_MATH_DEFINES=1 -w -I"$REF/src" -I"$REF/lib" -I"$PYINC" -c "$REF/foldcomp/foldcomp.cxx" -o "$OUT/py_module.o" &
  wait
  mkdir -p "$OUT/pymod"
  g++ -shared -o "$OUT/pymod/foldcomp$PYEXT" $POBJS "$OUT/py_module.o" -lz
  rm -f "$OUT"/*.o
  echo "built $OUT/pymod/foldcomp$PYEXT"
fi
