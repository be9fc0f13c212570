#!/bin/bash
# TEST INFRASTRUCTURE — builds tests/emu/libminppo_emu.so: the csrc/ kernel sources compiled by g++
# against the SIMT emulator shim (no GPU, no hipcc).  Usage: tests/emu/build_emu.sh [extra g++ flags]
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
SRC="$ROOT/minppo_amd/csrc"
OUT="$HERE/libminppo_emu.so"
FILES=""
for f in $(cat "$SRC/SOURCES.txt"); do FILES="$FILES $SRC/$f"; done
g++ -std=c++17 -O2 -g -fPIC -shared -Wl,-Bsymbolic -x c++ -DMPPO_EMU=1 -I"$HERE" -I"$SRC" -Wno-attributes -Wno-unused-value "$@" \
    $FILES "$HERE/emu_runtime.cpp" $( [ -f "$HERE/emu_stubs.cpp" ] && echo "$HERE/emu_stubs.cpp" ) -o "$OUT.tmp.$$"
mv -f "$OUT.tmp.$$" "$OUT"   # (atomic: another process - a rank of a multi-process test - may be loading the library while this one rebuilds it)
echo "built $OUT"
