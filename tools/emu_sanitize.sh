#!/bin/bash
# tools/emu_sanitize.sh [pytest arguments...] - the kernel sources under AddressSanitizer + UndefinedBehaviorSanitizer: the SIMT emulator build
# (tests/emu/build_emu.sh: csrc/*.hip compiled by g++, lanes as fibers) is rebuilt with -fsanitize=address,undefined and the emulator-backed
# parity tests run against it (GPU sanitizers are not available on this pool; the emulator executes the same indexing, LDS carving and table
# walks as the device code).  Output is NOT captured (-s): UBSan reports are printed, not fatal.  The normal build is restored afterwards.
#   tools/emu_sanitize.sh tests/test_convex_pairs.py tests/test_cylinder.py tests/test_kernels_physics.py -k "forward_matches or export_style or geom_pair"
cd "$(dirname "$0")/.." || exit 1
LIB=tests/emu/libminppo_emu.so
cp "$LIB" /tmp/libminppo_emu.keep.$$ 2>/dev/null
bash tests/emu/build_emu.sh -fsanitize=address,undefined -fno-omit-frame-pointer -O1 || exit 1
ASAN=$(gcc -print-file-name=libasan.so); UBSAN=$(gcc -print-file-name=libubsan.so)
LOG=${EMU_SANITIZE_LOG:-/tmp/emu_sanitize.log}
LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 UBSAN_OPTIONS=print_stacktrace=1 \
  python -m pytest -x -q -s -m "not gpu" "$@" > "$LOG" 2>&1
rc=$?
if [ -f /tmp/libminppo_emu.keep.$$ ]; then mv /tmp/libminppo_emu.keep.$$ "$LIB"; else bash tests/emu/build_emu.sh > /dev/null; fi
echo "pytest exit $rc; UBSan reports: $(grep -c 'runtime error' "$LOG"); AddressSanitizer reports: $(grep -c 'ERROR: AddressSanitizer' "$LOG"); log: $LOG"
tail -n 2 "$LOG"
exit $rc
