#!/bin/bash
# The randomised differential scenarios of tests/stress.py on the AddressSanitizer build of the
# kernel source under the CPU shim (tests/emu, `make asan`): out-of-bounds reads and writes of
# device buffers, which a GPU would swallow silently, stop the run.
#   tests/emu_asan_stress.sh [seconds] [seed] [team]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
make -C "$ROOT/tests/emu" --no-print-directory asan >/dev/null
export MM_STRESS_ENGINE=emu_small MM_EMU_SMALL_LIB="$ROOT/tests/emu/libmm_engine_emu_small_asan.so"
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1
export LD_PRELOAD="$(gcc -print-file-name=libasan.so)"
exec python "$ROOT/tests/stress.py" "$@"
