#!/bin/bash
# CPU-side sanitizer pass (GPU AddressSanitizer is not available on this pool): the kernels' per-lane bodies as compiled
# for the CPU emulation harness, and the C oracle, under -fsanitize=address,undefined; then the host and oracle test files.
# Found so far: one left shift of a negative value in the binary-GCD inner loop (csrc/fe.hpp), since fixed.
set -e
cd "$(dirname "$0")/.."
ASAN=$(gcc -print-file-name=libasan.so)
FLAGS="-O1 -g -fPIC -shared -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer"
cp tests/emu/libp2e_emu.so /tmp/p2e_emu_orig.so; cp oracle/libp2e_oracle.so /tmp/p2e_oracle_orig.so
trap 'cp /tmp/p2e_emu_orig.so tests/emu/libp2e_emu.so; cp /tmp/p2e_oracle_orig.so oracle/libp2e_oracle.so' EXIT
(cd tests/emu && g++ -std=c++17 $FLAGS -Wall -Wno-unused-function -o libp2e_emu.so p2e_emu.cpp)
(cd oracle && gcc -std=gnu11 $FLAGS -Wall -Wextra -o libp2e_oracle.so p2e_oracle.c)
LD_PRELOAD=$ASAN ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest tests/test_oracle.py tests/test_host.py tests/test_curve_programs.py -x -q -m "not gpu" -k "not c_client and not bingcd_inversion_round and not against_barrett"
