#!/bin/bash
# AddressSanitizer + UBSan over the CPU builds (GPU sanitizers are not available on this pool): the oracle,
# the host-side operator state machine and the device arithmetic compiled for the host (tests/twin).
set -e
cd "$(dirname "$0")/.."
T=${TMPDIR:-/tmp}
SAN="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared"
g++ $SAN -std=c++17 -ffp-contract=off -fno-fast-math -I/opt/rocm/include -o $T/librpf_twin_asan.so tests/twin/rpf_twin.cpp rawphotoforge_amd/csrc/rpf_tables.cpp -lm
gcc $SAN -ffp-contract=off -o $T/librpf_oracle_asan.so oracle/rpf_oracle.c -lquadmath -lm
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
  RPF_ASAN_DIR=$T python tools/sanitize_cpu.py
