#!/bin/bash
# AddressSanitizer + UBSan run of the host-side readers (libpymasc_io.so) over their CPU test-suite.
# GPU sanitizers are unavailable on the pool; this covers the native host code.
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer \
    -o /tmp/libpymasc_io_asan.so pymasc_amd/csrc/io/*.cpp -lz -pthread
ASAN=$(g++ -print-file-name=libasan.so)
STD=$(g++ -print-file-name=libstdc++.so.6)      # preloaded too, or ASan cannot intercept __cxa_throw under python
PYMASC_AMD_IO_LIB=/tmp/libpymasc_io_asan.so LD_PRELOAD="$ASAN $STD" ASAN_OPTIONS=detect_leaks=0 \
    python -m pytest tests/test_io_readers.py -x -q -s -m "not gpu" -k "not host" -p no:cacheprovider
