#!/bin/bash
# builds and runs the sanitizer fuzz of the host gzip decoder in a scratch directory (the decoder's source is copied next to the stub
# smi_internal.h, because a quoted #include looks in the including file's own directory first)
set -e
here=$(cd "$(dirname "$0")" && pwd)
work=${1:-/tmp/smi_asan}
mkdir -p "$work"
cp "$here/host_inflate_fuzz.cpp" "$here/smi_internal.h" "$here/../../sicelore-2.1_amd/csrc/smi_inflate_host.hip" "$work/"
cd "$work"
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -I. -I"$here/../../include" -x c++ host_inflate_fuzz.cpp -o host_inflate_fuzz -lz -lpthread
ASAN_OPTIONS=detect_leaks=0 ./host_inflate_fuzz
