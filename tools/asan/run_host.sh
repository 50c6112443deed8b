#!/bin/bash
# AddressSanitizer over the HOST side of the library (the GPU side cannot be instrumented on this pool: no xnack, no GPU ASan).
# Builds libsicelore_mi_hostasan.so (host code instrumented, device code as shipped: -fno-gpu-sanitize) and runs the CPU tests that go
# through the host-side entry points -- BGZF / BAM / FASTQ index / gene tagger / gene counts / finalize / read names / clustering /
# region grouping / the host gzip decoder -- and the malformed-input fuzz (tests/test_host_fuzz.py) with the sanitizer's runtime preloaded.
# A report aborts the test run.  usage: tools/asan/run_host.sh [extra pytest args]
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
rt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
make -s -j8 -C "$root/sicelore-2.1_amd/csrc" ARCH=gfx950 VARIANT=hostasan EXTRA="-fsanitize=address -fno-gpu-sanitize -shared-libsan -g -fno-omit-frame-pointer"
cd "$root"
LD_PRELOAD=$rt ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 SMI_LIBRARY=$root/sicelore-2.1_amd/csrc/libsicelore_mi_hostasan.so \
  python -m pytest tests/test_host_fuzz.py tests/test_bam.py tests/test_hostio.py tests/test_gene_tagger.py tests/test_gene_counts.py tests/test_capi_cpu.py \
  tests/test_finalize.py tests/test_inflate_host.py tests/test_scan_stats.py tests/test_read_name.py tests/test_cluster.py tests/test_group.py \
  -x -q -p no:cacheprovider "$@"
