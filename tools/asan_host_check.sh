#!/bin/bash
# Build the HOST side of libmi355_carla.so with AddressSanitizer (the gfx950 code objects are unchanged) and drive its GPU-free entry points under it.
# usage: tools/asan_host_check.sh      (CPU container; a few minutes for the first build)
cd "$(dirname "$0")/.." || exit 1
python carla-ppo_amd/mi355/build.py --asan || exit 1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:abort_on_error=1:handle_segv=0 LD_PRELOAD=$RT MI355_LIB=$PWD/build/asan/libmi355_carla.so python tools/asan_host_check.py
