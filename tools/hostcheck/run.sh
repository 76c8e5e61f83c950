#!/bin/bash
# Development: the library's host code under sanitizers (device code as always).
#   bash tools/hostcheck/run.sh build   (here, no GPU needed): exp/libsxe_asan.so + exp/api_walk_asan
#                                       (tools/hostcheck/api_walk.cpp), exp/libsxe_ubsan.so
#   bash tools/hostcheck/run.sh run     (on the GPU box): the API walk under the address sanitizer, then the
#                                       GPU test suite and a few thousand random scenes with the library
#                                       built with the undefined-behaviour sanitizer (loaded through SXE_LIB)
cd "$(dirname "$0")/../.." || exit 1
RT=$(dirname "$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)")
case "$1" in
build)
  mkdir -p exp
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -ffp-contract=off -fPIC -shared -std=c++17 -Wall -Wno-unused-function \
      -Xarch_host -fsanitize=address -o exp/libsxe_asan.so renderer_amd/csrc/sxe_api.hip renderer_amd/csrc/sxe_geometry.hip \
      renderer_amd/csrc/sxe_raster.hip -L"$RT" -l:libclang_rt.asan-x86_64.so -Wl,-rpath,"$RT" || exit 1
  /opt/rocm/lib/llvm/bin/clang++ -std=c++17 -O1 -g -fsanitize=address -shared-libsan -Wall -Wextra -I include tools/hostcheck/api_walk.cpp \
      -o exp/api_walk_asan -L exp -l:libsxe_asan.so -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$RT" -Wl,-rpath-link,/opt/rocm/lib || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -g -ffp-contract=off -fPIC -shared -std=c++17 -Wall -Wno-unused-function \
      -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -o exp/libsxe_ubsan.so renderer_amd/csrc/sxe_api.hip \
      renderer_amd/csrc/sxe_geometry.hip renderer_amd/csrc/sxe_raster.hip -L"$RT" -l:libclang_rt.ubsan_standalone-x86_64.so -Wl,-rpath,"$RT" || exit 1
  ;;
run)
  ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 timeout -k 10 300 ./exp/api_walk_asan || exit 1
  export SXE_LIB=$PWD/exp/libsxe_ubsan.so UBSAN_OPTIONS=print_stacktrace=1
  timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread || exit 1
  timeout -k 10 300 python -u tools/fuzz_parity.py 3000 3401 | tail -1 || exit 1
  timeout -k 10 300 python -u tools/fuzz_paths.py 600 3402 | tail -1
  ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
