#!/bin/bash
# Host side of liboa_icp.so under the sanitizers ("race detection / sanitizers"):
# worker pool, condition variables, the process-wide allocation cache, thread-local error state, per-device threads, the
# multi-device loop with its fault hooks.  Device code is NOT instrumented (-fno-gpu-sanitize).
#   tools/sanitize_host.sh build <asan|ubsan|tsan>        here (no GPU needed): tools/_san/liboa_icp_<kind>.so
#   tools/sanitize_host.sh run <asan|ubsan|tsan> [pytest args]   on the GPU box: the GPU suite (or the given selection) on that build
# asan = AddressSanitizer + UBSan.  ROCm's compiler-rt intercepts hsa_amd_memory_pool_allocate for its device-side ASan; with
# the stock (uninstrumented) ROCr of this image that interceptor fails at the first device allocation ("out of memory:
# allocator is trying to allocate 0x400000 bytes"), so on this image only ubsan and tsan run (profiles/r04d_sanitizers.txt).
REPO="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
SAN="$REPO/tools/_san"; mkdir -p "$SAN"
KIND="${2:-ubsan}"
RTDIR=$(dirname "$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)")
case "$KIND" in
  asan)  FLAGS="-fsanitize=address,undefined"; RT="$RTDIR/libclang_rt.asan-x86_64.so" ;;
  ubsan) FLAGS="-fsanitize=undefined";         RT="$RTDIR/libclang_rt.ubsan_standalone-x86_64.so" ;;
  tsan)  FLAGS="-fsanitize=thread";            RT="$RTDIR/libclang_rt.tsan-x86_64.so" ;;
  *) echo "kind: asan | ubsan | tsan"; exit 2 ;;
esac
case "$1" in
build)
  # the HOST unit instrumented (all the host code lives there), linked with the kernel families' ordinary objects
  # (build/obj/oa_fam_*.o: device code + launch stubs only; `python -c "import __graft_entry__ as g; g.build_hip()"` makes them)
  FAMS=$(ls "$REPO"/build/obj/oa_fam_*.o | grep -v "oa_fam_exp")
  cd "$REPO/object_alignment_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -g -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form \
      -fno-slp-vectorize -fPIC -fvisibility=hidden -pthread $FLAGS -fno-gpu-sanitize -fno-omit-frame-pointer -c oa_icp.hip -o "$SAN/oa_icp_$KIND.o" && \
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden -pthread $FLAGS -fno-gpu-sanitize -shared-libsan \
      -o "$SAN/liboa_icp_$KIND.so" "$SAN/oa_icp_$KIND.o" $FAMS && ls -la "$SAN/liboa_icp_$KIND.so" ;;
driver)
  # the C driver (tools/san_driver.c) against the instrumented library: no Python in the process (TSan cannot be preloaded
  # into this image's python; it runs here)
  /opt/rocm/lib/llvm/bin/clang -O1 -g $FLAGS -shared-libsan -fno-omit-frame-pointer -I "$REPO/include" "$REPO/tools/san_driver.c" \
      -o "$SAN/san_driver_$KIND" -L "$SAN" -l:liboa_icp_$KIND.so -lm -lpthread -Wl,-rpath,"$SAN" -Wl,-rpath,"$RTDIR" || exit 1
  mkdir -p "$REPO/gpurun_out"; rm -f "$REPO"/gpurun_out/sand_$KIND.*
  cd "$REPO" && GPU_MAX_HW_QUEUES=12 OA_MULTI_THREADS=1 OA_MULTI_OWN_STREAMS=1 OA_EXCHANGE_TIMEOUT_S=5 \
    LD_LIBRARY_PATH="/opt/rocm/lib:$LD_LIBRARY_PATH" \
    ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:halt_on_error=0:log_path=$REPO/gpurun_out/sand_$KIND" \
    UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0:log_path=$REPO/gpurun_out/sand_$KIND" \
    TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:ignore_noninstrumented_modules=1:second_deadlock_stack=1:log_path=$REPO/gpurun_out/sand_$KIND" \
    "$SAN/san_driver_$KIND" 2>&1 | tail -8
  n=$(ls "$REPO"/gpurun_out/ | grep -c "^sand_$KIND\.")
  echo "--- $KIND (C driver): $n report file(s); lines that mention liboa_icp / oa_icp.hip: $(cat "$REPO"/gpurun_out/sand_$KIND.* 2>/dev/null | grep -c 'oa_icp')"
  cat "$REPO"/gpurun_out/sand_$KIND.* 2>/dev/null | grep -E "SUMMARY|runtime error" | sort | uniq -c | sort -rn | head -20 ;;
run)
  shift; shift
  mkdir -p "$REPO/gpurun_out"; rm -f "$REPO"/gpurun_out/san_$KIND.*
  cd "$REPO" && OA_ICP_LIB_DEBUG=1 OA_ICP_LIB="$SAN/liboa_icp_$KIND.so" LD_PRELOAD="$RT" \
    ASAN_OPTIONS="detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:halt_on_error=0:log_path=$REPO/gpurun_out/san_$KIND" \
    UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=0:log_path=$REPO/gpurun_out/san_$KIND" \
    TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:ignore_noninstrumented_modules=1:second_deadlock_stack=1:log_path=$REPO/gpurun_out/san_$KIND" \
    python -m pytest "${@:-tests}" -q -m gpu -p no:cacheprovider 2>&1 | tail -8
  n=$(ls "$REPO"/gpurun_out/ | grep -c "^san_$KIND\.")
  echo "--- $KIND: $n report file(s); lines that mention liboa_icp / oa_icp.hip: $(cat "$REPO"/gpurun_out/san_$KIND.* 2>/dev/null | grep -c 'oa_icp')"
  cat "$REPO"/gpurun_out/san_$KIND.* 2>/dev/null | grep -E "SUMMARY|runtime error" | sort | uniq -c | sort -rn | head -20 ;;
*) echo "usage: $0 build|run|driver <asan|ubsan|tsan> [pytest args]"; exit 2 ;;
esac
