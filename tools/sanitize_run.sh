#!/bin/bash
# Runs GPU parity tests against the host-sanitizer builds of the C-ABI shim (make -C few-shot-music-generation_amd/csrc san).
#   gpurun -- tools/sanitize_run.sh      (results: gpurun_out/san_ubsan.log, gpurun_out/san_asan.log)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TESTS="tests/test_gpu_parity.py::test_forward_backward_every_tensor tests/test_gpu_parity.py::test_ten_update_trajectory tests/test_gpu_parity.py::test_eval_batch_equals_eval_steps_and_chunks tests/test_gpu_parity.py::test_errors tests/test_gpu_parity.py::test_param_roundtrip_and_opt_state_roundtrip tests/test_gpu_parity.py::test_persistent_kernel_timeout_falls_back_and_repeats_the_step tests/test_gpu_parity.py::test_maml_step_and_eval_match_oracle tests/test_gpu_parity.py::test_indexed_step_on_a_device_resident_table_equals_the_token_step tests/test_gpu_parity.py::test_sample_matches_oracle_greedy_decode"
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 FSMG_LIB=$R/few-shot-music-generation_amd/lib/libfsmg_ubsan.so \
  LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so 2>/dev/null || true) \
  timeout 600 python -m pytest $TESTS -x -q -m gpu > $O/san_ubsan.log 2>&1; echo "ubsan rc=$?" >> $O/san_ubsan.log
ASAN_RT=/nonexistent
[ -f "$ASAN_RT" ] || ASAN_RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1 FSMG_LIB=$R/few-shot-music-generation_amd/lib/libfsmg_asan.so LD_PRELOAD=$ASAN_RT \
  timeout 900 python -m pytest $TESTS -x -q -m gpu > $O/san_asan.log 2>&1; echo "asan rc=$?" >> $O/san_asan.log
tail -n 4 $O/san_ubsan.log; tail -n 4 $O/san_asan.log
