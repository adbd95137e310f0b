#!/bin/bash
# Runs GPU parity tests against the host-sanitizer builds of the C-ABI shim (make -C few-shot-music-generation_amd/csrc san).
#   gpurun -- 'bash tools/sanitize_run.sh'      (results: gpurun_out/san_ubsan.log, gpurun_out/san_asan_abi.log)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
TESTS="tests/test_gpu_parity.py::test_forward_backward_every_tensor tests/test_gpu_parity.py::test_ten_update_trajectory tests/test_gpu_parity.py::test_eval_batch_equals_eval_steps_and_chunks tests/test_gpu_parity.py::test_errors tests/test_gpu_parity.py::test_param_roundtrip_and_opt_state_roundtrip tests/test_gpu_parity.py::test_persistent_kernel_timeout_falls_back_and_repeats_the_step tests/test_gpu_parity.py::test_maml_step_and_eval_match_oracle tests/test_gpu_parity.py::test_indexed_step_on_a_device_resident_table_equals_the_token_step tests/test_gpu_parity.py::test_sample_matches_oracle_greedy_decode tests/test_gpu_parity.py::test_fused_softmax_matches_the_cross_entropy_pass tests/test_gpu_parity.py::test_fused_softmax_falls_back_when_a_logit_leaves_its_range tests/test_gpu_parity.py::test_xcd_partitioned_schedule_gives_the_same_bits tests/test_gpu_parity.py::test_split_update_gives_the_same_bits_and_every_reader_waits_for_it tests/test_gpu_parity.py::test_xov_selfcheck_passes_on_this_runtime_and_a_fault_parks_the_order tests/test_gpu_parity.py::test_lazy_column_split_copies_are_refreshed_before_anybody_reads_them tests/test_unigram.py"
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 FSMG_LIB=$R/few-shot-music-generation_amd/lib/libfsmg_ubsan.so \
  LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so 2>/dev/null || true) \
  timeout 600 python -m pytest $TESTS -x -q -m gpu > $O/san_ubsan.log 2>&1; echo "ubsan rc=$?" >> $O/san_ubsan.log
# ASan: PyTorch-ROCm exits silently at import under the ASan runtime, so the C-ABI is driven from plain C++ (tools/abi_asan_smoke.cpp; build it here:
#   hipcc -fsanitize=address -shared-libsan -g -Iinclude tools/abi_asan_smoke.cpp -Lfew-shot-music-generation_amd/lib -lfsmg_asan -o tools/abi_asan_smoke.bin)
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 LD_LIBRARY_PATH=$R/few-shot-music-generation_amd/lib:$(dirname $(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)) \
  timeout 600 $R/tools/abi_asan_smoke.bin > $O/san_asan_abi.log 2>&1; echo "asan rc=$?" >> $O/san_asan_abi.log
tail -n 4 $O/san_ubsan.log; tail -n 4 $O/san_asan_abi.log
