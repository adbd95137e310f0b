#!/bin/bash
# one iteration on the XCD-local bf16-split recurrences: rebuild the library, the experiment objects and tools/xcd_chain_bench, run it on the GPU box
# and print the [4] / [5] lines.   usage: tools/p16_iter.sh <variant> <tag> [HID=1024|512] [extra env, e.g. "RPX=15 CONC=1"]
set -e
cd /root/repo/few-shot-music-generation_amd/csrc
make -j6 2>&1 | grep -i "error" -A5 | head -20 || true
make experiments 2>&1 | grep -i "error" -A5 | head -20 || true
cd /root/repo
hipcc -w --offload-arch=gfx950 -O3 -std=c++17 -Ifew-shot-music-generation_amd/csrc -Iinclude -c tools/xcd_chain_bench.cpp -o /tmp/xcb.o
hipcc --offload-arch=gfx950 /tmp/xcb.o few-shot-music-generation_amd/build/exp/lstm_xcd.o few-shot-music-generation_amd/build/exp/lstm_step.o few-shot-music-generation_amd/build/exp/gemm.o -o tools/xcd_chain_bench.bin
HID=${3:-1024}
/usr/local/graft/bin/gpurun --timeout 600 -- "cd \$GRAFT_REPO_ROOT; mkdir -p gpurun_out; HID=$HID BX3=1 VARIANT=$1 $4 timeout 400 tools/xcd_chain_bench.bin > gpurun_out/p16_$2.log 2>&1; echo rc=\$? >> gpurun_out/p16_$2.log" 2>&1 | grep "status\|left"
grep "B=45\|phase\|wave\|rc=\|FAIL\|CONC" gpurun_out/p16_$2.log | head -70
