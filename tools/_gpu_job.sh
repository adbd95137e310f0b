#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O; cd $R
cat > /tmp/dbg.py <<'PY'
import sys, os
sys.path.insert(0, os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'few-shot-music-generation_amd', 'src')); sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from fsmg.binding import FsmgModel
for hid, ms in ((1024, 45), (1024, 64), (512, 100), (1024, 33)):
    cfg = dict(input_size=500, embedding_size=32, hidden_size=hid, n_layers=2, max_len=16, lr=1e-3, lr_decay=0.5, n_decay=1000, max_grad_norm=5.0, seed=1)
    m = FsmgModel(cfg, max_sequences=ms)
    print(hid, ms, 'xcd_bx3', m.debug_read('xcd_bx3', 1)[0])
PY
echo normal; python /tmp/dbg.py 2>&1 | tail -4
echo ubsan; UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 FSMG_LIB=$R/few-shot-music-generation_amd/lib/libfsmg_ubsan.so LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.ubsan_standalone-x86_64.so) python /tmp/dbg.py 2>&1 | tail -6
