"""The C-ABI library: builds for gfx950, loads without a GPU, exports exactly what include/fsmg.h
declares, and refuses to run without a device (no CPU fallback).  No compute calls here."""
import os
import re
import subprocess

import pytest

from conftest import ROOT, small_config

HEADER = os.path.join(ROOT, 'include', 'fsmg.h')


@pytest.fixture(scope='module')
def lib():
    from fsmg.build import build
    build()
    from fsmg.binding import load_library
    return load_library()


def _declared():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(fsmg_[a-z_]+)\s*\(', text)))


def test_header_and_binding_and_library_agree(lib):
    from fsmg.binding import SIGNATURES, library_path
    declared = _declared()
    assert declared == sorted(SIGNATURES), (set(declared) ^ set(SIGNATURES))
    out = subprocess.check_output(['nm', '-D', '--defined-only', library_path()], universal_newlines=True)
    exported = sorted(set(re.findall(r' T (fsmg_[a-z_]+)$', out, flags=re.M)))
    assert exported == declared, (set(exported) ^ set(declared))
    assert lib.fsmg_version() == 600


def test_library_contains_gfx950_code_object():
    from fsmg.binding import library_path
    blob = open(library_path(), 'rb').read()
    assert b'gfx950' in blob and b'k_lstm_fwd_step' in blob and b'k_gemm' in blob


def test_state_bytes_matches_padded_layout(lib):
    from fsmg.binding import FsmgModel
    cfg = small_config(input_size=10000, max_len=128, embedding_size=250, hidden_size=512)
    Ep, Hp, V1p, V1 = 256, 512, 10004, 10001
    r64 = lambda n: (n + 63) // 64 * 64
    n_flat = r64(V1 * Ep) + r64((Ep + Hp) * 4 * Hp) + r64(4 * Hp) + r64(Hp * V1p) + r64(V1p)
    assert FsmgModel.state_bytes(cfg) == (4 * n_flat + 16) * 4


@pytest.mark.skipif(os.path.exists('/dev/kfd'), reason='a GPU is present')
def test_create_fails_loudly_without_a_device(lib):
    from fsmg.binding import FsmgError, FsmgModel
    with pytest.raises(FsmgError, match='NO_DEVICE'):
        FsmgModel(small_config())


@pytest.mark.skipif(os.path.exists('/dev/kfd'), reason='a GPU is present')
def test_plugin_has_no_cpu_fallback(lib):
    from fsmg.binding import FsmgError
    from models.lstm_baseline import LSTMBaseline
    with pytest.raises(FsmgError, match='no CPU fallback'):
        LSTMBaseline(small_config())


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'few-shot-music-generation_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text, os.path.join(dirpath, f)


def test_chain_kernels_keep_their_landing_registers():
    """tools/check_xcd_asm.py on a fresh hipcc -S of csrc/lstm_xcd.hip: the row-group-chain kernels' in-flight loads land in
    fixed registers nothing else touches, and the compiler put no vmcnt wait of its own into their time loops."""
    import subprocess
    import sys
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_xcd_asm.py')], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, universal_newlines=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-3000:]
