"""Build libfsmg.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

PKG_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(PKG_DIR, 'csrc')
# FSMG_LIB: load another build of the same ABI (the host-sanitizer builds of `make san`, tools/sanitize_run.sh)
LIB = os.environ.get('FSMG_LIB') or os.path.join(PKG_DIR, 'lib', 'libfsmg.so')


def build(verbose=False, jobs=None):
    """make -C csrc; returns the path of the shared library."""
    jobs = jobs or min(8, os.cpu_count() or 1)
    proc = subprocess.run(['make', '-C', CSRC, '-j%d' % jobs], stdout=subprocess.PIPE,
                          stderr=subprocess.STDOUT, universal_newlines=True)
    if verbose or proc.returncode != 0:
        print(proc.stdout)
    if proc.returncode != 0 or not os.path.isfile(os.path.join(PKG_DIR, 'lib', 'libfsmg.so')):
        raise RuntimeError('building libfsmg.so failed (hipcc --offload-arch=gfx950):\n' + proc.stdout[-4000:])
    return LIB


if __name__ == '__main__':
    print(build(verbose=True))
