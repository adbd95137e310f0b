"""ctypes wrapper of oracle/cpu_ref.cpp, the plain C++/OpenMP restatement of the reference step (TEST INFRASTRUCTURE
ONLY: bench.py's second cpu_baseline variant and tests/test_cpu_ref.py).  Build with `make -C oracle`."""
import ctypes as C
import os

import numpy as np

from . import lstm_oracle as O

LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_build', 'libcpuref.so')


def build():
    import subprocess
    subprocess.run(["make", "-C", os.path.dirname(os.path.dirname(LIB))], check=True,
                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return LIB


class CpuRef(object):
    def __init__(self, config, params, threads=0, clip_norm_mode='tf1_slices'):
        if not os.path.isfile(LIB):
            raise RuntimeError('oracle/_build/libcpuref.so not built (make -C oracle)')
        lib = C.CDLL(LIB)
        lib.cpuref_create.restype = C.c_void_p
        lib.cpuref_create.argtypes = [C.c_int] * 5 + [C.c_float] * 3 + [C.c_int] * 2
        lib.cpuref_destroy.argtypes = [C.c_void_p]
        lib.cpuref_param.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_float), C.c_longlong, C.c_int]
        lib.cpuref_eval.restype = C.c_float
        lib.cpuref_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.cpuref_train.restype = C.c_float
        lib.cpuref_train.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        self._lib, self.threads, self.config = lib, int(threads), dict(config)
        d = O.model_dims(config)
        self._h = lib.cpuref_create(d['start'], d['T'], d['E'], d['H'], d['L'], float(config['lr']), float(config['max_grad_norm']),
                                    float(config['n_decay']), int(clip_norm_mode == 'tf1_slices'), int(threads))
        self.shapes = dict(O.param_shapes(config))
        for k, v in params.items():
            a = np.ascontiguousarray(v, dtype=np.float32)
            assert lib.cpuref_param(self._h, k.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size, 1) == 0, k

    def get_params(self):
        out = {}
        for k, shape in self.shapes.items():
            a = np.empty(shape, np.float32)
            assert self._lib.cpuref_param(self._h, k.encode(), a.ctypes.data_as(C.POINTER(C.c_float)), a.size, 0) == 0
            out[k] = a
        return out

    @staticmethod
    def _rows(tokens):
        a = np.ascontiguousarray(O.flatten_first_two_dims(tokens), dtype=np.int32)
        return a, a.shape[0]

    def train(self, support, query):
        s, ns = self._rows(support)
        q, nq = self._rows(query)
        return float(self._lib.cpuref_train(self._h, s.ctypes.data, ns, q.ctypes.data, nq))

    def eval(self, query):
        q, nq = self._rows(query)
        return float(self._lib.cpuref_eval(self._h, q.ctypes.data, nq))

    def __del__(self):
        try:
            self._lib.cpuref_destroy(self._h)
        except Exception:
            pass
