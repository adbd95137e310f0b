"""ctypes binding of libfsmg -- exactly the entry points declared in include/fsmg.h.

This is "the reference-side binding a maintainer would add" (INTEGRATION.md): the
`models.lstm_baseline.LSTMBaseline` plugin calls these instead of a TensorFlow session.
"""
import ctypes as C
import os

import numpy as np

from fsmg.build import LIB

FSMG_GRAD_TAIL = 16
CLIP_MODES = {'tf1_slices': 0, 'dense': 1}
FSMG_CONFIG_VERSION = 4
GEMM_KINDS = {'auto': 0, 'bx3': 1, 'f32': 2}
SCHEDULES = {'auto': 0, 'single_stream': 1, 'two_stream': 2, 'xcd_partitioned': 3}
RECURRENCES = {'auto': 0, 'per_step': 1, 'column_split': 2, 'xcd_local': 3}

ERRORS = {-1: 'FSMG_ERR_INVALID', -2: 'FSMG_ERR_NO_DEVICE', -3: 'FSMG_ERR_HIP', -4: 'FSMG_ERR_NOMEM',
          -5: 'FSMG_ERR_NAME', -6: 'FSMG_ERR_SIZE', -7: 'FSMG_ERR_TOKEN_RANGE', -8: 'FSMG_ERR_STATE',
          -9: 'FSMG_ERR_TIMEOUT', -10: 'FSMG_ERR_SOFTMAX_RANGE'}
# the step was skipped on the device and the handle has changed how it runs the next one: repeat the call (include/fsmg.h)
RETRY_CODES = (-9, -10)


class FsmgError(RuntimeError):
    def __init__(self, code, message):
        super(FsmgError, self).__init__('%s: %s' % (ERRORS.get(code, code), message))
        self.code = code


class FsmgConfig(C.Structure):
    _fields_ = [('input_size', C.c_int32), ('max_len', C.c_int32), ('embedding_size', C.c_int32),
                ('hidden_size', C.c_int32), ('n_layers', C.c_int32), ('lr', C.c_float),
                ('max_grad_norm', C.c_float), ('n_decay', C.c_float), ('clip_norm_mode', C.c_int32),
                ('device', C.c_int32), ('max_sequences', C.c_int32), ('use_graph', C.c_int32),
                ('stream', C.c_void_p), ('state_arena', C.c_void_p), ('state_arena_bytes', C.c_uint64),
                ('config_version', C.c_int32), ('gemm', C.c_int32), ('schedule', C.c_int32), ('recurrence', C.c_int32),
                ('dp_split_backward', C.c_int32), ('reserved', C.c_int32 * 7)]


class FsmgStats(C.Structure):
    _fields_ = [('timeouts', C.c_int64), ('steps_skipped_timeout', C.c_int64), ('steps_skipped_token_range', C.c_int64),
                ('xcd_launches', C.c_int64), ('persistent_launches', C.c_int64), ('step_launches', C.c_int64),
                ('persistent_path', C.c_int32), ('fallback_steps_left', C.c_int32), ('steps_skipped_peer_failure', C.c_int64),
                ('xov_selfcheck_mismatches', C.c_int64), ('softmax_range_rows', C.c_int64), ('steps_skipped_softmax_range', C.c_int64),
                ('aux_stream_tries', C.c_int32), ('reserved0', C.c_int32)]


_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)
_F32P = C.POINTER(C.c_float)
# name -> (restype, argtypes); must list every symbol include/fsmg.h declares
SIGNATURES = {
    'fsmg_version': (C.c_int, []),
    'fsmg_last_error': (C.c_char_p, [_P]),
    'fsmg_state_bytes': (C.c_uint64, [C.POINTER(FsmgConfig)]),
    'fsmg_create': (C.c_int, [C.POINTER(FsmgConfig), C.POINTER(_P)]),
    'fsmg_destroy': (C.c_int, [_P]),
    'fsmg_synchronize': (C.c_int, [_P]),
    'fsmg_init_params': (C.c_int, [_P, C.c_uint64]),
    'fsmg_num_params': (C.c_int, [_P]),
    'fsmg_param_info': (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'fsmg_set_param': (C.c_int, [_P, C.c_char_p, _F32P, C.c_int64]),
    'fsmg_get_param': (C.c_int, [_P, C.c_char_p, _F32P, C.c_int64]),
    'fsmg_set_opt_state': (C.c_int, [_P, C.c_char_p, _F32P, _F32P, C.c_int64]),
    'fsmg_get_opt_state': (C.c_int, [_P, C.c_char_p, _F32P, _F32P, C.c_int64]),
    'fsmg_set_step': (C.c_int, [_P, C.c_int64]),
    'fsmg_get_step': (C.c_int, [_P, C.POINTER(C.c_int64)]),
    'fsmg_get_grad': (C.c_int, [_P, C.c_char_p, _F32P, C.c_int64]),
    'fsmg_train_step': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F32P]),
    'fsmg_forward_backward': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    'fsmg_grad_buffer': (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    'fsmg_grad_bucket': (C.c_int, [_P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int64)]),
    'fsmg_stream_wait_bucket': (C.c_int, [_P, _P, C.c_int32]),
    'fsmg_apply_update': (C.c_int, [_P, C.c_float, _F32P]),
    'fsmg_comm_unique_id': (C.c_int, [C.c_char_p]),
    'fsmg_comm_init': (C.c_int, [_P, C.c_char_p, C.c_int32, C.c_int32]),
    'fsmg_comm_attach': (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    'fsmg_comm_broadcast_state': (C.c_int, [_P, C.c_int32]),
    'fsmg_comm_release': (C.c_int, [_P]),
    'fsmg_upload_table': (C.c_int, [_P, C.c_int32, _P, C.c_int64]),
    'fsmg_forward_backward_indexed': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32]),
    'fsmg_train_step_indexed': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, _F32P]),
    'fsmg_maml_forward_backward': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32]),
    'fsmg_maml_step': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _F32P]),
    'fsmg_maml_eval': (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, _F32P]),
    'fsmg_maml_forward_backward_indexed': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float]),
    'fsmg_maml_step_indexed': (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, _F32P]),
    'fsmg_eval_step': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _F32P]),
    'fsmg_eval_batch': (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _F32P]),
    'fsmg_sample': (C.c_int, [_P, C.c_int32, _I32P]),
    'fsmg_read_losses': (C.c_int, [_P, _F32P, C.c_int32]),
    'fsmg_get_stats': (C.c_int, [_P, C.POINTER(FsmgStats)]),
    'fsmg_debug_read': (C.c_int, [_P, C.c_char_p, _F32P, C.c_int64]),
    'fsmg_debug_dims': (C.c_int, [_P, _I32P]),
    'fsmg_debug_set': (C.c_int, [_P, C.c_char_p, C.c_int64]),
    'fsmg_debug_clock_begin': (C.c_int, [_P, C.c_int32]),
    'fsmg_debug_clock_end': (C.c_int, [_P, _F32P]),
    'fsmg_unigram_create': (C.c_int, [C.c_int32, C.c_int32, C.POINTER(_P)]),
    'fsmg_unigram_destroy': (C.c_int, [_P]),
    'fsmg_unigram_last_error': (C.c_char_p, [_P]),
    'fsmg_unigram_nll': (C.c_int, [_P, _P, C.c_int64, C.c_int32, _F32P]),
    'fsmg_unigram_train': (C.c_int, [_P, _P, C.c_int64, C.c_int32, _F32P]),
    'fsmg_unigram_get_counts': (C.c_int, [_P, _F32P, C.c_int64]),
    'fsmg_unigram_set_counts': (C.c_int, [_P, _F32P, C.c_int64]),
    'fsmg_unigram_argmax': (C.c_int, [_P, _I32P]),
    'fsmg_debug_step_profile': (C.c_int, [_P, C.c_int32, C.POINTER(C.c_uint64), C.c_int64, _I32P, _I32P]),
    'fsmg_timing_enable': (C.c_int, [_P, C.c_int32]),
    'fsmg_timing_select': (C.c_int, [_P, C.c_char_p]),
    'fsmg_timing_read': (C.c_int, [_P, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    'fsmg_timing_reset': (C.c_int, [_P]),
}

_lib = None


def library_path():
    return LIB


def load_library():
    """dlopen the in-tree libfsmg.so; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB):
            raise FsmgError(-2, 'libfsmg.so not built (%s); run `python -c "import __graft_entry__ as g; g.build()"` '
                                'or `make -C few-shot-music-generation_amd/csrc` -- there is no CPU fallback' % LIB)
        # PyTorch-ROCm ships its own libamdhip64; whichever HIP runtime is loaded first serves the whole
        # process.  Load torch's first (when torch is installed) so that torch.cuda / RCCL and libfsmg share
        # one runtime regardless of import order -- the reverse order leaves torch with "No HIP GPUs".
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def _f32p(a):
    return a.ctypes.data_as(_F32P)


def _tok_ptr(tokens):
    """host numpy int32 array or a raw device address (int) -> (void*, on_device, keepalive)"""
    if isinstance(tokens, (int, np.integer)):
        return C.c_void_p(int(tokens)), 1, None
    a = np.ascontiguousarray(tokens, dtype=np.int32)
    return C.c_void_p(a.ctypes.data), 0, a


class FsmgModel(object):
    """One model handle == one LSTM language model resident on one MI355X."""

    def __init__(self, config, device=0, stream=None, state_arena=None, state_arena_bytes=0,
                 max_sequences=0, clip_norm_mode='tf1_slices', use_graph=True, gemm=None, schedule=None, recurrence=None,
                 dp_split_backward=None):
        self._lib = load_library()
        # schedule / arithmetic knobs: explicit arguments, else optional keys of the model config, else the library's choice
        gemm = gemm or config.get('gemm', 'auto')
        schedule = schedule or config.get('schedule', 'auto')
        recurrence = recurrence or config.get('recurrence', 'auto')
        if dp_split_backward is None:
            dp_split_backward = config.get('dp_split_backward', False)
        self.cfg = FsmgConfig(
            input_size=int(config['input_size']), max_len=int(config['max_len']),
            embedding_size=int(config['embedding_size']), hidden_size=int(config['hidden_size']),
            n_layers=int(config['n_layers']), lr=float(config['lr']),
            max_grad_norm=float(config['max_grad_norm']), n_decay=float(config['n_decay']),
            clip_norm_mode=CLIP_MODES[clip_norm_mode], device=int(device), max_sequences=int(max_sequences),
            use_graph=int(bool(use_graph)), stream=stream, state_arena=state_arena,
            state_arena_bytes=int(state_arena_bytes), config_version=FSMG_CONFIG_VERSION, gemm=GEMM_KINDS[gemm],
            schedule=SCHEDULES[schedule], recurrence=RECURRENCES[recurrence], dp_split_backward=(2 if dp_split_backward == 2 and dp_split_backward is not True else int(bool(dp_split_backward))))
        self.max_len = int(config['max_len'])
        handle = _P()
        rc = self._lib.fsmg_create(C.byref(self.cfg), C.byref(handle))
        if rc != 0:
            raise FsmgError(rc, self._lib.fsmg_last_error(None).decode())
        self._h = handle
        self.param_shapes = {}
        for i in range(self._lib.fsmg_num_params(self._h)):
            name = C.create_string_buffer(64)
            rows, cols = C.c_int64(), C.c_int64()
            self._ck(self._lib.fsmg_param_info(self._h, i, name, 64, C.byref(rows), C.byref(cols)))
            self.param_shapes[name.value.decode()] = (rows.value, cols.value)

    @staticmethod
    def state_bytes(config):
        lib = load_library()
        cfg = FsmgConfig(input_size=int(config['input_size']), max_len=int(config['max_len']),
                         embedding_size=int(config['embedding_size']), hidden_size=int(config['hidden_size']),
                         n_layers=int(config['n_layers']), lr=1.0, max_grad_norm=1.0, n_decay=1.0,
                         config_version=FSMG_CONFIG_VERSION)
        return int(lib.fsmg_state_bytes(C.byref(cfg)))

    def _ck(self, rc):
        if rc != 0:
            raise FsmgError(rc, self._lib.fsmg_last_error(self._h).decode())

    def close(self):
        if getattr(self, '_h', None):
            self._lib.fsmg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters -------------------------------------------------------------------------
    def _shape(self, name):
        if name not in self.param_shapes:
            raise FsmgError(-5, "unknown parameter '%s'" % name)
        rows, cols = self.param_shapes[name]
        vector = name.startswith('bias_') or name == 'softmax_b'        # cols == 1 alone cannot tell a vector from an [n, 1] matrix (E = 1)
        return (rows,) if vector else (rows, cols)

    def init_params(self, seed):
        self._ck(self._lib.fsmg_init_params(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))

    def set_param(self, name, value):
        a = np.ascontiguousarray(value, dtype=np.float32)
        self._ck(self._lib.fsmg_set_param(self._h, name.encode(), _f32p(a), a.size))

    def get_param(self, name):
        out = np.empty(self._shape(name), np.float32)
        self._ck(self._lib.fsmg_get_param(self._h, name.encode(), _f32p(out), out.size))
        return out

    def set_params(self, params):
        for k, v in params.items():
            self.set_param(k, v)

    def get_params(self):
        return {k: self.get_param(k) for k in self.param_shapes}

    def get_grad(self, name):
        out = np.empty(self._shape(name), np.float32)
        self._ck(self._lib.fsmg_get_grad(self._h, name.encode(), _f32p(out), out.size))
        return out

    def get_opt_state(self, name):
        m = np.empty(self._shape(name), np.float32)
        v = np.empty(self._shape(name), np.float32)
        self._ck(self._lib.fsmg_get_opt_state(self._h, name.encode(), _f32p(m), _f32p(v), m.size))
        return m, v

    def set_opt_state(self, name, m, v):
        m = np.ascontiguousarray(m, dtype=np.float32)
        v = np.ascontiguousarray(v, dtype=np.float32)
        self._ck(self._lib.fsmg_set_opt_state(self._h, name.encode(), _f32p(m), _f32p(v), m.size))

    @property
    def step(self):
        s = C.c_int64()
        self._ck(self._lib.fsmg_get_step(self._h, C.byref(s)))
        return s.value

    @step.setter
    def step(self, value):
        self._ck(self._lib.fsmg_set_step(self._h, int(value)))

    # -- hot path -----------------------------------------------------------------------------
    def _episode_shape(self, support, query, shape):
        if shape is not None:
            return shape
        n, k, t = support.shape
        n2, q, t2 = query.shape
        if n != n2 or t != self.max_len or t2 != self.max_len:
            raise ValueError('support %r / query %r do not match max_len=%d' % (support.shape, query.shape, self.max_len))
        return n, k, q

    def train_step(self, support, query, shape=None, want_loss=True):
        """support [N,K,T], query [N,Q,T] int32 (numpy) -- or raw device addresses with shape=(N,K,Q)."""
        n, k, q = self._episode_shape(support, query, shape)
        sp, dev, _k1 = _tok_ptr(support)
        qp, _, _k2 = _tok_ptr(query)
        loss = C.c_float()
        self._ck(self._lib.fsmg_train_step(self._h, sp, qp, n, k, q, dev, C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def forward_backward(self, support, query, shape=None):
        n, k, q = self._episode_shape(support, query, shape)
        sp, dev, _k1 = _tok_ptr(support)
        qp, _, _k2 = _tok_ptr(query)
        self._ck(self._lib.fsmg_forward_backward(self._h, sp, qp, n, k, q, dev))

    def grad_buffer(self):
        ptr, count = _P(), C.c_int64()
        self._ck(self._lib.fsmg_grad_buffer(self._h, C.byref(ptr), C.byref(count)))
        return ptr.value, count.value

    def grad_bucket(self, bucket):
        ptr, count = _P(), C.c_int64()
        self._ck(self._lib.fsmg_grad_bucket(self._h, int(bucket), C.byref(ptr), C.byref(count)))
        return ptr.value, count.value

    def stream_wait_bucket(self, stream_handle, bucket):
        self._ck(self._lib.fsmg_stream_wait_bucket(self._h, C.c_void_p(stream_handle), int(bucket)))

    def apply_update(self, grad_scale=1.0, want_loss=True):
        loss = C.c_float()
        self._ck(self._lib.fsmg_apply_update(self._h, float(grad_scale), C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    # -- gradient exchange inside the library (RCCL) ---------------------------------------------------
    @staticmethod
    def comm_unique_id():
        '''rank 0: 128 opaque bytes to hand to every rank (ncclGetUniqueId)'''
        lib = load_library()
        buf = C.create_string_buffer(128)
        rc = lib.fsmg_comm_unique_id(buf)
        if rc != 0:
            raise FsmgError(rc, lib.fsmg_last_error(None).decode())
        return buf.raw

    def comm_init(self, unique_id, world_size, rank):
        '''collective: every rank of the job calls it with the id rank 0 made; from then on train_step / train_step_indexed /
        maml_step exchange the gradients themselves (grad_scale 1 / world_size)'''
        self._ck(self._lib.fsmg_comm_init(self._h, C.c_char_p(bytes(unique_id)), int(world_size), int(rank)))

    def comm_broadcast_state(self, root=0):
        self._ck(self._lib.fsmg_comm_broadcast_state(self._h, int(root)))

    def comm_release(self):
        self._ck(self._lib.fsmg_comm_release(self._h))

    # -- device-resident episode table ----------------------------------------------------------------
    def upload_table(self, table_id, table):
        a = np.ascontiguousarray(table, dtype=np.int32)
        if a.ndim != 2 or a.shape[1] != self.max_len:
            raise ValueError('token table %r does not match max_len=%d' % (a.shape, self.max_len))
        self._ck(self._lib.fsmg_upload_table(self._h, int(table_id), C.c_void_p(a.ctypes.data), a.shape[0]))

    @staticmethod
    def _idx(support_idx, query_idx):
        s = np.ascontiguousarray(support_idx, dtype=np.int32)
        q = np.ascontiguousarray(query_idx, dtype=np.int32)
        if s.ndim != 2 or q.ndim != 2 or s.shape[0] != q.shape[0]:
            raise ValueError('index arrays must be [N, K] and [N, Q]')
        return s, q

    def forward_backward_indexed(self, table_id, support_idx, query_idx):
        s, q = self._idx(support_idx, query_idx)
        self._ck(self._lib.fsmg_forward_backward_indexed(self._h, int(table_id), C.c_void_p(s.ctypes.data), C.c_void_p(q.ctypes.data),
                                                         s.shape[0], s.shape[1], q.shape[1]))

    def train_step_indexed(self, table_id, support_idx, query_idx, want_loss=True):
        s, q = self._idx(support_idx, query_idx)
        loss = C.c_float()
        self._ck(self._lib.fsmg_train_step_indexed(self._h, int(table_id), C.c_void_p(s.ctypes.data), C.c_void_p(q.ctypes.data),
                                                   s.shape[0], s.shape[1], q.shape[1], C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    # -- cfg-E: MAML-style inner / outer loop ------------------------------------------------------
    def maml_forward_backward(self, support, query, inner_steps, inner_lr, shape=None):
        n, k, q = self._episode_shape(support, query, shape)
        sp, dev, _k1 = _tok_ptr(support)
        qp, _, _k2 = _tok_ptr(query)
        self._ck(self._lib.fsmg_maml_forward_backward(self._h, sp, qp, n, k, q, int(inner_steps), float(inner_lr), dev))

    def maml_step(self, support, query, inner_steps, inner_lr, shape=None, want_loss=True):
        n, k, q = self._episode_shape(support, query, shape)
        sp, dev, _k1 = _tok_ptr(support)
        qp, _, _k2 = _tok_ptr(query)
        loss = C.c_float()
        self._ck(self._lib.fsmg_maml_step(self._h, sp, qp, n, k, q, int(inner_steps), float(inner_lr), dev,
                                          C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def maml_forward_backward_indexed(self, table_id, support_idx, query_idx, inner_steps, inner_lr):
        s, q = self._idx(support_idx, query_idx)
        self._ck(self._lib.fsmg_maml_forward_backward_indexed(self._h, int(table_id), C.c_void_p(s.ctypes.data), C.c_void_p(q.ctypes.data),
                                                              s.shape[0], s.shape[1], q.shape[1], int(inner_steps), float(inner_lr)))

    def maml_step_indexed(self, table_id, support_idx, query_idx, inner_steps, inner_lr, want_loss=True):
        s, q = self._idx(support_idx, query_idx)
        loss = C.c_float()
        self._ck(self._lib.fsmg_maml_step_indexed(self._h, int(table_id), C.c_void_p(s.ctypes.data), C.c_void_p(q.ctypes.data),
                                                  s.shape[0], s.shape[1], q.shape[1], int(inner_steps), float(inner_lr),
                                                  C.byref(loss) if want_loss else None))
        return loss.value if want_loss else None

    def maml_eval(self, support, query, inner_steps, inner_lr, shape=None):
        n, k, q = self._episode_shape(support, query, shape)
        sp, dev, _k1 = _tok_ptr(support)
        qp, _, _k2 = _tok_ptr(query)
        nll = C.c_float()
        self._ck(self._lib.fsmg_maml_eval(self._h, sp, qp, n, k, q, int(inner_steps), float(inner_lr), dev, C.byref(nll)))
        return nll.value

    def eval_step(self, query, shape=None):
        if shape is None:
            n, q, t = query.shape
            if t != self.max_len:
                raise ValueError('query %r does not match max_len=%d' % (query.shape, self.max_len))
        else:
            n, q = shape
        qp, dev, _k = _tok_ptr(query)
        nll = C.c_float()
        self._ck(self._lib.fsmg_eval_step(self._h, qp, n, q, dev, C.byref(nll)))
        return nll.value

    def eval_batch(self, queries, shape=None):
        """queries [n_episodes,N,Q,T] -> float32 [n_episodes]"""
        if shape is None:
            ne, n, q, t = queries.shape
            if t != self.max_len:
                raise ValueError('queries %r do not match max_len=%d' % (queries.shape, self.max_len))
        else:
            ne, n, q = shape
        qp, dev, _k = _tok_ptr(queries)
        out = np.empty(ne, np.float32)
        self._ck(self._lib.fsmg_eval_batch(self._h, qp, ne, n, q, dev, _f32p(out)))
        return out

    def sample(self, num):
        out = np.empty(max(int(num), 1), np.int32)
        self._ck(self._lib.fsmg_sample(self._h, int(num), out.ctypes.data_as(_I32P)))
        return [int(t) for t in out[:int(num)]]

    def read_losses(self, n):
        out = np.empty(n, np.float32)
        self._ck(self._lib.fsmg_read_losses(self._h, _f32p(out), n))
        return out

    def stats(self):
        st = FsmgStats()
        self._ck(self._lib.fsmg_get_stats(self._h, C.byref(st)))
        return {name: int(getattr(st, name)) for name, _ in FsmgStats._fields_}

    def synchronize(self):
        self._ck(self._lib.fsmg_synchronize(self._h))

    # -- introspection ---------------------------------------------------------------------
    def debug_dims(self):
        d = (C.c_int32 * 5)()
        self._ck(self._lib.fsmg_debug_dims(self._h, d))
        return dict(Ep=d[0], Hp=d[1], V1p=d[2], B=d[3], T=d[4])

    def debug_set(self, what, value):
        """run-time knob of the handle (include/fsmg.h: chain_spin_limit, fallback_steps, persistent, eager, inplace_dlogits,
        upd_split, xov_selfcheck, xov_selfcheck_fault)"""
        self._ck(self._lib.fsmg_debug_set(self._h, what.encode(), int(value)))

    def clock_begin(self, microseconds):
        """start the shader-clock probe (its own stream); issue the work to be measured next, then clock_end()"""
        self._ck(self._lib.fsmg_debug_clock_begin(self._h, int(microseconds)))

    def clock_end(self):
        ghz = C.c_float()
        self._ck(self._lib.fsmg_debug_clock_end(self._h, C.byref(ghz)))
        return float(ghz.value)

    def debug_read(self, what, count):
        out = np.empty(int(count), np.float32)
        self._ck(self._lib.fsmg_debug_read(self._h, what.encode(), _f32p(out), out.size))
        return out

    def step_profile(self, which):
        """-> uint64 [n_blocks, n_waves, 8] s_memtime stamps of one instrumented recurrent step kernel"""
        buf = np.zeros(1 << 20, np.uint64)
        nb, nw = C.c_int32(), C.c_int32()
        self._ck(self._lib.fsmg_debug_step_profile(self._h, int(which), buf.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                   buf.size, C.byref(nb), C.byref(nw)))
        return buf[:nb.value * nw.value * 8].reshape(nb.value, nw.value, 8)

    def timing_enable(self, on=True):
        self._ck(self._lib.fsmg_timing_enable(self._h, int(bool(on))))

    def timing_select(self, kernel_class=None):
        self._ck(self._lib.fsmg_timing_select(self._h, kernel_class.encode() if kernel_class else None))

    def timing_reset(self):
        self._ck(self._lib.fsmg_timing_reset(self._h))

    def timing_read(self, kernel_class):
        ms, n = C.c_double(), C.c_int64()
        self._ck(self._lib.fsmg_timing_read(self._h, kernel_class.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value


class FsmgUnigram(object):
    """Device-resident unigram counts (include/fsmg.h fsmg_unigram_*): the graph of the reference's UnigramModel
    (/root/reference/src/models/unigram_model.py:26-39) -- scatter_add histogram, gather / reduce_sum, -mean(log)."""

    def __init__(self, input_size, device=0):
        self._lib = load_library()
        self.input_size = int(input_size)
        handle = _P()
        rc = self._lib.fsmg_unigram_create(self.input_size, int(device), C.byref(handle))
        if rc != 0:
            raise FsmgError(rc, self._lib.fsmg_unigram_last_error(None).decode())
        self._h = handle

    def _ck(self, rc):
        if rc != 0:
            raise FsmgError(rc, self._lib.fsmg_unigram_last_error(self._h).decode())

    def close(self):
        if getattr(self, '_h', None):
            self._lib.fsmg_unigram_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _words(words):
        if isinstance(words, tuple):                  # (device address, count)
            return C.c_void_p(int(words[0])), int(words[1]), 1, None
        a = np.ascontiguousarray(words, dtype=np.int32).ravel()
        return C.c_void_p(a.ctypes.data), a.size, 0, a

    def nll(self, words):
        ptr, n, dev, keep = self._words(words)
        out = C.c_float()
        self._ck(self._lib.fsmg_unigram_nll(self._h, ptr, n, dev, C.byref(out)))
        return float(out.value)

    def train(self, words, want_loss=True):
        ptr, n, dev, keep = self._words(words)
        out = C.c_float()
        self._ck(self._lib.fsmg_unigram_train(self._h, ptr, n, dev, C.byref(out) if want_loss else None))
        return float(out.value) if want_loss else None

    def get_counts(self):
        out = np.empty(self.input_size, np.float32)
        self._ck(self._lib.fsmg_unigram_get_counts(self._h, _f32p(out), out.size))
        return out

    def set_counts(self, counts):
        a = np.ascontiguousarray(counts, dtype=np.float32)
        if a.size != self.input_size:
            raise ValueError('counts must have input_size entries')
        self._ck(self._lib.fsmg_unigram_set_counts(self._h, _f32p(a), a.size))

    def argmax(self):
        w = C.c_int32()
        self._ck(self._lib.fsmg_unigram_argmax(self._h, C.byref(w)))
        return int(w.value)
