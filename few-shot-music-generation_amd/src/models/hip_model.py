"""Device plumbing shared by the HIP-backed plugins: handle creation, checkpoints, init.

Plays the role /root/reference/src/models/tf_model.py plays for the TensorFlow plugins
(session, Saver, initialisation) with the same observable behaviour:
  * save(dir)            -> <dir>/<name>/<name>-<global_step>.npz, newest 10 kept
                            (tf_model.py:96-97,106-114; file format is ours);
  * recover_or_init(dir) -> restore the latest checkpoint under <dir>/<name> if there is one --
                            every tensor whose name AND shape match ("optimistic restore",
                            tf_model.py:28-75) -- then initialise what was not restored
                            (Glorot-uniform / zeros); works with dir == ''.
A checkpoint holds what TF's held: weights, Adam m/v, global_step.
PyTorch is used for device memory, the stream and torch.distributed only.
"""
import glob
import os
import re

import numpy as np

from fsmg.binding import FsmgError, FsmgModel
from models.base_model import BaseModel

MAX_TO_KEEP = 10


def episode_sequences(config):
    """Sequences per training episode for fsmg_config.max_sequences: the explicit `max_sequences` key, else N x (K + Q) from the
    merged YAMLs (model: batch_size = N artists per episode, task: support_size = K, query_size = Q; reference
    src/data/episode.py:36-60 draws exactly that many songs), else 0 = the library's default (45)."""
    explicit = int(config.get('max_sequences', 0) or 0)
    if explicit > 0:
        return explicit
    if all(k in config for k in ('batch_size', 'support_size', 'query_size')):
        return int(config['batch_size']) * (int(config['support_size']) + int(config['query_size']))
    return 0


class HIPModel(BaseModel):
    def __init__(self, config):
        super(HIPModel, self).__init__(config)
        import torch
        if not torch.cuda.is_available():
            raise FsmgError(-2, 'no MI355X visible to PyTorch-ROCm: the HIP path has no CPU fallback')
        self._torch = torch
        self._device = int(config.get('device', os.environ.get('LOCAL_RANK', 0)))
        torch.cuda.set_device(self._device)
        nbytes = FsmgModel.state_bytes(config)
        # params | grads (+tail) | adam m | adam v, owned by the torch allocator so that the gradient
        # region can be handed to torch.distributed (RCCL) as an ordinary tensor
        self._arena = torch.zeros(nbytes + 256, dtype=torch.uint8, device='cuda:%d' % self._device)
        base = self._arena.data_ptr()
        pad = (-base) % 256
        # a dedicated (non-default) torch stream: its handle is a real hipStream_t the library can launch
        # on, and collectives issued under `stream_context()` are ordered with the library's kernels
        self._stream = torch.cuda.Stream(device=self._device)
        # sequences per training episode, when the merged YAMLs say (N-way x (K + Q)): sizes the activations up front and
        # picks the recurrent kernels built for that row count (fsmg_config.max_sequences)
        max_sequences = episode_sequences(config)
        self._model = FsmgModel(config, device=self._device, stream=self._stream.cuda_stream,
                                state_arena=base + pad, state_arena_bytes=nbytes,
                                max_sequences=max_sequences,
                                clip_norm_mode=config.get('clip_norm_mode', 'tf1_slices'),
                                use_graph=bool(config.get('use_graph', True)))
        gptr, gcount = self._model.grad_buffer()
        off = gptr - base
        self.grad_tensor = self._arena[off:off + 4 * gcount].view(torch.float32)
        # the same memory as three buckets (softmax grads | everything else | tail scalars) for the overlapped exchange
        self.grad_buckets = []
        for b in range(3):
            bptr, bcount = self._model.grad_bucket(b)
            self.grad_buckets.append(self._arena[bptr - base:bptr - base + 4 * bcount].view(torch.float32))
        self._comm_stream = torch.cuda.Stream(device=self._device)
        self._initialised = False
        self._train_calls = 0
        self._eval_calls = 0
        self._scalars = None
        if config.get('checkpt_dir'):
            os.makedirs(config['checkpt_dir'], exist_ok=True)
            self._scalars = open(os.path.join(config['checkpt_dir'], 'scalars.jsonl'), 'a')

    # -- engine interface used by fsmg.dist.EpisodeParallel ------------------------------------
    def forward_backward(self, support, query, **kw):
        self._model.forward_backward(support, query, **kw)

    def forward_backward_indexed(self, table_id, support_idx, query_idx):
        self._model.forward_backward_indexed(table_id, support_idx, query_idx)

    def maml_forward_backward(self, support, query, inner_steps, inner_lr, **kw):
        self._model.maml_forward_backward(support, query, inner_steps, inner_lr, **kw)

    def fused_train_step(self, support, query, want_loss=True, table=None, **kw):
        """forward + backward + clip + Adam as ONE captured graph (the single-process step: nothing to exchange in between)"""
        if table is not None:
            return self._model.train_step_indexed(table, support, query, want_loss=want_loss)
        return self._model.train_step(support, query, want_loss=want_loss, **kw)

    def apply_update(self, grad_scale=1.0, want_loss=True):
        return self._model.apply_update(grad_scale, want_loss=want_loss)

    def fused_maml_step(self, support, query, inner_steps, inner_lr, want_loss=True, table=None, **kw):
        if table is not None:       # support / query are row indices into the device-resident split table
            return self._model.maml_step_indexed(table, support, query, inner_steps, inner_lr, want_loss=want_loss)
        return self._model.maml_step(support, query, inner_steps, inner_lr, want_loss=want_loss, **kw)

    def maml_forward_backward_indexed(self, table_id, support_idx, query_idx, inner_steps, inner_lr):
        self._model.maml_forward_backward_indexed(table_id, support_idx, query_idx, inner_steps, inner_lr)

    # -- the gradient exchange inside the library (fsmg_comm_*: RCCL calls issued by libfsmg on its own stream) ----------
    library_comm = False

    def attach_library_comm(self):
        """Every rank of an initialised torch.distributed job: rank 0's ncclUniqueId travels through the job's own store, each
        rank joins with ncclCommInitRank, and from then on train steps are ONE library call per rank -- PyTorch is left with
        the memory (model config key `dp_exchange: 'library'`)."""
        import os
        import torch.distributed as dist
        world, rank = dist.get_world_size(), dist.get_rank()
        if world > 1 and os.environ.get('FSMG_ALLOW_LIBRARY_RCCL', '0') != '1':
            # EXPERIMENTAL: no box with two GPUs has run this path yet (tests/test_dist_hip.py covers it with one rank, and with
            # two as soon as two devices are visible); the torch-issued exchange is the tested default
            raise RuntimeError("dp_exchange: 'library' (RCCL calls issued by libfsmg) has not run with more than one rank yet; "
                               "set FSMG_ALLOW_LIBRARY_RCCL=1 to use it, or keep dp_exchange: 'torch'")
        ids = [FsmgModel.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ids, src=0)
        self._model.comm_init(ids[0], world, rank)
        self.library_comm = True

    @property
    def engine(self):
        return self._model

    def stream_context(self):
        return self._torch.cuda.stream(self._stream)

    def comm_context(self, bucket):
        """context of the communication stream, after making it wait until `bucket` of the pending backward is final"""
        self._model.stream_wait_bucket(self._comm_stream.cuda_stream, bucket)
        return self._torch.cuda.stream(self._comm_stream)

    def _log_scalar(self, tag, value, step):
        # the reference writes TensorBoard summaries (lstm_baseline.py:106-111,126-131); no TB here -> JSONL
        if self._scalars is not None:
            self._scalars.write('{"tag": "%s", "step": %d, "value": %.9g}\n' % (tag, step, value))
            self._scalars.flush()

    # -- checkpoints ----------------------------------------------------------------------------
    def _checkpt_prefix(self, checkpt_path):
        directory = os.path.join(checkpt_path, self.name)
        os.makedirs(directory, exist_ok=True)
        return os.path.join(directory, self.name)

    def save(self, checkpt_path):
        step = self._model.step
        blob = {'global_step': np.int64(step)}
        for name in self._model.param_shapes:
            blob['param/' + name] = self._model.get_param(name)
            m, v = self._model.get_opt_state(name)
            blob['adam_m/' + name], blob['adam_v/' + name] = m, v
        prefix = self._checkpt_prefix(checkpt_path)
        tmp = '%s-%d.tmp.npz' % (prefix, step)
        np.savez(tmp, **blob)
        os.replace(tmp, '%s-%d.npz' % (prefix, step))
        for old in self._checkpoints(os.path.dirname(prefix))[:-MAX_TO_KEEP]:
            os.remove(old[1])

    def _checkpoints(self, directory):
        found = []
        for path in glob.glob(os.path.join(directory, '%s-*.npz' % self.name)):
            m = re.search(r'-(\d+)\.npz$', path)
            if m:
                found.append((int(m.group(1)), path))
        return sorted(found)

    def _recover(self, checkpt_path):
        if not checkpt_path:
            return set()
        found = self._checkpoints(os.path.join(checkpt_path, self.name))
        if not found:
            return self._recover_tf(checkpt_path)
        path = found[-1][1]
        print('recovering %s from %s' % (self.name, path))
        restored = set()
        with np.load(path) as blob:
            for name, shape in self._model.param_shapes.items():
                want = self._model._shape(name)
                key = 'param/' + name
                if key in blob and blob[key].shape == want:
                    self._model.set_param(name, blob[key])
                    restored.add(name)
                    km, kv = 'adam_m/' + name, 'adam_v/' + name
                    if km in blob and kv in blob and blob[km].shape == want and blob[kv].shape == want:
                        self._model.set_opt_state(name, blob[km], blob[kv])
            if 'global_step' in blob:
                self._model.step = int(blob['global_step'])
        return restored

    def _recover_tf(self, checkpt_path):
        """A checkpoint the REFERENCE wrote (tf.train.Saver's tensor bundle, tf_model.py:96-97,106-125): weights, Adam slots and
        global_step are taken over -- every tensor whose name and shape match (optimistic_restore, tf_model.py:28-75)."""
        from models import tf_checkpoint as TC
        prefix = TC.latest_checkpoint(os.path.join(checkpt_path, self.name))
        if prefix is None:
            return set()
        print('recovering %s from %s (TensorFlow checkpoint)' % (self.name, prefix))
        params, adam_m, adam_v, step = TC.map_variables(TC.read_bundle(prefix), int(self._config['n_layers']))
        restored = set()
        for name, shape in self._model.param_shapes.items():
            want = self._model._shape(name)
            if name in params and tuple(params[name].shape) == want:
                self._model.set_param(name, params[name].astype(np.float32))
                restored.add(name)
                if name in adam_m and name in adam_v and tuple(adam_m[name].shape) == want and tuple(adam_v[name].shape) == want:
                    self._model.set_opt_state(name, adam_m[name].astype(np.float32), adam_v[name].astype(np.float32))
        if step is not None:
            self._model.step = step
        return restored

    def recover_or_init(self, init_path):
        missing = [n for n in self._model.param_shapes]
        # initialise everything first (also zeroes Adam state and the step), then overlay the checkpoint
        self._model.init_params(int(self._config.get('seed', 0)))
        restored = self._recover(init_path)
        print('Initializing vars:')
        print([n for n in missing if n not in restored])
        self._initialised = True

    def _require_init(self):
        if not self._initialised:
            raise RuntimeError('call recover_or_init() before train/eval/sample')
