"""Episode-parallel training over the GPUs of one node.

One process per GPU.  Every outer step each rank runs forward+backward on ITS episode,
the flat fp32 gradient buffer (plus its tail scalars: squared embedding-slice norm and
loss) is summed with ONE all-reduce (RCCL over xGMI when the backend is "nccl"), and
every rank then applies the identical clip + Adam update with grad_scale = 1/world, so
replicas stay bit-identical without ever broadcasting parameters again.

The reference has no distributed code; R = 1 is its semantics, R > 1 is "R episodes per
Adam step" = the gradient of the mean loss over the R episodes' rows.

`engine` is anything with forward_backward(support, query), grad_tensor (a torch tensor
aliasing the flat gradient buffer) and apply_update(grad_scale) -> loss: the HIP model
(models.hip_model.HIPModel) in production, a CPU stand-in in the gloo tests.
"""
import torch
import torch.distributed as dist

RETRY_CODES = (-9, -10)        # FSMG_ERR_TIMEOUT, FSMG_ERR_SOFTMAX_RANGE (fsmg.binding.RETRY_CODES; repeated here: no import of the binding on CPU-only ranks)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun); returns (rank, world)."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world


class EpisodeParallel(object):
    def __init__(self, engine, group=None, bucketed=None):
        self.engine = engine
        self.group = group
        import os
        # bucketed exchange on the communication stream (default) or ONE all-reduce on the compute stream
        self.bucketed = (os.environ.get('FSMG_DP_BUCKETS', '1') != '0') if bucketed is None else bool(bucketed)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.exchange = True            # False: skip the gradient exchange (bench.py's "what does the exchange cost" leg; replicas diverge)

    def broadcast_parameters(self, tensor):
        """One-time parameter / optimiser-state broadcast from rank 0 (after init or restore)."""
        if self.world > 1:
            ctx = getattr(self.engine, 'stream_context', None)
            if ctx is not None:
                with ctx():
                    dist.broadcast(tensor, src=0, group=self.group)
            else:
                dist.broadcast(tensor, src=0, group=self.group)

    def train_step(self, support, query, want_loss=True, **kw):
        """kw: shape=(N, K, Q) for raw device token addresses; table=id: support / query are row indices into the uploaded split
        table; maml=(inner_steps, inner_lr) selects the cfg-E step (per-rank
        inner SGD on the support rows, no communication; the query-set gradients are exchanged exactly like a plain step's)"""
        try:
            return self._train_step_once(support, query, want_loss, **kw)
        except Exception as e:
            # "the step was skipped on the device, repeat it" (include/fsmg.h: FSMG_ERR_TIMEOUT = -9, FSMG_ERR_SOFTMAX_RANGE = -10): a
            # persistent kernel of SOME rank timed out, or a row of some rank's logits left the fused softmax's range.  The indicator
            # travelled in the reduced gradient tail, so every rank skipped the update, raised here with the same status code and has
            # switched (one launch per time step / the cross-entropy pass) -- every rank repeats the step, in lock-step
            if getattr(e, 'code', None) not in RETRY_CODES:
                raise
            return self._train_step_once(support, query, want_loss, **kw)

    def _train_step_once(self, support, query, want_loss=True, maml=None, table=None, **kw):
        if self.world == 1 and maml is None and hasattr(self.engine, 'fused_train_step'):
            return self.engine.fused_train_step(support, query, want_loss=want_loss, table=table, **kw)
        if self.world == 1 and maml is not None and table is not None and hasattr(self.engine, 'fused_maml_step'):
            return self.engine.fused_maml_step(support, query, maml[0], maml[1], want_loss=want_loss, table=table)
        if getattr(self.engine, 'library_comm', False):
            # the library owns the exchange (fsmg_comm_init): one call per step and rank, collectives issued by libfsmg
            if maml is not None:
                return self.engine.fused_maml_step(support, query, maml[0], maml[1], want_loss=want_loss, table=table, **kw)
            return self.engine.fused_train_step(support, query, want_loss=want_loss, table=table, **kw)
        if table is not None and maml is not None:      # cfg-E on the device-resident table: per-rank inner loop, then the exchange below
            self.engine.maml_forward_backward_indexed(table, support, query, maml[0], maml[1])
        elif table is not None:               # support / query are [N,K] / [N,Q] row indices into a device-resident split table
            self.engine.forward_backward_indexed(table, support, query)
        elif maml is not None:
            self.engine.maml_forward_backward(support, query, maml[0], maml[1], **kw)
        else:
            self.engine.forward_backward(support, query, **kw)
        buckets = getattr(self.engine, 'grad_buckets', None)
        if not self.exchange:
            pass
        elif self.world > 1 and buckets is not None and self.bucketed:
            # overlapped exchange: each bucket is reduced on the communication stream as soon as it is final
            # (bucket 0 = softmax gradients, ready while BPTT / the weight-gradient GEMMs still run)
            works = []
            for b, tensor in enumerate(buckets):
                with self.engine.comm_context(b):
                    works.append(dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            with self.engine.stream_context():
                for w in works:
                    w.wait()                 # the compute stream waits for the collectives, the host does not
        elif self.world > 1:
            ctx = getattr(self.engine, 'stream_context', None)
            if ctx is not None:
                with ctx():                  # same stream as the HIP kernels: no host sync needed
                    dist.all_reduce(self.engine.grad_tensor, op=dist.ReduceOp.SUM, group=self.group)
            else:
                dist.all_reduce(self.engine.grad_tensor, op=dist.ReduceOp.SUM, group=self.group)
        return self.engine.apply_update(1.0 / self.world, want_loss=want_loss)

    def mean_scalar(self, value):
        """Mean of a host float over ranks (validation NLL sharded over ranks)."""
        if self.world == 1:
            return value
        t = torch.tensor([value], dtype=torch.float64)
        if dist.get_backend(self.group) == 'nccl':
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t.item()) / self.world
