#!/usr/bin/env python3
"""bench.py -- episodes/s of the LSTM-baseline train step on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1] = "cfg-B"): synthetic vocab 10 000 (V1 = 10 001), seq_len
128, 5-way / 5-shot / 4 query songs -> B = 45 sequences per episode, LSTM E=250 H=512 L=1, lr 5e-3, clip 5,
Glorot-uniform parameters, a pool of 256 episodes resident in HBM, a fresh episode per step.  One "step" =
one full train call: token staging, forward, BPTT, clip_by_global_norm, Adam, global_step++ (nothing is
skipped; the per-step loss stays in a device ring and is read back after the timed region).

With N > 1 every rank trains on its own episode per step and one RCCL all-reduce sums the flat gradient
buffer (weak scaling: per-GPU work fixed); value = N * K / max-over-ranks time.

Prints ONE JSON line (rank 0).  `roofline` is the dominant kernel (the dW = out^T * dlogits GEMM,
k_gemm<XC,XC>, one launch per step) timed with HIP events on the library's stream over a
repeat of the timed steps; `kernels` (extra) is a per-class breakdown from a second, fully instrumented pass;
`cpu_baseline` is the oracle's torch-CPU restatement of the same step ("port": TensorFlow cannot run here)
on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(ROOT, 'few-shot-music-generation_amd', 'src')
for p in (ROOT, SRC):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402

CFG_B = dict(name='lstm_baseline', seed=1234, input_size=10000, max_len=128, embedding_size=250,
             hidden_size=512, n_layers=1, lr=5e-3, max_grad_norm=5, n_decay=10000)
N_WAY, K_SHOT, Q_QUERY = 5, 5, 4
# other BASELINE.json configurations (parity-test cases; selectable for diagnostics, never the headline)
OTHER = {
    'cfg-C': (dict(name='lstm_baseline', seed=1234, input_size=4708, max_len=50, embedding_size=250, hidden_size=1024,
                   n_layers=2, lr=5e-3, max_grad_norm=5, n_decay=10000), 5, 5, 4),
    'cfg-D': (dict(CFG_B), 20, 1, 4),
    # cfg-B with 4 / 8 episodes per Adam step on ONE GPU (rows batched: 20-way / 40-way x (5+4) = 180 / 360 sequences);
    # same update rule as episode-parallel training over 4 / 8 ranks.  `value` then counts steps, not episodes.
    'cfg-Bx4': (dict(CFG_B), 20, 5, 4),
    'cfg-Bx8': (dict(CFG_B), 40, 5, 4),
}
POOL = 256
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
DOMINANT = 'gemm_dw'
CLASSES = ['gemm_zx', 'lstm_fwd', 'gemm_logits', 'ce', 'gemm_dhout', 'gemm_dw', 'lstm_bwd', 'gemm_dk',
           'gemm_dx', 'embed_grad', 'update']


def synthetic_episodes(n_episodes, N, K, Q, T, vocab, seed):
    """SURVEY.md 8(d) throughput input: token ids i.i.d. uniform on [0, vocab) from RandomState(seed)"""
    rng = np.random.RandomState(seed)
    return [(rng.randint(0, vocab, size=(N, K, T)).astype(np.int32), rng.randint(0, vocab, size=(N, Q, T)).astype(np.int32))
            for _ in range(n_episodes)]


def hbm_traffic():
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes (FETCH_SIZE x2 per the gfx950
    correction + WRITE_SIZE), recorded under profiles/ -- counters cannot be read from inside this process"""
    try:
        with open(os.path.join(ROOT, 'profiles', 'r01_dw_gemm_hbm_traffic.json')) as f:
            return json.load(f)['traffic_bytes_per_launch']
    except Exception:
        return None


def algorithmic_gflop(cfg, B):
    """per-class algorithmic FLOPs of one train episode: 2*M*N*K of the unpadded shapes"""
    T, E, H, V1, L = cfg['max_len'], cfg['embedding_size'], cfg['hidden_size'], cfg['input_size'] + 1, cfg['n_layers']
    n = B * T
    g = {'gemm_zx': 2 * n * (E + (L - 1) * H) * 4 * H, 'lstm_fwd': 2 * n * H * 4 * H * L, 'gemm_logits': 2 * n * H * V1,
         'gemm_dhout': 2 * n * H * V1, 'gemm_dw': 2 * n * H * V1, 'lstm_bwd': 2 * n * H * 4 * H * L,
         'gemm_dk': 2 * n * (E + H + (L - 1) * 2 * H) * 4 * H, 'gemm_dx': 2 * n * (E + (L - 1) * H) * 4 * H}
    return {k: v / 1e9 for k, v in g.items()}


T_START = time.time()


def log(msg):
    sys.stderr.write('[bench %7.1fs] %s\n' % (time.time() - T_START, msg))
    sys.stderr.flush()


def cpu_baseline(cfg, pool, budget_s=20.0):
    """oracle torch-CPU restatement timed on the host cores: bounded sample of the same workload"""
    import torch
    from oracle import lstm_oracle as O
    from oracle.torch_ref import TorchRef
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cores = min(avail, int(os.environ.get('FSMG_CPU_THREADS', 16)))   # measured on the 2x EPYC 9575F host: 8/16/32/64/128 threads -> 0.49/0.57/0.45/0.24/0.09 episodes/s
    ref = TorchRef(cfg, O.glorot_init(cfg, cfg['seed'], np.float32), dtype=torch.float32, threads=cores)
    log('cpu_baseline: %d threads (of %d available), warm-up step' % (cores, avail))
    ref.train(*pool[0])                                   # warm-up
    log('cpu_baseline: timing')
    n, t0 = 0, time.perf_counter()
    while True:
        ref.train(*pool[(n + 1) % len(pool)])
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or n >= 64:
            break
    return {'value': n / dt, 'unit': 'episodes/s', 'cores': cores, 'kind': 'port',
            'sample': '%d train episodes of the same cfg-B workload (torch-CPU fp32 restatement, %d threads, %.1f s)'
                      % (n, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    ap.add_argument('--config', default='cfg-B', choices=['cfg-B', 'cfg-C', 'cfg-D', 'cfg-Bx4', 'cfg-Bx8'])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fsmg.dist import EpisodeParallel, init_from_env
    from models.lstm_baseline import LSTMBaseline

    rank, world = init_from_env('nccl')
    if world != args.gpus:
        raise SystemExit('launched with WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    local = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local)
    global N_WAY, K_SHOT, Q_QUERY
    base = CFG_B
    if args.config != 'cfg-B':
        base, N_WAY, K_SHOT, Q_QUERY = OTHER[args.config]
    cfg = dict(base, device=local)
    B = N_WAY * (K_SHOT + Q_QUERY)

    pool_host = synthetic_episodes(POOL, N_WAY, K_SHOT, Q_QUERY, cfg['max_len'], cfg['input_size'], seed=1234 + rank)
    d_sup = torch.from_numpy(np.stack([s for s, _ in pool_host])).cuda()
    d_qry = torch.from_numpy(np.stack([q for _, q in pool_host])).cuda()
    sup_stride, qry_stride = d_sup[0].numel() * 4, d_qry[0].numel() * 4

    log('rank %d/%d: pool on device, creating model' % (rank, world))
    model = LSTMBaseline(cfg)
    model.recover_or_init('')
    log('model ready')
    par = EpisodeParallel(model)
    eng = model.engine
    shape = (N_WAY, K_SHOT, Q_QUERY)

    def step(i):
        e = i % POOL
        par.train_step(d_sup.data_ptr() + e * sup_stride, d_qry.data_ptr() + e * qry_stride,
                       want_loss=False, shape=shape)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
        if i == 0:
            torch.cuda.synchronize()
            log('first step done')
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    barrier()
    elapsed = time.perf_counter() - t0
    log('timed region done: %.3f s for %d steps' % (elapsed, args.steps))
    # roofline leg: the same K steps again with HIP events bracketing every launch of the dominant kernel on
    # the library's stream (event timing needs eager launches, so the step's hipGraph replay is off here;
    # the kernel, its arguments and its data are identical to the timed region's)
    eng.timing_select(DOMINANT)
    eng.timing_enable(True)
    eng.timing_reset()
    for i in range(args.steps):
        step(args.warmup + args.steps + i)
    dom_ms, dom_n = eng.timing_read(DOMINANT)
    eng.timing_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    losses = eng.read_losses(min(args.steps, 1024)) if args.steps > 0 else np.zeros(1)

    out = None
    if rank == 0:
        gf = algorithmic_gflop(cfg, B)
        value = world * args.steps / elapsed
        total_gflop = 3 * (gf['gemm_zx'] + gf['lstm_fwd'] + gf['gemm_logits'])
        dom_avg_ms = dom_ms / max(dom_n, 1)
        achieved = gf[DOMINANT] / dom_avg_ms if dom_n else 0.0        # GFLOP/ms == TFLOP/s
        out = {
            'metric': 'episodes/s (LSTM-baseline train step, synthetic vocab=10k seq_len=128 5-way/5-shot h=512)',
            'value': value, 'unit': 'episodes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / max(args.steps, 1), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': (args.config + ' (diagnostic run, not the headline workload) -- ' if args.config != 'cfg-B' else '') + 'cfg-B: synthetic V=10000 (V1=10001) T=128 5-way 5-shot 4-query (B=45 sequences/episode), '
                                   'LSTM E=250 H=512 L=1, full train step (fwd+BPTT+clip+Adam), one episode per GPU per step',
                       'episodes_per_step': world, 'parallelism': 'episode-parallel x%d, 1 RCCL all-reduce/step' % world},
            'roofline': {'bound': 'mfma', 'kernel': 'k_gemm<XC,XC> = k_gemm<1, 1> (dW = out^T * dlogits, M=512 N=10004 K=5760)',
                         'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': hbm_traffic(),
                         'avg_launch_ms': dom_avg_ms, 'launches': dom_n, 'algorithmic_gflop_per_launch': gf[DOMINANT]},
            'step_mfma_frac': (total_gflop / (1e3 * elapsed / max(args.steps, 1))) / PEAK_F32_MFMA_TFLOPS,
            'final_loss': float(losses[-1]), 'first_loss': float(losses[0]),
        }
    if rank == 0:
        # the other half of BASELINE.json's metric: the validation path (query-only forward, batched 16 episodes
        # per call like train.evaluate does); inputs resident in HBM, NLLs read back per call
        n_ev, reps = 16, 5
        qptr = d_qry.data_ptr()
        eng.eval_batch(qptr, shape=(n_ev, N_WAY, Q_QUERY))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            nll = eng.eval_batch(qptr + r * n_ev * qry_stride, shape=(n_ev, N_WAY, Q_QUERY))
        torch.cuda.synchronize()
        out['eval'] = {'episodes_per_s': n_ev * reps / (time.perf_counter() - t0), 'batch_episodes': n_ev,
                       'mean_val_nll': float(np.mean(nll)), 'unit': 'eval episodes/s (query-only forward, 20 sequences/episode)'}
    if rank == 0 and world == 1:
        # the reference's calling convention: host numpy episodes in, the loss read back every step (one 23 KB H2D
        # token copy + one synchronising 4-byte D2H per step) -- PCIe-inclusive, never the headline value
        n_h = min(args.steps, 30)
        t0 = time.perf_counter()
        for i in range(n_h):
            eng.train_step(*pool_host[i % POOL], want_loss=True)
        out['host_synchronous'] = {'episodes_per_s': n_h / (time.perf_counter() - t0),
                                   'note': 'host token buffers + per-step loss readback (reference train() semantics)'}
    if rank == 0 and world == 1 and not args.no_breakdown:
        # second, fully instrumented pass: every kernel class bracketed by HIP events (extra information)
        eng.timing_select(None)
        eng.timing_enable(True)
        eng.timing_reset()
        nb = min(args.steps, 10)
        for i in range(nb):
            step(i)
        torch.cuda.synchronize()
        gf = algorithmic_gflop(cfg, B)
        kernels = {}
        for c in CLASSES:
            ms, n = eng.timing_read(c)
            if n:
                per_step = ms / nb
                kernels[c] = {'ms_per_step': per_step}
                if c in gf:
                    kernels[c]['tflops'] = gf[c] / per_step
                    kernels[c]['frac_mfma_peak'] = gf[c] / per_step / PEAK_F32_MFMA_TFLOPS
        eng.timing_enable(False)
        out['kernels'] = kernels
        log('breakdown done')
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline(base, pool_host)
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
