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
buffer (weak scaling: per-GPU work fixed); value = N * K / max-over-ranks time.  `python bench.py --gpus N` WITHOUT a
torchrun environment starts its own N ranks (re-executes itself under torch.distributed.run on 127.0.0.1, rank r on GPU r)
and prints rank 0's line; under torchrun (RANK / WORLD_SIZE set) it is one rank of the job.  For N > 1 the timed loop runs
the DEFAULT exchange schedule only -- "graph_end" (one graph per backward pass, the three gradient
buckets reduced on the communication stream when it ends) --; FSMG_BENCH_SCHEDULES=all (or a comma list) times the others
behind it IN THE SAME RUN (`schedules`), each only after graph_end has produced a guarded result: "split_bucket0" (two graphs: bucket 0 = softmax gradients, 56 %
of the bytes, is released behind the projection-gradient GEMMs and travels under BPTT), "split_after_chain" (the same with
the cut behind the last recurrent chain: an XCD-local chain needs every CU, so a collective started in front of it only
delays it; behind it bucket 0 travels beside the weight- / input-gradient GEMMs) and "one_collective" (a single
all-reduce on the compute stream); `value` is the DEFAULT schedule's ("graph_end": what the shipped configuration runs; the others are listed under `schedules`), `comm.exposed_ms` what each one adds to the same loop without
any exchange.  FSMG_BENCH_LIBRARY_RCCL=1 adds "library_rccl": the collectives issued by libfsmg itself (fsmg_comm_init).

Prints ONE JSON line (rank 0, the LAST line of stdout).  `roofline` is the kernel BASELINE.json's north star sets a
target for -- the fused LSTM cell (recurrent 4x GEMV on MFMA + gate nonlinearities + state update: k_lstm_fwd_xcd and
k_lstm_bwd_xcd at hidden size 512), 2 launches per step, each running T = 128 dependent time steps -- timed with HIP
events on the library's stream over a repeat of the timed steps in the schedule the timed region used (single stream
at cfg-B; event timing only replaces the hipGraph replay by the same launches issued eagerly).  `kernels` is the
per-class breakdown of a fully instrumented pass (every GEMM with its own fraction of the MFMA peak); `guard` proves
that the timed region did the work it claims (global_step advanced by exactly warmup + steps, no step skipped, no
persistent-kernel time-out); `cpu_baseline` is the oracle's CPU restatement of the same step ("port": TensorFlow cannot
run here) on a bounded sample of the same workload.
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
    # "cfg-C: ... T = 50 (YAML) and 128" -- the same two layers of 1024 at the headline's sequence length (637 GFLOP per episode)
    'cfg-C-T128': (dict(name='lstm_baseline', seed=1234, input_size=4708, max_len=128, embedding_size=250, hidden_size=1024,
                        n_layers=2, lr=5e-3, max_grad_norm=5, n_decay=10000), 5, 5, 4),
    'cfg-D': (dict(CFG_B), 20, 1, 4),
    # the reference's OWN shipped defaults (src/config/lstm_baseline.yaml: E=250, H=200, L=1; lyrics.yaml: max_len 50; 5shot.yaml),
    # vocabulary sized like the headline workload -- what a user gets who drops the plugin in with the reference's YAMLs
    'ref-default': (dict(name='lstm_baseline', seed=1234, input_size=10000, max_len=50, embedding_size=250, hidden_size=200,
                         n_layers=1, lr=5e-3, max_grad_norm=5, n_decay=10000), 5, 5, 4),
    # cfg-B with 4 / 8 episodes per Adam step on ONE GPU (rows batched: 20-way / 40-way x (5+4) = 180 / 360 sequences);
    # same update rule as episode-parallel training over 4 / 8 ranks.  `value` then counts steps, not episodes.
    'cfg-Bx4': (dict(CFG_B), 20, 5, 4),
    'cfg-Bx8': (dict(CFG_B), 40, 5, 4),
    # BASELINE.json configs[4]: MAML-style inner/outer loop (models.maml_lstm; 1 inner clipped-SGD step on the support rows,
    # outer clip+Adam on the query gradient), freemidi-sized vocabulary, 2-layer LSTM h=1024, 5-way/5-shot
    'cfg-E': (dict(name='maml_lstm', seed=1234, input_size=4708, max_len=50, embedding_size=250, hidden_size=1024,
                   n_layers=2, lr=5e-3, max_grad_norm=5, n_decay=10000, inner_steps=1, inner_lr=0.1), 5, 5, 4),
}
POOL = 256
PEAK_F32_MFMA_TFLOPS = 157.3       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
# the GEMMs' bound: fp32 products assembled from 6 bf16 MFMAs (k_gemm_bx3, csrc/gemm.hip) -> dense bf16 peak / 6
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_BX3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
HBM_ACHIEVABLE_TBPS = 6.3          # ~6.3 of the 8.0 TB/s spec is achievable
REPEATS = max(1, int(os.environ.get('FSMG_BENCH_REPEATS', 5)))       # timed regions of K steps each; value = their median
ARITHMETIC_GEMM = ('fp32 storage, fp32 accumulation everywhere.  GEMMs (default, FSMG_GEMM=bx3): every operand value is split EXACTLY into three '
                   'bf16 numbers and the six partial products >= 2^-23 of the fp32 product are summed in fp32 by v_mfma_f32_32x32x16_bf16 -- '
                   'error against fp64 below the fp32-MFMA kernel\'s on every shape of the step (tools/gemm_bench.cpp, BX3=0/1 with verify; '
                   'tests/test_gpu_parity.py::test_bf16_split_gemm_is_no_less_accurate_than_the_fp32_mfma_gemm); FSMG_GEMM=f32 selects '
                   'v_mfma_f32_32x32x2_f32 (timed in the same run: roofline.alt_gemm_f32_mfma_value).  ')
# the fused cell's arithmetic depends on the kernel family the timed schedule ran (the line said 4x4x1 for every schedule)
ARITHMETIC_CELL = {
    'bf16x3': 'Fused LSTM cell (this schedule: k_lstm_*_xcd16 at hidden 512, k_lstm_*_pair16 at hidden 1024): the recurrent contraction as the same exact three-way bf16 split, six products per '
              'fp32 product on v_mfma_f32_16x16x32_bf16, fp32 accumulation; gates, state update and gate gradients in fp32.',
    'f32': 'Fused LSTM cell (this schedule): v_mfma_f32_4x4x1_16B_f32 (fp32 MFMA); gates, state update and gate gradients in fp32.',
    'f32_16x16x4': 'Fused LSTM cell (this schedule: column-split persistent kernels): v_mfma_f32_16x16x4_f32 (fp32 MFMA).',
}
CELL_CLASSES = ('lstm_fwd', 'lstm_bwd')      # the fused LSTM cell: the north star's target kernel
CLASSES = ['gemm_zx', 'lstm_fwd', 'gemm_logits', 'ce', 'gemm_dhout', 'gemm_dw', 'lstm_bwd', 'gemm_dk',
           'gemm_dx', 'embed_grad', 'update']


def synthetic_episodes(n_episodes, N, K, Q, T, vocab, seed):
    """SURVEY.md 8(d) throughput input: token ids i.i.d. uniform on [0, vocab) from RandomState(seed)"""
    rng = np.random.RandomState(seed)
    return [(rng.randint(0, vocab, size=(N, K, T)).astype(np.int32), rng.randint(0, vocab, size=(N, Q, T)).astype(np.int32))
            for _ in range(n_episodes)]


def padded_zipf_episodes(n_episodes, N, K, Q, T, vocab, seed):
    """What real lyrics / MIDI episodes look like instead of SURVEY 8(d)'s uniform pool: word ids Zipf(1.1), song length U[T/4, T], zero
    padded to T -- a third of a pass's positions hold token 0, the next twenty words 50-400 each (the embedding
    gradient and the occurrence table are the kernels that care)."""
    rng = np.random.RandomState(seed)

    def songs(n):
        a = np.minimum(rng.zipf(1.1, size=(N, n, T)) - 1, vocab - 1).astype(np.int32)
        lens = rng.randint(max(1, T // 4), T + 1, size=(N, n))
        a[np.arange(T)[None, None, :] >= lens[:, :, None]] = 0
        return a
    return [(songs(K), songs(Q)) for _ in range(n_episodes)]


def hbm_traffic(variant=None):
    """(bytes, source): HBM bytes per launch of the fused-cell kernels as RECORDED by the committed rocprofv3 PMC passes
    (tools/pmc_passes.sh -> profiles/r05_pmc.json: raw FETCH_SIZE + WRITE_SIZE per kernel INSTANTIATION, separate passes) -- counters
    cannot be read from inside this process, so this is not a measurement of this run; `variant` (e.g. 'xcd16<4', 'xcd16<2', 'xcd<2')
    selects the record of the kernel family the timed region ran (the round-4 record averaged two instantiations).
    No x2 on FETCH_SIZE: the guide's gfx950 correction is calibrated for 16-B/lane streaming loads, and these kernels' raw
    FETCH_SIZE (47.7 MB forward) already equals their algorithmic read (Z: 47.2 MB).  (None, None) when no record exists."""
    try:
        pmc_name = 'r06_pmc.json' if os.path.isfile(os.path.join(ROOT, 'profiles', 'r06_pmc.json')) else 'r05_pmc.json'
        with open(os.path.join(ROOT, 'profiles', pmc_name)) as f:
            rec = json.load(f)['kernels']
        per = {}
        for d in ('fwd', 'bwd'):
            hits = [v for k, v in rec.items() if ('k_lstm_%s_%s' % (d, variant)) in k and v.get('FETCH_SIZE_KB') is not None and v.get('WRITE_SIZE_KB') is not None]
            if hits:
                per[d] = 1024.0 * (hits[0]['FETCH_SIZE_KB'] + hits[0]['WRITE_SIZE_KB'])
        if variant and len(per) == 2:
            return sum(per.values()) / 2, ('recorded (profiles/%s, k_lstm_*_%s...>: raw FETCH_SIZE + WRITE_SIZE per launch, mean of forward '
                                           '%.1f MB and backward %.1f MB)' % (pmc_name, variant, per['fwd'] / 1e6, per['bwd'] / 1e6))
    except Exception:
        pass
    for name in ('r04_lstm_cell_pmc.json', 'r03_lstm_cell_pmc.json', 'r02_lstm_cell_pmc.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                rec = json.load(f)
            per = {k: rec['fetch_size_kb'][k] * 1024 + rec['write_size_kb'].get(k, 0.0) * 1024 for k in rec['fetch_size_kb']}
            return sum(per.values()) / max(len(per), 1), 'recorded (profiles/%s: raw FETCH_SIZE + WRITE_SIZE per launch, mean of forward and backward)' % name
        except Exception:
            continue
    return None, None


def algorithmic_gflop(cfg, B):
    """per-class algorithmic FLOPs of one train episode: 2*M*N*K of the unpadded shapes"""
    T, E, H, V1, L = cfg['max_len'], cfg['embedding_size'], cfg['hidden_size'], cfg['input_size'] + 1, cfg['n_layers']
    n = B * T
    g = {'gemm_zx': 2 * n * (E + (L - 1) * H) * 4 * H, 'lstm_fwd': 2 * n * H * 4 * H * L, 'gemm_logits': 2 * n * H * V1,
         'gemm_dhout': 2 * n * H * V1, 'gemm_dw': 2 * n * H * V1, 'lstm_bwd': 2 * n * H * 4 * H * L,
         'gemm_dk': 2 * n * (E + H + (L - 1) * 2 * H) * 4 * H, 'gemm_dx': 2 * n * (E + (L - 1) * H) * 4 * H}
    return {k: v / 1e9 for k, v in g.items()}


def step_roofline(cfg, B, gf, ms_per_step, fused_softmax=False):
    """Blended bound of the whole train step: every dense contraction at its own arithmetic's peak (GEMMs: fp32 products from six
    bf16 MFMAs = dense bf16 peak / 6; fused cell: fp32 MFMA), the HBM-bound passes (cross entropy: logits read + dlogits written;
    clip + Adam: 7 P floats; activations written once and read once) at the achievable HBM rate -- summed as if nothing overlapped.
    frac = that bound over the measured step: a slower-but-overlapped recurrence cannot read as a regression here, nor hide one."""
    T, E, H, V1, L = cfg['max_len'], cfg['embedding_size'], cfg['hidden_size'], cfg['input_size'] + 1, cfg['n_layers']
    n = B * T
    gemm_gf = sum(v for k, v in gf.items() if k.startswith('gemm_'))
    cell_gf = gf['lstm_fwd'] + gf['lstm_bwd']
    P = V1 * E + sum(((E if l == 0 else H) + H) * 4 * H + 4 * H for l in range(L)) + H * V1 + V1
    # SURVEY.md 8(d)'s algorithmic bytes: parameters read once forward and once backward, the optimizer's 7 P floats, the tokens,
    # the saved activations written and read once.  The logits are NOT in it ("never materialise logits"): the step
    # does materialise them -- written by the projection, read by the cross entropy, dlogits written (in place since round 5) and
    # read by both projection-gradient GEMMs -- which is listed beside it, not priced as necessary
    by_alg = 2 * 4 * P + 7 * 4 * P + 2 * 4 * n + 2 * (6 * L * H + E) * 4 * n
    # classic softmax: logits written + read, dlogits written (in place) + read by dH and dW = 5 passes; fused softmax (round 5, DESIGN.md
    # 10.8): exp(logit) written by the projection's epilogue, read by dH and dW = 3 passes, plus c_r * h_r written and read once
    by_logits = (3 * 4 * n * V1 + 2 * 4 * n * H) if fused_softmax else 5 * 4 * n * V1
    t_gemm, t_cell, t_hbm = gemm_gf / PEAK_BX3_TFLOPS, cell_gf / PEAK_F32_MFMA_TFLOPS, by_alg / (HBM_ACHIEVABLE_TBPS * 1e9)
    out = {'bound_ms': t_gemm + t_cell + t_hbm, 'frac': (t_gemm + t_cell + t_hbm) / ms_per_step,
           'gemm_ms_at_bf16_peak_over_6': t_gemm, 'cell_ms_at_fp32_mfma_peak': t_cell, 'hbm_ms_at_%.1f_TBps' % HBM_ACHIEVABLE_TBPS: t_hbm,
           'hbm_bytes': by_alg, 'hbm_bytes_algorithmic': by_alg,
           'hbm_bytes_logits_round_trips': by_logits,
           'hbm_bytes_moved_model': by_alg + by_logits,
           'frac_r04_definition': (t_gemm + t_cell + (2 * 4 * n * V1 + 7 * 4 * P + 2 * (6 * L * H + E) * 4 * n) / (HBM_ACHIEVABLE_TBPS * 1e9)) / ms_per_step,
           'note': 'sum of the three bounds (no overlap assumed) over the measured ms_per_step; hbm_bytes = SURVEY.md 8(d) algorithmic bytes (no logits); '
                   'hbm_bytes_moved_model adds what the step materialises around the vocabulary projection (classic: logits written + read, dlogits written + read twice; '
                   'fused softmax: exp(logit) written once + read twice, c*h written + read; '
                   'split-K slabs not included); frac_r04_definition = the round-4 figure, whose HBM term counted one logits round trip as necessary'}
    try:
        pmc_name = 'r06_pmc.json' if os.path.isfile(os.path.join(ROOT, 'profiles', 'r06_pmc.json')) else 'r05_pmc.json'
        with open(os.path.join(ROOT, 'profiles', pmc_name)) as f:
            rec = json.load(f)
        if rec.get('step_hbm_bytes_measured') and rec.get('workload', '').startswith('cfg-B') and (E, H, L, T) == (250, 512, 1, 128):
            out['hbm_bytes_measured_recorded'] = rec['step_hbm_bytes_measured']
            out['hbm_bytes_measured_source'] = 'profiles/' + pmc_name + ': sum over the kernels of one train step of FETCH_SIZE + WRITE_SIZE (separate --pmc passes), recorded -- not a measurement of this run'
    except Exception:
        pass
    return out


def other_configs(log, steps=20, warmup=5):
    """(+ cfg-B and the reference's default dims on a pool of zero-padded Zipf songs: `--pool padded-zipf`)
    cfg-C, cfg-D's per-rank workload (20-way 1-shot, 100 rows), cfg-E (MAML-style step) and the reference's default dims for `steps`
    train steps each, plus cfg-B in the SERIAL order (FSMG_XCD_OVERLAP=0: the fp32 XCD-local fused cell chip-wide, the number the
    north star's >= 0.30 is about) with the cell kernels event-timed.  Each leg is this script again in a process of its own
    (`--config X --steps 20 --warmup 5`, one timed region, 1.5 s): what a user gets is a fresh process, and a leg must not depend on
    what the process did before it (round 5 found handles 5, 7, 9 of a process 32-45 % slower through the priority of a stream:
    DESIGN.md 10.4 -- fixed, the isolation stays).  Per leg: value, ms_per_step, guard.ok, roofline_step.frac."""
    import subprocess
    legs = [(n, n, {}) for n in ('cfg-C', 'cfg-C-T128', 'cfg-D', 'cfg-E', 'ref-default')] + [('cfg-B-serial-order', 'cfg-B', {'FSMG_XCD_OVERLAP': '0'}),
                                                                                  ('cfg-B-padded-zipf-pool', 'cfg-B', {'_pool': 'padded-zipf', '_val': '400'}),
                                                                                  ('ref-default-padded-zipf-pool', 'ref-default', {'_pool': 'padded-zipf'})]
    res = {}
    for name, config, env_over in legs:
        t_leg = time.perf_counter()
        env_over = dict(env_over)
        pool = env_over.pop('_pool', 'uniform')
        val_steps = env_over.pop('_val', None)
        env = dict(os.environ, FSMG_BENCH_REPEATS='1', **env_over)
        cmd = [sys.executable, os.path.abspath(__file__), '--config', config, '--steps', str(steps), '--warmup', str(warmup), '--pool', pool,
               '--no-cpu-baseline', '--no-breakdown', '--no-other-configs', '--no-extras'] + (['--val-nll', val_steps] if val_steps else [])
        try:
            proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, env=env, timeout=180)
            line = [l for l in proc.stdout.splitlines() if l.startswith('{"metric"')][-1]
            d = json.loads(line)
        except Exception as e:                         # noqa: BLE001 -- a leg that fails is reported, the others still run
            res[name] = {'error': repr(e)[:300]}
            log('other_configs: %s FAILED (%r)' % (name, e))
            continue
        r = {'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'],
             'guard': {'ok': bool(d['guard']['ok']), 'advanced_by': d['guard']['advanced_by'], 'timeouts': d['guard']['timeouts'],
                       'persistent_path': d['guard']['persistent_path']},
             'workload': d['config']['workload']}
        if d.get('roofline_step'):
            r['roofline_step'] = {'frac': d['roofline_step']['frac']}
        if d.get('val'):
            r['val'] = d['val']
        if name == 'cfg-B-serial-order' and d.get('roofline'):
            rf = d['roofline']
            r['fused_cell'] = {'achieved': rf['achieved'], 'peak': rf['peak'], 'unit': rf['unit'], 'frac': rf['frac'], 'kernel_variant': rf.get('kernel_variant'),
                               'us_per_time_step': {'lstm_fwd': rf['forward']['us_per_time_step'], 'lstm_bwd': rf['backward']['us_per_time_step']},
                               'note': 'fp32 XCD-local kernels (k_lstm_fwd_xcd / k_lstm_bwd_xcd, v_mfma_f32_4x4x1) on the whole chip, serial order: '
                                       'the north star\'s fused-cell fraction; the headline step runs the bf16-split chains packed on 3 XCDs beside the GEMMs (roofline).  '
                                       '>= 0.30 at 45 rows is not reachable with a per-step cross-CU hand-off (2.0 us per step needed, 2.2 / 2.35 measured; DESIGN.md 10.6); '
                                       'it is met from 100 rows on (cfg-D on the bf16-split kernels: 0.43) and against the XCDs the cell occupies in the partitioned order'}
        r['leg_s'] = time.perf_counter() - t_leg
        res[name] = r
        log('other_configs: %s %.1f episodes/s (%.3f ms/step, guard %s, %.1f s)' % (name, r['value'], r['ms_per_step'], r['guard']['ok'], r['leg_s']))
    return res


T_START = time.time()


def log(msg):
    sys.stderr.write('[bench %7.1fs] %s\n' % (time.time() - T_START, msg))
    sys.stderr.flush()


def _affinity():
    return len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)


def cpu_baseline(cfg, pool, shape, budget_s=18.0):
    """The reference's CPU path, timed as this repository's CPU restatements of the identical op graph ("port": TensorFlow
    cannot run on either box) on a bounded sample of the SAME workload, same episodes as the GPU run.
    Variants: torch-CPU (MKL GEMMs, the closest stand-in for TF-Eigen) and, when its recipe has been built
    (oracle/Makefile -> oracle/_build/libcpuref.so), the plain C++/OpenMP restatement.  Reports host cores used/available,
    train and eval episodes/s and the NLL agreement with the GPU on the first sample episode."""
    import torch
    from oracle import lstm_oracle as O
    from oracle.torch_ref import TorchRef
    avail = _affinity()
    # thread count: the T = 128 recurrence is a chain of small [45 x 762] x [762 x 2048] matmuls (94 MFLOP each); beyond
    # ~16 threads the per-op fork/join and cross-CCD traffic outweigh the extra FLOPs (measured on the 2x EPYC 9575F host,
    # 8/16/32/64/128 threads: 0.49/0.57/0.45/0.24/0.09 episodes/s), so more cores make THIS graph slower, not faster
    cores = min(avail, int(os.environ.get('FSMG_CPU_THREADS', 16)))
    try:
        torch.set_num_interop_threads(1)          # one op at a time: the graph is a dependency chain anyway
    except RuntimeError:
        pass                                      # already fixed by earlier parallel work in this process
    params = O.glorot_init(cfg, cfg['seed'], np.float32)
    ref = TorchRef(cfg, params, dtype=torch.float32, threads=cores)
    log('cpu_baseline[torch]: %d threads (of %d available), warm-up step' % (cores, avail))
    first_eval = ref.eval(pool[0][1])
    agree_eps, agree_losses = [0], [float(ref.train(*pool[0]))]          # warm-up; every train loss is kept for the GPU-vs-CPU agreement
    log('cpu_baseline[torch]: timing')
    n, t0 = 0, time.perf_counter()
    while True:
        agree_eps.append((n + 1) % len(pool))
        agree_losses.append(float(ref.train(*pool[(n + 1) % len(pool)])))
        n += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s * 0.75 or n >= 64:
            break
    agree_eval_ep = len(pool) - 1                        # a held-out episode: the train loop above never reaches it (<= 65 of 256)
    agree_eval = float(ref.eval(pool[agree_eval_ep][1]))
    ne, t1 = 0, time.perf_counter()
    while True:
        ref.eval(pool[(ne + 1) % len(pool)][1])
        ne += 1
        de = time.perf_counter() - t1
        if de >= budget_s * 0.25 or ne >= 64:
            break
    variants = [{'name': 'torch-cpu', 'train_episodes_per_s': n / dt, 'eval_episodes_per_s': ne / de, 'threads': cores,
                 'train_eps': n, 'eval_eps': ne}]
    try:
        from oracle.cpu_ref import CpuRef
        cx = CpuRef(cfg, params, threads=min(avail, int(os.environ.get('FSMG_CXX_THREADS', 32))))
        log('cpu_baseline[cxx]: %d threads, timing' % cx.threads)
        cx.train(*pool[0])
        nc, t2 = 0, time.perf_counter()
        while True:
            cx.train(*pool[(nc + 1) % len(pool)])
            nc += 1
            dc = time.perf_counter() - t2
            if dc >= budget_s * 0.5 or nc >= 64:
                break
        variants.append({'name': 'c++/openmp', 'train_episodes_per_s': nc / dc, 'threads': cx.threads, 'train_eps': nc})
    except Exception as e:                                 # not built on this box: the torch variant stands alone
        log('cpu_baseline[cxx]: unavailable (%s)' % str(e)[:80])
    best = max(variants, key=lambda v: v['train_episodes_per_s'])
    honest = None
    try:
        honest = cpu_honest_line(cfg, shape, variants[0], cores, avail, torch)
    except Exception as e:                                 # noqa: BLE001 -- extra information only
        log('cpu_baseline[honest line]: failed (%s)' % str(e)[:120])
    return {'value': best['train_episodes_per_s'], 'unit': 'episodes/s', 'cores': best['threads'], 'kind': 'port', 'honest_line': honest,
            'cores_used': best['threads'], 'cores_available': avail, 'cores_of': '%d of %d' % (best['threads'], avail), 'variants': variants,
            'train_eps': n, 'eval_eps': ne, 'first_eval_nll': float(first_eval),
            'agreement_train_episodes': agree_eps, 'agreement_train_losses': agree_losses, 'agreement_eval_episode': agree_eval_ep, 'agreement_eval_nll': agree_eval,
            'sample': '%d train + %d eval episodes of the same %d-way %d-shot workload (fp32 CPU restatement of the reference '
                      'graph, best of %s; %.1f s)' % (n, ne, shape[0], shape[1], '/'.join(v['name'] for v in variants), dt + de)}


def host_peak_model(avail):
    """fp32 peak of the host the CPU leg runs on, from what the OS says: physical cores x 64 FLOP/cycle (two 512-bit FMA pipes) x max clock.
    A model (the inputs are reported), good to the factor that matters for `host_peak_frac`."""
    import re
    import subprocess
    mhz, tpc, model = None, 1, ''
    try:
        txt = subprocess.run(['lscpu'], stdout=subprocess.PIPE, universal_newlines=True, timeout=10).stdout
        m = re.search(r'CPU max MHz:\s*([0-9.]+)', txt) or re.search(r'CPU MHz:\s*([0-9.]+)', txt)
        mhz = float(m.group(1)) if m else None
        m = re.search(r'Thread\(s\) per core:\s*(\d+)', txt)
        tpc = int(m.group(1)) if m else 1
        m = re.search(r'Model name:\s*(.+)', txt)
        model = m.group(1).strip() if m else ''
    except Exception:                                      # noqa: BLE001
        pass
    if not mhz:
        try:
            mhz = max(float(l.split(':')[1]) for l in open('/proc/cpuinfo') if l.startswith('cpu MHz'))
        except Exception:                                  # noqa: BLE001
            mhz = 3000.0
    cores = max(1, avail // max(tpc, 1))
    return {'model': model, 'physical_cores_available': cores, 'threads_per_core': tpc, 'clock_mhz': mhz, 'flop_per_cycle_per_core': 64,
            'peak_tflops': cores * 64 * mhz * 1e6 / 1e12}


def cpu_honest_line(cfg, shape, torch_variant, cores, avail, torch):
    """the CPU proxy is capped at 16 threads because the RECURRENCE stops scaling there -- but 77 % of the step's
    FLOPs are the three contractions over the vocabulary (logits, dH, dW), which do scale.  Those three are timed alone at the
    proxy's thread count and at every available core; `composed` = the measured step with its three big contractions re-priced at
    all cores (an estimate of a per-op thread policy, labelled as such), and host_peak_frac says how far either is from the host."""
    N, K, Q = shape
    n, H, V1 = N * (K + Q) * cfg['max_len'], cfg['hidden_size'], cfg['input_size'] + 1
    a = torch.randn(n, H); w = torch.randn(H, V1); g = torch.randn(n, V1)

    def big3():
        t0 = time.perf_counter()
        for _ in range(2):
            _ = a @ w; _ = g @ w.t(); _ = a.t() @ g
        return (time.perf_counter() - t0) / 2
    old = torch.get_num_threads()
    torch.set_num_threads(cores); big3(); t_few = big3()
    scan = {}
    for nt in sorted({min(avail, x) for x in (32, 64, 128, avail // 2, avail)} - {cores}):
        torch.set_num_threads(nt); big3(); scan[nt] = big3()
    torch.set_num_threads(old)
    scan[cores] = t_few
    best_nt = min(scan, key=scan.get)
    t_all = scan[best_nt]
    gflop3 = 3 * 2.0 * n * H * V1 / 1e9
    t_step = 1.0 / torch_variant['train_episodes_per_s']
    t_comp = max(t_step - t_few + t_all, 1e-9)
    gf_step = 3 * 2.0 * n * ((cfg['embedding_size'] + H) * 4 * H + (cfg['n_layers'] - 1) * 2 * H * 4 * H + H * V1) / 1e9
    peak = host_peak_model(avail)
    return {'big3_gflop': gflop3, 'big3_s_by_threads': {str(k): v for k, v in sorted(scan.items())}, 'big3_best_threads': best_nt,
            'big3_tflops_best': gflop3 / t_all / 1e3, 'step_s_measured_at_%d_threads' % cores: t_step,
            'composed_episodes_per_s': 1.0 / t_comp, 'composed_note': 'measured torch-CPU step with its three vocabulary contractions re-priced at their best thread count (estimate)',
            'host_peak': peak, 'host_peak_frac_measured': gf_step / t_step / 1e3 / peak['peak_tflops'],
            'host_peak_frac_composed': gf_step / t_comp / 1e3 / peak['peak_tflops']}


def val_nll_leg(cfg, eng, step, args, shape, device):
    """The metric's other half (BASELINE.json: "episodes/s + val NLL"; reference src/train/train.py:27-33,80-81 evaluate()): train on until
    global_step = args.val_nll, then the mean query NLL of 8 held-out episodes of the same distribution (a pool seeded differently from the
    training pool), next to the fp32 CPU restatement's value AT THE SAME PARAMETERS (read back from the device) and to the untrained model's."""
    import torch
    from fsmg.binding import FsmgModel
    from oracle import lstm_oracle as O
    from oracle.torch_ref import TorchRef
    N, K, Q = shape
    gen = padded_zipf_episodes if args.pool == 'padded-zipf' else synthetic_episodes
    held = gen(8, N, K, Q, cfg['max_len'], cfg['input_size'], seed=987654)
    i = 0
    while eng.step < args.val_nll:
        step(args.warmup + i)
        i += 1
    torch.cuda.synchronize()
    trained = eng.step
    losses = eng.read_losses(min(trained, 1024))
    qs = np.stack([q for _, q in held])
    gpu = [float(x) for x in eng.eval_batch(qs)]
    ref = TorchRef(cfg, eng.get_params(), dtype=torch.float32, threads=min(_affinity(), 16))
    cpu = [float(ref.eval(q)) for _, q in held]
    fresh = FsmgModel(cfg, device=device, max_sequences=N * (K + Q))
    fresh.init_params(cfg['seed'])
    untrained = float(np.mean(fresh.eval_batch(qs)))
    fresh.close()
    g, c = float(np.mean(gpu)), float(np.mean(cpu))
    return {'val_nll_gpu': g, 'val_nll_cpu': c, 'rel_diff': abs(g - c) / abs(c), 'per_episode_max_rel_diff': max(abs(a - b) / abs(b) for a, b in zip(gpu, cpu)),
            'val_nll_untrained': untrained, 'train_steps': int(trained), 'held_out_episodes': len(held),
            'train_loss_first': float(losses[0]), 'train_loss_last': float(losses[-1]),
            'note': 'mean query NLL (nats / token) of %d held-out %s episodes after %d train steps; cpu = the fp32 torch-CPU restatement at the parameters read '
                    'back from the device; bar 1e-4 relative' % (len(held), args.pool, trained)}


def exchange_plans(world, env, same_gpu=False):
    """The gradient-exchange schedules a run times, in order: [(name, model-config overrides)].  One GPU: nothing to exchange.  N > 1: by
    DEFAULT only "graph_end" -- what `train.train` and the shipped configuration run (one graph per backward pass, the three buckets
    reduced on the communication stream when it ends); it is the conservative one: no RCCL kernel ever runs beside a persistent chain
    that needs every CU of its XCDs.  FSMG_BENCH_SCHEDULES=all, or a comma list of split_bucket0 / split_after_chain / one_collective /
    library_rccl, adds the others BEHIND it (main() times them only after graph_end has produced a guarded result on every rank); the
    library-owned exchange (fsmg_comm_*: one rank is all it has ever seen) additionally needs FSMG_BENCH_LIBRARY_RCCL=1 and real GPUs."""
    if world == 1:
        return [('single_gpu', {})]
    optional = [('split_bucket0', {'dp_split_backward': True}), ('split_after_chain', {'dp_split_backward': 2}), ('one_collective', {'bucketed': False})]
    if env.get('FSMG_BENCH_LIBRARY_RCCL', '0') == '1' and not same_gpu:
        optional.append(('library_rccl', {'dp_exchange': 'library', 'dp_split_backward': True}))
    want = [w.strip() for w in env.get('FSMG_BENCH_SCHEDULES', '').split(',') if w.strip()]
    plans = [('graph_end', {})]
    if 'all' in want:
        plans += optional
    else:
        plans += [pl for pl in optional if pl[0] in want]
    return plans


def self_launch(args):
    """`python bench.py --gpus N` outside torchrun: start N ranks of this script under torch.distributed.run (rank r -> GPU r,
    rendezvous on 127.0.0.1), pass their output through and print rank 0's JSON line LAST."""
    import socket
    import subprocess
    same_gpu = os.environ.get('FSMG_BENCH_SAME_GPU', '0') == '1'
    if not same_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit('bench.py --gpus %d: only %d GPU(s) visible (FSMG_BENCH_SAME_GPU=1 runs all ranks on GPU 0 over gloo: '
                             'a dry run of the launcher, its numbers mean nothing)' % (args.gpus, have))
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC: RCCL needs it on this driver
    env.setdefault('OMP_NUM_THREADS', str(max(1, _affinity() // max(args.gpus, 1))))
    env['FSMG_BENCH_SELF_LAUNCHED'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log('self-launch: %s' % ' '.join(cmd[1:]))
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, universal_newlines=True, env=env)
    lines = proc.stdout.splitlines()
    result = None
    for i in range(len(lines) - 1, -1, -1):
        if lines[i].startswith('{"metric"'):
            result = lines.pop(i)
            break
    for line in lines:
        print(line, file=sys.stderr)
    if result is None:
        raise SystemExit('bench.py --gpus %d: the ranks printed no result line (exit code %d)' % (args.gpus, proc.returncode))
    sys.stdout.flush()
    print(result)
    return proc.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-breakdown', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the clock probe, the validation leg and the host-path legs (the other_configs sub-runs)')
    ap.add_argument('--val-nll', type=int, default=0, metavar='STEPS',
                    help='after the timed region: train on until global_step = STEPS, then the mean query NLL of 8 HELD-OUT episodes (LSTMBaseline.eval) '
                         'beside the CPU restatement\'s value at the same parameters -> `val` (the metric\'s "val NLL" half; meaningful on --pool padded-zipf)')
    ap.add_argument('--pool', default='uniform', choices=['uniform', 'padded-zipf'],
                    help='token pool: SURVEY 8(d)\'s i.i.d. uniform ids (the headline), or zero-padded Zipf songs (what real data looks like; a diagnostic leg)')
    ap.add_argument('--config', default='cfg-B', choices=['cfg-B', 'cfg-C', 'cfg-C-T128', 'cfg-D', 'cfg-Bx4', 'cfg-Bx8', 'cfg-E', 'ref-default'])
    args = ap.parse_args()
    if args.gpus > 1 and 'RANK' not in os.environ and int(os.environ.get('WORLD_SIZE', '1')) == 1:
        sys.exit(self_launch(args))
    # stdout carries the result line and nothing else: what the plugins print on the way (recover_or_init's "Initializing vars")
    # goes to stderr with the progress log
    # (at the descriptor level too: gloo / RCCL print their connection banners from C++)
    sys.stdout.flush()
    result_stream = os.fdopen(os.dup(1), 'w')
    os.dup2(2, 1)
    sys.stdout = sys.stderr

    import torch
    import torch.distributed as dist
    from fsmg.dist import EpisodeParallel, init_from_env

    # FSMG_BENCH_SAME_GPU=1 (diagnostic): all ranks on GPU 0 over gloo -- exercises the N > 1 code path on a 1-GPU box (RCCL
    # refuses two ranks on one device); the numbers of such a run mean nothing
    same_gpu = os.environ.get('FSMG_BENCH_SAME_GPU', '0') == '1'
    rank, world = init_from_env('gloo' if same_gpu else 'nccl')
    if world != args.gpus:
        raise SystemExit('launched with WORLD_SIZE=%d but --gpus %d' % (world, args.gpus))
    local = 0 if same_gpu else int(os.environ.get('LOCAL_RANK', 0))
    if same_gpu:
        os.environ['LOCAL_RANK'] = '0'
    torch.cuda.set_device(local)
    global N_WAY, K_SHOT, Q_QUERY
    base = CFG_B
    if args.config != 'cfg-B':
        base, N_WAY, K_SHOT, Q_QUERY = OTHER[args.config]
    cfg = dict(base, device=local)
    B = N_WAY * (K_SHOT + Q_QUERY)
    maml = (cfg['inner_steps'], cfg['inner_lr']) if args.config == 'cfg-E' else None

    pool_host = (padded_zipf_episodes if args.pool == 'padded-zipf' else synthetic_episodes)(POOL, N_WAY, K_SHOT, Q_QUERY, cfg['max_len'], cfg['input_size'], seed=1234 + rank)
    d_sup = torch.from_numpy(np.stack([s for s, _ in pool_host])).cuda()
    d_qry = torch.from_numpy(np.stack([q for _, q in pool_host])).cuda()
    sup_stride, qry_stride = d_sup[0].numel() * 4, d_qry[0].numel() * 4

    log('rank %d/%d: pool on device, creating model' % (rank, world))
    if maml:
        from models.maml_lstm import MAMLLSTM as Model
    else:
        from models.lstm_baseline import LSTMBaseline as Model
    shape = (N_WAY, K_SHOT, Q_QUERY)
    kw = dict(maml=maml) if maml else {}

    def make(over):
        """a fresh model + its episode-parallel driver; `over`: model-config keys the plugin hands to fsmg_config (gemm,
        schedule, recurrence, dp_split_backward) plus `bucketed` for the exchange"""
        over = dict(over)
        bucketed = over.pop('bucketed', None)
        m = Model(dict(cfg, max_sequences=N_WAY * (K_SHOT + Q_QUERY), **over))
        m.recover_or_init('')
        return m, EpisodeParallel(m, bucketed=bucketed)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(par_, eng_, tag):
        """W untimed warm-up steps, then REPEATS regions of EXACTLY K steps, each between barrier + synchronize on both sides;
        returns this rank's seconds per region and the guard that proves every region did the work it claims.  A time-out
        inside a region puts the handle on per-step launches: the region is reported as it was measured, and the persistent
        path is re-armed for the NEXT one (instead of poisoning the rest of the sample for `fallback` steps)."""
        def step_(i):
            e = i % POOL
            par_.train_step(d_sup.data_ptr() + e * sup_stride, d_qry.data_ptr() + e * qry_stride,
                            want_loss=False, shape=shape, **kw)
        step0 = eng_.step
        for i in range(args.warmup):
            step_(i)
            if i == 0:
                torch.cuda.synchronize()
                log('%s: first step done' % tag)
        els, rearmed, regions = [], 0, []
        step_before = eng_.step
        for r in range(REPEATS):
            barrier()
            s0 = eng_.step
            t0 = time.perf_counter()
            for i in range(args.steps):
                step_(args.warmup + r * args.steps + i)
            barrier()
            els.append(time.perf_counter() - t0)
            st = eng_.stats()
            regions.append({'advanced_by': eng_.step - s0, 'timeouts': st['timeouts'], 'persistent_path': bool(st['persistent_path'])})
            if not st['persistent_path'] and r + 1 < REPEATS:
                eng_.debug_set('persistent', 1)
                rearmed += 1
        log('%s: %d timed regions done: %s s for %d steps each' % (tag, REPEATS, ' '.join('%.4f' % e for e in els), args.steps))
        # ---- guard: a step whose persistent kernel times out (or whose batch is rejected) is SKIPPED on the device -- no Adam,
        # no global_step -- so a poisoned region would time no-ops
        step_after = eng_.step
        stats = eng_.stats()
        g = {'rank': rank, 'global_step_before': step_before, 'global_step_after': step_after,
             'advanced_by': step_after - step_before, 'expected': args.steps * REPEATS, 'regions': regions, 'rearmed': rearmed,
             'warmup_advanced_by': step_before - step0, 'timeouts': stats['timeouts'],
             'steps_skipped_timeout': stats['steps_skipped_timeout'], 'steps_skipped_token_range': stats['steps_skipped_token_range'],
             'persistent_path': bool(stats['persistent_path']), 'fallback_steps_left': stats['fallback_steps_left'],
             'xcd_local_kernels': stats['xcd_launches'] > 0,
             'xov_selfcheck_mismatches': stats.get('xov_selfcheck_mismatches'), 'softmax_range_rows': stats.get('softmax_range_rows'),
             'steps_skipped_softmax_range': stats.get('steps_skipped_softmax_range'),
             # [passes whose gated projection was recomputed and compared, passes under the XCD-partitioned order, the period]
             'xov_selfcheck': [int(v) for v in eng_.debug_read('xov_selfcheck', 3)], 'aux_stream_tries': stats.get('aux_stream_tries'),
             'fused_softmax_taken': bool(eng_.debug_read('fused_softmax', 2)[1]),
             'ok': (step_after - step_before == args.steps * REPEATS and step_before - step0 == args.warmup and stats['timeouts'] == 0
                    and stats['steps_skipped_timeout'] == 0 and stats['steps_skipped_token_range'] == 0
                    and not stats.get('xov_selfcheck_mismatches') and not stats.get('softmax_range_rows') and not stats.get('steps_skipped_softmax_range'))}
        return els, g

    plans = exchange_plans(world, os.environ, same_gpu)
    if any(pl[0] == 'library_rccl' for pl in plans):
        os.environ['FSMG_ALLOW_LIBRARY_RCCL'] = '1'
    schedules, built, failed_plans, skipped_plans = {}, {}, {}, {}
    keeper = None          # the ONE handle that stays alive: the first plan that ran, replaced by the first whose guard holds (what `used` picks)

    def drop(m_):
        try:
            m_.engine.close()
        except Exception:                      # noqa: BLE001
            pass
    for name, env_over in plans:
        # the opt-in schedules are timed only once the default one has produced a guarded result on every rank: the first minutes
        # on an N-GPU node belong to the schedule the shipped configuration runs, not to fall-backs of its variants
        if name != plans[0][0] and not (plans[0][0] in schedules and schedules[plans[0][0]]['guard_ok']):
            skipped_plans[name] = 'not timed: the default schedule %r did not produce a guarded result first' % plans[0][0]
            continue
        # a schedule that cannot be built or run on this box must not cost the run its result line: every rank reports whether
        # it got through, and the schedule counts only if all did (the first plan is the one the step has always used)
        err = None
        try:
            m_, p_ = make(env_over)
            el, g = timed(p_, m_.engine, name)
        except Exception as e:                  # noqa: BLE001 -- reported below, per rank
            err = '%s: %s' % (type(e).__name__, e)
            log('schedule %s FAILED on rank %d: %s' % (name, rank, err))
        if world > 1:
            errs = [None] * world
            dist.all_gather_object(errs, err)
        else:
            errs = [err]
        if any(errs):
            if not schedules and name == plans[-1][0]:
                raise SystemExit('no exchange schedule ran: %r' % errs)
            failed_plans[name] = errs
            continue
        per = [el]                                 # per rank: seconds of each of the REPEATS regions
        gs = [g]
        if world > 1:
            gs = [None] * world
            dist.all_gather_object(gs, g)
            per = [None] * world
            dist.all_gather_object(per, el)
        worst = [max(p[r] for p in per) for r in range(REPEATS)]          # a region ends when its slowest rank has ended
        ms = sorted(1e3 * w / max(args.steps, 1) for w in worst)
        med = ms[len(ms) // 2] if len(ms) % 2 else 0.5 * (ms[len(ms) // 2 - 1] + ms[len(ms) // 2])
        schedules[name] = {'value': world * 1e3 / med, 'ms_per_step': med,
                           'ms_per_step_min': ms[0], 'ms_per_step_max': ms[-1], 'repeats': REPEATS,
                           'ms_per_step_regions': [1e3 * w / max(args.steps, 1) for w in worst],
                           'per_rank_ms_per_step': [1e3 * sorted(p)[len(p) // 2] / max(args.steps, 1) for p in per],
                           'guard_ok': all(x['ok'] for x in gs), 'guard_per_rank': gs}
        # Only one handle is kept alive between plans: nothing a plan measures may depend on the handles of the plans before it
        # (with five handles alive, every odd one ran 32 % slower until the auxiliary stream lost its priority)
        if keeper is None:
            keeper = name
            built[name] = (m_, p_, el)
        elif not schedules[keeper]['guard_ok'] and schedules[name]['guard_ok']:
            drop(built.pop(keeper)[0])
            keeper = name
            built[name] = (m_, p_, el)
        else:
            drop(m_)
            del m_, p_
    # `value` is the schedule a user gets from the default configuration (plans[0]: one episode-parallel pass, buckets released when
    # the backward pass ends) -- the other schedules of an N > 1 run are listed beside it, never picked for the headline
    used = keeper
    model, par, local_elapsed = built[used]
    eng = model.engine
    elapsed = schedules[used]['ms_per_step'] * max(args.steps, 1) / 1e3
    per_rank = [t * max(args.steps, 1) / 1e3 for t in schedules[used]['per_rank_ms_per_step']]
    guard = dict(schedules[used]['guard_per_rank'][0])
    guard['ok'] = schedules[used]['guard_ok']
    if not guard['ok']:
        log('GUARD FAILED: %r' % schedules[used]['guard_per_rank'])

    def step(i):
        e = i % POOL
        par.train_step(d_sup.data_ptr() + e * sup_stride, d_qry.data_ptr() + e * qry_stride,
                       want_loss=False, shape=shape, **kw)

    world_info = None
    without_exchange_ms = None
    extras_failed = {}                         # optional legs that raised: reported, never fatal
    if world > 1:
        props = torch.cuda.get_device_properties(local)
        mine = {'rank': rank, 'local_device': local, 'name': props.name, 'uuid': str(getattr(props, 'uuid', '')),
                'pci_bus_id': getattr(props, 'pci_bus_id', None)}
        devs = [None] * world
        dist.all_gather_object(devs, mine)
        world_info = {'size': dist.get_world_size(), 'backend': dist.get_backend(), 'devices': devs,
                      'distinct_devices': len({(d['local_device'], d['uuid'], d['pci_bus_id']) for d in devs}),
                      'launcher': 'bench.py self-launch' if os.environ.get('FSMG_BENCH_SELF_LAUNCHED') == '1' else 'torchrun',
                      'same_gpu_dry_run': same_gpu}
        # the same loop with NO exchange (replicas diverge: timing only, on its own model) -> what each schedule's exchange adds
        try:
            m0, p0 = make({})
            p0.exchange = False
            el0, _ = timed(p0, m0.engine, 'no_exchange')
            per0 = [None] * world
            dist.all_gather_object(per0, el0)
            w0 = sorted(max(p[r] for p in per0) for r in range(REPEATS))
            without_exchange_ms = 1e3 * w0[len(w0) // 2] / max(args.steps, 1)
            del m0, p0
        except Exception as e:                 # noqa: BLE001 -- an extra leg must not cost the result line
            log('no_exchange leg failed: %r' % (e,))
            extras_failed['no_exchange'] = repr(e)
    # ---- roofline leg: the same K steps again with HIP events around every launch of the fused-cell kernels on the library's
    # stream, in the schedule of the timed region (event timing replaces the hipGraph replay by the same launches, eagerly)
    cell = {}
    if not maml:
        try:
            for cls in CELL_CLASSES:
                eng.timing_select(cls)
                eng.timing_enable(True)
                eng.timing_reset()
                for i in range(args.steps):
                    step(args.warmup + args.steps + i)
                cell[cls] = eng.timing_read(cls)
                eng.timing_enable(False)
        except Exception as e:                 # noqa: BLE001
            log('roofline leg failed: %r' % (e,))
            extras_failed['roofline'] = repr(e)
            cell = {}
            try:
                eng.timing_enable(False)
            except Exception:                  # noqa: BLE001
                pass
    # sustained shader clock beside the step: a one-wave probe on its own stream compares the shader-clock counter with the 100 MHz
    # real-time counter while a few train steps run (the peaks of every `frac` are quoted at the 2.4 GHz spec clock)
    clock_ghz = None
    try:
        if args.no_extras:
            raise StopIteration
        n_clk = max(4, min(args.steps, 12))
        eng.synchronize()
        for i in range(2):
            step(i)
        eng.clock_begin(int(0.8 * n_clk * 1e3 * schedules[used]['ms_per_step']))
        for i in range(n_clk + 2):
            step(2 + i)
        clock_ghz = eng.clock_end()
        eng.synchronize()
    except StopIteration:
        pass
    except Exception as e:                     # noqa: BLE001
        log('clock probe failed: %r' % (e,))
        extras_failed['clock_probe'] = repr(e)
    # communication (N > 1): the all-reduce of the flat gradient buffer alone on the library's stream, and per schedule what the
    # exchange adds to the same timed loop without any exchange (= the exposed, un-overlapped part)
    comm = None
    if world > 1:
        ar_ms, nbytes = None, None
        try:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with model.stream_context():
                dist.all_reduce(model.grad_tensor, op=dist.ReduceOp.SUM)
                ev0.record()
                for _ in range(5):
                    dist.all_reduce(model.grad_tensor, op=dist.ReduceOp.SUM)
                ev1.record()
            torch.cuda.synchronize()
            nbytes = int(model.grad_tensor.numel() * 4)
            ar_ms = ev0.elapsed_time(ev1) / 5
        except Exception as e:                 # noqa: BLE001
            log('standalone all-reduce leg failed: %r' % (e,))
            extras_failed['allreduce_standalone'] = repr(e)
        comm = {'allreduce_ms_standalone': ar_ms, 'bytes': nbytes,
                'allreduce_busbw_GBps': 2.0 * (world - 1) / world * nbytes / (ar_ms * 1e-3) / 1e9 if ar_ms else None,
                'ms_per_step_without_exchange': without_exchange_ms,
                'exposed_ms': ({n: schedules[n]['ms_per_step'] - without_exchange_ms for n in schedules}
                               if without_exchange_ms is not None else None),
                'note': 'exposed_ms = ms_per_step of the schedule minus the same K-step loop with the exchange switched off (timing only)'}
    losses = eng.read_losses(min(args.steps, 1024)) if args.steps > 0 else np.zeros(1)

    out = None
    if rank == 0:
        gf = algorithmic_gflop(cfg, B)
        value = world * args.steps / elapsed
        total_gflop = 3 * (gf['gemm_zx'] + gf['lstm_fwd'] + gf['gemm_logits'])
        names = {'cfg-B': 'cfg-B: synthetic V=10000 (V1=10001) T=128 5-way 5-shot 4-query (B=45 sequences/episode), LSTM E=250 H=512 L=1',
                 'ref-default': 'ref-default: the reference\'s shipped YAML defaults (lstm_baseline.yaml E=250 H=200 L=1, lyrics.yaml T=50, 5shot.yaml 5-way 5-shot 4-query), '
                                'synthetic V=10000 -- a diagnostic run, not the headline workload'}
        wl = names.get(args.config, '%s (diagnostic run, not the headline workload): V=%d T=%d %d-way %d-shot %d-query, LSTM E=%d H=%d L=%d'
                       % (args.config, cfg['input_size'], cfg['max_len'], N_WAY, K_SHOT, Q_QUERY, cfg['embedding_size'],
                          cfg['hidden_size'], cfg['n_layers']))
        out = {
            'metric': 'episodes/s (LSTM-baseline train step, synthetic vocab=10k seq_len=128 5-way/5-shot h=512)',
            'value': value, 'unit': 'episodes/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / max(args.steps, 1), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            # the type the path computes in: fp32 values, fp32 accumulation; with the default GEMM kind every product is assembled from
            # bf16 pieces on the bf16 matrix pipe (exact three-way split, six products) -- said here, not only in `arithmetic`
            'dtype': 'f32' if os.environ.get('FSMG_GEMM', 'bx3') == 'f32' else 'f32 (bf16x3-split products, fp32 accumulate)',
            'arithmetic': ARITHMETIC_GEMM, 'gemm_kind': os.environ.get('FSMG_GEMM', 'bx3'), 'data': 'synthetic' if args.pool == 'uniform' else 'synthetic (zero-padded Zipf songs: a diagnostic pool, not SURVEY 8(d)\'s)',
            'config': {'workload': wl + (', MAML-style step (1 inner clipped-SGD step on the support rows + outer clip+Adam on the query gradient)'
                                         if maml else ', full train step (fwd+BPTT+clip+Adam)') + ', one episode per GPU per step',
                       'episodes_per_step': world, 'parallelism': 'episode-parallel x%d, 1 RCCL all-reduce/step' % world},
            'spread': {'repeats': REPEATS, 'ms_per_step_median': schedules[used]['ms_per_step'], 'ms_per_step_min': schedules[used]['ms_per_step_min'],
                       'ms_per_step_max': schedules[used]['ms_per_step_max'], 'ms_per_step_regions': schedules[used]['ms_per_step_regions'],
                       'note': 'value / ms_per_step = the MEDIAN of %d timed regions of `steps` steps each (barrier + synchronize around every region)' % REPEATS},
            'guard': guard,
            # whole-step algorithmic TFLOP/s over the fp32-MFMA peak (the bound of round 1's arithmetic; > 1 is possible now
            # that the GEMMs run on the bf16 pipe) and over the blended bound (GEMM FLOPs at peak/6 of bf16, cell FLOPs at fp32 MFMA)
            'step_mfma_frac': None if maml else (total_gflop / (1e3 * elapsed / max(args.steps, 1))) / PEAK_F32_MFMA_TFLOPS,
            'step_tflops': None if maml else total_gflop / (1e3 * elapsed / max(args.steps, 1)),
            'roofline_step': None if maml else step_roofline(cfg, B, gf, 1e3 * elapsed / max(args.steps, 1), bool(guard.get('fused_softmax_taken'))),
            'final_loss': float(losses[-1]), 'first_loss': float(losses[0]),
            'per_rank_ms_per_step': [1e3 * t / max(args.steps, 1) for t in per_rank], 'comm': comm,
            'schedule_used': used,
            'schedules': {n: {k: v for k, v in sc.items() if k != 'guard_per_rank'} for n, sc in schedules.items()},
            'schedules_failed': failed_plans, 'schedules_skipped': skipped_plans,
            'guard_per_rank': schedules[used]['guard_per_rank'], 'world': world_info,
        }
        if cell:
            T = cfg['max_len']
            tot_ms = sum(ms for ms, _ in cell.values())
            tot_n = sum(n for _, n in cell.values())
            gflop = sum(gf[c] for c in cell) * args.steps                 # both directions, every launch of the repeat
            ach = gflop / tot_ms if tot_ms > 0 else 0.0
            steps_per_launch = {c: (T * cfg['n_layers'] * args.steps) / max(n, 1) for c, (ms, n) in cell.items()}
            try:
                cell_bx3 = bool(eng.debug_read('xcd_bx3', 1)[0])
            except Exception:                  # noqa: BLE001
                cell_bx3 = False
            if cell_bx3 and cfg['hidden_size'] > 512:      # hidden 1024, three row groups per weight copy and up (csrc/lstm_pair16.h, round 6)
                cell_kernel = ('k_lstm_fwd_pair16 + k_lstm_bwd_pair16 (recurrent [B x H] x [H x 4H] contraction as an exact three-way bf16 split on '
                               'v_mfma_f32_16x16x32_bf16, fp32 accumulation; K_h planes 0-1 in the registers of an XCD pair, plane 2 in 128 KiB of LDS per CU')
            elif cell_bx3:    # hidden 512, > 64 rows: the recurrent product as six bf16 products per fp32 product (csrc/lstm_xcd.hip)
                cell_kernel = ('k_lstm_fwd_xcd16 + k_lstm_bwd_xcd16 (recurrent [B x H] x [H x 4H] contraction as an exact three-way bf16 split on '
                               'v_mfma_f32_16x16x32_bf16, fp32 accumulation')
            elif cfg['hidden_size'] > 512:
                cell_kernel = 'k_lstm_fwd_pair* + k_lstm_bwd_pair* (recurrent [B x H] x [H x 4H] contraction on v_mfma_f32_4x4x1_16B_f32, K_h in the registers of an XCD pair'
            elif 192 < cfg['hidden_size'] <= 256:
                cell_kernel = ('k_lstm_fwd_slice + k_lstm_bwd_slice (hidden size padded to 256: two register-resident copies of K_h per XCD, sixteen row slices, '
                               'recurrent [B x H] x [H x 4H] contraction on v_mfma_f32_4x4x1_16B_f32')
            elif cfg['hidden_size'] != 512:
                cell_kernel = ('k_lstm_fwd_chain + k_lstm_bwd_rs / k_lstm_bwd_chain (column-split persistent kernels: gate columns over the chip, cross-XCD hand-off; '
                               'recurrent [B x H] x [H x 4H] contraction on v_mfma_f32_16x16x4_f32')
            else:
                cell_kernel = 'k_lstm_fwd_xcd + k_lstm_bwd_xcd (recurrent [B x H] x [H x 4H] contraction on v_mfma_f32_4x4x1_16B_f32'
            cell_arith = 'bf16x3' if cell_bx3 else ('f32' if (cfg['hidden_size'] in (512, 1024) or 192 < cfg['hidden_size'] <= 256) else 'f32_16x16x4')
            out['arithmetic'] = ARITHMETIC_GEMM + ARITHMETIC_CELL[cell_arith]
            out['roofline'] = {
                'bound': 'mfma',
                # the matrix pipe the timed cell kernels issue on, and the fraction of THAT pipe's bound for fp32-equivalent work;
                # `frac` stays the north star's definition (fp32-MFMA peak) so that rounds compare
                'pipe': ('v_mfma_f32_16x16x32_bf16 (bf16 pipe, six products per fp32 product: bound %.0f TFLOP/s fp32-equivalent)' % PEAK_BX3_TFLOPS) if cell_bx3
                        else 'v_mfma_f32_4x4x1_16B_f32 (fp32 pipe: bound %.1f TFLOP/s)' % PEAK_F32_MFMA_TFLOPS,
                'peak_pipe_used': PEAK_BX3_TFLOPS if cell_bx3 else PEAK_F32_MFMA_TFLOPS,
                'frac_pipe_used': ach / (PEAK_BX3_TFLOPS if cell_bx3 else PEAK_F32_MFMA_TFLOPS),
                'forward_us_per_time_step': 1e3 * cell['lstm_fwd'][0] / (T * cfg['n_layers'] * args.steps),
                'backward_us_per_time_step': 1e3 * cell['lstm_bwd'][0] / (T * cfg['n_layers'] * args.steps),
                'kernel': 'fused LSTM cell: %s + gate nonlinearities / gate gradients + state update, %d dependent time steps per launch)'
                          % (cell_kernel, int(steps_per_launch['lstm_fwd'])),
                'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_F32_MFMA_TFLOPS,
                'launches': tot_n, 'avg_launch_ms': tot_ms / max(tot_n, 1),
                # what one launch has to move (per direction, mean of the two): forward reads Z and writes the four gates, c and h; backward reads
                # gates, c and dH and writes dZ -- 4H + 4H + 2H floats per row and time step either way
                'algorithmic_bytes_per_launch': 4.0 * B * T * 10 * cfg['hidden_size'] * cfg['n_layers'] / max(cell['lstm_fwd'][1] / max(args.steps, 1), 1),
                'traffic_note': None if cfg['hidden_size'] > 512 else ('forward = its algorithmic bytes; the bf16-split backward hands dh over as a reduce-scatter (32 partial [16 x 512] tiles per '
                                 'XCD and time step, 1 MB, plus the sentinel refill): about a third of those L2 writes are written back (WRITE_SIZE 334 MB against '
                                 '47 MB of dZ) = 0.7 TB/s during a latency-bound chain (DESIGN.md 9.3)') if cell_bx3 else None,
                'algorithmic_gflop_per_launch': gf['lstm_fwd'] / max(cell['lstm_fwd'][1] / max(args.steps, 1), 1),
                'forward': {'avg_launch_ms': cell['lstm_fwd'][0] / max(cell['lstm_fwd'][1], 1), 'us_per_time_step': 1e3 * cell['lstm_fwd'][0] / (T * cfg['n_layers'] * args.steps),
                            'frac': gf['lstm_fwd'] * args.steps / cell['lstm_fwd'][0] / PEAK_F32_MFMA_TFLOPS},
                'backward': {'avg_launch_ms': cell['lstm_bwd'][0] / max(cell['lstm_bwd'][1], 1), 'us_per_time_step': 1e3 * cell['lstm_bwd'][0] / (T * cfg['n_layers'] * args.steps),
                             'frac': gf['lstm_bwd'] * args.steps / cell['lstm_bwd'][0] / PEAK_F32_MFMA_TFLOPS},
                'note': 'algorithmic 2*B*H*4H FLOP per time step (SURVEY.md 8d, recurrent-only) over the HIP-event time of the launches in the timed '
                        'schedule; latency-bound chain: us_per_time_step is the figure to watch (0.60 us at the MFMA peak)'
                        + ('; peak = the fp32 MFMA peak the other kernel family is priced against -- this family runs on the bf16 pipe, whose '
                           'bound for fp32-equivalent work is %.0f TFLOP/s (frac_bf16_split)' % PEAK_BX3_TFLOPS if cell_bx3 else '')}
            if clock_ghz:
                out['roofline'].update({'clock_ghz': clock_ghz, 'spec_clock_ghz': 2.4, 'frac_at_sustained_clock': ach / (PEAK_F32_MFMA_TFLOPS * clock_ghz / 2.4),
                                        'clock_note': 'shader clock sustained while train steps run (s_memtime against the 100 MHz s_memrealtime, one probe wave on its own stream); peak and frac are at the 2.4 GHz spec clock'})
            if cell_bx3:
                out['roofline']['frac_bf16_split'] = ach / PEAK_BX3_TFLOPS
            try:
                xp = eng.debug_read('xcd_partitioned', 2)
            except Exception:                  # noqa: BLE001
                xp = [0.0, 8.0]
            variant = (('pair16<3' if cfg['hidden_size'] > 512 else ('xcd16<4' if xp[0] else 'xcd16<2')) if cell_bx3
                       else ('xcd<2' if cfg['hidden_size'] == 512 else None))
            out['roofline']['traffic'], out['roofline']['traffic_source'] = hbm_traffic(variant)
            out['roofline']['kernel_variant'] = variant
            if xp[0]:
                # XCD-partitioned order: the chains run on a few XCDs BESIDE the projection / dW GEMMs, slower per step than alone on
                # the whole chip -- by design.  Both ways of pricing them: against the whole chip's peak (frac) and against the peak
                # of the XCDs they occupy; the step as a whole is priced by roofline_step
                nx = max(int(xp[1]), 1)
                out['roofline'].update({'schedule': 'xcd_partitioned', 'xcds_occupied': nx,
                                        'frac_of_occupied_xcds_peak': ach / (PEAK_F32_MFMA_TFLOPS * nx / 8.0),
                                        'frac_bf16_split_of_occupied_xcds': ach / (PEAK_BX3_TFLOPS * nx / 8.0)})
    if rank == 0 and world == 1 and args.config == 'cfg-B' and not args.no_other_configs:
        # the other BASELINE.json configurations and the serial-order fused cell on the driver's record -- 20 steps
        # each on fresh handles inside this run (one timed region between two synchronisations; a diagnostic, never the headline)
        try:
            out['other_configs'] = other_configs(log)
            oc = out['other_configs']
            # scalars INSIDE roofline / config: the driver's record keeps those two dicts and drops extra top-level keys
            ser = oc.get('cfg-B-serial-order', {})
            if out.get('roofline') is not None and ser.get('fused_cell'):
                out['roofline'].update({'serial_order_fused_cell_frac': ser['fused_cell']['frac'],
                                        'serial_order_fused_cell_pipe': 'v_mfma_f32_4x4x1_16B_f32 (fp32 MFMA), the whole chip, nothing beside it',
                                        'serial_order_fwd_us_per_time_step': ser['fused_cell']['us_per_time_step']['lstm_fwd'],
                                        'serial_order_bwd_us_per_time_step': ser['fused_cell']['us_per_time_step']['lstm_bwd'],
                                        'serial_order_value': ser.get('value')})
            for leg in ('cfg-C', 'cfg-C-T128', 'cfg-D', 'cfg-E', 'ref-default'):
                if 'value' in oc.get(leg, {}):
                    out['config']['other_%s_episodes_per_s' % leg] = oc[leg]['value']
            val = oc.get('cfg-B-padded-zipf-pool', {}).get('val')
            if val:
                out['config'].update({'val_nll': val['val_nll_gpu'], 'val_nll_cpu_same_parameters': val['val_nll_cpu'], 'val_nll_rel_diff': val['rel_diff'],
                                      'val_nll_untrained': val['val_nll_untrained'], 'val_nll_train_steps': val['train_steps'],
                                      'val_nll_note': val['note']})
        except Exception as e:                 # noqa: BLE001
            log('other_configs leg failed: %r' % (e,))
            extras_failed['other_configs'] = repr(e)
    if rank == 0 and not maml and not args.no_extras:
        try:
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
                           'mean_val_nll': float(np.mean(nll)), 'unit': 'eval episodes/s (query-only forward, %d sequences/episode)' % (N_WAY * Q_QUERY)}
        except Exception as e:                 # noqa: BLE001 -- an extra leg must not cost the result line
            log('eval leg failed: %r' % (e,))
            extras_failed['eval'] = repr(e)
    if rank == 0 and world == 1 and not maml and not args.no_extras:
        try:
            # the reference's calling convention: host numpy episodes in, the loss read back every step (one 23 KB H2D
            # token copy + one synchronising 4-byte D2H per step) -- PCIe-inclusive, never the headline value
            n_h = min(args.steps, 30)
            t0 = time.perf_counter()
            for i in range(n_h):
                eng.train_step(*pool_host[i % POOL], want_loss=True)
            out['host_synchronous'] = {'episodes_per_s': n_h / (time.perf_counter() - t0),
                                       'note': 'host token buffers + per-step loss readback (reference train() semantics)'}
            # what train.train does here: the split's token table resident in HBM, an episode = 45 row indices from the host
            # sampler (180 B H2D), losses read from the device ring once per log line
            table = np.concatenate([np.concatenate([s.reshape(-1, cfg['max_len']), q.reshape(-1, cfg['max_len'])]) for s, q in pool_host[:32]])
            eng.upload_table(0, table)
            rng = np.random.RandomState(7)
            idx = [(rng.randint(0, table.shape[0], size=(N_WAY, K_SHOT)), rng.randint(0, table.shape[0], size=(N_WAY, Q_QUERY))) for _ in range(64)]
            n_i = max(args.steps, 30)
            for i in range(3):
                eng.train_step_indexed(0, *idx[i], want_loss=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_i):
                eng.train_step_indexed(0, *idx[i % 64], want_loss=False)
            _ = eng.read_losses(min(n_i, 1024))
            out['host_indexed_deferred'] = {'episodes_per_s': n_i / (time.perf_counter() - t0),
                                            'note': 'host sampler indices into the device-resident table, losses read once per window (train.train fast path)'}
        except Exception as e:                 # noqa: BLE001 -- an extra leg must not cost the result line
            log('host_paths leg failed: %r' % (e,))
            extras_failed['host_paths'] = repr(e)
    if rank == 0 and world == 1 and not args.no_breakdown and not maml:
        try:
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
                        kernels[c]['frac_mfma_peak'] = gf[c] / per_step / PEAK_F32_MFMA_TFLOPS      # of the fp32-MFMA peak
                        if c.startswith('gemm_') and os.environ.get('FSMG_GEMM', 'bx3') != 'f32':
                            kernels[c]['frac_bx3_bound'] = gf[c] / per_step / PEAK_BX3_TFLOPS          # of bf16 peak / 6
            eng.timing_enable(False)
            out['kernels'] = kernels
            # the GEMMs as one family (63 % of the step): fp32-equivalent TFLOP/s over the instrumented pass against their own bound
            gemm = [c for c in kernels if c.startswith('gemm_')]
            g_ms = sum(kernels[c]['ms_per_step'] for c in gemm)
            g_gf = sum(gf[c] for c in gemm)
            bx3 = os.environ.get('FSMG_GEMM', 'bx3') != 'f32'
            out['roofline_gemm'] = {'bound': 'mfma', 'kernel': 'k_gemm_bx3h / k_gemm_bx3w / k_gemm_bx3 (all seven dense contractions of the step; 256 x 256 tiles for the three large ones)' if bx3 else 'k_gemm (fp32 MFMA)',
                                    'achieved': g_gf / g_ms, 'peak': PEAK_BX3_TFLOPS if bx3 else PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s (fp32-equivalent)',
                                    'frac': g_gf / g_ms / (PEAK_BX3_TFLOPS if bx3 else PEAK_F32_MFMA_TFLOPS), 'ms_per_step': g_ms,
                                    'note': 'peak = dense bf16 MFMA peak / 6 products per fp32 product at the 2.4 GHz spec clock; the chip sustains 1.92 GHz beside this kernel'}
            log('breakdown done')
        except Exception as e:                 # noqa: BLE001 -- an extra leg must not cost the result line
            log('breakdown leg failed: %r' % (e,))
            extras_failed['breakdown'] = repr(e)
    if rank == 0 and world == 1 and not args.no_breakdown and not maml and os.environ.get('FSMG_GEMM', 'bx3') != 'f32':
        # the same timed loop with every GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32): what the bf16-split GEMMs buy
        try:
            alt = Model(dict(cfg, gemm='f32'))
            alt.recover_or_init('')
            par_alt = EpisodeParallel(alt)
            def step_alt(i):
                e = i % POOL
                par_alt.train_step(d_sup.data_ptr() + e * sup_stride, d_qry.data_ptr() + e * qry_stride, want_loss=False, shape=shape)
            n_alt = min(args.steps, 60)
            for i in range(min(args.warmup, 10)):
                step_alt(i)
            torch.cuda.synchronize()
            a0 = alt.engine.step
            t0 = time.perf_counter()
            for i in range(n_alt):
                step_alt(10 + i)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            l_alt = alt.engine.read_losses(min(n_alt, 1024))
            out['alt_gemm_f32_mfma'] = {'value': n_alt / dt, 'unit': 'episodes/s', 'ms_per_step': 1e3 * dt / n_alt, 'steps': n_alt,
                                        'advanced_by': alt.engine.step - a0, 'final_loss': float(l_alt[-1]),
                                        'note': 'same workload and schedule, every GEMM on v_mfma_f32_32x32x2_f32 (fsmg_config.gemm = FSMG_GEMM_F32)'}
            if out.get('roofline') is not None:
                out['roofline']['alt_gemm_f32_mfma_value'] = n_alt / dt
                out['roofline']['alt_gemm_f32_mfma_note'] = 'episodes/s of the same loop with every GEMM on v_mfma_f32_32x32x2_f32 (true fp32 MFMA): what the bf16-split GEMMs buy'
            del alt, par_alt
        except Exception as e:                 # noqa: BLE001
            log('fp32-MFMA GEMM leg failed: %r' % (e,))
            extras_failed['alt_gemm_f32_mfma'] = repr(e)
        log('fp32-MFMA GEMM leg done')
    if rank == 0 and world == 1 and args.val_nll > 0 and not maml:
        try:
            out['val'] = val_nll_leg(base, eng, step, args, shape, local)
        except Exception as e:                 # noqa: BLE001 -- an extra leg must not cost the result line
            log('val-nll leg failed: %r' % (e,))
            extras_failed['val_nll'] = repr(e)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not maml:
        out['cpu_baseline'] = cpu_baseline(base, pool_host, shape)
        try:
            # GPU against CPU AFTER training (at Glorot init every model scores ln V1, right or wrong): a fresh handle
            # from the same initial parameters takes the torch-CPU variant's train steps -- same episodes, same order -- and the two are
            # compared on every train loss and on the query NLL of a held-out episode at the parameters both have reached
            from fsmg.binding import FsmgModel
            from oracle import lstm_oracle as O
            cb = out['cpu_baseline']
            fresh = FsmgModel(base, device=local, max_sequences=N_WAY * (K_SHOT + Q_QUERY))
            fresh.init_params(0)
            fresh.set_params(O.glorot_init(base, base['seed'], np.float32))
            g_first = float(fresh.eval_step(pool_host[0][1]))
            g_losses = [float(fresh.train_step(*pool_host[e])) for e in cb['agreement_train_episodes']]
            g_after = float(fresh.eval_step(pool_host[cb['agreement_eval_episode']][1]))
            fresh.close()
            c_losses, c_after = cb['agreement_train_losses'], cb['agreement_eval_nll']
            cb['nll_agreement'] = {
                'after_train_updates': len(c_losses), 'gpu': g_after, 'cpu': c_after, 'rel_diff': abs(g_after - c_after) / abs(c_after),
                'train_loss_max_rel_diff': max(abs(a_ - b_) / abs(b_) for a_, b_ in zip(g_losses, c_losses)),
                'gpu_train_losses': g_losses, 'cpu_train_losses': c_losses,
                'at_init': {'gpu': g_first, 'cpu': cb['first_eval_nll'], 'ln_V1': float(np.log(base['input_size'] + 1))},
                'note': 'same Glorot init, the same %d train episodes in the same order on both sides (fp32 torch-CPU restatement vs the HIP path), then the query NLL '
                        'of held-out episode %d; bar 1e-4 relative (BASELINE.json north_star)' % (len(c_losses), cb['agreement_eval_episode'])}
            # scalars (the driver's record keeps cpu_baseline's scalar keys)
            cb.update({'nll_after_train_gpu': g_after, 'nll_after_train_cpu': c_after, 'nll_after_train_rel_diff': cb['nll_agreement']['rel_diff'],
                       'nll_after_train_updates': len(c_losses), 'train_loss_max_rel_diff': cb['nll_agreement']['train_loss_max_rel_diff'],
                       'nll_agreement_ok': bool(cb['nll_agreement']['rel_diff'] <= 1e-4 and cb['nll_agreement']['train_loss_max_rel_diff'] <= 1e-4)})
            val = (out.get('other_configs') or {}).get('cfg-B-padded-zipf-pool', {}).get('val') or out.get('val')
            if val:
                cb.update({'val_nll_gpu': val['val_nll_gpu'], 'val_nll_cpu': val['val_nll_cpu'], 'val_nll_rel_diff': val['rel_diff'],
                           'val_nll_train_steps': val['train_steps']})
        except Exception as e:                 # noqa: BLE001 -- an extra leg must not cost the result line
            log('nll agreement leg failed: %r' % (e,))
            extras_failed['nll_agreement'] = repr(e)
    elif rank == 0:
        out['cpu_baseline'] = None
    if rank == 0:
        out['extras_failed'] = extras_failed
        sys.stderr.flush()
        print(json.dumps(out), file=result_stream)
        result_stream.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
