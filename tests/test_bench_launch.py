"""bench.py as the driver starts it: `python bench.py --gpus N ...` with NO torchrun environment must start its own N ranks
(VERDICT r02 next #1).  The GPU test is a dry run of that launcher on ONE GPU (FSMG_BENCH_SAME_GPU=1: all ranks on GPU 0, gloo
instead of RCCL -- its numbers mean nothing, its structure is what the 8-GPU run will print)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(extra)
    return env


def test_bench_refuses_more_ranks_than_gpus_with_a_clear_message():
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip('needs a box with fewer than 2 GPUs')
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=300, env=_clean_env())
    assert proc.returncode != 0
    assert 'GPU(s) visible' in proc.stderr and 'FSMG_BENCH_SAME_GPU' in proc.stderr, proc.stderr[-2000:]


def test_default_exchange_plan_is_graph_end_alone_and_the_others_are_opt_in():
    """the first multi-GPU minutes go to the conservative schedule the shipped configuration runs."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.exchange_plans(1, {}) == [('single_gpu', {})]
    assert [n for n, _ in bench.exchange_plans(8, {})] == ['graph_end']
    assert [n for n, _ in bench.exchange_plans(8, {'FSMG_BENCH_SCHEDULES': 'all'})] == ['graph_end', 'split_bucket0', 'split_after_chain', 'one_collective']
    assert [n for n, _ in bench.exchange_plans(8, {'FSMG_BENCH_SCHEDULES': 'one_collective,graph_end'})] == ['graph_end', 'one_collective']
    assert [n for n, _ in bench.exchange_plans(2, {'FSMG_BENCH_SCHEDULES': 'all', 'FSMG_BENCH_LIBRARY_RCCL': '1'})][-1] == 'library_rccl'
    assert 'library_rccl' not in [n for n, _ in bench.exchange_plans(2, {'FSMG_BENCH_SCHEDULES': 'all', 'FSMG_BENCH_LIBRARY_RCCL': '1'}, same_gpu=True)]
    # ... and train.train's episode-parallel driver takes the same default (src/fsmg/dist.py: bucketed exchange when the backward graph has ended)
    from fsmg.dist import EpisodeParallel, RETRY_CODES
    assert RETRY_CODES == (-9, -10)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_and_reports_every_schedule():
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                           '--no-cpu-baseline', '--no-breakdown'], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          universal_newlines=True, timeout=900, env=_clean_env(FSMG_BENCH_SAME_GPU='1', HSA_ENABLE_IPC_MODE_LEGACY='0', FSMG_BENCH_REPEATS='2',
                                                                               FSMG_BENCH_SCHEDULES='all'))
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-4000:])
    printed = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(printed) == 1, printed[:5]                   # ONE JSON line on stdout: progress and the plugins' prints go to stderr
    out = json.loads(printed[0])
    assert out['n_gpus'] == 2 and out['steps'] == 3 and out['warmup'] == 1 and out['scaling'] == 'weak'
    assert out['world']['size'] == 2 and out['world']['launcher'] == 'bench.py self-launch'
    assert out['world']['backend'] == 'gloo' and out['world']['same_gpu_dry_run'] is True
    assert [d['rank'] for d in out['world']['devices']] == [0, 1]
    # graph_end first; the opt-in schedules only behind a guarded graph_end (two processes time-slicing one GPU may time out: then they are skipped, and said so)
    assert list(out['schedules'])[0] == 'graph_end' and not out['schedules_failed']
    assert set(out['schedules']) | set(out['schedules_skipped']) == {'graph_end', 'split_bucket0', 'split_after_chain', 'one_collective'}
    assert bool(out['schedules_skipped']) == (not out['schedules']['graph_end']['guard_ok'])
    assert out['schedule_used'] in out['schedules']
    # the headline is the DEFAULT schedule (what the shipped configuration runs), not the fastest of the four; median of the repeats
    if out['schedules']['graph_end']['guard_ok'] or not any(s['guard_ok'] for s in out['schedules'].values()):
        assert out['schedule_used'] == 'graph_end'
    used = out['schedules'][out['schedule_used']]
    assert abs(out['ms_per_step'] - used['ms_per_step']) < 1e-9 and used['repeats'] == 2 and len(used['ms_per_step_regions']) == 2
    assert used['ms_per_step_min'] <= used['ms_per_step'] <= used['ms_per_step_max'] and out['spread']['repeats'] == 2
    assert abs(out['value'] - 2 * 3 / (out['ms_per_step'] * 3e-3)) < 1e-6 * out['value']
    assert len(out['guard_per_rank']) == 2 and {g['rank'] for g in out['guard_per_rank']} == {0, 1}
    for g in out['guard_per_rank']:                         # two processes time-slicing one GPU may time out and recover: report, not hide
        assert g['expected'] == 3 * 2 and 'timeouts' in g and 'fallback_steps_left' in g and 'rearmed' in g
    assert set(out['comm']['exposed_ms']) == set(out['schedules']) and out['comm']['ms_per_step_without_exchange'] > 0
    assert out['comm']['bytes'] > 0 and out['comm']['allreduce_ms_standalone'] > 0


@pytest.mark.gpu
def test_single_gpu_bench_prints_one_json_line_with_the_contract_keys():
    """`python bench.py --steps K --warmup W` as the driver runs it at N = 1 (short, without the CPU leg): stdout is ONE JSON line."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '2', '--no-cpu-baseline'],
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True, timeout=900, env=_clean_env(FSMG_BENCH_REPEATS='2'))
    assert proc.returncode == 0, (proc.stdout[-2000:], proc.stderr[-4000:])
    printed = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(printed) == 1, printed[:5]
    out = json.loads(printed[0])
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
                'config', 'roofline', 'cpu_baseline'):
        assert key in out, key
    assert out['n_gpus'] == 1 and out['steps'] == 4 and out['warmup'] == 2 and out['unit'] == 'episodes/s' and out['dtype'] == 'f32 (bf16x3-split products, fp32 accumulate)'
    assert abs(out['value'] - 1e3 / out['ms_per_step']) < 1e-6 * out['value']
    r = out['roofline']
    assert r['bound'] == 'mfma' and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and r['achieved'] > 0
    assert out['guard']['ok'], out['guard']
