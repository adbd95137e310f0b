"""`python -um train.train --data=... --model=... --task=... [--checkpt_dir=...] [--init_dir=...]`

Same entrypoint, flags, YAML merge order (data <- task <- model, later wins) and log lines as
/root/reference/src/train/train.py:36-126; run it from this `src/` directory exactly like the
reference.  Differences, all deliberate:
  * flags are parsed inside main() (Q15) and YAML is read with safe_load (Q1);
  * sample directories are created with exist_ok (Q10);
  * when launched under torchrun (WORLD_SIZE > 1) training is episode-parallel: rank r trains on
    episodes r, r+R, ... of the single train stream, validation episodes are dealt round-robin to
    the ranks and averaged, and only rank 0 prints / checkpoints / writes samples;
  * validation uses the plugin's batched `eval_many` when it has one (same per-episode values).
"""
import argparse
import os
import pprint
import sys
from importlib import import_module

import yaml

from data.episode import ShardedEpisodeSampler, load_sampler_from_config

PP = pprint.PrettyPrinter(depth=6)
EVAL_CHUNK = 16          # episodes per eval_many call


def build_parser():
    parser = argparse.ArgumentParser(description='Train a model.')
    for flag in ('data', 'model', 'task', 'checkpt_dir', 'init_dir'):
        parser.add_argument('--' + flag, dest=flag, default='')
    return parser


def load_config(args):
    """data.yaml <- task.yaml <- model.yaml, then the driver-injected keys (train.py:49-53)."""
    config = {}
    for path in (args.data, args.task, args.model):
        with open(path, 'r') as f:
            config.update(yaml.safe_load(f) or {})
    config['dataset_path'] = os.path.abspath(config['dataset_path'])
    config['checkpt_dir'] = args.checkpt_dir
    return config


def load_model_from_config(config):
    Model = getattr(import_module(config['model_module_name']), config['model_class_name'])
    return Model(config)


def write_seq(seq, dir, name):
    if isinstance(seq, str):
        with open(os.path.join(dir, name + '.txt'), 'w') as f:
            f.write(seq)
    else:
        seq.write(os.path.join(dir, name + '.mid'))


def evaluate(model, episode_sampler, n_episodes):
    """Mean of model.eval over n_episodes fresh episodes (train.py:27-33)."""
    total, done = 0.0, 0
    many = getattr(model, 'eval_many', None)
    while done < n_episodes:
        n = min(EVAL_CHUNK, n_episodes - done) if many else 1
        episodes = [episode_sampler.get_episode() for _ in range(n)]
        total += sum(many(episodes)) if many else model.eval(episodes[0])
        done += n
    return total / n_episodes


def sharded_validate(model, sampler, n, rank=0, world=1, parallel=None):
    """Mean NLL over exactly n fresh episodes of `sampler`'s stream (train.py:27-33 with the episodes dealt over the ranks).
    The n episodes are dealt round-robin: rank r takes ceil((n - r) / world) of them, every rank contributes the SUM of its NLLs,
    so the mean is over exactly n episodes whatever n % world is; a rank that got one episode fewer than the busiest rank skips
    one position, so every rank's copy of the stream ends at the same place."""
    mine = (n - rank + world - 1) // world if world > 1 else n
    nll = evaluate(model, sampler, mine) if mine > 0 else 0.0
    if world > 1 and mine < (n + world - 1) // world:
        sampler.next_indices()          # keep every rank's copy of the stream at the same position
    if world > 1 and parallel:
        return parallel.mean_scalar(nll * mine) * world / n
    return nll


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world > 1:
        from fsmg.dist import init_from_env
        rank, world = init_from_env()
    chief = rank == 0

    def say(*a):
        if chief:
            print(*a)
            sys.stdout.flush()

    say('Args:')
    say(PP.pformat(vars(args)))
    config = load_config(args)
    say('Config:')
    say(PP.pformat(config))

    # Building a sampler validates songs, tokenises them and PERSISTS sidecars / split files next to the dataset.  Under
    # torchrun the chief does that alone; the other ranks wait and then load what it wrote (identical datasets on every
    # rank, no two writers on one file).
    if world > 1 and not chief:
        import torch.distributed as dist
        dist.barrier()
    episode_sampler = {}
    for split in config['splits']:
        config['split'] = split
        sampler = load_sampler_from_config(config)
        episode_sampler[split] = ShardedEpisodeSampler(sampler, rank, world) if world > 1 else sampler
    if world > 1 and chief:
        import torch.distributed as dist
        dist.barrier()

    config['input_size'] = episode_sampler['train'].get_num_unique_words()
    if not config['input_size'] > 0:
        raise RuntimeError('error reading data: %d unique tokens processed' % config['input_size'])
    say('Num unique words: %d' % config['input_size'])

    n_train, print_every_n, val_every_n = config['n_train'], config['print_every_n'], config['val_every_n']
    n_val, n_test, n_samples, max_len = config['n_val'], config['n_test'], config['n_samples'], config['max_len']

    model = load_model_from_config(config)
    model.recover_or_init(args.init_dir)

    def validate(split, n):
        return sharded_validate(model, episode_sampler[split], n, rank, world, getattr(model, '_parallel', None))

    say('Iter: %d, val-nll: %.3e' % (0, validate('val', n_val)))

    # Fast path (same numbers, same log lines): the train split's packed token table lives in HBM, an episode travels as
    # N*(K+Q) row indices, and the per-step loss stays in the device's ring until the next log line needs the window's
    # mean -- the reference's loop (train.py:83-98) pays a blocking device->host read per step for `avg_loss +=`.
    train_sampler = episode_sampler['train']
    fast = (hasattr(model, 'attach_table') and hasattr(model, 'train_indexed') and hasattr(model, 'recent_losses')
            and hasattr(train_sampler, 'next_indices') and os.environ.get('FSMG_TRAIN_SYNC', '0') == '0')
    if fast:
        model.attach_table('train', train_sampler.token_table())
    RING = 1024
    avg_loss, pending, counted, skipped_total = 0., 0, 0, 0
    step_mark = model.global_step() if (fast and hasattr(model, 'global_step')) else None

    def drain():                                   # fold the losses still on the device into avg_loss
        nonlocal avg_loss, pending, counted, skipped_total, step_mark
        if not pending:
            return
        # A step whose persistent kernel timed out (own or a peer rank's) or whose batch held a bad token is SKIPPED on the
        # device: no ring slot, no global_step.  Only the steps that really ran have a loss to read; the others are counted
        # and reported instead of being filled in with stale ring entries.
        done = pending
        if step_mark is not None:
            now = model.global_step()
            done = max(0, min(pending, now - step_mark))
            step_mark = now
        if done:
            losses = model.recent_losses(done)
            avg_loss += float(sum(losses))
            counted += done
            if hasattr(model, 'log_deferred_losses'):
                model.log_deferred_losses(losses)
        if done < pending:
            skipped_total += pending - done
            say('warning: %d train episode(s) were skipped on the device (persistent-kernel time-out or token-range error); '
                '%d so far' % (pending - done, skipped_total))
        pending = 0

    for i in range(1, n_train + 1):
        if fast:
            model.train_indexed('train', *train_sampler.next_indices())
            pending += 1
            if pending == RING:
                drain()
        else:
            avg_loss += model.train(train_sampler.get_episode())
            counted += 1

        if i % val_every_n == 0:            # val_every_n may be a float (Q11)
            drain()
            say('Iter: %d, val-nll: %.3e' % (i, validate('val', n_val)))
            if args.checkpt_dir != '' and chief:
                model.save(args.checkpt_dir)

        if i % print_every_n == 0:
            drain()
            # mean over the steps of this window that ran (= print_every_n unless the device skipped some)
            say('Iter: %d, loss: %.3e' % (i, avg_loss / max(counted, 1)))
            avg_loss, counted = 0., 0

    say('Train Avg NLL: %.3e' % validate('train', n_test))
    say('Validation Avg NLL: %.3e' % validate('val', n_test))
    say('Test Avg NLL: %.3e' % validate('test', n_test))

    if not chief:
        return
    samples_dir = os.path.join(args.checkpt_dir, 'samples')
    os.makedirs(samples_dir, exist_ok=True)
    for i in range(n_samples):
        curr_sample_dir = os.path.join(samples_dir, 'sample_%d' % i)
        os.makedirs(curr_sample_dir, exist_ok=True)
        episode = episode_sampler['test'].get_episode()
        support_set = episode.support[0]
        sample = model.sample(support_set, max_len)
        for j in range(support_set.shape[0]):
            write_seq(episode_sampler['test'].detokenize(support_set[j]), curr_sample_dir, 'support_%d' % j)
        write_seq(episode_sampler['test'].detokenize(sample), curr_sample_dir, 'model_sample')


if __name__ == '__main__':
    main()
