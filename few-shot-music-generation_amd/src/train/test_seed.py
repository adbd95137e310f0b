"""`python -um train.test_seed --data=... --task=... [--expect=LOSS]` -- seed / regression check.

Counterpart of /root/reference/src/train/test_seed.py:17-65: build the train sampler and the LSTM baseline from
`config/lstm_baseline_test_seed.yaml`, run N_UPDATES train steps and report the loss of the last one.  The
reference compares it with constants that are only valid for its own TensorFlow build, datasets and Python 2
hash order; here the expected value is an argument (tests/ derive it from the CPU oracle on
the committed fixture), and without --expect the loss is only printed.
"""
import argparse
import os
import sys

import yaml

from data.episode import load_sampler_from_config
from train.train import load_model_from_config

N_UPDATES = 10
EPSILON = 0.001


def main(argv=None):
    ap = argparse.ArgumentParser(description='Seed test for the LSTM baseline.')
    ap.add_argument('--data', default='')
    ap.add_argument('--task', default='')
    ap.add_argument('--model', default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                    'config', 'lstm_baseline_test_seed.yaml'))
    ap.add_argument('--expect', type=float, default=None)
    args = ap.parse_args(argv)
    config = {}
    for path in (args.data, args.task, args.model):
        with open(path, 'r') as f:
            config.update(yaml.safe_load(f) or {})
    config['dataset_path'] = os.path.abspath(config['dataset_path'])
    config['split'] = 'train'
    sampler = load_sampler_from_config(config)
    config['input_size'] = sampler.get_num_unique_words()
    model = load_model_from_config(config)          # no checkpt_dir key -> no scalar log, like the reference
    model.recover_or_init('')
    loss = None
    for _ in range(N_UPDATES):
        loss = model.train(sampler.get_episode())
    print('loss after %d updates: %.7f' % (N_UPDATES, loss))
    if args.expect is not None and abs(loss - args.expect) > EPSILON:
        print('FAILED: expected %.7f +- %g' % (args.expect, EPSILON))
        return 1
    return 0


if __name__ == '__main__':
    sys.exit(main())
