"""A trivial plugin used by the CPU tests of train.train (stands in for any `models/` plugin)."""
from models.base_model import BaseModel


class FakeModel(BaseModel):
    calls = []

    def __init__(self, config):
        super(FakeModel, self).__init__(config)
        self.steps = 0
        FakeModel.calls = [('init', config['input_size'])]

    def train(self, episode):
        self.steps += 1
        FakeModel.calls.append(('train', episode.support.shape, episode.query.shape))
        return 10.0 / self.steps

    def eval(self, episode):
        FakeModel.calls.append(('eval', episode.query.shape))
        return 2.5

    def sample(self, support_set, num):
        return [1, 2, 3][:num] + [0] * max(0, num - 3)

    def save(self, checkpt_path):
        FakeModel.calls.append(('save', checkpt_path))

    def recover_or_init(self, init_path):
        FakeModel.calls.append(('recover_or_init', init_path))
