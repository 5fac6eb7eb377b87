"""Weight initialisation the reference's training pipeline applies before fitting
(src/ml/pipelines/default.py:36-57), restated for use outside Lightning (bench, tests)."""
import torch
import torch.nn as nn


def init_like_training_setup(model: nn.Module, seed: int = 0) -> nn.Module:
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    for m in model.modules():
        if isinstance(m, (nn.Conv1d, nn.Conv2d)):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.Linear):
            nn.init.normal_(m.weight, 0, 0.01)
            nn.init.constant_(m.bias, 0)
    torch.random.set_rng_state(gen_state)
    return model
