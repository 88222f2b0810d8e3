"""CPU: the Hydra-shaped config composition used by main_1d.py / main_2d.py and
the synthetic Markov-pair generator."""
import os

import torch

from tests.conftest import DROPIN


def test_compose_defaults_groups_and_overrides():
    from rpde.config import compose, instantiate
    cfg = compose(os.path.join(DROPIN, "conf"), "config", ["model=fno_2d/fno_2d", "training.epochs=3", "model.width=16"])
    assert cfg.model["_target_"] == "models.fno.FNO2d" and cfg.model.width == 16 and cfg.model.in_channels == 1
    assert cfg.training.epochs == 3 and cfg.training.batch_size == 16 and cfg.dataset.pde == "ns"
    model = instantiate(cfg.model)
    assert type(model).__name__ == "FNO2d" and model.width == 16
    cfg = compose(os.path.join(DROPIN, "conf"))
    m = instantiate(cfg.model)
    assert type(m).__name__ == "FFNO2D" and m.n_modes == 20 and "in_proj.weight_g" in m.state_dict()


def test_synthetic_pairs_are_standardised_and_mixed_resolution():
    from utils.synthetic import markov_pairs
    pairs = markov_pairs({32: 3, 64: 2}, 2, seed=0)
    assert [p[0].shape[-1] for p in pairs] == [32, 32, 32, 64, 64]
    x = torch.stack([p[0] for p in pairs[:3]])
    assert abs(float(x.mean())) < 1e-5 and abs(float(x.std()) - 1) < 1e-3
    assert pairs[0][0].shape == pairs[0][1].shape == (1, 32, 32)
    p1 = markov_pairs({48: 4}, 1, seed=1)
    assert p1[0][0].shape == (1, 48)
