"""`python -m yolo_v1_amd.train` -- the user-facing entry point (reference train.py:144-209) -- runs the path bench.py
measures (fused HIP SGD + hipGraph replay) and that path is the eager one bit for bit."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(tmp, *extra):
    return ["--backbone", "resnet", "--S", "7", "--batch-size", "4", "--epochs", "2", "--iters-per-epoch", "4",
            "--save-dir", str(tmp)] + list(extra)


def test_train_main_default_is_graphed_and_equals_eager_bit_for_bit(tmp_path):
    from yolo_v1_amd import train
    graphed = train.main(_args(tmp_path / "g"), return_losses=True)
    eager = train.main(_args(tmp_path / "e", "--eager"), return_losses=True)
    assert len(graphed) == 8 and all(np.isfinite(graphed))
    assert graphed == eager, (graphed, eager)                    # same kernels, same order: identical fp32 losses
    # torch.optim.SGD (train.py:84) instead of the fused kernel: same update up to fp32 rounding of momentum*buf + g
    tsgd = train.main(_args(tmp_path / "t", "--torch-sgd"), return_losses=True)
    np.testing.assert_allclose(graphed, tsgd, rtol=2e-3)
    for d in ("g", "e", "t"):
        assert os.path.exists(tmp_path / d / "resnet_sgd_S7_yolo.pth")            # train.py:209
    sd = torch.load(tmp_path / "g" / "resnet_sgd_S7_yolo.pth", weights_only=True)
    sde = torch.load(tmp_path / "e" / "resnet_sgd_S7_yolo.pth", weights_only=True)
    assert all(k.startswith("module.") for k in sd)
    for k in sd:
        assert torch.equal(sd[k], sde[k]), k                     # weights, running statistics, num_batches_tracked
    assert int(sd["module.bn1.num_batches_tracked"]) == 8        # the capture warm-up left no trace


def test_train_main_loader_feeds_static_graph_inputs_and_validates(tmp_path):
    from yolo_v1_amd import train
    common = ["--backbone", "densenet", "--S", "7", "--batch-size", "4", "--epochs", "1", "--iters-per-epoch", "3",
              "--loader", "--workers", "0", "--val-synthetic", "8", "--little-val-num", "8"]
    graphed = train.main(common + ["--save-dir", str(tmp_path / "g")], return_losses=True)
    eager = train.main(common + ["--save-dir", str(tmp_path / "e"), "--eager"], return_losses=True)
    assert len(graphed) == 3 and graphed == eager
    assert len(set(graphed)) == 3                                # three different batches went through the static buffers
    assert os.path.exists(tmp_path / "g" / "densenet_sgd_S7_yolo.pth")
    log = open(tmp_path / "g" / "train.log").read() if os.path.exists(tmp_path / "g" / "train.log") else ""
    assert "start evaluate" in log or log == ""
