"""CPU: loadSRLModel's reference surface (/root/reference/state_representation/models.py:38-107,196-217): exp_config.json keys
(state-dim, losses, n_actions, model-type, multi-view -> 6 channels, split-dimensions, inverse-model-type), the same assertion
messages on non-conform configs, the pickled-PCA baseline as ONE GEMM equal to sklearn's transform."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from state_representation.models import SRLNeuralNetwork, SRLPCA, CustomCNN, loadSRLModel


def write_cfg(folder, cfg):
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "exp_config.json"), "w") as f:
        json.dump(cfg, f)


def test_custom_cnn_checkpoint_with_exp_config(tmp_path):
    folder = str(tmp_path / "logs" / "kuka" / "autoencoder")
    write_cfg(folder, {"state-dim": 5, "losses": ["autoencoder", "inverse"], "n_actions": 6, "model-type": "custom_cnn", "multi-view": True,
                       "split-dimensions": {"autoencoder": 0, "inverse": 0}, "inverse-model-type": "mlp"})
    net = CustomCNN(5, n_channels=6, img_shape=(64, 64))
    torch.save(net.state_dict(), os.path.join(folder, "srl_model.pth"))
    m = loadSRLModel(os.path.join(folder, "srl_model.pth"), cuda=False, img_shape=(64, 64))
    assert isinstance(m, SRLNeuralNetwork) and m.state_dim == 5 and m.n_channels == 6          # multi-view -> 6 channels
    assert m.losses == ["autoencoder", "inverse"] and m.n_actions == 6 and m.split_dimensions is None and m.inverse_model_type == "mlp"
    imgs = np.random.RandomState(0).randint(0, 256, size=(3, 64, 64, 6)).astype(np.uint8)
    ref = SRLNeuralNetwork(5, cuda=False, n_channels=6, img_shape=(64, 64), state_dict=net.state_dict(), backend="torch")
    assert np.array_equal(m.getStates(imgs).numpy(), ref.getStates(imgs).numpy())
    assert m.getState(imgs[0]).shape == (5,)


def test_non_conform_configs_fail_like_the_reference(tmp_path):
    folder = str(tmp_path / "a")
    write_cfg(folder, {"losses": ["autoencoder"], "n_actions": 6, "model-type": "custom_cnn"})
    with pytest.raises(AssertionError, match="up to date model"):
        loadSRLModel(os.path.join(folder, "srl_model.pth"))
    write_cfg(folder, {"state-dim": 3, "n_actions": 6, "model-type": "custom_cnn"})           # no losses, not pca
    with pytest.raises(AssertionError, match="up to date model"):
        loadSRLModel(os.path.join(folder, "srl_model.pth"))
    write_cfg(folder, {"state-dim": 3, "losses": ["autoencoder"], "model-type": "custom_cnn"})  # no n_actions, not supervised
    with pytest.raises(AssertionError, match="up to date model"):
        loadSRLModel(os.path.join(folder, "srl_model.pth"))
    write_cfg(folder, {"state-dim": 3, "losses": ["autoencoder"], "n_actions": 6, "model-type": "resnet"})
    with pytest.raises(NotImplementedError):
        loadSRLModel(os.path.join(folder, "srl_model.pth"))
    with pytest.raises(AssertionError, match="state_dim must be > 0"):
        loadSRLModel(None, state_dim=0)


@pytest.mark.parametrize("whiten", [False, True])
def test_pca_baseline_is_one_gemm_equal_to_sklearn(tmp_path, whiten):
    from sklearn.decomposition import PCA
    rs = np.random.RandomState(1)
    imgs = rs.randint(0, 256, size=(40, 16, 16, 3)).astype(np.uint8)
    pca = PCA(n_components=4, whiten=whiten).fit(imgs.reshape(40, -1))
    folder = str(tmp_path / "logs" / "kuka" / "baselines" / "pca")
    write_cfg(folder, {"state-dim": 4})
    with open(os.path.join(folder, "srl_model.pkl"), "wb") as f:
        pickle.dump(pca, f)
    m = loadSRLModel(os.path.join(folder, "srl_model.pkl"), cuda=False)
    assert isinstance(m, SRLPCA) and m.state_dim == 4
    ref = pca.transform(imgs.reshape(40, -1))
    out = m.getStates(imgs).numpy()
    assert out.dtype == np.float64 and np.abs(out - ref).max() <= 1e-9 * max(1.0, np.abs(ref).max())
    assert np.abs(m.getState(imgs[7]) - ref[7]).max() <= 1e-9 * max(1.0, np.abs(ref).max())
