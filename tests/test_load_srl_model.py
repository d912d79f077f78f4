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
    write_cfg(folder, {"state-dim": 5, "losses": ["forward", "inverse"], "n_actions": 6, "model-type": "custom_cnn", "multi-view": True,
                       "split-dimensions": {"forward": 0, "inverse": 0}, "inverse-model-type": "mlp"})
    net = CustomCNN(5, n_channels=6, img_shape=(64, 64))
    torch.save(net.state_dict(), os.path.join(folder, "srl_model.pth"))
    m = loadSRLModel(os.path.join(folder, "srl_model.pth"), cuda=False, img_shape=(64, 64))
    assert isinstance(m, SRLNeuralNetwork) and m.state_dim == 5 and m.n_channels == 6          # multi-view -> 6 channels
    assert m.losses == ["forward", "inverse"] and m.n_actions == 6 and m.split_dimensions is None and m.inverse_model_type == "mlp"
    imgs = np.random.RandomState(0).randint(0, 256, size=(3, 64, 64, 6)).astype(np.uint8)
    ref = SRLNeuralNetwork(5, cuda=False, n_channels=6, img_shape=(64, 64), state_dict=net.state_dict(), backend="torch")
    assert np.array_equal(m.getStates(imgs).numpy(), ref.getStates(imgs).numpy())
    assert m.getState(imgs[0]).shape == (5,)


def test_srl_zoo_style_checkpoint_keys(tmp_path):
    """What the reference's SRLNeuralNetwork.load feeds load_state_dict (models.py:150-169): SRLModules' state_dict — the encoder under
    `model.` next to the loss heads.  The heads are dropped, the encoder loads strictly; autoencoder-family losses and split models are
    different networks in srl_zoo and are refused."""
    folder = str(tmp_path / "logs" / "kuka" / "inverse_forward")
    cfg = {"state-dim": 4, "losses": ["inverse", "forward", "reward"], "n_actions": 6, "model-type": "custom_cnn"}
    write_cfg(folder, cfg)
    net = CustomCNN(4, n_channels=3, img_shape=(64, 64))
    sd = {"model." + k: v for k, v in net.state_dict().items()}
    sd.update({"inverse_net.weight": torch.zeros(6, 8), "inverse_net.bias": torch.zeros(6), "forward_net.weight": torch.zeros(4, 10),
               "forward_net.bias": torch.zeros(4), "reward_net.0.weight": torch.zeros(16, 8), "reward_net.0.bias": torch.zeros(16)})
    path = os.path.join(folder, "srl_model.pth")
    torch.save(sd, path)
    m = loadSRLModel(path, cuda=False, img_shape=(64, 64))
    imgs = np.random.RandomState(0).randint(0, 256, size=(2, 64, 64, 3)).astype(np.uint8)
    ref = SRLNeuralNetwork(4, cuda=False, img_shape=(64, 64), state_dict=net.state_dict(), backend="torch")
    assert np.array_equal(m.getStates(imgs).numpy(), ref.getStates(imgs).numpy())
    sd["optimizer_step"] = torch.zeros(1)                                     # something that is neither encoder nor head
    torch.save(sd, path)
    with pytest.raises(KeyError):
        loadSRLModel(path, cuda=False, img_shape=(64, 64))
    del sd["optimizer_step"], sd["model.fc.weight"]                           # an incomplete encoder fails the strict load
    torch.save(sd, path)
    with pytest.raises(RuntimeError):
        loadSRLModel(path, cuda=False, img_shape=(64, 64))
    torch.save(net.state_dict(), path)
    for bad in (dict(cfg, losses=["autoencoder", "inverse"]), dict(cfg, losses=["vae"]), dict(cfg, losses=["dae"]),
                dict(cfg, **{"split-dimensions": {"inverse": 2, "forward": 2}})):
        write_cfg(folder, bad)
        with pytest.raises(NotImplementedError):
            loadSRLModel(path, cuda=False, img_shape=(64, 64))


def test_non_dict_split_dimensions_means_no_split(tmp_path):
    """state_representation/models.py:159 takes the SRLModulesSplit path only for an OrderedDict: any other value of `split-dimensions`
    (srl_zoo is recalled to write -1 when not splitting; also null) loads the plain SRLModules layout."""
    folder = str(tmp_path / "logs" / "kuka" / "inverse")
    net = CustomCNN(3, n_channels=3, img_shape=(64, 64))
    path = os.path.join(folder, "srl_model.pth")
    for value in (-1, None, {"inverse": 0}):
        write_cfg(folder, {"state-dim": 3, "losses": ["inverse"], "n_actions": 6, "model-type": "custom_cnn", "split-dimensions": value})
        torch.save({"model." + k: v for k, v in net.state_dict().items()}, path)
        m = loadSRLModel(path, cuda=False, img_shape=(64, 64))
        assert isinstance(m, SRLNeuralNetwork) and m.split_dimensions is None


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
