"""SRL encoder restatement (state_representation/models.py): CPU shape/semantics tests, and a GPU test of the
device-resident raw_pixels -> state pipeline against the same network evaluated on the CPU in float32."""
import numpy as np
import pytest
import torch

from state_representation.models import CustomCNN, SRLNeuralNetwork, loadSRLModel, preprocess


def test_custom_cnn_shapes_follow_the_input_size():
    assert CustomCNN(3, 3, (224, 224)).flat_dim == 6 * 6 * 64
    assert CustomCNN(3, 3, (64, 64)).flat_dim == 1 * 1 * 64
    net = loadSRLModel(state_dim=5, img_shape=(64, 64))
    s = net.getState(np.zeros((64, 64, 3), np.uint8))
    assert s.shape == (5,) and s.dtype == np.float32
    assert net.getStates(np.zeros((7, 64, 64, 3), np.uint8)).shape == (7, 5)


def test_preprocess_layout_and_normalisation():
    img = np.zeros((2, 4, 6, 3), np.uint8)
    img[0, 1, 2] = (255, 0, 128)
    x = preprocess(torch.from_numpy(img))
    assert x.shape == (2, 3, 6, 4)                                      # N, C, W, H (models.py:188 transpose)
    assert abs(float(x[0, 0, 2, 1]) - (1.0 - 0.485) / 0.229) < 1e-6
    assert abs(float(x[0, 1, 2, 1]) - (0.0 - 0.456) / 0.224) < 1e-6
    assert abs(float(x[0, 2, 2, 1]) - (128 / 255.0 - 0.406) / 0.225) < 1e-6


@pytest.mark.gpu
def test_device_resident_pixels_to_state_pipeline():
    from srlhip.pixel_env import PixelStateVecEnv
    torch.manual_seed(0)
    enc = SRLNeuralNetwork(8, cuda=True, img_shape=(64, 64))
    env = PixelStateVecEnv("KukaButtonGymEnv-v0", 256, enc, seed=1)
    states = env.reset()
    assert states.shape == (256, 8) and states.is_cuda
    cpu = SRLNeuralNetwork(8, cuda=False, img_shape=(64, 64), state_dict=enc.model.state_dict())
    ref = cpu.getStates(env.images.cpu().numpy())
    assert np.abs(states.cpu().numpy() - ref.numpy()).max() < 2e-3        # float32 MIOpen vs CPU convolutions
    for _ in range(20):
        states, rew, done = env.step()
    assert torch.isfinite(states).all() and states.std() > 0
    ref = cpu.getStates(env.images.cpu().numpy())
    assert np.abs(states.cpu().numpy() - ref.numpy()).max() < 2e-3
    env.close()


def test_batchnorm_folding_is_equivalent():
    torch.manual_seed(1)
    net = SRLNeuralNetwork(4, img_shape=(64, 64))
    for m in net.model.modules():                       # non-trivial BN statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    net = SRLNeuralNetwork(4, img_shape=(64, 64), state_dict=net.model.state_dict())
    img = np.random.RandomState(0).randint(0, 256, size=(5, 64, 64, 3)).astype(np.uint8)
    with torch.no_grad():
        ref = net.model.getStates(preprocess(torch.from_numpy(img)))
    assert np.abs(net.getStates(img).numpy() - ref.numpy()).max() < 1e-4
