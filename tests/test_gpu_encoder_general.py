"""Layered HIP encoder (csrc/encoder_general.hip, through the C-ABI) for the frame shapes the fused 64x64x3 kernel does
not cover — the reference's 224x224x3 observation (kuka_button_gym_env.py:21-22), 6-channel multi_view frames
(:401-417), non-square frames — against the plain PyTorch float32 forward of the same network on the CPU
(state_representation/models.py:178-193).  Same tolerance as the fused kernel: 2e-5 of the largest state component."""
import os

import numpy as np
import pytest
import torch

from srlhip import _lib
from state_representation.models import SRLNeuralNetwork

pytestmark = pytest.mark.gpu
TOL = 2e-5


def make_nets(state_dim, seed, img_shape, n_channels):
    torch.manual_seed(seed)
    net = SRLNeuralNetwork(state_dim, img_shape=img_shape, n_channels=n_channels, backend="torch")
    for m in net.model.modules():                       # non-trivial BatchNorm statistics: exercises the folding
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    sd = net.model.state_dict()
    return (SRLNeuralNetwork(state_dim, cuda=True, img_shape=img_shape, n_channels=n_channels, state_dict=sd, backend="hip"),
            SRLNeuralNetwork(state_dim, cuda=False, img_shape=img_shape, n_channels=n_channels, state_dict=sd, backend="torch"))


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


def frames(n, shape, ch, seed):
    rs = np.random.RandomState(seed)
    imgs = rs.randint(0, 256, size=(n,) + shape + (ch,)).astype(np.uint8)
    imgs[0, :, :shape[1] // 3] = 0                       # flat borders: zero padding in normalised space
    if n > 2:
        imgs[1] = 255
        imgs[2, shape[0] // 6:, : shape[1] // 2] = rs.randint(0, 256, size=ch).astype(np.uint8)
    return imgs


def test_supported_shapes():
    assert _lib.encoder_supported(224, 224, 3) and _lib.encoder_supported(64, 64, 6) and _lib.encoder_supported(64, 64, 3)
    assert _lib.encoder_supported(96, 128, 3) and _lib.encoder_supported(224, 224, 6)
    assert not _lib.encoder_supported(64, 64, 4) and not _lib.encoder_supported(16, 16, 3) and not _lib.encoder_supported(2048, 64, 3)


@pytest.mark.parametrize("shape,ch,state_dim,n", [((224, 224), 3, 3, 5), ((64, 64), 6, 5, 37), ((224, 224), 6, 2, 3), ((96, 128), 3, 4, 9),
                                                 ((128, 96), 6, 200, 4), ((75, 61), 3, 3, 6), ((48, 56), 3, 2, 70)])
def test_layered_encoder_matches_torch_fp32(shape, ch, state_dim, n):
    gpu, cpu = make_nets(state_dim, 11 + state_dim, shape, ch)
    assert gpu.backend == "hip"
    imgs = frames(n, shape, ch, n)
    out = gpu.getStates(imgs).cpu().numpy()
    ref = cpu.getStates(imgs).numpy()
    assert out.shape == (n, state_dim) and np.isfinite(out).all()
    assert rel_err(out, ref) < TOL, rel_err(out, ref)
    assert not gpu.hip.overflow()
    # a second, larger batch through the same handle (the scratch planes grow), then the first one again
    more = frames(n + 7, shape, ch, n + 1)
    assert rel_err(gpu.getStates(more).cpu().numpy(), cpu.getStates(more).numpy()) < TOL
    assert rel_err(gpu.getStates(imgs).cpu().numpy(), ref) < TOL


def test_layered_path_agrees_with_the_fused_kernel_at_64x64x3():
    gpu, cpu = make_nets(6, 5, (64, 64), 3)
    imgs = frames(300, (64, 64), 3, 3)
    fused = gpu.getStates(imgs).cpu().numpy()
    os.environ["SRLHIP_ENCODER_GENERAL"] = "1"
    try:
        layered_net = SRLNeuralNetwork(6, cuda=True, img_shape=(64, 64), state_dict=cpu.model.state_dict(), backend="hip")
    finally:
        del os.environ["SRLHIP_ENCODER_GENERAL"]
    layered = layered_net.getStates(imgs).cpu().numpy()
    ref = cpu.getStates(imgs).numpy()
    assert rel_err(layered, ref) < TOL and rel_err(fused, ref) < TOL
    assert rel_err(layered, fused) < TOL


def test_multi_view_frames_from_the_rasteriser():
    """6-channel frames as KukaButtonGymEnv(multi_view=True) renders them (kuka_button_gym_env.py:401-417)"""
    gpu, cpu = make_nets(4, 2, (64, 64), 6)
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0, cfg.obs_mode, cfg.img_h, cfg.img_w, cfg.multi_view, cfg.io_device = 64, 3, _lib.OBS_RAW_PIXELS, 64, 64, 1, 1
    h = _lib.Handle(cfg)
    img = torch.zeros((64, 64, 64, 6), dtype=torch.uint8, device="cuda")
    h.reset(obs_out=img.data_ptr()); h.sync()
    assert img[..., 3:].float().std() > 1.0               # the second camera rendered something
    out = gpu.getStates(img).cpu().numpy()
    assert rel_err(out, cpu.getStates(img.cpu().numpy()).numpy()) < TOL
    h.close()


@pytest.mark.parametrize("shape,multi_view,use_graph", [((224, 224), False, False), ((64, 64), True, True), ((112, 112), False, True)])
def test_pixel_pipeline_on_the_layered_encoder(shape, multi_view, use_graph):
    """stepper -> rasteriser -> layered encoder on one stream (PixelStateVecEnv), eager and replayed from a HIP graph:
    the states are the CPU float32 forward of the frames the env holds."""
    from srlhip.pixel_env import PixelStateVecEnv
    ch = 6 if multi_view else 3
    gpu, cpu = make_nets(5, 21, shape, ch)
    env = PixelStateVecEnv("KukaButtonGymEnv-v0", 24, gpu, seed=5, img_shape=shape, env_kwargs={"multi_view": multi_view}, use_graph=use_graph)
    states = env.reset()
    torch.cuda.synchronize()
    assert rel_err(states.cpu().numpy(), cpu.getStates(env.images.cpu().numpy()).numpy()) < TOL
    for t in range(6):
        states, rew, done = env.step()
        torch.cuda.synchronize()
        assert rel_err(states.cpu().numpy(), cpu.getStates(env.images.cpu().numpy()).numpy()) < TOL, t
    assert (len(env._graphs) == 1) == use_graph
    env.close()


@pytest.mark.parametrize("shape,n", [((224, 224), 4), ((75, 61), 6), ((48, 56), 33)])
def test_int8_first_layer_of_the_layered_path(shape, n, monkeypatch):
    """SRLHIP_ENCODER_L1=i8 (read when the handle is created): layer 1 of the layered path on the int8 matrix pipe, the measured
    alternative (slower than the split-f16 form in this kernel — DESIGN.md, NOTES section U — so not the default): same bar."""
    monkeypatch.setenv("SRLHIP_ENCODER_L1", "i8")
    gpu, cpu = make_nets(3, 5, shape, 3)
    imgs = frames(n, shape, 3, n)
    out, ref = gpu.getStates(imgs).cpu().numpy(), cpu.getStates(imgs).numpy()
    assert rel_err(out, ref) < TOL, rel_err(out, ref)
    assert not gpu.hip.overflow()
    monkeypatch.delenv("SRLHIP_ENCODER_L1")
    gpu2, _ = make_nets(3, 5, shape, 3)
    assert rel_err(gpu2.getStates(imgs).cpu().numpy(), out) < TOL
