"""Fused HIP encoder (csrc/encoder.hip, through the C-ABI) against the plain PyTorch float32 forward of the same
network on the CPU — the reference the op restates (state_representation/models.py:178-193).  Tolerance: the kernel
computes on the float16 matrix pipe with every operand split into hi + lo/2048 (22 significant bits per product,
float32 accumulation), so it is held to 2e-5 relative to the largest state component; MIOpen's own float32
forward differs from the CPU by about as much."""
import os

import numpy as np
import pytest
import torch

from state_representation.models import SRLNeuralNetwork

pytestmark = pytest.mark.gpu
TOL = 2e-5


def make_net(state_dim, seed, cuda):
    torch.manual_seed(seed)
    net = SRLNeuralNetwork(state_dim, img_shape=(64, 64), backend="torch")
    for m in net.model.modules():                       # non-trivial BatchNorm statistics: exercises the folding
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.5); m.running_var.uniform_(0.5, 2.0); m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0, 0.2)
    sd = net.model.state_dict()
    return (SRLNeuralNetwork(state_dim, cuda=cuda, img_shape=(64, 64), state_dict=sd, backend="hip" if cuda else "torch"),
            SRLNeuralNetwork(state_dim, cuda=False, img_shape=(64, 64), state_dict=sd, backend="torch"))


def rel_err(a, b):
    return float(np.abs(a - b).max() / max(1.0, np.abs(b).max()))


@pytest.mark.parametrize("state_dim,n", [(3, 1), (5, 37), (2, 600), (200, 64), (300, 5)])
def test_hip_encoder_matches_torch_fp32(state_dim, n):
    gpu, cpu = make_net(state_dim, 7 + state_dim, True)
    assert gpu.backend == "hip"
    rs = np.random.RandomState(n)
    imgs = rs.randint(0, 256, size=(n, 64, 64, 3)).astype(np.uint8)
    imgs[0, :, :20] = 0                                   # flat borders: zero padding in normalised space
    if n > 2:
        imgs[1] = 255
        imgs[2, 10:50, 5:60] = rs.randint(0, 256, size=3).astype(np.uint8)
    out = gpu.getStates(imgs).cpu().numpy()
    ref = cpu.getStates(imgs).numpy()
    assert out.shape == (n, state_dim) and np.isfinite(out).all()
    assert rel_err(out, ref) < TOL, rel_err(out, ref)
    assert not gpu.hip.overflow()


def test_hip_encoder_on_rasterised_frames_and_foreign_stream():
    from srlhip.pixel_env import PixelStateVecEnv
    gpu, cpu = make_net(8, 1, True)
    env = PixelStateVecEnv("KukaButtonGymEnv-v0", 300, gpu, seed=2)
    states = env.reset()
    for _ in range(5):
        states, rew, done = env.step()
    torch.cuda.synchronize()
    ref = cpu.getStates(env.images.cpu().numpy()).numpy()
    assert rel_err(states.cpu().numpy(), ref) < TOL
    # the PyTorch-ROCm forward of the same weights agrees too (looser: MIOpen picks its own algorithms)
    assert rel_err(gpu.getStatesTorch(env.images).cpu().numpy(), ref) < 2e-3
    # explicit stream: enqueue on a side stream, then order torch's stream behind it
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    out = gpu.getStates(env.images, stream=side.cuda_stream)
    torch.cuda.current_stream().wait_stream(side)
    assert rel_err(out.cpu().numpy(), ref) < TOL
    env.close()


def test_graph_replayed_pixel_step_equals_eager_step():
    """PixelStateVecEnv replays stepper + rasteriser + encoder of one step from a HIP graph (srlhip_graph_*): same
    states / rewards / dones / frames as three eager launches, with device-sampled and with given actions."""
    from srlhip.pixel_env import PixelStateVecEnv
    gpu, _ = make_net(6, 3, True)
    envs = [PixelStateVecEnv("KukaButtonGymEnv-v0", 200, gpu, seed=4, use_graph=flag) for flag in (True, False)]
    for e in envs:
        e.reset()
    acts = torch.zeros((200,), dtype=torch.int32, device="cuda")
    for t in range(12):
        outs = []
        for e in envs:
            if t % 3 == 2:
                acts.copy_(torch.full((200,), t % 6, dtype=torch.int32))
                s, r, d = e.step(acts)
            else:
                s, r, d = e.step()
            torch.cuda.synchronize()
            outs.append((s.cpu().numpy().copy(), r.cpu().numpy().copy(), d.cpu().numpy().copy(), e.images.cpu().numpy().copy()))
        for a, b in zip(*outs):
            assert np.array_equal(a, b), t
    assert len(envs[0]._graphs) == 2 and not envs[1]._graphs
    for e in envs:
        e.close()


@pytest.mark.parametrize("img_shape", [(64, 64), (96, 96)])
def test_vec_env_with_a_learned_srl_model(img_shape, tmp_path):
    """createEnvs / HipVecEnv with a learned SRL model (registered_srl[...] is SRLType.SRL; the reference's
    MultiprocessSRLModel path, rl_baselines/utils.py:162-216): the observation is encoder(render()) — checked against a
    raw_pixels twin stepped with the same seeds and actions whose frames go through the same network on the CPU.
    64x64 frames take the fused HIP encoder, 96x96 the layered one (csrc/encoder_general.hip)."""
    from srlhip.vec_env import HipVecEnv
    n, sd = 48, 5
    gpu, cpu = make_net(sd, 11, True) if img_shape == (64, 64) else (None, None)
    if gpu is None:
        torch.manual_seed(3)
        gpu = SRLNeuralNetwork(sd, cuda=True, img_shape=img_shape)
        cpu = SRLNeuralNetwork(sd, cuda=False, img_shape=img_shape, state_dict=gpu.model.state_dict(), backend="torch")
    assert gpu.backend == "hip"
    kw = {"img_shape": img_shape, "random_target": True}
    env = HipVecEnv("KukaButtonGymEnv-v0", n, seed=9, env_kwargs=dict(kw, srl_model="autoencoder"), encoder=gpu, log_dir=str(tmp_path))
    twin = HipVecEnv("KukaButtonGymEnv-v0", n, seed=9, env_kwargs=dict(kw, srl_model="raw_pixels"))
    assert env.observation_space.shape == (sd,) and env.observation_space.dtype == np.float32
    tol = TOL
    obs, frames = env.reset(), twin.reset()
    assert obs.shape == (n, sd) and rel_err(obs, cpu.getStates(frames).numpy()) < tol
    rs = np.random.RandomState(0)
    for t in range(25):
        a = rs.randint(6, size=n)
        obs, r, d, info = env.step(a)
        frames, r2, d2, _ = twin.step(a)
        assert np.array_equal(r, r2) and np.array_equal(d, d2)
        assert rel_err(obs, cpu.getStates(frames).numpy()) < tol, t
    assert len(env.get_images()) == n and env.get_images()[0].shape == img_shape + (3,)
    env.close(); twin.close()
    # the createEnvs entry point: a Namespace like rl_baselines.train builds, random-initialised encoder of state_dim 4
    import argparse
    from rl_baselines.utils import createEnvs
    args = argparse.Namespace(env="MobileRobotGymEnv-v0", num_cpu=8, seed=0, log_dir=None, srl_model="robotic_priors", num_stack=1)
    venv = createEnvs(args, env_kwargs={"state_dim": 4, "img_shape": img_shape})
    o = venv.reset()
    assert o.shape == (8, 4) and np.isfinite(o).all()
    o, r, d, _ = venv.step([0] * 8)
    assert o.shape == (8, 4)
    venv.close()


def test_hip_encoder_rejects_what_it_does_not_cover():
    from srlhip import _lib
    with pytest.raises(RuntimeError):
        SRLNeuralNetwork(2, cuda=True, img_shape=(1040, 64), backend="hip")                   # beyond the C-ABI's 1024-pixel side
    net = SRLNeuralNetwork(2, cuda=True, img_shape=(1040, 64))                # auto: falls to the PyTorch-ROCm forward, on the GPU
    assert net.backend == "torch" and net.getStates(np.zeros((2, 1040, 64, 3), np.uint8)).is_cuda
    assert SRLNeuralNetwork(2, cuda=True, img_shape=(224, 224)).backend == "hip"              # the reference's frame size: layered HIP path
    gpu, _ = make_net(2, 0, True)
    with pytest.raises(_lib.SrlHipError):
        gpu.hip.forward(0, 4, 0)                                           # null buffers
    assert gpu.getStates(np.zeros((0, 64, 64, 3), np.uint8)).shape == (0, 2)


def test_two_waves_per_simd_variant_in_a_subprocess():
    """SRLHIP_ENCODER_WAVES=8 selects the MG = 4 instantiation of encoder_fwd_k (an experiment knob read once per process): the same
    parity bar, in a child process — a variant the library can be switched to must not go untested."""
    import subprocess
    import sys
    code = ("import sys, numpy as np; sys.path[:0] = {!r}; import torch; "
            "from state_representation.models import SRLNeuralNetwork; torch.manual_seed(0); "
            "g = SRLNeuralNetwork(5, cuda=True, img_shape=(64, 64), backend='hip'); "
            "c = SRLNeuralNetwork(5, cuda=False, img_shape=(64, 64), state_dict=g.model.state_dict(), backend='torch'); "
            "x = np.random.RandomState(1).randint(0, 256, size=(300, 64, 64, 3)).astype(np.uint8); "
            "a, b = g.getStates(x).cpu().numpy(), c.getStates(x).numpy(); "
            "print('ERR', float(np.abs(a - b).max() / max(1.0, np.abs(b).max())), g.hip.overflow())").format(sys.path[:4])
    env = dict(os.environ, SRLHIP_ENCODER_WAVES="8")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    err = float(out.stdout.split("ERR")[1].split()[0])
    assert err < TOL and "False" in out.stdout.split("ERR")[1]
