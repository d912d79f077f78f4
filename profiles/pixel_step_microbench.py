"""Config-4 step loop (run on the GPU box): eager launches vs the HIP-graph replay of stepper + rasteriser + encoder."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "robotics-rl-srl_amd"))
import torch
from srlhip.pixel_env import PixelStateVecEnv
from state_representation.models import SRLNeuralNetwork
torch.manual_seed(0)
enc = SRLNeuralNetwork(3, cuda=True, img_shape=(64, 64))
for flag in (False, True, False, True):
    env = PixelStateVecEnv("KukaButtonGymEnv-v0", 4096, enc, seed=0, use_graph=flag)
    env.reset()
    for _ in range(20): env.step()
    torch.cuda.synchronize(); env.h.sync()
    t0 = time.perf_counter()
    for _ in range(200): env.step()
    env.h.sync(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    print("use_graph={}: {:.1f} us per 4096-env pixel step -> {:.2e} env-steps/s".format(flag, dt * 1e6, 4096 / dt))
    env.close()
