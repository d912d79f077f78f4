import os, sys, time
sys.path.insert(0, os.path.join(os.getcwd(), "robotics-rl-srl_amd")); sys.path.insert(0, os.getcwd())
import torch
from srlhip.pixel_env import PixelStateVecEnv
from state_representation.models import SRLNeuralNetwork
torch.manual_seed(0)
enc = SRLNeuralNetwork(3, cuda=True, img_shape=(64, 64))
for g in (False, True):
    env = PixelStateVecEnv("KukaButtonGymEnv-v0", 4096, enc, seed=0, use_graph=g)
    env.reset()
    for _ in range(300): env.step()
    env.h.sync(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(1000): env.step()
    env.h.sync(); torch.cuda.synchronize()
    print("use_graph", g, "ms per VecEnv step", (time.perf_counter() - t0))
    env.close()
