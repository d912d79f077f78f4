"""Per-kernel times of the layered encoder on 1024 frames of 224x224x3 (HIP events around each forward; rocprofv3 gives the split):
run under `rocprofv3 --kernel-trace --stats` or alone (prints ms per forward).  SRLHIP_LIB selects an experiment build."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "robotics-rl-srl_amd"))
import torch
from state_representation.models import SRLNeuralNetwork
torch.manual_seed(0)
net = SRLNeuralNetwork(3, cuda=True, img_shape=(224, 224), backend="hip")
imgs = torch.randint(0, 256, (1024, 224, 224, 3), dtype=torch.uint8, device="cuda")
for _ in range(2): net.getStates(imgs)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): net.getStates(imgs)
e1.record(); torch.cuda.synchronize()
print(os.environ.get("SRLHIP_LIB", "product")[-14:], "ms per 1024 frames:", round(e0.elapsed_time(e1) / 5, 4))
