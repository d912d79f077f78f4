import sys, time
sys.path.insert(0, 'robotics-rl-srl_amd')
import numpy as np, torch
from state_representation.models import SRLNeuralNetwork
torch.manual_seed(0)
net = SRLNeuralNetwork(8, cuda=True, img_shape=(64, 64), backend="hip")
imgs = torch.randint(0, 256, (4096, 64, 64, 3), dtype=torch.uint8, device="cuda")
for fn, name in ((net.getStates, "hip"), (net.getStatesTorch, "torch")):
    for _ in range(3): fn(imgs)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): out = fn(imgs)
    e1.record(); torch.cuda.synchronize()
    print(name, "ms per 4096-batch:", e0.elapsed_time(e1) / 10)
a = net.getStates(imgs).cpu().numpy(); b = net.getStatesTorch(imgs).cpu().numpy()
cpu = SRLNeuralNetwork(8, cuda=False, img_shape=(64, 64), state_dict=net.model.state_dict(), backend="torch")
c = cpu.getStates(imgs[:256].cpu().numpy()).numpy()
print("hip vs cpu", np.abs(a[:256]-c).max()/np.abs(c).max(), "miopen vs cpu", np.abs(b[:256]-c).max()/np.abs(c).max(), "scale", np.abs(c).max())
print("overflow", net.hip.overflow())
