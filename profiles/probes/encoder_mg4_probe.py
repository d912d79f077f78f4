"""Which pooled features does an encoder build get wrong?  FC = identity (state_dim 64), so the outputs ARE the 64 features.
usage (GPU box): SRLHIP_LIB=... SRLHIP_ENCODER_WAVES=8 python profiles/probes/encoder_mg4_probe.py"""
import os, sys
import numpy as np
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [os.path.join(REPO, "robotics-rl-srl_amd"), REPO]
import torch
from state_representation.models import SRLNeuralNetwork
torch.manual_seed(0)
c = SRLNeuralNetwork(64, cuda=False, img_shape=(64, 64), backend='torch')
with torch.no_grad():
    fc = [m for m in c.model.modules() if isinstance(m, torch.nn.Linear)][-1]
    fc.weight.copy_(torch.eye(64)); fc.bias.zero_()
g = SRLNeuralNetwork(64, cuda=True, img_shape=(64, 64), state_dict=c.model.state_dict(), backend='hip')
x = np.random.RandomState(1).randint(0, 256, size=(512, 64, 64, 3)).astype(np.uint8)
a, b = g.getStates(x).cpu().numpy(), c.getStates(x).numpy()
err = np.abs(a - b)
print("lib", os.environ.get("SRLHIP_LIB", "product"), "waves", os.environ.get("SRLHIP_ENCODER_WAVES", "4"))
print("max err", float(err.max()), "ref max", float(np.abs(b).max()), "overflow", g.hip.overflow())
bad = err > 1e-4 * max(1.0, np.abs(b).max())
print("wrong entries", int(bad.sum()), "of", bad.size, "; images with a wrong feature", int(bad.any(axis=1).sum()), "of", len(x))
print("wrong per channel:", bad.sum(axis=0).tolist())
if bad.any():
    i, ch = np.argwhere(bad)[0]
    print("first: image", int(i), "channel", int(ch), "got", float(a[i, ch]), "want", float(b[i, ch]))
    print("image indices wrong (first 40):", np.nonzero(bad.any(axis=1))[0][:40].tolist())
    r = a[bad] / np.where(b[bad] == 0, 1, b[bad])
    print("got/want ratio quantiles:", np.quantile(r, [0, 0.25, 0.5, 0.75, 1]).tolist(), "; got==0 count", int((a[bad] == 0).sum()))
