"""Layered HIP encoder (csrc/encoder_general.hip) against the PyTorch-ROCm forward (MIOpen) on the same frames."""
import os, sys, json
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "robotics-rl-srl_amd"))
import torch
from state_representation.models import SRLNeuralNetwork


def timed(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


for shape, ch, n in (((224, 224), 3, 512), ((224, 224), 3, 4096), ((64, 64), 6, 4096), ((224, 224), 6, 512), ((64, 64), 3, 4096)):
    if shape == (64, 64) and ch == 3: os.environ["SRLHIP_ENCODER_GENERAL"] = "1"
    net = SRLNeuralNetwork(3, cuda=True, img_shape=shape, n_channels=ch, backend="hip")
    os.environ.pop("SRLHIP_ENCODER_GENERAL", None)
    imgs = torch.randint(0, 256, (n,) + shape + (ch,), dtype=torch.uint8, device="cuda")
    out = torch.empty((n, 3), dtype=torch.float32, device="cuda")
    hip_ms = timed(lambda: net.getStates(imgs, out=out), 10)
    torch_ms = timed(lambda: net.getStatesTorch(imgs), 3)
    g = (shape[0] // 2) * (shape[1] // 2)
    flops = 2.0 * n * (g * 64 * 49 * ch + (g // 4) * 64 * 576)        # conv1 + conv2 (conv3 and the FC are < 3 %)
    print(json.dumps({"shape": list(shape) + [ch], "frames": n, "hip_ms": round(hip_ms, 4), "torch_ms": round(torch_ms, 4),
                      "alg_tflops": round(flops / hip_ms / 1e9, 1), "frames_per_s": round(n / hip_ms * 1e3)}))
