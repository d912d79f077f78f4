"""Per-chunk timing of the Kuka rollout (4096 envs, Philox random agent): us per VecEnv step for consecutive launches
from reset, for the kernel selected by SRLHIP_KUKA_KERNEL.  Separates what depends on the episode phase (contact-free
start, contacts / resets later) from what depends on the device state (clocks): a second, freshly reset handle is
timed right after the first."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "robotics-rl-srl_amd"))
import torch
from srlhip import _lib

n = 4096
dev = torch.device("cuda", 0)
clocks = []


def sample_clocks(stop):
    while not stop.is_set():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = [l for l in out.splitlines() if "sclk" in l or "Power" in l]
            clocks.append((time.perf_counter(), " | ".join(x.split(":", 1)[-1].strip() for x in sclk[:3])))
        except Exception as exc:
            clocks.append((time.perf_counter(), repr(exc)))
        time.sleep(0.05)


def run(kern, chunk, chunks, tag):
    os.environ["SRLHIP_KUKA_KERNEL"] = kern
    cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
    cfg.num_envs, cfg.seed0, cfg.rng_mode, cfg.auto_reset, cfg.io_device = n, 0, _lib.RNG_PHILOX, 1, 1
    h = _lib.Handle(cfg)
    rew = torch.zeros((chunk, n), dtype=torch.float32, device=dev)
    done = torch.zeros((chunk, n), dtype=torch.uint8, device=dev)
    h.reset(obs_out=0)
    h.sync()
    line = []
    for c in range(chunks):
        h.timing_begin()
        h.rollout(chunk, out=(0, rew.data_ptr(), done.data_ptr(), 0))
        ms = h.timing_end()
        line.append("%.1f" % (ms * 1e3 / chunk))
    print(tag, kern, "chunk", chunk, "us/step:", " ".join(line), flush=True)
    h.close()


stop = threading.Event()
th = threading.Thread(target=sample_clocks, args=(stop,)); th.start()
t0 = time.perf_counter()
for kern in sys.argv[1:] or ["group"]:
    run(kern, 128, 16, "A")
    run(kern, 128, 6, "B(fresh)")
    run(kern, 1024, 6, "C(fresh)")
    time.sleep(1.0)
    run(kern, 128, 6, "D(after 1 s idle)")
stop.set(); th.join()
print("clock samples (s since start):")
for t, c in clocks[::4]:
    print("  %.2f %s" % (t - t0, c))
