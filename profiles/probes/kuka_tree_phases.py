"""Where a step of the full-model Kuka kernel spends its time: the profiling build of the tree kernels (make -C robotics-rl-srl_amd/csrc prof:
-DSRL_TREE_PROF, shader-clock stamps at the phase boundaries of tphysics_step, accumulated by lane 0 of a workgroup) run on the
bench configuration — 4096 envs, Philox random agent, T fused steps.  Usage (GPU box, repo root):
    SRLHIP_LIB=robotics-rl-srl_amd/csrc/build/libsrlhip_prof.so python profiles/probes/kuka_tree_phases.py [T]
Prints cycles per phase and per step for workgroups 0 and 511 (the device printf of the kernel), then the launch time."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "robotics-rl-srl_amd"))
import torch
from srlhip import _lib

assert "prof" in os.environ.get("SRLHIP_LIB", ""), "set SRLHIP_LIB to the profiling build"
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
NO_OUT = len(sys.argv) > 2 and sys.argv[2] == "noout"       # no observation / reward / done planes: what the output stores cost
n = 4096
dev = torch.device("cuda", 0)
cfg = _lib.default_config(_lib.ENV_KUKA_BUTTON)
cfg.num_envs, cfg.seed0, cfg.rng_mode, cfg.auto_reset, cfg.io_device = n, 0, _lib.RNG_PHILOX, 1, 1
h = _lib.Handle(cfg)
rew = torch.zeros((T, n), dtype=torch.float32, device=dev)
done = torch.zeros((T, n), dtype=torch.uint8, device=dev)
obs = torch.zeros((T, n, 3), dtype=torch.float32, device=dev)
h.reset(obs_out=0)
h.sync()
for rep in range(2):
    h.timing_begin()
    h.rollout(T, out=(0, 0, 0, 0) if NO_OUT else (obs.data_ptr(), rew.data_ptr(), done.data_ptr(), 0))
    ms = h.timing_end()
    print("launch {}: T {} {:.3f} ms, {:.2f} us per step".format(rep, T, ms, ms * 1e3 / T), flush=True)
h.close()
names = ["start of the physics step (what is left of the old phase 0 once 20 and 21 are stamped)", "IK", "collision + motor targets", "RNEA sums", "CRBA", "Gauss-Jordan 12x12", "row setup",
         "150 sweeps, free steps", "sweeps + setup, steps with generic rows", "integrate + refresh (sincos, FK)",
         "counters, reward, termination", "episode statistics, auto-reset, observation + output stores",
         "generic: candidates -> row definitions", "generic: W J of every slot", "generic: own bank-B row (diagonal, rhs, couplings)", "generic: the 150 sweeps",
         "generic: outputs", "NUMBER of steps with generic rows (a count, not cycles)", "NUMBER of steps with a joint-limit row", "NUMBER of steps with a contact row",
         "loop back-edge + action sampling", "noise draw + action mapping (step_command)"]
print("phase names:", {i: s for i, s in enumerate(names)})
