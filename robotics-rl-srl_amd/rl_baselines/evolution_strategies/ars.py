"""rl_baselines/evolution_strategies/ars.py — Augmented Random Search (arXiv 1803.07055) with the reference's
surface (ARSModel: customArguments / getAction / getActionProba / makeEnv / train / save / load, same hyper-parameters
and update rule, ars.py:17-220), evaluated on the GPU.

The reference evaluates the 2 x n_population perturbed linear policies with a Python loop over envs around a
SubprocVecEnv step (ars.py:152-186).  Here the population is the batch: env 2k runs M + noise * delta_k, env 2k+1 runs
M - noise * delta_k; one batched matmul + argmax (or softmax sampling) produces all actions as an int32 tensor that the
stepper reads in place (DeviceVecEnv, io_device = 1), on the stepper's own HIP stream — no host round trip per step.
Finished directions receive the `None` action (-1) exactly like the reference ("do nothing, as we are done")."""
import pickle
import time

import numpy as np
import torch

from srlhip.device_env import DeviceVecEnv, DeviceVecFrameStack, DeviceVecNormalize


class ARSModel(object):
    def __init__(self):
        self.n_population = None
        self.top_population = None
        self.step_size = None
        self.exploration_noise = None
        self.continuous_actions = None
        self.max_step_amplitude = None
        self.deterministic = None
        self.M = None                     # the linear policy (numpy, like the reference: pickled by save())

    def save(self, save_path, _locals=None):
        assert self.M is not None, "Error: must train or load model before use"
        with open(save_path, "wb") as f:
            pickle.dump(self.__dict__, f)

    @classmethod
    def load(cls, load_path, args=None):
        with open(load_path, "rb") as f:
            class_dict = pickle.load(f)
        loaded_model = ARSModel()
        loaded_model.__dict__ = class_dict
        return loaded_model

    def customArguments(self, parser):
        parser.add_argument('--num-population', help='Number of population (each one has 2 envs)', type=int, default=10)
        parser.add_argument('--exploration-noise', help='The standard deviation of the exploration noise', type=float,
                            default=0.02)
        parser.add_argument('--step-size', help='The step size for param update', type=float, default=0.02)
        parser.add_argument('--top-population', help='Number of top population to use in update', type=int, default=2)
        parser.add_argument('--algo-type', help='"v1" is standard ARS, "v2" is for rolling average normalization.',
                            type=str, default="v2", choices=["v1", "v2"])
        parser.add_argument('--max-step-amplitude', type=float, default=10,
                            help='Set the maximum update vectors amplitude (mesured in factors of step_size)')
        parser.add_argument('--deterministic', action='store_true', default=False,
                            help='do a deterministic approach for the actions on the output of the policy')
        return parser

    @classmethod
    def getOptParam(cls):
        return {"top_population": (int, (1, 5)), "exploration_noise": (float, (0, 0.1)), "step_size": (float, (0, 0.1)),
                "max_step_amplitude": (float, (1, 100))}

    # ---- host-side single-policy interface (replay / enjoy: rl_baselines/evolution_strategies/ars.py:77-104) ------
    def _scores(self, observation, delta=0):
        assert self.M is not None, "Error: must train or load model before use"
        return np.atleast_2d(np.asarray(observation, dtype=np.float64)) @ (self.M + delta)

    def getActionProba(self, observation, dones=None, delta=0):
        """Continuous actions: the linear policy's output; discrete: its softmax, one row per observation."""
        scores = self._scores(observation, delta)
        if self.continuous_actions:
            return scores
        z = np.exp(scores - scores.max(axis=1, keepdims=True))
        return z / z.sum(axis=1, keepdims=True)

    def getAction(self, observation, dones=None, delta=0):
        if self.continuous_actions:
            return self._scores(observation, delta)
        if self.deterministic:
            return self._scores(observation, delta).argmax(axis=1)
        cdf = np.cumsum(self.getActionProba(observation, delta=delta), axis=1)          # inverse-CDF sampling, all rows at once
        return (np.random.random_sample((len(cdf), 1)) * cdf[:, -1:] < cdf).argmax(axis=1)

    # ---- env assembly: ars.py:107-126 with the device-resident stack ----------------------------------------------
    @classmethod
    def makeEnv(cls, args, env_kwargs=None, load_path_normalise=None):
        if "num_population" in args.__dict__:
            args.num_cpu = args.num_population * 2
        env_kwargs = dict(env_kwargs or {})
        env_kwargs.setdefault("srl_model", getattr(args, "srl_model", "ground_truth"))
        envs = DeviceVecEnv(args.env, args.num_cpu, seed=args.seed, env_kwargs=env_kwargs,
                            device_id=getattr(args, "device_id", 0))
        envs = DeviceVecFrameStack(envs, getattr(args, "num_stack", 1))
        if getattr(args, "srl_model", "ground_truth") != "raw_pixels" and getattr(args, "algo_type", "v2") == "v2":
            envs = DeviceVecNormalize(envs, norm_obs=True, norm_reward=False)
            if load_path_normalise is not None:        # replay: frozen statistics of the training run (rl_baselines/utils.py:232-237)
                envs.training = False
                envs.load_running_average(load_path_normalise)
        return envs

    @staticmethod
    def batched_actions(obs, M, delta, noise, active, continuous, deterministic, generator=None):
        """obs [2P, D] (env 2k = +delta_k, env 2k+1 = -delta_k), M [D, A], delta [P, D, A] -> actions for every env.
        Discrete: int32 [2P] with -1 for finished directions.  Continuous: float32 [2P, A] (zeros when finished)."""
        P = delta.shape[0]
        sign = torch.tensor([1.0, -1.0], dtype=M.dtype, device=M.device).view(1, 2, 1, 1)
        W = M.unsqueeze(0).unsqueeze(0) + noise * sign * delta.unsqueeze(1)              # [P, 2, D, A]
        out = torch.matmul(obs.view(P, 2, 1, -1).to(M.dtype), W).view(2 * P, -1)       # [2P, A]
        if continuous:
            return (out * active.unsqueeze(-1).to(out.dtype)).to(torch.float32).contiguous()
        if deterministic:
            a = torch.argmax(out, dim=1)
        else:
            a = torch.multinomial(torch.softmax(out, dim=1), 1, generator=generator).squeeze(1)
        return torch.where(active, a, torch.full_like(a, -1)).to(torch.int32).contiguous()

    def train(self, args, callback=None, env_kwargs=None, train_kwargs=None):
        assert args.top_population <= args.num_population, \
            "Cannot select top %d, from population of %d." % (args.top_population, args.num_population)
        assert args.num_population > 1, "The population cannot be less than 2."
        env = self.makeEnv(args, env_kwargs)
        args.__dict__.update(train_kwargs or {})
        continuous = bool(getattr(args, "continuous_actions", False))
        action_space = int(np.prod(env.action_space.shape)) if continuous else env.action_space.n
        obs_dim = int(np.prod(env.observation_space.shape))
        self.n_population, self.top_population = args.num_population, args.top_population
        self.step_size, self.exploration_noise = args.step_size, args.exploration_noise
        self.continuous_actions, self.max_step_amplitude = continuous, args.max_step_amplitude
        self.deterministic = bool(getattr(args, "deterministic", False))
        self.M = np.zeros((obs_dim, action_space))
        num_updates = int(args.num_timesteps) // args.num_population * 2
        P, dev = self.n_population, env.device
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(args.seed))
        M = torch.zeros((obs_dim, action_space), dtype=torch.float64, device=dev)
        start_time, step = time.time(), 0
        self.history = []
        log_dir = getattr(args, "log_dir", None)
        with torch.cuda.stream(env.torch_stream):          # policy math and stepper kernels on ONE stream: no host syncs
            while step < num_updates:
                r = torch.zeros((P, 2), dtype=torch.float64, device=dev)
                delta = torch.randn((P, obs_dim, action_space), dtype=torch.float64, device=dev, generator=gen)
                done = torch.zeros(2 * P, dtype=torch.bool, device=dev)
                live_steps = torch.zeros((), dtype=torch.int64, device=dev)      # env steps taken while some direction was still running
                step0 = step
                obs = env.reset()
                while True:
                    actions = self.batched_actions(obs, M, delta, self.exploration_noise, ~done, continuous,
                                                   self.deterministic, gen)
                    live_steps += (~done).any().to(torch.int64)
                    obs, reward, new_done = env.step(actions)
                    step += P
                    done = done | (new_done != 0)
                    # cumulate the reward for every direction that is not finished (ars.py:178-180: after the update of `done`)
                    r += (reward.to(torch.float64) * (~done).to(torch.float64)).view(P, 2)
                    if callback is not None:
                        callback(locals(), globals())
                    if (step // P) % 16 == 0 and bool(done.all()):      # the only device->host read: every 16 env steps
                        step = step0 + P * int(live_steps)                # the <= 15 idle steps since the last direction ended do not count
                        break
                    if (step / P + 1) % 500 == 0:
                        print("{} steps - {:.2f} FPS".format(step, step / (time.time() - start_time)))
                idx = torch.argsort(r.max(dim=1).values, descending=True)[:self.top_population]
                top = r[idx]
                delta_sum = ((top[:, 0] - top[:, 1]).view(-1, 1, 1) * delta[idx]).sum(0)
                # the normalisation of step_size guards against zero variance on sparse rewards (ars.py:196-199)
                denom = torch.clamp(self.top_population * top.std(unbiased=False), min=1.0 / self.max_step_amplitude)
                M = M + (self.step_size / denom) * delta_sum
                self.M = M.cpu().numpy()                   # callbacks may save() the model at any time
                self.history.append(float(r.mean()))
        self.M = M.cpu().numpy()
        if log_dir is not None and hasattr(env, "save_running_average"):      # the policy was trained on normalised observations
            env.save_running_average(log_dir)
        env.close()
        return self
