"""rl_baselines/evolution_strategies/cma_es.py — CMA-ES with the reference's surface (CMAESModel: customArguments /
getAction / getActionProba / makeEnv / train / save / load, cma_es.py:24-140), evaluated on the GPU.

The reference asks the `cma` package for `num_population` parameter vectors of a small PyTorch MLP (observation -> 100 ->
actions, cma_es.py:98-105,307-326), then loops over the population in Python — set the parameters, forward ONE observation
— around a SubprocVecEnv step, `None` for finished members (cma_es.py:114-135).  Here the population is the batch:
member k drives env k, all forwards are two batched matmuls on the stepper's own HIP stream (DeviceVecEnv, io_device = 1),
the int32 action tensor is read by the stepper in place, and the only host read is a `done.all()` every 16 steps.

The `cma` package is not available in this environment; the strategy itself — (mu/mu_w, lambda)-CMA-ES with cumulative
step-size adaptation and rank-one + rank-mu covariance updates, Hansen's tutorial parameter settings, i.e. what
`cma.CMAEvolutionStrategy(x0, sigma0, {'popsize': n})` runs — is restated in `CMAES` below as torch tensor code on the same
device.  Learned-from-pixels policies (the reference's CNNPolicyPytorch) are not part of the device loop."""
import math
import pickle
import time

import numpy as np
import torch

from srlhip.device_env import DeviceVecEnv, DeviceVecFrameStack, DeviceVecNormalize


class CMAES(object):
    """ask() -> [lambda, n] candidates; tell(candidates, losses) (minimisation).  float64 tensors on `device`."""

    def __init__(self, x0, sigma0, popsize, device, seed=0):
        n = len(x0)
        self.n, self.lam, self.device = n, int(popsize), device
        self.mean = torch.as_tensor(np.asarray(x0, dtype=np.float64), device=device)
        self.sigma = float(sigma0)
        self.mu = self.lam // 2
        w = math.log(self.mu + 0.5) - torch.log(torch.arange(1, self.mu + 1, dtype=torch.float64, device=device))
        self.weights = w / w.sum()
        self.mueff = float(1.0 / (self.weights ** 2).sum())
        self.cc = (4 + self.mueff / n) / (n + 4 + 2 * self.mueff / n)
        self.cs = (self.mueff + 2) / (n + self.mueff + 5)
        self.c1 = 2 / ((n + 1.3) ** 2 + self.mueff)
        self.cmu = min(1 - self.c1, 2 * (self.mueff - 2 + 1 / self.mueff) / ((n + 2) ** 2 + self.mueff))
        self.damps = 1 + 2 * max(0.0, math.sqrt((self.mueff - 1) / (n + 1)) - 1) + self.cs
        self.chiN = math.sqrt(n) * (1 - 1 / (4 * n) + 1 / (21 * n * n))
        self.pc = torch.zeros(n, dtype=torch.float64, device=device)
        self.ps = torch.zeros(n, dtype=torch.float64, device=device)
        self.C = torch.eye(n, dtype=torch.float64, device=device)
        self.B = torch.eye(n, dtype=torch.float64, device=device)
        self.D = torch.ones(n, dtype=torch.float64, device=device)
        self.gen, self.eigen_gen = 0, 0
        self.rng = torch.Generator(device=device)
        self.rng.manual_seed(int(seed))
        self.xbest, self.fbest = self.mean.clone(), float("inf")

    def ask(self):
        z = torch.randn((self.lam, self.n), dtype=torch.float64, device=self.device, generator=self.rng)
        self._y = (z * self.D) @ self.B.T                       # N(0, C)
        return self.mean + self.sigma * self._y

    def tell(self, x, losses):
        losses = torch.as_tensor(losses, dtype=torch.float64, device=self.device)
        order = torch.argsort(losses)
        if float(losses[order[0]]) < self.fbest:
            self.fbest, self.xbest = float(losses[order[0]]), x[order[0]].clone()
        y = self._y[order[:self.mu]]
        yw = (self.weights.unsqueeze(1) * y).sum(0)
        self.mean = self.mean + self.sigma * yw
        self.gen += 1
        invsqrtC_yw = self.B @ ((self.B.T @ yw) / self.D)
        self.ps = (1 - self.cs) * self.ps + math.sqrt(self.cs * (2 - self.cs) * self.mueff) * invsqrtC_yw
        hsig = float(self.ps.norm()) / math.sqrt(1 - (1 - self.cs) ** (2 * self.gen)) / self.chiN < 1.4 + 2 / (self.n + 1)
        self.pc = (1 - self.cc) * self.pc + (math.sqrt(self.cc * (2 - self.cc) * self.mueff) * yw if hsig else 0.0)
        rank_mu = (y.T * self.weights) @ y
        self.C = ((1 - self.c1 - self.cmu) * self.C + self.c1 * (torch.outer(self.pc, self.pc) + (0.0 if hsig else self.cc * (2 - self.cc)) * self.C)
                  + self.cmu * rank_mu)
        self.sigma *= math.exp((self.cs / self.damps) * (float(self.ps.norm()) / self.chiN - 1))
        if self.gen - self.eigen_gen > self.lam / ((self.c1 + self.cmu) * self.n * 10):      # O(n^2) amortised, as in the tutorial
            self.eigen_gen = self.gen
            self.C = torch.triu(self.C) + torch.triu(self.C, 1).T
            d, self.B = torch.linalg.eigh(self.C)
            self.D = torch.sqrt(torch.clamp(d, min=1e-20))


class BatchedMLP(object):
    """The reference's MLPPolicyPytorch(in, [100], out) for a whole population: parameters [P, n] in nn.Module.parameters()
    order (fc_in.weight [H, D], fc_in.bias [H], fc_out.weight [A, H], fc_out.bias [A]) -> scores [P, A]."""

    def __init__(self, in_dim, out_dim, hidden=100):
        self.D, self.H, self.A = int(in_dim), int(hidden), int(out_dim)
        self.n_params = self.H * self.D + self.H + self.A * self.H + self.A

    def split(self, params):
        D, H, A = self.D, self.H, self.A
        o = 0
        w1 = params[..., o:o + H * D].reshape(params.shape[:-1] + (H, D)); o += H * D
        b1 = params[..., o:o + H]; o += H
        w2 = params[..., o:o + A * H].reshape(params.shape[:-1] + (A, H)); o += A * H
        b2 = params[..., o:o + A]
        return w1, b1, w2, b2

    def forward(self, params, obs):
        """params [P, n], obs [P, D] -> [P, A]"""
        w1, b1, w2, b2 = self.split(params)
        h = torch.relu(torch.bmm(w1, obs.to(params.dtype).unsqueeze(-1)).squeeze(-1) + b1)
        return torch.bmm(w2, h.unsqueeze(-1)).squeeze(-1) + b2


class CMAESModel(object):
    def __init__(self):
        self.policy = None              # BatchedMLP (shape only; the parameters are best_model)
        self.n_population = None
        self.mu = None
        self.sigma = None
        self.continuous_actions = None
        self.deterministic = None
        self.es = None
        self.best_model = None          # numpy parameter vector of the best member seen (cma's result.xbest)

    def save(self, save_path, _locals=None):
        assert self.policy is not None, "Error: must train or load model before use"
        d = dict(self.__dict__)
        d["es"] = None                  # device tensors / generator stay out of the pickle
        with open(save_path, "wb") as f:
            pickle.dump(d, f)

    @classmethod
    def load(cls, load_path, args=None):
        with open(load_path, "rb") as f:
            class_dict = pickle.load(f)
        loaded_model = CMAESModel()
        loaded_model.__dict__ = class_dict
        return loaded_model

    def customArguments(self, parser):
        parser.add_argument('--num-population', help='Number of population', type=int, default=20)
        parser.add_argument('--mu', type=float, default=0, help='inital location for gaussian sampling of network parameters')
        parser.add_argument('--sigma', type=float, default=0.14, help='inital scale for gaussian sampling of network parameters')
        parser.add_argument('--cuda', action='store_true', default=False, help='use gpu for the neural network')
        parser.add_argument('--deterministic', action='store_true', default=False,
                            help='do a deterministic approach for the actions on the output of the policy')
        return parser

    @classmethod
    def getOptParam(cls):
        return {"sigma": (float, (0, 0.2))}

    # ---- host-side single-policy interface (replay / enjoy) ----------------------------------------------------------
    def _scores(self, observation):
        assert self.policy is not None and self.best_model is not None, "Error: must train or load model before use"
        obs = torch.as_tensor(np.atleast_2d(np.asarray(observation, dtype=np.float64)))
        params = torch.as_tensor(self.best_model, dtype=torch.float64).unsqueeze(0).expand(len(obs), -1)
        return self.policy.forward(params, obs).numpy()

    def getActionProba(self, observation, dones=None):
        scores = self._scores(observation)
        if self.continuous_actions:
            return scores
        z = np.exp(scores - scores.max(axis=1, keepdims=True))
        return z / z.sum(axis=1, keepdims=True)

    def getAction(self, observation, dones=None):
        if self.continuous_actions:
            return self._scores(observation)
        if self.deterministic:
            return self._scores(observation).argmax(axis=1)
        cdf = np.cumsum(self.getActionProba(observation), axis=1)
        return (np.random.random_sample((len(cdf), 1)) * cdf[:, -1:] < cdf).argmax(axis=1)

    # ---- env assembly: cma_es.py:79-80 (createEnvs with allow_early_resets) on the device-resident stack ----------------
    @classmethod
    def makeEnv(cls, args, env_kwargs=None, load_path_normalise=None):
        env_kwargs = dict(env_kwargs or {})
        env_kwargs.setdefault("srl_model", getattr(args, "srl_model", "ground_truth"))
        if env_kwargs["srl_model"] == "raw_pixels":
            raise NotImplementedError("CMA-ES on the device stack evaluates state policies (MLP); the reference's CNN-from-pixels "
                                      "policy is not part of it")
        envs = DeviceVecEnv(args.env, args.num_cpu, seed=args.seed, env_kwargs=env_kwargs, device_id=getattr(args, "device_id", 0))
        envs = DeviceVecFrameStack(envs, getattr(args, "num_stack", 1))
        envs = DeviceVecNormalize(envs, norm_obs=True, norm_reward=False)
        if load_path_normalise is not None:
            envs.training = False
            envs.load_running_average(load_path_normalise)
        return envs

    def train(self, args, callback=None, env_kwargs=None, train_kwargs=None):
        args.num_cpu = args.num_population
        env = self.makeEnv(args, env_kwargs=env_kwargs)
        args.__dict__.update(train_kwargs or {})
        continuous = bool(getattr(args, "continuous_actions", False))
        action_space = int(np.prod(env.action_space.shape)) if continuous else env.action_space.n
        self.policy = BatchedMLP(int(np.prod(env.observation_space.shape)), action_space)
        self.n_population, self.mu, self.sigma = args.num_population, args.mu, args.sigma
        self.continuous_actions, self.deterministic = continuous, bool(getattr(args, "deterministic", False))
        P, dev = self.n_population, env.device
        self.es = CMAES(self.policy.n_params * [self.mu], self.sigma, P, dev, seed=int(args.seed))
        self.best_model = np.array(self.policy.n_params * [self.mu], dtype=np.float64)
        num_updates = int(args.num_timesteps)
        gen = torch.Generator(device=dev)
        gen.manual_seed(int(args.seed) + 1)
        start_time, step = time.time(), 0
        self.history = []
        log_dir = getattr(args, "log_dir", None)
        with torch.cuda.stream(env.torch_stream):          # policy math and stepper kernels on ONE stream: no host syncs
            while step < num_updates:
                obs = env.reset()
                r = torch.zeros(P, dtype=torch.float64, device=dev)
                population = self.es.ask()                  # [P, n_params]
                done = torch.zeros(P, dtype=torch.bool, device=dev)
                live = torch.zeros((), dtype=torch.int64, device=dev)
                k = 0
                while True:
                    scores = self.policy.forward(population, obs)
                    if continuous:
                        actions = (scores * (~done).unsqueeze(-1).to(scores.dtype)).to(torch.float32).contiguous()
                    else:
                        a = torch.argmax(scores, dim=1) if self.deterministic else \
                            torch.multinomial(torch.softmax(scores, dim=1), 1, generator=gen).squeeze(1)
                        actions = torch.where(done, torch.full_like(a, -1), a).to(torch.int32).contiguous()      # None: "do nothing, as we are done"
                    live += (~done).sum()                   # step += np.sum(~done) (cma_es.py:127)
                    obs, reward, new_done = env.step(actions)
                    done = done | (new_done != 0)
                    r += reward.to(torch.float64) * (~done).to(torch.float64)      # cumulate the reward of every member that is not finished
                    k += 1
                    if callback is not None:
                        callback(locals(), globals())
                    if k % 16 == 0 and bool(done.all()):    # the only device->host read
                        break
                step += int(live)
                print("{} steps - {:.2f} FPS".format(step, step / (time.time() - start_time)))
                self.es.tell(population, -r)
                self.best_model = self.es.xbest.cpu().numpy()
                self.history.append(float(r.mean()))
        if log_dir is not None:
            env.save_running_average(log_dir)
        env.close()
        return self
