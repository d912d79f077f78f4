"""python -m environments.dataset_generator — same CLI, seeding and on-disk output as
/root/reference/environments/dataset_generator.py:120-274, but the `--num-cpu` "threads"
are lanes of batched GPU handles instead of OS processes (the reference forks num_cpu workers, :166-190; here `--device-ids`
spreads the lanes over the node's GPUs in contiguous blocks, one handle per GPU, all stepping at once): thread t still runs its own
episode list with the reference's per-episode seeds (:80-86), every env is re-seeded and
reset individually when its episode ends (srlhip_seed / srlhip_reset with a mask), and a
random agent samples actions from a per-thread action-space RandomState exactly like
`env.action_space.seed(seed); env.action_space.sample()`."""
from __future__ import division, absolute_import, print_function

import argparse
import glob
import os
import shutil
import time

import numpy as np

from environments import ThreadingType
from environments.registry import registered_env
from srlhip import _lib
from srlhip.recorder import EpisodeSaver


def convertImagePath(args, path, record_id_start):
    image_name = path.split("/")[-1]
    new_record_id = record_id_start + int(path.split("/")[-2].split("_")[-1])
    return args.name + "/record_{:03d}".format(new_record_id) + "/" + image_name


def episode_seeds(args, thread_num):
    """dataset_generator.py:80-83"""
    n_ep = args.num_episode // args.num_cpu + 1 * (args.num_episode % args.num_cpu > thread_num)
    rem = args.num_episode % args.num_cpu
    return [args.seed + i + args.num_episode // args.num_cpu * thread_num + (thread_num if thread_num <= rem else rem)
            for i in range(n_ep)]


class _Fleet(object):
    """The generator's lanes ("threads" 0..n-1) spread over one handle per GPU, contiguous blocks in thread order.  Every method takes /
    returns arrays over ALL lanes; a thread's episode seeds depend on its number only (episode_seeds), so the dataset does not depend
    on how many GPUs generate it."""

    def __init__(self, make_cfg, n, device_ids):
        from srlhip.vec_env import shard_bounds
        device_ids = list(device_ids)[:max(1, min(len(device_ids), n))]
        self.n, self.parts = n, []
        for (lo, hi), dev in zip(shard_bounds(n, len(device_ids)), device_ids):
            self.parts.append((_lib.Handle(make_cfg(hi - lo, dev, lo)), lo, hi))
        h0 = self.parts[0][0]
        self.cfg, self.kind, self.kuka = h0.cfg, h0.cfg.env_kind, h0.cfg.env_kind >= _lib.ENV_KUKA_BUTTON
        self.obs_dim, self.action_dim, self.num_actions = h0.obs_dim, h0.action_dim, h0.num_actions

    def _cat(self, field):
        return np.concatenate([h.get_state(field) for h, _, _ in self.parts], axis=-1)

    def seed_and_reset(self, seeds, mask):
        m = mask.astype(np.uint8)
        for h, lo, hi in self.parts:
            if m[lo:hi].any():
                h.seed(seeds[lo:hi], mask=m[lo:hi])
                h.reset(mask=m[lo:hi], obs_out=np.zeros((hi - lo, self.obs_dim), np.float32))

    def step(self, actions):
        """one step of every lane: launched on every GPU first (srlhip_step_async), then collected -> done uint8 [n]"""
        for h, lo, hi in self.parts:
            h.step_async(actions[lo:hi])
        return np.concatenate([h.step_wait()[2] for h, _, _ in self.parts])

    def last_reward(self):
        return self._cat(_lib.F_LAST_REWARD)

    def render(self):
        return np.concatenate([h.render() for h, _, _ in self.parts])

    def ground_truth_and_target(self):
        if self.kuka:
            return self._cat(_lib.F_KUKA_GRIPPER).T.copy(), self._cat(_lib.F_KUKA_BUTTON_POS).T.copy()
        x, y = self._cat(_lib.F_POS_X), self._cat(_lib.F_POS_Y)
        cur = self._cat(_lib.F_CUR_TARGET)
        tx = np.where(cur > 0, self._cat(_lib.F_TARGET2_X), self._cat(_lib.F_TARGET_X))
        ty = np.where(cur > 0, self._cat(_lib.F_TARGET2_Y), self._cat(_lib.F_TARGET_Y))
        if self.kind == _lib.ENV_MOBILE_1D:
            return x[:, None], tx[:, None]
        if self.kind == _lib.ENV_MOBILE_LINE:
            return np.stack([x, y], 1), (tx - 0.2)[:, None]
        return np.stack([x, y], 1), np.stack([tx, ty], 1)

    def close(self):
        for h, _, _ in self.parts:
            h.close()


def run_batched(args):
    from srlhip.vec_env import parse_device_ids
    env_cls = registered_env[args.env][0]
    n = args.num_cpu

    def make_cfg(count, device, first):
        cfg = _lib.default_config(env_cls.ENV_KIND)
        cfg.num_envs, cfg.device_id, cfg.first_env_id = count, device, first
        cfg.is_discrete, cfg.random_target = int(not args.continuous_actions), int(args.random_target)
        cfg.shape_reward, cfg.multi_view = int(args.shape_reward), int(args.multi_view)     # force_down: the env class's own ctor default (srlhip_default_config), the reference never passes it
        cfg.max_distance = args.max_distance
        cfg.obs_mode, cfg.rng_mode, cfg.auto_reset = _lib.OBS_GROUND_TRUTH, _lib.RNG_MT19937, 0
        cfg.img_h = cfg.img_w = args.img_size                       # frames come from srlhip_render (tile rasteriser)
        return cfg

    fleet = _Fleet(make_cfg, n, parse_device_ids(getattr(args, "device_ids", None)) or [args.device_id])
    cfg = fleet.cfg
    partition = n > 1
    savers = None
    if not args.no_record_data:
        import importlib
        globals_ = importlib.import_module(env_cls.__module__).getGlobals()
        savers = [EpisodeSaver(args.name + ("_part-" + str(t) if partition else ""), args.max_distance, -1,
                               globals_=globals_, relative_pos=True, path=args.save_path) for t in range(n)]
    seeds = [episode_seeds(args, t) for t in range(n)]
    ep_idx = np.zeros(n, dtype=np.int64)
    active = np.array([len(s) > 0 for s in seeds])
    arng = [np.random.RandomState() for _ in range(n)]
    adim = fleet.action_dim
    t_ep = np.zeros(n, dtype=np.int64)

    def start_episodes(mask):
        sd = np.zeros(n, dtype=np.int64)
        for i in np.nonzero(mask)[0]:
            sd[i] = seeds[i][ep_idx[i]]
            arng[i].seed(int(sd[i]) % 2 ** 32)                 # env.action_space.seed(seed)
        fleet.seed_and_reset(sd, mask)
        if savers is not None:
            gt, tgt = fleet.ground_truth_and_target()
            frames_rgb = fleet.render()
            for i in np.nonzero(mask)[0]:
                savers[i].reset(frames_rgb[i], tgt[i], gt[i])
        t_ep[mask] = 0

    start_episodes(active.copy())
    frames, start_time = 0, time.time()
    while active.any():
        if cfg.is_discrete:
            actions = np.full(n, -1, dtype=np.int32)
            for i in np.nonzero(active)[0]:
                actions[i] = arng[i].randint(fleet.num_actions)
        else:
            actions = np.zeros((n, adim), dtype=np.float32)
            for i in np.nonzero(active)[0]:
                actions[i] = arng[i].uniform(-1, 1, adim).astype(np.float32)
        done = fleet.step(actions).astype(bool) & active
        frames += int(active.sum())
        t_ep[active] += 1
        if savers is not None:
            gt, _ = fleet.ground_truth_and_target()
            rew = fleet.last_reward()
            frames_rgb = fleet.render()
            for i in np.nonzero(active)[0]:
                r = float(rew[i]) if args.shape_reward else int(rew[i])
                a = int(actions[i]) if cfg.is_discrete else actions[i]
                savers[i].step(frames_rgb[i], a, r, bool(done[i]), gt[i])
        if done.any():
            for i in np.nonzero(done)[0]:
                print("Episode finished after {} timesteps".format(t_ep[i] + 1))
                ep_idx[i] += 1
                if ep_idx[i] >= len(seeds[i]):
                    active[i] = False
            restart = done & active
            if restart.any():
                start_episodes(restart)
    fps = frames / max(time.time() - start_time, 1e-9)
    print("{:.2f} FPS".format(fps))
    fleet.close()
    return fps


# flag table: (flags, kwargs) — names, defaults and meaning as in the reference CLI (:123-151)
_FLAGS = [
    (("--num-cpu",), dict(type=int, default=1, help="envs stepped together (lanes of one GPU handle)")),
    (("--num-episode",), dict(type=int, default=50, help="episodes to generate")),
    (("--save-path",), dict(type=str, default="srl_zoo/data/", help="output root folder")),
    (("--name",), dict(type=str, default="kuka_button", help="dataset folder name")),
    (("--env",), dict(type=str, default="KukaButtonGymEnv-v0", choices=list(registered_env.keys()))),
    (("--display",), dict(action="store_true", help="ignored (no GUI)")),
    (("--no-record-data",), dict(action="store_true")),
    (("--max-distance",), dict(type=float, default=0.28, help="negative reward beyond this distance to the goal")),
    (("-c", "--continuous-actions"), dict(action="store_true")),
    (("--seed",), dict(type=int, default=0)),
    (("-f", "--force"), dict(action="store_true", help="overwrite an existing dataset (and stale part folders)")),
    (("-r", "--random-target"), dict(action="store_true")),
    (("--multi-view",), dict(action="store_true")),
    (("--shape-reward",), dict(action="store_true")),
    (("--reward-dist",), dict(action="store_true", help="print the reward histogram at the end")),
    (("--run-ppo2",), dict(action="store_true", help="unsupported: needs stable-baselines")),
    (("--ppo2-timesteps",), dict(type=int, default=1000)),
    (("--toward-target-timesteps-proportion",), dict(type=float, default=0.0)),
    (("--device-id",), dict(type=int, default=0, help="HIP device ordinal")),
    (("--device-ids",), dict(type=str, default=None, help="GPUs the --num-cpu lanes are spread over: 'all' or e.g. 0,1,2,3 (default: --device-id only)")),
    (("--img-size",), dict(type=int, default=224, help="recorded frame size (reference: 224)")),
]


def build_parser():
    parser = argparse.ArgumentParser(description="Deterministic dataset generator for SRL training (batched on the MI355Xs of one node)")
    for flags, kw in _FLAGS:
        parser.add_argument(*flags, **kw)
    return parser


def merge_parts(args):
    """Fuse <name>_part-T folders into <name>/ with globally renumbered record_XXX folders and concatenated
    npz arrays — the layout the reference leaves behind (:203-263)."""
    root = args.save_path + args.name
    parts = sorted(glob.glob(root + "_part-[0-9]*"), key=lambda a: int(a.rsplit("-", 1)[1]))
    for cfg_file in ("dataset_config.json", "env_globals.json"):
        os.rename(os.path.join(parts[0], cfg_file), os.path.join(root, cfg_file))
    merged = {"ground_truth.npz": {}, "preprocessed_data.npz": {}}
    next_record = 0
    for part in parts:
        first = next_record
        for record in sorted(glob.glob(part + "/record_[0-9]*"), key=lambda a: int(a.rsplit("_", 1)[1])):
            os.renames(record, "{}/record_{:03d}".format(root, next_record))
            next_record += 1
        for fname, acc in merged.items():
            with np.load(os.path.join(part, fname)) as data:
                for key in data.files:
                    arr = data[key]
                    if key == "images_path":
                        arr = np.array([convertImagePath(args, path, first) for path in arr])
                    acc.setdefault(key, []).append(arr)
        shutil.rmtree(part)
    for fname, acc in merged.items():
        np.savez(os.path.join(root, fname), **{k: np.concatenate(v) for k, v in acc.items()})


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert args.num_cpu > 0, "Error: number of cpu must be positive and non zero"
    assert args.max_distance > 0, "Error: max distance must be positive and non zero"
    assert args.num_episode > 0, "Error: number of episodes must be positive and non zero"
    assert not (args.reward_dist and args.shape_reward), "Error: cannot display the reward distribution for continuous reward"
    assert not (registered_env[args.env][3] is ThreadingType.NONE and args.num_cpu != 1), \
        "Error: cannot have more than 1 CPU for the environment {}".format(args.env)
    assert not args.run_ppo2, "Error: --run-ppo2 needs stable-baselines, which this build does not ship"
    if args.num_cpu > args.num_episode:
        args.num_cpu = args.num_episode
        print("num_cpu cannot be greater than num_episode, defaulting to {} cpus.".format(args.num_cpu))
    # seeds 0 and 1 must give unrelated datasets, not shifted copies (:166)
    args.seed = np.random.RandomState(args.seed).randint(int(1e10))
    root = args.save_path + args.name
    record = not args.no_record_data
    if record and os.path.exists(root):
        assert args.force, "Error: save directory '{}' already exists".format(root)
        for stale in [root] + glob.glob(root + "_part-[0-9]*"):
            shutil.rmtree(stale)
    if record:
        os.makedirs(root)
    run_batched(args)
    if record and args.num_cpu > 1:
        merge_parts(args)
    if args.reward_dist:
        rewards, counts = np.unique(np.load(root + "/preprocessed_data.npz")["rewards"], return_counts=True)
        print("reward distribution:")
        for reward, frac in zip(rewards, counts / counts.sum()):
            print(" ", reward, "{:.2f}%".format(frac * 100))


if __name__ == '__main__':
    main()
