"""rl_baselines/random_agent.py:28-42 — the random-agent loop the headline metric times."""
import time


class RandomAgentModel(object):
    def customArguments(self, parser):
        parser.add_argument('--num-cpu', help='Number of envs (lanes of the GPU handles)', type=int, default=1)
        parser.add_argument('--device-ids', help="GPUs the envs are sharded over: 'all' or e.g. 0,1,2,3 (default: device 0)", type=str, default=None)
        parser.add_argument('--persistent', action='store_true', default=False,
                            help='no kernel launch per step: a resident kernel per GPU takes the steps through mapped memory (KukaButtonGymEnv, ground truth)')
        return parser

    def makeEnv(self, args, env_kwargs=None, load_path_normalise=None):
        from rl_baselines.utils import createEnvs
        return createEnvs(args, env_kwargs=env_kwargs, load_path_normalise=load_path_normalise)

    def train(self, args, callback=None, env_kwargs=None, train_kwargs=None):
        env = self.makeEnv(args, env_kwargs=env_kwargs)
        obs = env.reset()
        num_updates = int(args.num_timesteps) // args.num_cpu
        start_time = time.time()
        fps = 0.0
        for step in range(num_updates):
            actions = [env.action_space.sample() for _ in range(args.num_cpu)]
            obs, reward, done, info = env.step(actions)
            if callback is not None:
                callback(locals(), globals())
            if (step + 1) % 500 == 0:
                total_steps = step * args.num_cpu
                fps = total_steps / (time.time() - start_time)
                print("{} steps - {:.2f} FPS".format(total_steps, fps))
        env.close()
        return fps
