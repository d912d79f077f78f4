"""CPU reference-path stand-in for BASELINE config 1 — TEST INFRASTRUCTURE / bench cpu_baseline ONLY.

Times the CPU oracle (physics only, no 224x224 rendering -> a GENEROUS stand-in: the real reference is slower)
under an emulation of stable-baselines' SubprocVecEnv protocol, which is how the reference runs N envs
(rl_baselines/utils.py:216-220): one worker process per env, multiprocessing.Pipe, one pickled message per env
per step, auto-reset in the worker, driven by the random-agent loop of rl_baselines/random_agent.py:35-42."""
import multiprocessing as mp
import os
import time

import numpy as np


def _worker(conn, kind, seed):
    from oracle import mobile_oracle
    env = mobile_oracle.MobileOracleEnv(kind)
    env.seed(seed)
    while True:
        cmd, data = conn.recv()
        if cmd == "step":
            obs, reward, done = env.step(data)
            if done:
                obs = env.reset()
            conn.send((obs, reward, done, {}))
        elif cmd == "reset":
            conn.send(env.reset())
        else:
            conn.close()
            return


def mobile_subproc_fps(num_cpu=4, n_steps=2048, warmup=256, kind=0, seed0=0):
    ctx = mp.get_context("fork")
    pipes, procs = [], []
    for i in range(num_cpu):
        parent, child = ctx.Pipe()
        p = ctx.Process(target=_worker, args=(child, kind, seed0 + i), daemon=True)
        p.start()
        child.close()
        pipes.append(parent)
        procs.append(p)
    for c in pipes:
        c.send(("reset", None))
    [c.recv() for c in pipes]
    rng = np.random.RandomState(0)

    def run(k):
        for _ in range(k):
            actions = rng.randint(4, size=num_cpu)
            for c, a in zip(pipes, actions):
                c.send(("step", int(a)))
            [c.recv() for c in pipes]
    run(warmup)
    t0 = time.perf_counter()
    run(n_steps)
    dt = time.perf_counter() - t0
    for c in pipes:
        c.send(("close", None))
    for p in procs:
        p.join(timeout=5)
    return {"value": n_steps * num_cpu / dt, "unit": "env-steps/s", "cores": num_cpu, "kind": "port",
            "sample": "oracle/mobile_oracle.py under a SubprocVecEnv-protocol emulation: {} worker processes, Pipe + pickle "
                      "per step, {} steps after {} warm-up, physics only (no rendering); host has {} logical cores".format(
                          num_cpu, n_steps, warmup, os.cpu_count())}


def _kuka_worker(conn, seed):
    from oracle import kuka_clib
    env = kuka_clib.OracleEnv(seed)
    while True:
        cmd, data = conn.recv()
        if cmd == "step":
            obs, reward, done = env.step(data)
            if done:
                obs = env.reset()
            conn.send((obs, reward, done, {}))
        elif cmd == "reset":
            conn.send(env.reset())
        else:
            env.close()
            conn.close()
            return


def kuka_subproc_fps(num_cpu=None, n_steps=None, warmup=16, seed0=0, budget_s=10.0):
    """KukaButtonGymEnv-v0 (ground_truth obs) behind the SubprocVecEnv protocol, random agent
    (rl_baselines/random_agent.py:35-42): one worker process per env, num_cpu = os.cpu_count() by default."""
    from oracle import clib
    clib.build()
    num_cpu = num_cpu or os.cpu_count() or 1
    ctx = mp.get_context("fork")
    pipes, procs = [], []
    for i in range(num_cpu):
        parent, child = ctx.Pipe()
        p = ctx.Process(target=_kuka_worker, args=(child, seed0 + i), daemon=True)
        p.start()
        child.close()
        pipes.append(parent)
        procs.append(p)
    for c in pipes:
        c.send(("reset", None))
    [c.recv() for c in pipes]
    rng = np.random.RandomState(0)

    def one():
        actions = rng.randint(6, size=num_cpu)
        for c, a in zip(pipes, actions):
            c.send(("step", int(a)))
        return sum(1 for c in pipes if c.recv()[2])

    for _ in range(warmup):
        one()
    t0 = time.perf_counter()
    steps, ends = 0, 0
    while (n_steps is None and time.perf_counter() - t0 < budget_s) or (n_steps is not None and steps < n_steps):
        ends += one()
        steps += 1
    dt = time.perf_counter() - t0
    for c in pipes:
        c.send(("close", None))
    for p in procs:
        p.join(timeout=5)
    return {"value": steps * num_cpu / dt, "unit": "env-steps/s", "cores": num_cpu, "kind": "port",
            "sample": "oracle/kuka_oracle.c behind a SubprocVecEnv-protocol emulation: {} worker processes (one env each), Pipe + "
                      "pickle per step, {} VecEnv steps after {} warm-up, {} episode ends each followed by the literal 505-step "
                      "reset in the worker, physics only (no rendering); host has {} logical cores".format(
                          num_cpu, steps, warmup, ends, os.cpu_count())}


if __name__ == "__main__":
    print(mobile_subproc_fps())
