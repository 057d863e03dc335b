#!/usr/bin/env python3
"""Evaluation driver: loads a checkpoint written by train_synth.py (or by the reference's train.py:
the file format is the same) and runs episodes with the call sequence of the reference's test.py
(test.py:84-105): select_action -> env.step until done, buffer.clear() after every episode."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="CartPole-v1")
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--checkpoint-dir", default="PPO_preTrained")
    ap.add_argument("--random-seed", type=int, default=0)
    ap.add_argument("--run-num-pretrained", type=int, default=0)
    ap.add_argument("--episodes", type=int, default=10)
    ap.add_argument("--max-ep-len", type=int, default=400)
    ap.add_argument("--action-std", type=float, default=0.1)
    ap.add_argument("--quiet", action="store_true")
    args = ap.parse_args(argv)
    from PPO import PPO
    from ppo_amd import envs

    env = envs.make(args.env, n_envs=1)
    continuous = env.action_space.n is None
    state_dim = env.observation_space.shape[0]
    action_dim = env.action_space.shape[0] if continuous else env.action_space.n
    agent = PPO(state_dim, action_dim, 0.0003, 0.001, 0.99, 80, 0.2, continuous,
                args.action_std if continuous else None)
    path = args.checkpoint or os.path.join(
        args.checkpoint_dir, args.env, f"PPO_{args.env}_{args.random_seed}_{args.run_num_pretrained}.pth")
    if not args.quiet:
        print("loading network from : " + path)
    agent.load(path)
    total = 0.0
    for ep in range(1, args.episodes + 1):
        ep_reward = 0.0
        state = env.reset()
        for _ in range(args.max_ep_len):
            action = agent.select_action(state)
            state, reward, done, _ = env.step(action)
            ep_reward += reward
            if done:
                break
        agent.buffer.clear()
        total += ep_reward
        if not args.quiet:
            print(f"Episode: {ep} \t\t Reward: {round(ep_reward, 2)}")
    env.close()
    avg = total / args.episodes
    if not args.quiet:
        print("average test reward : " + str(round(avg, 2)))
    return avg


if __name__ == "__main__":
    test()
