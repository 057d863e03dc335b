#!/usr/bin/env python3
"""Training driver for the MI355X PPO engine on its device-side environments.

Two modes, one agent object:

  --n-envs 1   the reference's protocol, step by step (train.py:168-238): one numpy state in, one
               action out, `buffer.rewards.append(float)`, `buffer.is_terminals.append(bool)`,
               `update()` every `update_timestep` steps, `decay_action_std` every
               `action_std_decay_freq` steps, CSV log `episode,timestep,reward`, checkpoint every
               `save_model_freq` steps.  This is the call sequence an unchanged train.py produces.

  --n-envs N   the same agent driven with N lockstep envs per call: states, actions, rewards and
               terminals are device tensors and never visit the host; `update()` runs after every
               `horizon` lockstep steps on the N x horizon samples.

Log files and checkpoints use the reference's names and formats (PPO_logs/<env>/PPO_<env>_log_<n>.csv,
PPO_preTrained/<env>/PPO_<env>_<seed>_<run>.pth), so plot_graph.py / test.py read them.
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--env", default="CartPole-v1")
    ap.add_argument("--n-envs", type=int, default=1)
    ap.add_argument("--horizon", type=int, default=128, help="lockstep steps per update (N > 1)")
    ap.add_argument("--max-ep-len", type=int, default=400)
    ap.add_argument("--max-training-timesteps", type=int, default=int(1e5))
    ap.add_argument("--update-timestep", type=int, default=None, help="single-env mode; default 4 * max_ep_len")
    ap.add_argument("--k-epochs", type=int, default=40)
    ap.add_argument("--eps-clip", type=float, default=0.2)
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--lr-actor", type=float, default=0.0003)
    ap.add_argument("--lr-critic", type=float, default=0.001)
    ap.add_argument("--action-std", type=float, default=0.6)
    ap.add_argument("--action-std-decay-rate", type=float, default=0.05)
    ap.add_argument("--min-action-std", type=float, default=0.1)
    ap.add_argument("--action-std-decay-freq", type=int, default=int(2.5e5))
    ap.add_argument("--random-seed", type=int, default=0)
    ap.add_argument("--run-num-pretrained", type=int, default=0)
    ap.add_argument("--log-dir", default="PPO_logs")
    ap.add_argument("--checkpoint-dir", default="PPO_preTrained")
    ap.add_argument("--save-model-freq", type=int, default=int(2e4))
    ap.add_argument("--inplace", action="store_true",
                    help="N > 1: let the env write rewards/terminals straight into the rollout buffer rows "
                         "(next_state_slot / act_inplace / claim_env_slots) instead of select_action + appends")
    ap.add_argument("--fused-rollout", action="store_true",
                    help="N > 1, CartPole: the whole rollout enqueued by one library call (ppo_cartpole_rollout)")
    ap.add_argument("--quiet", action="store_true")
    return ap.parse_args(argv)


def _paths(args):
    log_dir = os.path.join(args.log_dir, args.env)
    os.makedirs(log_dir, exist_ok=True)
    run_num = len([f for f in os.listdir(log_dir) if os.path.isfile(os.path.join(log_dir, f))])
    log_path = os.path.join(log_dir, f"PPO_{args.env}_log_{run_num}.csv")
    ck_dir = os.path.join(args.checkpoint_dir, args.env)
    os.makedirs(ck_dir, exist_ok=True)
    ck_path = os.path.join(ck_dir, f"PPO_{args.env}_{args.random_seed}_{args.run_num_pretrained}.pth")
    return log_path, ck_path


def train(args):
    import torch
    from PPO import PPO
    from ppo_amd import envs

    say = (lambda *a: None) if args.quiet else print
    env = envs.make(args.env, n_envs=args.n_envs)
    continuous = env.action_space.n is None
    state_dim = env.observation_space.shape[0]
    action_dim = env.action_space.shape[0] if continuous else env.action_space.n
    log_path, ck_path = _paths(args)
    say("training environment name : " + args.env, "| envs in lockstep :", args.n_envs)
    say("logging at : " + log_path)
    say("save checkpoint path : " + ck_path)
    if args.random_seed:
        torch.manual_seed(args.random_seed)
        env.seed(args.random_seed)

    agent = PPO(state_dim, action_dim, args.lr_actor, args.lr_critic, args.gamma, args.k_epochs,
                args.eps_clip, continuous, args.action_std if continuous else None)
    t_start = time.time()
    log_f = open(log_path, "w+")
    log_f.write("episode,timestep,reward\n")
    history = []          # (timestep, average episode reward) per log line

    if args.n_envs == 1:
        # ---------------------------------------------------------- the reference's own loop shape
        update_timestep = args.update_timestep or args.max_ep_len * 4
        print_freq, log_freq = args.max_ep_len * 4, args.max_ep_len * 2
        print_sum = print_n = log_sum = log_n = 0
        time_step = i_episode = 0
        while time_step <= args.max_training_timesteps:
            state = env.reset()
            ep_reward = 0.0
            for _ in range(args.max_ep_len):
                action = agent.select_action(state)
                state, reward, done, _ = env.step(action)
                agent.buffer.rewards.append(reward)
                agent.buffer.is_terminals.append(done)
                time_step += 1
                ep_reward += reward
                if time_step % update_timestep == 0:
                    agent.update()
                if continuous and time_step % args.action_std_decay_freq == 0:
                    agent.decay_action_std(args.action_std_decay_rate, args.min_action_std)
                if time_step % log_freq == 0 and log_n:
                    avg = round(log_sum / log_n, 4)
                    log_f.write(f"{i_episode},{time_step},{avg}\n")
                    log_f.flush()
                    history.append((time_step, avg))
                    log_sum = log_n = 0
                if time_step % print_freq == 0 and print_n:
                    say(f"Episode : {i_episode} \t\t Timestep : {time_step} \t\t "
                        f"Average Reward : {round(print_sum / print_n, 2)}")
                    print_sum = print_n = 0
                if time_step % args.save_model_freq == 0:
                    agent.save(ck_path)
                    say("model saved at : " + ck_path)
                if done:
                    break
            print_sum += ep_reward
            print_n += 1
            log_sum += ep_reward
            log_n += 1
            i_episode += 1
    else:
        # ---------------------------------------------------------- N envs in lockstep
        N, H = args.n_envs, args.horizon
        agent.buffer.reserve(N * H)
        obs = env.reset()
        time_step = i_episode = 0
        next_save = args.save_model_freq
        next_decay = args.action_std_decay_freq
        while time_step <= args.max_training_timesteps:
            if args.fused_rollout:
                agent.cartpole_rollout(env, H)
            elif args.inplace:
                for _ in range(H):
                    agent.next_state_slot(N).copy_(obs)
                    actions = agent.act_inplace(N)
                    r_row, d_row = agent.buffer.claim_env_slots(N)
                    obs = env.step(actions, obs_out=obs, reward_out=r_row, done_out=d_row)[0]
            else:
                for _ in range(H):
                    actions = agent.select_action(obs)             # [N] device tensor
                    obs, rewards, dones, _ = env.step(actions)
                    agent.buffer.rewards.append(rewards)
                    agent.buffer.is_terminals.append(dones)
            time_step += N * H
            agent.update()
            if continuous and time_step >= next_decay:
                agent.decay_action_std(args.action_std_decay_rate, args.min_action_std)
                next_decay += args.action_std_decay_freq
            if hasattr(env, "pop_episode_stats"):
                s, n = env.pop_episode_stats()
                if n:
                    i_episode += n
                    avg = round(s / n, 4)
                    log_f.write(f"{i_episode},{time_step},{avg}\n")
                    log_f.flush()
                    history.append((time_step, avg))
                    say(f"Episode : {i_episode} \t\t Timestep : {time_step} \t\t Average Reward : {avg}")
            if time_step >= next_save:
                agent.save(ck_path)
                say("model saved at : " + ck_path)
                next_save += args.save_model_freq
    log_f.close()
    env.close()
    agent.save(ck_path)
    say("Total training time  : ", round(time.time() - t_start, 2), "s")
    return agent, history, ck_path, log_path


if __name__ == "__main__":
    train(parse_args())
