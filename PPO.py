"""Drop-in module name: `from PPO import PPO` (train.py:12, test.py:12, make_gif.py:13 of the
reference) resolves to the MI355X engine when this repository is first on sys.path."""
from ppo_amd.ppo import PPO, ActorCritic, RolloutBuffer, device  # noqa: F401
