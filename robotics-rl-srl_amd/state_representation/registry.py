"""state_representation/registry.py — which srl_model names are produced by the environment itself and which by a
learned encoder on top of raw pixels (same table as the reference; the env restriction lists name the Kuka family)."""
from state_representation import SRLType

KUKA_ENVS = ["KukaButtonGymEnv"]

# format NAME: (SRLType, LIMITED_TO_ENV)
registered_srl = {
    "raw_pixels": (SRLType.ENVIRONMENT, None),
    "ground_truth": (SRLType.ENVIRONMENT, None),
    "joints": (SRLType.ENVIRONMENT, KUKA_ENVS),
    "joints_position": (SRLType.ENVIRONMENT, KUKA_ENVS),
}
for _name in ("robotic_priors", "inverse", "forward", "multi_view_srl", "srl_combination", "supervised", "autoencoder",
              "autoencoder_inverse", "autoencoder_reward", "autoencoder_forward", "random", "random_inverse",
              "reward_inverse", "srl_splits", "srl_split_forward", "srl_3_splits", "reward", "vae", "dae", "pca"):
    registered_srl[_name] = (SRLType.SRL, None)
