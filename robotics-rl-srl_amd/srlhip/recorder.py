"""EpisodeSaver-compatible recorder (state_representation/episode_saver.py:13-162):
same folder layout and npz keys that srl_zoo trains from —
  <path>/<name>/record_XXX/frameNNNNNN.jpg, preprocessed_data.npz{rewards, actions, episode_starts},
  ground_truth.npz{target_positions, ground_truth_states, images_path}, dataset_config.json, env_globals.json.
JPEGs are written with Pillow (cv2 is not installed); like the reference (which runs
cv2.COLOR_BGR2RGB on an RGB array before cv2.imwrite) the stored file is the RGB frame."""
import json
import os

import numpy as np


def _json_safe(d):
    out = {}
    for k, v in (d or {}).items():
        if isinstance(k, str) and (v is None or isinstance(v, (bool, int, float, str))):
            out[k] = v
    return out


class EpisodeSaver(object):
    def __init__(self, name, max_dist, state_dim=-1, globals_=None, learn_every=3, learn_states=False, path='data/',
                 relative_pos=False):
        if learn_states:
            raise NotImplementedError("online SRL training over zmq (state_representation/client.py) is out of scope")
        self.name, self.path = name, path
        self.data_folder = path + name
        os.makedirs(self.data_folder, exist_ok=True)
        self.actions, self.rewards, self.target_positions = [], [], []
        self.episode_starts, self.ground_truth_states, self.images_path = [], [], []
        self.episode_step, self.episode_idx, self.episode_folder = 0, -1, None
        self.episode_success = False
        self.state_dim, self.n_steps = state_dim, 0
        with open("{}/dataset_config.json".format(self.data_folder), "w") as f:
            json.dump({'relative_pos': relative_pos, 'max_dist': str(max_dist)}, f)
        if globals_ is not None:
            with open("{}/env_globals.json".format(self.data_folder), "w") as f:
                json.dump(_json_safe(globals_), f)

    def saveImage(self, observation):
        image_path = "{}/{}/frame{:06d}".format(self.data_folder, self.episode_folder, self.episode_step)
        self.images_path.append("{}/{}/frame{:06d}".format(self.name, self.episode_folder, self.episode_step))
        if observation is None or np.size(observation) == 0:
            return                                  # ground-truth-only recording (no rasteriser output)
        from PIL import Image
        observation = np.asarray(observation, dtype=np.uint8)
        if observation.shape[2] > 3:
            Image.fromarray(observation[:, :, :3]).save("{}_1.jpg".format(image_path), quality=95)
            Image.fromarray(observation[:, :, 3:]).save("{}_2.jpg".format(image_path), quality=95)
        else:
            Image.fromarray(observation).save("{}.jpg".format(image_path), quality=95)

    def reset(self, observation, target_pos, ground_truth):
        if len(self.episode_starts) == 0 or self.episode_starts[-1] is False:
            self.episode_idx += 1
            self.episode_step = 0
            self.episode_success = False
            self.episode_folder = "record_{:03d}".format(self.episode_idx)
            os.makedirs("{}/{}".format(self.data_folder, self.episode_folder), exist_ok=True)
            self.episode_starts.append(True)
            self.target_positions.append(np.array(target_pos))
            self.ground_truth_states.append(np.array(ground_truth))
            self.saveImage(observation)

    def step(self, observation, action, reward, done, ground_truth_state):
        self.episode_step += 1
        self.n_steps += 1
        self.rewards.append(reward)
        self.actions.append(action)
        if reward > 0:
            self.episode_success = True
        if not done:
            self.episode_starts.append(False)
            self.ground_truth_states.append(np.array(ground_truth_state))
            self.saveImage(observation)
        else:
            self.save()

    def save(self):
        assert len(self.actions) == len(self.rewards) == len(self.episode_starts) == len(self.images_path)
        assert len(self.actions) == len(self.ground_truth_states)
        assert len(self.target_positions) == self.episode_idx + 1
        np.savez('{}/preprocessed_data.npz'.format(self.data_folder), rewards=np.array(self.rewards),
                 actions=np.array(self.actions), episode_starts=np.array(self.episode_starts))
        np.savez('{}/ground_truth.npz'.format(self.data_folder), target_positions=np.array(self.target_positions),
                 ground_truth_states=np.array(self.ground_truth_states), images_path=np.array(self.images_path))
