"""Definition of the humanoid environment — MI355X-native.

Keeps the surface of the reference's `minppo/env.py` (`HumanoidEnv(config)` with `reset`, `step`, `compute_reward`,
`is_done`, `get_obs`, `initial_qpos`, `actuator_ctrlrange`, `reward_config`, `observation_size`, `action_size`, `dt`;
`EnvState` / `EnvMetrics` records; `load_mjcf_model`), batched over environments: where the reference vmaps a
single-env function (`train.py:136,140`), every method here takes and returns `[N, ...]` torch CUDA tensors and runs the
cooperative rigid-body kernel (`csrc/k_physics.hip`) through the C ABI (`mppo_env_reset` / `mppo_env_step`).

`rng` arguments are accepted and ignored: `reset_noise_scale` is 0.0 in the reference (`env.py:87`), so its reset is
deterministic and its step keys are unused (`env.py:117-119`).
The debug viewer (`env.py:264-333`) is out of scope (needs the MuJoCo renderer and ffmpeg); `main` says so.
"""

from __future__ import annotations

import ctypes as C
import logging
from typing import Any, NamedTuple, Optional, Sequence

import numpy as np

from minppo_amd import _native as nat
from minppo_amd.config import Config, load_config_from_cli
from minppo_amd.model import CompiledModel, load_model
from minppo_amd.train import resolve_model, reward_cfg

logger = logging.getLogger(__name__)


def load_mjcf_model(kscale_id: str) -> CompiledModel:
    """Stands where the reference downloads and patches the MJCF (`env.py:27-50`): resolves the id (or a built-in
    model name / MJCF path) through the local robot table (minppo_amd/model.py)."""
    return load_model(kscale_id)


class EnvMetrics(NamedTuple):
    episode_returns: Any
    episode_lengths: Any
    returned_episode_returns: Any
    returned_episode_lengths: Any
    timestep: Any
    returned_episode: Any


class EnvState(NamedTuple):
    pipeline_state: Any  # [N, rec_dim] float32 record (layout: include/minppo_hip.h)
    obs: Any
    reward: Any
    done: Any
    metrics: EnvMetrics


class HumanoidEnv:
    """Batched humanoid environment on one GPU."""

    reset_noise_scale: float = 0.0

    def __init__(self, config: Config, *, lib: Optional[nat.Lib] = None, device: Any = "cuda:0") -> None:
        import torch

        self.torch = torch
        self.lib = lib if lib is not None else nat.load()
        self.device = torch.device(device)
        self._include_c_vals = config.environment.include_c_vals
        self._kscale_id = config.kscale_id
        self.cm = resolve_model(config)
        self._n_frames = config.environment.n_frames
        self._blob_host = np.frombuffer(self.cm.to_blob(self._include_c_vals), np.uint8).copy()
        self._blob_dev = torch.from_numpy(self._blob_host.copy()).to(self.device)
        self._model = C.c_void_p()
        with torch.cuda.device(self.device):
            self.lib.model_open(self._blob_host.ctypes.data, self._blob_host.size, self._blob_dev.data_ptr(), C.byref(self._model))
        if config.environment.jit_kernel:
            from minppo_amd import jit

            with torch.cuda.device(self.device):
                jit.specialize(self.lib, self._model, self.cm)
        self.dims = nat.ModelDims()
        self.lib.model_get_dims(self._model, C.byref(self.dims))
        self._action_size = self.cm.nu
        self.initial_qpos = torch.tensor(self.cm.t["qpos0"], dtype=torch.float32, device=self.device)
        self.reward_config = config.reward
        self._rc = reward_cfg(config)
        # "Currently unused" in the reference as well (env.py:107-113)
        self.actuator_ctrlrange = torch.tensor(self.cm.t["act_ctrlrange"], dtype=torch.float32, device=self.device)
        self._reset_rec = torch.zeros(self.dims.rec_dim, dtype=torch.float32, device=self.device)
        self._have_reset = False

    # -- PipelineEnv attributes ------------------------------------------------
    @property
    def observation_size(self) -> int:
        return self.dims.obs_dim

    @property
    def action_size(self) -> int:
        return self._action_size

    @property
    def dt(self) -> float:
        return float(self.dims.timestep) * self._n_frames

    def _stream(self) -> int:
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _metrics(self, N: int) -> EnvMetrics:
        t = self.torch
        z = lambda dt: t.zeros(N, dtype=dt, device=self.device)
        return EnvMetrics(z(t.float32), z(t.int32), z(t.float32), z(t.int32), z(t.int32), z(t.uint8))

    @staticmethod
    def _mstruct(m: EnvMetrics) -> nat.EnvMetrics:
        return nat.EnvMetrics(*[x.data_ptr() for x in m])

    # -- env.py:124-145 ---------------------------------------------------------
    def reset(self, rng: Any = None, num_envs: int = 1) -> EnvState:
        t = self.torch
        N = int(num_envs)
        state = t.empty(N, self.dims.rec_dim, dtype=t.float32, device=self.device)
        obs = t.empty(N, self.dims.obs_pad, dtype=t.float32, device=self.device)
        reward, done = t.empty(N, dtype=t.float32, device=self.device), t.empty(N, dtype=t.uint8, device=self.device)
        m = self._metrics(N)
        ms = self._mstruct(m)
        self.lib.env_reset(self._model, N, state.data_ptr(), self._reset_rec.data_ptr(), obs.data_ptr(), self.dims.obs_pad, reward.data_ptr(),
                           done.data_ptr(), C.byref(ms), self._stream())
        self._have_reset = True
        return EnvState(state, obs[:, :self.observation_size], reward, done.bool(), m)

    # -- env.py:148-196 -----------------------------------------------------------
    def step(self, env_state: EnvState, action: Any, rng: Any = None) -> EnvState:
        t = self.torch
        if not self._have_reset:
            self.reset(num_envs=1)
        state = env_state.pipeline_state.clone()
        N = state.shape[0]
        action = action.to(device=self.device, dtype=t.float32).contiguous()
        if action.shape != (N, self._action_size):
            raise ValueError(f"action must have shape ({N}, {self._action_size}), got {tuple(action.shape)}")
        obs = t.empty(N, self.dims.obs_pad, dtype=t.float32, device=self.device)
        reward, done = t.empty(N, dtype=t.float32, device=self.device), t.empty(N, dtype=t.uint8, device=self.device)
        m = EnvMetrics(*[x.clone() if x.dtype != t.bool else x.to(t.uint8) for x in env_state.metrics])
        ms = self._mstruct(m)
        self.lib.env_step(self._model, N, self._n_frames, C.byref(self._rc), state.data_ptr(), self._reset_rec.data_ptr(), action.data_ptr(),
                          self._action_size, obs.data_ptr(), self.dims.obs_pad, reward.data_ptr(), done.data_ptr(), C.byref(ms), self._stream())
        return EnvState(state, obs[:, :self.observation_size], reward, done.bool(), m)

    # -- env.py:245-261 / 238-242 / 199-235 on the state record (host-side views, for inspection) ------
    def get_obs(self, data: Any, action: Any = None) -> Any:
        return data[:, :self.observation_size]

    def is_done(self, state: Any) -> Any:
        z = state[:, 2]
        r = self.reward_config
        return ~((r.height_min_z < z) & (z < r.height_max_z))

    def compute_reward(self, state: Any, next_state: Any, action: Any) -> Any:
        t, r = self.torch, self.reward_config
        nq, nv, OP = self.cm.nq, self.cm.nv, self.dims.obs_pad
        p0 = t.linalg.norm(self.initial_qpos[None] - state[:, :nq], dim=-1)
        pos_r = t.exp(-r.original_pos_reward_exp_coefficient * p0) - r.original_pos_reward_subtraction_factor * t.clamp(p0, 0, r.original_pos_reward_max_diff_norm)
        z = state[:, 2]
        healthy = t.where(z < r.height_min_z, 0.0, 1.0)
        healthy = t.where(z > r.height_max_z, t.zeros_like(healthy), healthy)
        ctrl = -(action * action).sum(-1)
        vel = (next_state[:, OP + nv] - state[:, OP + nv]) / self.dt
        return r.weights_ctrl_cost * ctrl + r.weights_original_pos_reward * pos_r + r.weights_velocity * vel + r.weights_is_healthy * healthy

    def close(self) -> None:
        if getattr(self, "_model", None):
            self.lib.model_close(self._model)
            self._model = None


def main(args: Sequence[str] | None = None) -> None:
    """The reference's `minppo env` renders a video with the MuJoCo renderer + ffmpeg (`env.py:264-333`); that viewer is
    outside the training hot path and is not provided.  This entry point runs a short random-action rollout instead."""
    import sys

    import torch

    logging.basicConfig(level=logging.INFO)
    config = load_config_from_cli(sys.argv[1:] if args is None else args)
    env = HumanoidEnv(config)
    logger.info("Initialized environment with action size %d (rendering is not available in this build)", env.action_size)
    es = env.reset(num_envs=4)
    total = torch.zeros(4, device=env.device)
    for _ in range(int(config.visualization.max_steps)):
        es = env.step(es, torch.rand(4, env.action_size, device=env.device))
        total += es.reward
    logger.info("random-action rollout: mean return %.3f over %d steps", float(total.mean()), int(config.visualization.max_steps))


if __name__ == "__main__":
    main()
