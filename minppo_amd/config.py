"""Typed configuration tree for the MI355X-native minppo engine.

Mirrors the configuration *surface* of the reference (field names, defaults,
YAML + dot-list override grammar; reference `minppo/config.py:16-127`) without
OmegaConf, which is not available on the target image.  The merge order is the
reference's: structured defaults  <-  YAML file  <-  `a.b=c` dot-list overrides
(`minppo/config.py:120-123`).  The first positional argument is a path or a
bare config name resolved under `configs/` (`minppo/config.py:113-118`).

This module is host-side control plane: it is not accelerated and has no GPU
dependency.
"""

from __future__ import annotations

import dataclasses
import logging
import sys
from dataclasses import dataclass, field, fields, is_dataclass
from pathlib import Path
from typing import Any, Sequence

import yaml

logger = logging.getLogger(__name__)

CONFIG_ROOT_DIR = Path(__file__).parent / "configs"


class _Missing:
    """Sentinel for mandatory values (the reference uses `omegaconf.MISSING`)."""

    def __repr__(self) -> str:  # pragma: no cover - cosmetic
        return "???"

    def __bool__(self) -> bool:
        return False


MISSING: Any = _Missing()


class MissingMandatoryValue(ValueError):
    """Raised when a field that has no default is read before being set."""


@dataclass
class EnvironmentConfig:
    n_frames: int = field(default=1)
    backend: str = field(default="mjx")
    include_c_vals: bool = field(default=True)
    # Extension (not in the reference): name of a built-in robot model or path to
    # an MJCF file.  The reference downloads the MJCF by `kscale_id`; there is no
    # network on the target, so the engine resolves `kscale_id` through a local
    # table (minppo_amd/model.py) unless this is set.
    model: str = field(default="")
    # Extension: compile the environment kernel for THIS robot's dimensions at start-up (minppo_amd/jit.py: hipcc, about 20 s once,
    # cached) - the counterpart of the reference's jax.jit of its step function (env.py:123,147; train.py:306).  Bit-identical results (checked on the device
    # before use), 1.2x - 1.6x faster than the run-time-sized kernel; no effect for the robots the library was built for.
    jit_kernel: bool = field(default=False)


@dataclass
class VisualizationConfig:
    camera_name: str = field(default=MISSING)
    width: int = field(default=640)
    height: int = field(default=480)
    render_every: int = field(default=1)
    max_steps: int = field(default=1000)
    video_length: float = field(default=5.0)
    num_episodes: int = field(default=20)
    video_save_path: str = field(default="episode.mp4")


@dataclass
class RewardConfig:
    termination_height: float = field(default=-0.2)
    height_min_z: float = field(default=-0.2)
    height_max_z: float = field(default=2.0)
    is_healthy_reward: float = field(default=5)
    original_pos_reward_exp_coefficient: float = field(default=2)
    original_pos_reward_subtraction_factor: float = field(default=0.2)
    original_pos_reward_max_diff_norm: float = field(default=0.5)
    ctrl_cost_coefficient: float = field(default=0.1)
    weights_ctrl_cost: float = field(default=0.1)
    weights_original_pos_reward: float = field(default=4)
    weights_is_healthy: float = field(default=1)
    weights_velocity: float = field(default=1.25)


@dataclass
class ModelConfig:
    hidden_size: int = field(default=256)
    num_layers: int = field(default=2)
    use_tanh: bool = field(default=True)


@dataclass
class OptimizerConfig:
    lr: float = field(default=3e-4)
    max_grad_norm: float = field(default=0.5)


@dataclass
class ReinforcementLearningConfig:
    num_env_steps: int = field(default=10)
    gamma: float = field(default=0.99)
    gae_lambda: float = field(default=0.95)
    clip_eps: float = field(default=0.2)
    ent_coef: float = field(default=0.0)
    vf_coef: float = field(default=0.5)


@dataclass
class TrainingConfig:
    lr: float = field(default=3e-4)
    seed: int = field(default=1337)
    num_envs: int = field(default=2048)
    total_timesteps: int = field(default=1_000_000_000)
    num_minibatches: int = field(default=32)
    num_steps: int = field(default=10)
    update_epochs: int = field(default=4)
    anneal_lr: bool = field(default=True)
    model_save_path: str = field(default="trained_model.pkl")
    # Extension: MLP GEMM arithmetic. "f32" = exact f32 MFMA; "bf16" = bf16-in /
    # f32-accumulate MFMA (BASELINE config 4).  GAE and Adam are always f32.
    mlp_dtype: str = field(default="f32")
    rng_impl: str = field(default="philox")  # "philox": the engine's own streams; "threefry": jax.random-compatible streams following the reference's key plumbing (minppo_amd/jaxrng.py)
    # Extension (absent upstream): full-state checkpoints.  `checkpoint_path` is written every
    # `checkpoint_every` updates (0 = only at the end, "" = never); `resume_from` restores parameters, Adam moments,
    # step counters (LR schedule + RNG stream position), environment states and episode metrics before training.
    # Extension: `TrainOutput.metrics` in the REFERENCE's shape (train.py:283,287-289: the six `EnvMetrics` fields of every step of
    # every update, [num_updates, T, N] each) instead of the per-update device-side reductions.  Small runs only: every update then
    # synchronises and copies its [T, N] reward / done arrays out; refused above 1 GiB of history.
    keep_metrics_history: bool = field(default=False)
    checkpoint_path: str = field(default="")
    checkpoint_every: int = field(default=0)
    resume_from: str = field(default="")


@dataclass
class InferenceConfig:
    model_path: str = field(default=MISSING)


@dataclass
class Config:
    kscale_id: str = field(default=MISSING)
    environment: EnvironmentConfig = field(default_factory=EnvironmentConfig)
    visualization: VisualizationConfig = field(default_factory=VisualizationConfig)
    reward: RewardConfig = field(default_factory=RewardConfig)
    model: ModelConfig = field(default_factory=ModelConfig)
    opt: OptimizerConfig = field(default_factory=OptimizerConfig)
    rl: ReinforcementLearningConfig = field(default_factory=ReinforcementLearningConfig)
    training: TrainingConfig = field(default_factory=TrainingConfig)
    inference: InferenceConfig = field(default_factory=InferenceConfig)
    debug: bool = field(default=True)


# ---------------------------------------------------------------------------
# merge machinery (dataclass <- nested dict <- dot-list)
# ---------------------------------------------------------------------------


def _coerce(value: Any, typ: Any, path: str) -> Any:
    """Converts a YAML / dot-list scalar to the declared field type."""
    if isinstance(value, _Missing):
        return value
    if typ in (int, "int"):
        if isinstance(value, bool):
            raise ValueError(f"{path}: expected int, got bool")
        if isinstance(value, float):
            if value != int(value):
                raise ValueError(f"{path}: expected int, got {value!r}")
            return int(value)
        if isinstance(value, str):
            return int(float(value)) if ("e" in value.lower() or "." in value) else int(value.replace("_", ""))
        return int(value)
    if typ in (float, "float"):
        return float(value)
    if typ in (bool, "bool"):
        if isinstance(value, str):
            low = value.strip().lower()
            if low in ("true", "1", "yes", "on"):
                return True
            if low in ("false", "0", "no", "off"):
                return False
            raise ValueError(f"{path}: expected bool, got {value!r}")
        return bool(value)
    if typ in (str, "str"):
        return str(value)
    return value


def _merge_into(obj: Any, upd: dict, prefix: str = "") -> None:
    if not isinstance(upd, dict):
        raise ValueError(f"{prefix or '<root>'}: expected a mapping, got {type(upd).__name__}")
    known = {f.name: f for f in fields(obj)}
    for key, val in upd.items():
        path = f"{prefix}{key}"
        if key not in known:
            raise ValueError(f"Key '{path}' is not in the config schema")
        cur = getattr(obj, key)
        if is_dataclass(cur):
            _merge_into(cur, val if val is not None else {}, path + ".")
        else:
            setattr(obj, key, _coerce(val, known[key].type, path))


def _dotlist_to_dict(items: Sequence[str]) -> dict:
    out: dict = {}
    for item in items:
        if "=" not in item:
            raise ValueError(f"Override '{item}' is not of the form key.path=value")
        key, raw = item.split("=", 1)
        val = yaml.safe_load(raw) if raw != "" else ""
        node = out
        parts = key.strip().split(".")
        for p in parts[:-1]:
            node = node.setdefault(p, {})
            if not isinstance(node, dict):
                raise ValueError(f"Override '{item}' conflicts with an earlier scalar override")
        node[parts[-1]] = val
    return out


def to_dict(cfg: Any) -> dict:
    out = {}
    for f in fields(cfg):
        v = getattr(cfg, f.name)
        out[f.name] = to_dict(v) if is_dataclass(v) else ("???" if isinstance(v, _Missing) else v)
    return out


def to_yaml(cfg: Any) -> str:
    return yaml.safe_dump(to_dict(cfg), sort_keys=False)


def require(value: Any, name: str) -> Any:
    """Returns `value`, raising if it is still MISSING (OmegaConf raises on access)."""
    if isinstance(value, _Missing):
        raise MissingMandatoryValue(f"Missing mandatory value: {name}")
    return value


def make_config(yaml_dict: dict | None = None, overrides: Sequence[str] = ()) -> Config:
    cfg = Config()
    if yaml_dict:
        _merge_into(cfg, yaml_dict)
    if overrides:
        _merge_into(cfg, _dotlist_to_dict(overrides))
    return cfg


def load_config_from_cli(args: Sequence[str] | None = None) -> Config:
    """Same contract as the reference's `load_config_from_cli` (`minppo/config.py:106-127`)."""
    if args is None:
        args = sys.argv[1:]
    if len(args) < 1:
        raise ValueError("Usage: <config_name_or_path> (<additional_args> ...)")
    path, *other_args = args

    if Path(path).exists():
        with open(path, "r", encoding="utf-8") as f:
            raw = yaml.safe_load(f) or {}
    elif (config_path := CONFIG_ROOT_DIR / f"{path}.yaml").exists():
        with open(config_path, "r", encoding="utf-8") as f:
            raw = yaml.safe_load(f) or {}
    else:
        raise ValueError(f"Config file not found: {path}")

    cfg = make_config(raw, other_args)
    logger.info("Loaded config: %s", to_yaml(cfg))
    return cfg


def replace(cfg: Any, **changes: Any) -> Any:
    return dataclasses.replace(cfg, **changes)
