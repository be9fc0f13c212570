"""Train a model with a specified environment module — MI355X-native engine.

Same surface as the reference's `minppo/train.py`: `make_train(config) -> train(rng) -> TrainOutput`,
`main(args)`, `save_model(params, filename)`, the `Memory / RunnerState / UpdateState / TrainOutput`
records.  What differs is what runs underneath: instead of one jitted JAX program
(`minppo/train.py:306,310`) every update is one call into libminppo_hip.so, which enqueues the
rollout (policy MLP on f32 MFMA -> sample -> cooperative rigid-body step) and the PPO update (GAE ->
E x M minibatches of forward / clipped-PPO loss / backward -> global-norm clip -> Adam) as HIP kernels,
captured into a hipGraph and replayed.  PyTorch only holds the device arena, the stream and (for
multi-GPU) the rendezvous used to hand the RCCL id around.

Deviations from the reference, all deliberate and documented in DESIGN.md:
  * `rng` is an integer seed (or a 2-word key whose words are folded into one); the random streams
    are the engine's Philox streams, not JAX threefry.
  * `TrainOutput.metrics` holds per-update device-side reductions by default; `training.keep_metrics_history=true`
    (small runs) returns what the reference returns: an `EnvMetrics` of six `[num_updates, T, N]` arrays
    (`train.py:283,287-289`).
"""

from __future__ import annotations

import ctypes as C
import json
import logging
import math
import os
import pickle
from pathlib import Path
import sys
import time
from typing import Any, Callable, Dict, List, NamedTuple, Optional, Sequence

import numpy as np

from minppo_amd import _native as nat
from minppo_amd.config import Config, load_config_from_cli, require
from minppo_amd.model import CompiledModel, load_model

logger = logging.getLogger(__name__)


class Memory(NamedTuple):
    done: Any
    action: Any
    value: Any
    reward: Any
    log_prob: Any
    obs: Any
    info: Any


class TrainState(NamedTuple):
    """Stands where `flax.training.train_state.TrainState` stands (`train.py:126-130`)."""

    step: int
    params: dict
    opt_state: dict


class RunnerState(NamedTuple):
    train_state: TrainState
    env_state: Any
    last_obs: Any
    rng: Any


class UpdateState(NamedTuple):
    train_state: TrainState
    mem_batch: "Memory"
    advantages: Any
    targets: Any
    rng: Any


class TrainOutput(NamedTuple):
    runner_state: RunnerState
    metrics: Any


def save_model(params: dict, filename: str) -> None:
    """Same contract as the reference (`train.py:86-89`): pickle of the nested parameter dict."""
    d = os.path.dirname(filename)
    if d:
        os.makedirs(d, exist_ok=True)
    with open(filename, "wb") as f:
        pickle.dump(params, f)


# ---------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------


def _tensor_names(O: int, A: int, H: int, L: int = 2):
    """(name, shape) of every tensor in flat order: `L` hidden layers per MLP (`model.num_layers`, reference train.py:79,82), layer
    i + 1 of the actor is a_w{i+1} / a_b{i+1}, the output layer a_w{L+1} / a_b{L+1}; then log_std; then the critic alike."""
    def mlp(pref, last):
        for i in range(L + 1):
            n_in, n_out = (O if i == 0 else H), (last if i == L else H)
            yield f"{pref}_w{i + 1}", (n_in, n_out)
            yield f"{pref}_b{i + 1}", (n_out,)
    return [*mlp("a", A), ("log_std", (A,)), *mlp("c", 1)]


def param_slices(O: int, A: int, H: int, L: int = 2):
    """name -> (offset, shape) of the flat parameter vector (include/minppo_hip.h): every tensor starts on a 16-byte boundary;
    the alignment words in between hold zeros.  Returns (slices, total length incl. alignment words)."""
    out, off = {}, 0
    for name, shape in _tensor_names(O, A, H, L):
        out[name] = (off, shape)
        off += int(np.prod(shape))
        off = (off + 3) & ~3
    return out, off


def _orthogonal(rng: np.random.Generator, n_in: int, n_out: int, scale: float) -> np.ndarray:
    rows, cols = max(n_in, n_out), min(n_in, n_out)
    q, r = np.linalg.qr(rng.standard_normal((rows, cols)))
    q *= np.sign(np.diag(r))[None, :]
    return (scale * (q.T if n_in < n_out else q)).astype(np.float32)


def init_flat_params(seed: int, O: int, A: int, H: int, L: int = 2) -> np.ndarray:
    """`ActorCritic.init` (`train.py:63,68,80,112`): orthogonal kernels (gain sqrt 2 hidden, 0.01 heads),
    zero biases, zero log_std.  The generator is NumPy's PCG64 seeded with `seed` (not JAX threefry)."""
    rng = np.random.default_rng(seed)
    sl, total = param_slices(O, A, H, L)
    flat = np.zeros(total, np.float32)
    g = math.sqrt(2.0)
    for pref in ("a", "c"):
        for i in range(L + 1):
            off, (n_in, n_out) = sl[f"{pref}_w{i + 1}"]
            flat[off:off + n_in * n_out] = _orthogonal(rng, n_in, n_out, 0.01 if i == L else g).reshape(-1)
    return flat


def flat_to_tree(flat: np.ndarray, O: int, A: int, H: int, L: int = 2) -> dict:
    """The nested dict the reference pickles (`train.py:314`; Flax naming)."""
    sl, _ = param_slices(O, A, H, L)

    def get(name):
        off, shape = sl[name]
        return np.array(flat[off:off + int(np.prod(shape))].reshape(shape))

    def mlp(p):
        return {f"Dense_{i}": {"kernel": get(f"{p}_w{i + 1}"), "bias": get(f"{p}_b{i + 1}")} for i in range(L + 1)}

    return {"params": {"MLP_0": mlp("a"), "log_std": get("log_std"), "MLP_1": mlp("c")}}


def tree_to_flat(tree: dict, O: int, A: int, H: int, L: int = 2) -> np.ndarray:
    sl, total = param_slices(O, A, H, L)
    flat = np.zeros(total, np.float32)
    t = tree["params"]

    def put(name, arr):
        off, shape = sl[name]
        arr = np.asarray(arr, np.float32)
        if arr.shape != shape:
            raise ValueError(f"{name}: expected shape {shape}, got {arr.shape}")
        flat[off:off + arr.size] = arr.reshape(-1)

    put("log_std", t["log_std"])
    for p, key in (("a", "MLP_0"), ("c", "MLP_1")):
        for i in range(L + 1):
            put(f"{p}_w{i + 1}", t[key][f"Dense_{i}"]["kernel"])
            put(f"{p}_b{i + 1}", t[key][f"Dense_{i}"]["bias"])
    return flat


# ---------------------------------------------------------------------------
# engine wrapper
# ---------------------------------------------------------------------------

_REGION_DTYPES = {
    "count": "int32", "done": "uint8", "perm": "int32", "adv_sums": "float64", "episode_lengths": "int32",
    "returned_episode_lengths": "int32", "timestep": "int32", "returned_episode": "uint8", "jax_rng": "int32",
}


def reward_cfg(config: Config) -> nat.RewardCfg:
    r = config.reward
    return nat.RewardCfg(r.height_min_z, r.height_max_z, r.original_pos_reward_exp_coefficient, r.original_pos_reward_subtraction_factor,
                         r.original_pos_reward_max_diff_norm, r.weights_ctrl_cost, r.weights_original_pos_reward, r.weights_is_healthy,
                         r.weights_velocity)


def resolve_model(config: Config) -> CompiledModel:
    # `environment.backend` goes to brax's PipelineEnv in the reference (env.py:102: "mjx", or one of brax's own pipelines); the engine is a
    # restatement of the MJX pipeline and nothing else - any other value used to be read and ignored
    if config.environment.backend != "mjx":
        raise ValueError(f"environment.backend={config.environment.backend!r}: only the MJX pipeline ('mjx', the reference's default) exists in this engine")
    name = config.environment.model or require(config.kscale_id, "kscale_id")
    return load_model(name)


class Trainer:
    """One rank of the engine: robot model, HBM arena, engine handle and tensor views of its regions."""

    def __init__(self, config: Config, *, lib: Optional[nat.Lib] = None, device: Any = None, rank: int = 0, world_size: int = 1,
                 seed: Optional[int] = None, use_graph: bool = True, external_random: bool = False, stream: Any = None,
                 num_envs_local: Optional[int] = None, xp: str = "torch"):
        self.config = config
        self.lib = lib if lib is not None else nat.load()
        self.xp = xp
        self.rank, self.world_size = rank, world_size
        tr, rl = config.training, config.rl
        if rl.num_env_steps != tr.num_steps:
            # the reference would fail at the reshape (`train.py:260`; quirk C-1)
            raise ValueError(f"rl.num_env_steps ({rl.num_env_steps}) must equal training.num_steps ({tr.num_steps})")
        if not 1 <= config.model.num_layers <= 4:
            raise ValueError(f"model.num_layers = {config.model.num_layers}: the MI355X engine lays out 1 to 4 hidden layers (reference default 2, config.py:53)")
        self.L = int(config.model.num_layers)
        if tr.num_envs % world_size != 0:
            raise ValueError(f"training.num_envs ({tr.num_envs}) must be divisible by the number of ranks ({world_size})")
        self.num_envs_global = tr.num_envs
        self.N = num_envs_local if num_envs_local is not None else tr.num_envs // world_size
        self.T, self.M, self.E = tr.num_steps, tr.num_minibatches, tr.update_epochs
        # train.py:93-94 (global quantities)
        self.num_updates = tr.total_timesteps // tr.num_steps // tr.num_envs
        self.minibatch_size = tr.num_envs * tr.num_steps // tr.num_minibatches
        if self.minibatch_size * tr.num_minibatches != tr.num_steps * tr.num_envs:
            raise ValueError("`batch_size` must be equal to `num_steps * num_envs`")  # train.py:254-255
        self.seed = tr.seed if seed is None else int(seed)

        self.cm = resolve_model(config)
        blob = np.frombuffer(self.cm.to_blob(config.environment.include_c_vals), np.uint8)
        self._blob_host = blob.copy()
        if xp == "torch":
            import torch

            self.torch = torch
            self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
            self._blob_dev = torch.from_numpy(self._blob_host.copy()).to(self.device)
            self.stream = stream if stream is not None else torch.cuda.Stream(device=self.device)
            self._stream_ptr = self.stream.cuda_stream
        else:  # numpy "device" memory: the CPU emulator build used by the test-suite
            self.torch = None
            self.device = "cpu-emulator"
            buf = np.zeros(blob.size + 256, np.uint8)
            o = (-buf.ctypes.data) % 256
            self._blob_dev = buf[o:o + blob.size]
            self._blob_dev[:] = blob
            self.stream, self._stream_ptr = None, None
        self._model = C.c_void_p()
        with self._device_guard():  # (a specialised kernel is checked against the run-time-sized one on the device when the model is opened)
            self.lib.model_open(self._blob_host.ctypes.data, self._blob_host.size, nat.ptr(self._blob_dev), C.byref(self._model))
        if config.environment.jit_kernel and xp == "torch":  # (before the engine sizes its arena: the kernel's LDS / global-memory needs follow from it)
            from minppo_amd import jit

            with self._device_guard():
                jit.specialize(self.lib, self._model, self.cm, verbose=rank == 0)
        kind = C.c_int32(0)
        self.lib.model_is_specialized(self._model, C.byref(kind))
        self.env_kernel = ("run-time-sized", "library instantiation for this robot", "compiled for this robot at start-up")[kind.value]
        self.dims = nat.ModelDims()
        self.lib.model_get_dims(self._model, C.byref(self.dims))
        self.O, self.OP, self.A, self.H = self.dims.obs_dim, self.dims.obs_pad, self.dims.nu, config.model.hidden_size
        self.net = nat.Net(self.O, self.OP, self.A, self.H, int(config.model.use_tanh), int(tr.mlp_dtype == "bf16"), self.L)
        lr = tr.lr if tr.anneal_lr else config.opt.lr  # train.py:101 vs :123 (quirk C-3)
        self.ecfg = nat.EngineCfg(
            num_envs=self.N, num_steps=self.T, num_minibatches=self.M, update_epochs=self.E, n_frames=config.environment.n_frames,
            num_updates=max(self.num_updates, 1), world_size=world_size, rank=rank, gamma=rl.gamma, gae_lambda=rl.gae_lambda,
            loss=nat.LossCfg(rl.clip_eps, rl.vf_coef, rl.ent_coef),
            adam=nat.AdamCfg(lr, config.opt.max_grad_norm, 0.9, 0.999, 1e-5, int(tr.anneal_lr), 0, 0),
            reward=reward_cfg(config), net=self.net, seed=self.seed, use_graph=int(use_graph), external_random=int(external_random),
            rng_impl=self._rng_impl(tr.rng_impl), reserved0=0)
        nbytes = C.c_size_t()
        self.lib.engine_arena_bytes(self._model, C.byref(self.ecfg), C.byref(nbytes))
        self.arena_bytes = nbytes.value
        if xp == "torch":
            self.arena = self.torch.zeros(self.arena_bytes + 256, dtype=self.torch.uint8, device=self.device)
            o = (-self.arena.data_ptr()) % 256
            self.arena = self.arena[o:o + self.arena_bytes]
        else:
            raw = np.zeros(self.arena_bytes + 256, np.uint8)
            o = (-raw.ctypes.data) % 256
            self.arena = raw[o:o + self.arena_bytes]
        self._engine = C.c_void_p()
        self.lib.engine_create(self._model, C.byref(self.ecfg), nat.ptr(self.arena), self.arena_bytes, C.byref(self._engine))
        self.P = int(self.lib.param_count(C.byref(self.net)))
        self.updates_done = 0
        self._prepared = False
        self._keep_hist, self._hist_carry, self._hist_last = False, None, None
        self.comm_note = "single rank"  # why init_comm chose the transport it returned (drivers print it)
        self.set_params_flat(init_flat_params(self.seed, self.O, self.A, self.H, self.L))

    @staticmethod
    def _rng_impl(name: str) -> int:
        if name not in ("philox", "threefry"):
            raise ValueError(f"training.rng_impl must be 'philox' or 'threefry', got {name!r}")
        return 1 if name == "threefry" else 0

    def _seed_jax_rng(self) -> None:
        """The runner's carried key as the reference derives it from PRNGKey(seed) (train.py:110,142,285): three splits, the
        first two consumed by parameter init and env reset, the second key of the third is the scan's rng."""
        from minppo_amd import jaxrng

        rng = jaxrng.prng_key(self.seed)
        rng = jaxrng.split(rng)[0]   # rng, _rng = split(rng)        network.init
        rng = jaxrng.split(rng)[0]   # rng, reset_rng = split(rng)   reset_fn
        runner = jaxrng.split(rng)[1]  # rng, _rng = split(rng); RunnerState(..., _rng)
        reg = self.region("jax_rng")
        vals = np.zeros(reg.shape[0], np.int32)
        vals[:2] = runner.view(np.int32)
        self._write_region(reg, vals)

    # -- arena views ----------------------------------------------------------
    def region(self, name: str, shape: Optional[Sequence[int]] = None):
        off, nb = C.c_size_t(), C.c_size_t()
        self.lib.engine_region(self._engine, name.encode(), C.byref(off), C.byref(nb))
        dt = _REGION_DTYPES.get(name, "float32")
        raw = self.arena[off.value:off.value + nb.value]
        if self.xp == "torch":
            v = raw.view(getattr(self.torch, dt))
        else:
            v = raw.view(np.dtype(dt))
        return v.reshape(*shape) if shape is not None else v

    def traj(self) -> Dict[str, Any]:
        T, N = self.T, self.N
        return dict(obs=self.region("obs", (T + 1, N, self.OP)), action=self.region("action", (T, N, self.A)),
                    value=self.region("value", (T, N)), reward=self.region("reward", (T, N)), log_prob=self.region("log_prob", (T, N)),
                    done=self.region("done", (T, N)), adv=self.region("adv", (T, N)), target=self.region("target", (T, N)),
                    last_val=self.region("last_val", (N,)))

    def _to_host(self, x) -> np.ndarray:
        return x.detach().cpu().numpy() if self.xp == "torch" else np.array(x)

    def _sync(self) -> None:
        if self.xp == "torch":
            self.stream.synchronize()

    def set_params_flat(self, flat: np.ndarray) -> None:
        flat = np.ascontiguousarray(flat, np.float32)
        if flat.size != self.P:
            raise ValueError(f"expected {self.P} parameters, got {flat.size}")
        dst = self.region("params")
        if self.xp == "torch":
            self._sync()
            dst.copy_(self.torch.from_numpy(flat))
            self.torch.cuda.synchronize(self.device)
        else:
            dst[:] = flat

    def params_flat(self) -> np.ndarray:
        self._sync()
        return self._to_host(self.region("params")).copy()

    # -- full-state checkpoint (the reference only pickles the final parameters, train.py:86-89,314) ------
    # Everything that is carried from one update to the next lives in these arena regions: parameters, Adam moments,
    # the device counters (optimizer step = LR-schedule position, update index = RNG stream position), per-env physics
    # records, episode bookkeeping, and RunnerState.last_obs (slot 0 of `obs`).  Restoring them reproduces the
    # uninterrupted run bit for bit (tests/test_train_surface.py).
    _CKPT_REGIONS = ("params", "adam_m", "adam_v", "count", "state", "episode_returns", "episode_lengths", "returned_episode_returns",
                     "returned_episode_lengths", "timestep", "returned_episode", "jax_rng")
    _CKPT_VERSION = 2  # 2: 16-byte-aligned flat parameter layout

    def _ckpt_meta(self) -> Dict[str, Any]:
        return dict(version=self._CKPT_VERSION, model=self.cm.name, num_envs=self.N, num_steps=self.T, obs_dim=self.O, act_dim=self.A, hidden=self.H,
                    params=self.P, rec_dim=int(self.dims.rec_dim), seed=int(self.seed), rank=self.rank, world_size=self.world_size,
                    rng_impl="threefry" if self.ecfg.rng_impl == 1 else "philox")

    def save_checkpoint(self, path: str) -> None:
        self._sync()
        arrays = {name: self._to_host(self.region(name)).copy() for name in self._CKPT_REGIONS}
        arrays["last_obs"] = self._to_host(self.region("obs", (self.T + 1, self.N, self.OP))[0]).copy()
        meta = dict(self._ckpt_meta(), updates_done=int(self.updates_done))
        p = Path(path)
        if p.parent != Path(""):
            p.parent.mkdir(parents=True, exist_ok=True)
        tmp = p.with_name(p.name + ".tmp")
        with open(tmp, "wb") as f:
            np.savez(f, __meta__=np.frombuffer(json.dumps(meta).encode("utf-8"), np.uint8), **arrays)  # JSON, not pickle: loading never executes code
        os.replace(tmp, p)  # a crash never leaves a truncated checkpoint under the final name

    def load_checkpoint(self, path: str) -> None:
        with np.load(path) as z:
            try:
                meta = json.loads(z["__meta__"].tobytes().decode("utf-8"))
            except (UnicodeDecodeError, ValueError) as exc:
                raise ValueError(f"checkpoint {path}: metadata is not JSON (written by an older version?)") from exc
            want = self._ckpt_meta()
            # the seed keys the engine's Philox streams: only the same seed continues the same noise / permutation sequence
            meta.setdefault("rng_impl", "philox")  # (files written before the field existed: the engine's own streams)
            # rng_impl: a philox checkpoint carries an all-zero "jax_rng" region; continued with threefry it would silently draw from key (0, 0)
            for k in ("version", "model", "num_envs", "num_steps", "obs_dim", "act_dim", "hidden", "params", "rec_dim", "world_size", "rank", "seed", "rng_impl"):
                if meta.get(k) != want[k]:
                    raise ValueError(f"checkpoint {path}: {k} = {meta.get(k)!r} does not match this run ({want[k]!r})")
            self.reset()  # (re)builds the constant reset record; everything else is overwritten below
            self._sync()
            for name in self._CKPT_REGIONS:
                self._write_region(self.region(name), z[name])
            self._write_region(self.region("obs", (self.T + 1, self.N, self.OP))[0], z["last_obs"])
        self.updates_done = int(meta["updates_done"])
        self._hist_carry = None

    def _write_region(self, dst, src: np.ndarray) -> None:
        if tuple(dst.shape) != tuple(src.shape):
            raise ValueError(f"checkpoint region shape {src.shape} does not match {tuple(dst.shape)}")
        if self.xp == "torch":
            dst.copy_(self.torch.from_numpy(np.ascontiguousarray(src)))
            self.torch.cuda.synchronize(self.device)
        else:
            dst[...] = src

    @property
    def params(self) -> dict:
        return flat_to_tree(self.params_flat(), self.O, self.A, self.H, self.L)

    # -- multi-GPU ---------------------------------------------------------------
    def init_comm(self, mode: Optional[str] = None) -> str:
        """Connects this rank's engine to the other ranks' (one process per GPU, environments sharded, gradients summed
        per optimizer step).  Two transports, `mode` or $MPPO_ALLREDUCE:

          "peer" (default)  the engine's own exchange through hipIpc-mapped buffers, fused into the weight-gradient and Adam
                            launches (csrc/peer.h); the 64-byte handles travel through torch.distributed.  Works for ranks on
                            different GPUs of a node and for ranks that share one GPU.
          "rccl"            ncclAllReduce on the compute stream; the 128-byte id travels through torch.distributed.

        "peer" falls back to "rccl" when any rank cannot set the exchange up (the ranks agree through an all-reduce).  Returns
        the transport in use ("none" for a single rank)."""
        mode = (mode or os.environ.get("MPPO_ALLREDUCE", "peer")).lower()
        if mode not in ("peer", "rccl"):
            raise ValueError(f"MPPO_ALLREDUCE / mode must be 'peer' or 'rccl', got {mode!r}")
        if self.world_size == 1:
            if os.environ.get("MPPO_FORCE_COMM") == "1":  # single-rank communicator: exercises the RCCL path on one GPU
                host = np.zeros(128, np.uint8)
                self.lib.comm_unique_id(host.ctypes.data)
                with self.torch.cuda.device(self.device):
                    self.lib.engine_comm_init(self._engine, host.ctypes.data)
                self.comm_note = "MPPO_FORCE_COMM=1: single-rank RCCL communicator"
                return "rccl"
            return "none"
        import torch
        import torch.distributed as dist

        self.comm_note = "requested (mode / MPPO_ALLREDUCE=rccl)"

        on_gpu = self.xp == "torch" and dist.get_backend() == "nccl"

        def all_ok(ok: bool) -> bool:
            t = torch.tensor([1 if ok else 0], dtype=torch.int32)
            if on_gpu:
                t = t.to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return bool(int(t.item()))

        device_ctx = self._device_guard

        if mode == "peer":
            handle, err = np.zeros(64, np.uint8), None
            try:
                with device_ctx():
                    self.lib.engine_peer_export(self._engine, handle.ctypes.data)
            except nat.NativeError as exc:
                err = exc
            if all_ok(err is None):
                mine = torch.from_numpy(handle)
                if on_gpu:
                    mine = mine.to(self.device)
                gathered = [torch.zeros_like(mine) for _ in range(self.world_size)]
                dist.all_gather(gathered, mine)
                handles = np.concatenate([g.cpu().numpy() for g in gathered]).astype(np.uint8)
                # do several ranks drive the same GPU (a one-GPU box)?  Then nothing larger than one wave may wait for a peer (csrc/peer.h)
                ids = [None] * self.world_size
                dist.all_gather_object(ids, self._device_identity())
                shared = len(set(ids)) < len(ids)
                try:
                    with device_ctx():
                        self.lib.engine_peer_connect(self._engine, handles.ctypes.data, int(shared))
                except nat.NativeError as exc:
                    err = exc
                if all_ok(err is None):
                    dist.barrier()  # every rank has mapped every buffer before anybody's first update writes a flag
                    # one known all-reduce through the mapped buffers: a mapping that "succeeded" but carries no stores is found
                    # here, in seconds, and not as a time-out inside the first update
                    ok = C.c_int32(0)
                    try:
                        with device_ctx():
                            self.lib.engine_peer_selftest(self._engine, self._stream_ptr, C.byref(ok))
                    except nat.NativeError as exc:
                        err = exc
                    if all_ok(err is None and ok.value == 1):
                        self.comm_note = ("hipIpc buffers mapped on every rank, connect-time self-test exact on every rank (one fully checked exchange + "
                                          f"{os.environ.get('MPPO_PEER_SOAK', '2000')} more, every element compared on the device)")
                        return "peer"
                    err = err or "the self-test all-reduce timed out or returned a wrong sum"
                with device_ctx():
                    self.lib.engine_peer_disable(self._engine)
                dist.barrier()  # nobody unmaps while a peer still reads
            logger.warning("peer-to-peer exchange unavailable on some rank (%s); falling back to RCCL", err)
            self.comm_note = f"fallback: peer-to-peer exchange unavailable on some rank ({err or 'this rank was fine'})"
        host = np.zeros(128, np.uint8)
        if self.rank == 0:
            self.lib.comm_unique_id(host.ctypes.data)
        idbuf = torch.from_numpy(host)
        if on_gpu:
            idbuf = idbuf.to(self.device)
        dist.broadcast(idbuf, src=0)
        host = idbuf.cpu().numpy().copy()
        with device_ctx():  # (emulator build: the "communicator" is a shared-memory segment between the rank processes, tests/emu)
            self.lib.engine_comm_init(self._engine, host.ctypes.data)
        return "rccl"

    def _device_guard(self):
        import contextlib
        return self.torch.cuda.device(self.device) if self.xp == "torch" else contextlib.nullcontext()

    def _device_identity(self) -> str:
        if self.xp != "torch":
            return f"emulator-rank-{self.rank}"  # (rank processes of a CPU test: nothing is shared)
        import socket

        p = self.torch.cuda.get_device_properties(self.device)
        # both the uuid and the PCI address: ranks count as sharing a GPU only if every identifier the runtime offers agrees
        ident = (str(getattr(p, "uuid", "")), getattr(p, "pci_domain_id", None), getattr(p, "pci_bus_id", None), getattr(p, "pci_device_id", None))
        return f"{socket.gethostname()}/{ident}"

    def comm_mode(self) -> str:
        out = C.c_int32(0)
        self.lib.engine_comm_mode(self._engine, C.byref(out))
        return ("none", "rccl", "peer", "peer", "peer")[out.value]

    def peer_form(self) -> str:
        """How the peer-to-peer exchange is launched: "fused" (three launches per optimizer step), "split", "shared" (ranks on one GPU)."""
        out = C.c_int32(0)
        self.lib.engine_comm_mode(self._engine, C.byref(out))
        return {2: "fused", 3: "split", 4: "shared"}.get(out.value, "")

    def peer_latencies(self, iters: int = 2000) -> Optional[List[float]]:
        """COLLECTIVE (peer transport, every rank): the one-way latency in microseconds of a system-scope flag between rank 0 and every other
        rank - `iters` round trips of one word through the two ranks' exchange buffers, timed on the device (mppo_engine_peer_latency).  What
        ONE dependent trip of the fused exchange costs on this machine: the `t_link` of the efficiency model, measured.
        Returns the list on every rank (entry q - 1 = rank 0 <-> rank q), None without a connected exchange."""
        if self.world_size < 2 or self.comm_mode() != "peer" or not self._dist_ready():
            return None
        import torch
        import torch.distributed as dist

        out = []
        for q in range(1, self.world_size):
            us = C.c_double(0.0)
            if self.rank in (0, q):
                with self._device_guard():
                    self.lib.engine_peer_latency(self._engine, q if self.rank == 0 else 0, int(iters), int(self.rank == 0), self._stream_ptr, C.byref(us))
            t = torch.tensor([us.value if self.rank == 0 else 0.0], dtype=torch.float64)
            if dist.get_backend() == "nccl":
                t = t.to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)  # (also the barrier between the pairs)
            out.append(float(t[0].item()))
        return out

    def check_replicas(self) -> None:
        """COLLECTIVE (every rank): the replicas of a data-parallel run must hold bit-identical parameters after the same updates - a slice of
        the gradient is reduced once, by its owner, and broadcast.  Compares a checksum of the parameter bits over the ranks and raises on
        every rank if they differ (a torn or reordered store in the exchange that the tags did not catch would show here, within
        `checkpoint_every` updates, and not only at the end of the run)."""
        if self.world_size < 2 or not self._dist_ready():
            return
        import torch
        import torch.distributed as dist

        bits = self.params_flat().view(np.uint32).astype(np.uint64)
        chk = torch.tensor([int(bits.sum() % (1 << 62)), int((bits * (np.arange(bits.size, dtype=np.uint64) % 65521 + 1)).sum() % (1 << 62))], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            chk = chk.to(self.device)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not bool((lo == hi).all().item()):
            raise RuntimeError(f"rank {self.rank}: the ranks' parameter replicas differ (checksums {lo.tolist()} .. {hi.tolist()}): the gradient exchange delivered "
                               "different bytes to different ranks - this run's results are invalid")

    def check_peers(self, collective: bool = True) -> None:
        """Synchronises the device and raises if a wait for a peer rank ran into its time limit (MPPO_PEER_TIMEOUT_MS): the kernels
        then ran to their end on whatever was in the exchange buffers and the parameters are invalid - on THIS rank and, because a
        rank whose wait gave up still raises its own flags, on every rank that consumed its contribution.  So the check is a
        collective (every rank must call it): the ranks take the maximum of their error words over `torch.distributed` and raise
        together; none of them goes on to write a checkpoint of parameters that a peer's time-out has spoilt.  `collective=False`
        looks at this rank's word only (drivers without a process group)."""
        out, info = C.c_int32(0), (C.c_int32 * 8)()
        self.lib.engine_peer_status(self._engine, C.byref(out), info)
        worst, where = int(out.value), self.rank
        if collective and self.world_size > 1 and self._dist_ready():
            import torch
            import torch.distributed as dist

            # ONE packed value (count << 16 | rank): the maximum then names the rank that holds the largest count, not the largest rank id
            # among all ranks that gave up (round-4 advisor)
            t = torch.tensor([(min(int(out.value), (1 << 40) - 1) << 16) | (self.rank & 0xFFFF) if out.value else 0], dtype=torch.int64)
            if dist.get_backend() == "nccl":  # (an nccl group reduces device tensors only)
                if self.xp != "torch":
                    raise RuntimeError("check_peers: an nccl process group needs device tensors, but this Trainer holds NumPy memory (xp != 'torch'); pass collective=False")
                t = t.to(self.device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            worst, where = int(t[0].item()) >> 16, int(t[0].item()) & 0xFFFF
        if out.value:
            kind = {1: "the local gradient of rank", 2: "the reduced piece", 3: "the advantage sums of rank"}.get(info[0], "?")
            raise RuntimeError(f"rank {self.rank}: a wait for {kind} {info[1]} timed out (epoch {info[2]}, flag read {info[3]}; this rank: {info[4]} optimizer steps, "
                               f"{info[5]} updates, {info[6]} open arrivals, {info[7]} pieces per slice, {out.value} waits gave up); this run's results are invalid")
        if worst:
            raise RuntimeError(f"rank {self.rank}: a wait for a peer timed out on rank {where} ({worst} waits gave up there); the gradients it contributed "
                               f"to are invalid on every rank, this run's results are invalid")

    def check_status(self) -> None:
        """Synchronises and raises if the engine recorded a condition that invalidates the run: today one - a bucket of the two-launch
        permutation (csrc/k_perm.hip) overflowed, i.e. an epoch's index array was not a permutation (`count[3]`, made sticky by the
        end-of-update kernel; short of a 22-sigma event unreachable for num_envs * num_steps <= 131072, and guarded all the same)."""
        self._sync()
        if int(self._to_host(self.region("count"))[3]) != 0:
            raise RuntimeError(f"rank {self.rank}: a bucket of the permutation kernel overflowed during an update (csrc/k_perm.hip); an epoch trained on an index "
                               "array that was not a permutation - this run's results are invalid")

    @staticmethod
    def _dist_ready() -> bool:
        try:
            import torch.distributed as dist
        except ImportError:
            return False
        return dist.is_available() and dist.is_initialized()

    def barrier(self) -> None:
        """Host-side meeting point of the ranks after rank-asymmetric host work (a checkpoint written to a slow disk): with the
        peer-to-peer exchange a rank that enters the next update much later than its peers makes THEIR waits run into the time limit."""
        if self.world_size > 1 and self._dist_ready():
            import torch.distributed as dist

            dist.barrier()

    # -- stepping ----------------------------------------------------------------
    def reset(self) -> None:
        self._hist_carry = None  # (the metric words change under the history's feet: re-read before the next update)
        self.lib.engine_reset(self._engine, self._stream_ptr)
        if self.ecfg.rng_impl == 1:
            self._sync()
            self._seed_jax_rng()

    def update(self) -> None:
        if self.world_size > 1 and not self._prepared:
            self.prepare()
        if self._keep_hist and self._hist_carry is None:
            self._hist_carry = self._metric_words()  # the environments' metric words as they stand BEFORE this rollout
        self.lib.engine_update(self._engine, self._stream_ptr)
        self.updates_done += 1
        if self._keep_hist:
            self._hist_last = self._replay_metrics()

    # -- the reference's per-step metrics (opt-in; `training.keep_metrics_history`) --------------------------------------------
    _METRIC_FIELDS = (("episode_returns", np.float32), ("episode_lengths", np.int32), ("returned_episode_returns", np.float32),
                      ("returned_episode_lengths", np.int32), ("timestep", np.int32), ("returned_episode", np.uint8))

    def keep_metrics_history(self, on: bool = True) -> None:
        """From the next `update()` on, every update also produces `metrics_history()`: the six `EnvMetrics` fields (`env.py:53-59`) of
        every step of its rollout, `[T, N]` each - what the reference stores as `Memory.info` (`train.py:170,172`) and returns stacked
        over the updates (`train.py:283,287-289`).  The engine keeps only the LATEST metric words per environment on the device (and
        their per-update reductions, `rollout_stats`), so the `[T, N]` history is the bookkeeping of `env.py:183-194` replayed on the host
        from the rollout's own `reward` / `done` arrays and the words as they stood before the rollout - the same float32 operations in
        the same order as `env_kernel`'s epilogue, and the replay's last step must EQUAL the device's words (checked on every update).
        Each update then synchronises: for small runs / reference-shaped callers, not for throughput."""
        self._keep_hist, self._hist_carry, self._hist_last = bool(on), None, None

    def _metric_words(self) -> Dict[str, np.ndarray]:
        self._sync()
        return {k: self._to_host(self.region(k)).astype(dt).copy() for k, dt in self._METRIC_FIELDS}

    def _replay_metrics(self) -> Dict[str, np.ndarray]:
        T, N = self.T, self.N
        self._sync()
        reward = self._to_host(self.region("reward", (T, N))).astype(np.float32)
        done = self._to_host(self.region("done", (T, N))).astype(np.uint8)
        c = self._hist_carry
        ret, ln, rret, rln, ts = (c["episode_returns"].copy(), c["episode_lengths"].copy(), c["returned_episode_returns"].copy(),
                                  c["returned_episode_lengths"].copy(), c["timestep"].copy())
        h = {k: np.zeros((T, N), dt) for k, dt in self._METRIC_FIELDS}
        for t in range(T):  # env.py:183-194, float32 like the kernel (csrc/k_physics.hip epilogue)
            d = done[t] != 0
            nd, ndi = np.where(d, np.float32(0), np.float32(1)), np.where(d, 0, 1).astype(np.int32)
            new_ret, new_len = ret + reward[t], ln + 1
            ret, ln = new_ret * nd, new_len * ndi
            rret = rret * nd + new_ret * np.where(d, np.float32(1), np.float32(0))
            rln = rln * ndi + new_len * np.where(d, 1, 0).astype(np.int32)
            ts = ts + 1
            for k, v in (("episode_returns", ret), ("episode_lengths", ln), ("returned_episode_returns", rret), ("returned_episode_lengths", rln),
                         ("timestep", ts), ("returned_episode", d.astype(np.uint8))):
                h[k][t] = v
        now = self._metric_words()
        for k, _ in self._METRIC_FIELDS:
            if not np.array_equal(h[k][T - 1], now[k]):
                bad = int(np.flatnonzero(h[k][T - 1] != now[k])[0])
                raise RuntimeError(f"metrics history: the replayed `{k}` of environment {bad} after the rollout ({h[k][T - 1][bad]!r}) is not the device's word "
                                   f"({now[k][bad]!r}); the history of this update is not what the engine computed")
        self._hist_carry = now
        return h

    def metrics_history(self):
        """`EnvMetrics` of `[T, N]` arrays: the last update's rollout (needs `keep_metrics_history()` before that update)."""
        from minppo_amd.env import EnvMetrics  # (env.py imports this module)

        if not self._keep_hist or self._hist_last is None:
            raise RuntimeError("metrics_history(): call keep_metrics_history() before the update (training.keep_metrics_history=true)")
        return EnvMetrics(**self._hist_last)

    def prepare(self) -> None:
        """Several ranks, before the first update: capture the update (hipGraph) on every rank, then meet at a barrier, so that the
        ranks enter their first gradient exchange together rather than a capture time apart (csrc/peer.h bounds every wait).  A
        driver that does not use `torch.distributed` (handles gathered some other way, `learn_host_driven`-style callers) gets the
        capture only and is responsible for its own barrier between `prepare()` and the first `update()`."""
        self.lib.engine_prepare(self._engine, self._stream_ptr)
        self._sync()
        if self._dist_ready():
            import torch.distributed as dist

            dist.barrier()
        else:
            logger.warning("Trainer.prepare: torch.distributed is not initialised - no barrier between the ranks' captures and their first update; "
                           "the caller must provide one (the peer-to-peer exchange bounds every wait by MPPO_PEER_TIMEOUT_MS)")
        self._prepared = True

    def graph_active(self) -> bool:
        """True once `update()` replays a captured hipGraph (false: eager launches)."""
        out = C.c_int32(0)
        self.lib.engine_graph_active(self._engine, C.byref(out))
        return bool(out.value)

    def rollout(self) -> None:
        self.lib.engine_rollout(self._engine, self._stream_ptr)

    def learn(self) -> None:
        self.lib.engine_learn(self._engine, self._stream_ptr)
        self.updates_done += 1

    def learn_host_driven(self, allreduce_sum=None) -> None:
        """The learn phase with the collectives supplied by the caller: `allreduce_sum(array_view)` must sum the
        view in place over ranks.  Same stage calls, same order and same weighting as the engine's own
        `mppo_engine_learn` (csrc/engine.hip: do_learn); it exists so that the sharding arithmetic can be
        exercised with torch.distributed/gloo where RCCL is not available, and as a reference for integrators
        who drive the stages themselves.  Requires external_random-style permutations already in "perm"."""
        E, M, mb, B, W = self.E, self.M, (self.T * self.N) // self.M, self.T * self.N, self.world_size
        lib, s = self.lib, self._stream_ptr
        reg = {k: self.region(k) for k in ("adv", "perm", "adv_sums", "adv_stats", "params", "adam_m", "adam_v", "grad", "count", "losses", "obs",
                                           "action", "value", "log_prob", "target", "grad_ws", "adam_ws")}
        p = {k: nat.ptr(v) for k, v in reg.items()}
        lib.adv_sums(p["adv"], p["perm"], E * M, mb, p["adv_sums"], s)
        if W > 1:
            self._sync()
            allreduce_sum(reg["adv_sums"])
        lib.adv_stats_finalize(p["adv_sums"], E * M, float(mb * W), p["adv_stats"], s)
        batch = nat.Batch(p["obs"], self.OP, p["action"], self.A, p["value"], p["log_prob"], p["adv"], p["target"])
        lc = self.ecfg.loss
        ac = nat.AdamCfg(self.ecfg.adam.lr, self.ecfg.adam.max_grad_norm, 0.9, 0.999, 1e-5, self.ecfg.adam.anneal, mb * W * E, self.ecfg.num_updates)
        wsb = lib.grad_ws_bytes(C.byref(self.net), mb)
        for e in range(E):
            for k in range(M):
                st = e * M + k
                lib.minibatch_grad(C.byref(self.net), p["params"], C.byref(batch), p["perm"] + 4 * (e * B + k * mb), mb, p["adv_stats"] + 8 * st,
                                   1.0 / (mb * W), C.byref(lc), p["grad"], p["losses"] + 16 * st, p["grad_ws"], wsb, s)
                if W > 1:
                    self._sync()
                    allreduce_sum(reg["grad"])
                lib.clip_adam(self.P, p["params"], p["adam_m"], p["adam_v"], p["grad"], p["count"], st, C.byref(ac), p["adam_ws"], self.lib.adam_ws_bytes(self.P), s)
        self._sync()
        cnt = self.region("count")
        cnt[0] += E * M
        cnt[1] += 1
        obs = self.region("obs", (self.T + 1, self.N, self.OP))
        if self.xp == "torch":
            obs[0].copy_(obs[self.T])
        else:
            obs[0] = obs[self.T]
        self.updates_done += 1

    def rollout_stats(self, reduce: bool = False) -> Dict[str, float]:
        """Device-side reduction of the last rollout's `EnvMetrics` history (`env.py:183-194`, `train.py:170,283`):
        mean reward per env-step, fraction of done steps, number of episodes that ended, and the mean return / length
        of those episodes (`returned_episode_returns / _lengths` at the steps where `returned_episode` is set).
        `reduce=True` sums over the ranks first (a collective)."""
        self._sync()
        s = self._to_host(self.region("rollout_stats")).astype(np.float64)[:4]
        n = float(self.T * self.N)
        if reduce and self.world_size > 1:
            s = self._allreduce_host(s)
            n *= self.world_size
        ep = float(s[1])
        return {"mean_reward": float(s[0]) / n, "done_fraction": ep / n, "episodes": ep,
                "mean_episode_return": float(s[2]) / ep if ep > 0 else 0.0, "mean_episode_length": float(s[3]) / ep if ep > 0 else 0.0}

    def losses(self, reduce: bool = False) -> np.ndarray:
        """[E, M, 4] = (total, value, actor, entropy) of every minibatch of the last update.  With more than one rank
        the engine weights rows 1/(mb*world), so each rank holds its partial sums; `reduce=True` adds them over the
        ranks (a collective: every rank must call it) and returns the losses of the global minibatches."""
        self._sync()
        out = self._to_host(self.region("losses", (self.E, self.M, 4))).copy()
        if reduce and self.world_size > 1:
            out = self._allreduce_host(out)
        return out

    def _allreduce_host(self, a: np.ndarray) -> np.ndarray:
        import torch
        import torch.distributed as dist

        t = torch.from_numpy(np.ascontiguousarray(a))
        if dist.get_backend() == "nccl":
            t = t.to(self.device)
        dist.all_reduce(t)
        return t.cpu().numpy()

    def close(self) -> None:
        if getattr(self, "_engine", None) is not None and self._engine:
            self._sync()
            self.lib.engine_destroy(self._engine)
            self._engine = None
        if getattr(self, "_model", None) is not None and self._model:
            self.lib.model_close(self._model)
            self._model = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass


def _seed_from_rng(rng: Any) -> int:
    if rng is None:
        return 0
    a = np.asarray(rng)
    if a.ndim == 0:
        return int(a)
    a = a.astype(np.uint64).reshape(-1)
    s = 0
    for w in a:
        s = (s * 0x100000001B3 + int(w)) & 0xFFFFFFFFFFFFFFFF
    return s


_METRICS_HISTORY_MAX_BYTES = 1 << 30  # training.keep_metrics_history: refuse more than 1 GiB of [num_updates, T, N] history


def make_train(config: Config, **trainer_kwargs: Any) -> Callable[[Any], TrainOutput]:
    """`make_train(config) -> train(rng)` (`minppo/train.py:92-291`)."""
    num_updates = config.training.total_timesteps // config.training.num_steps // config.training.num_envs
    minibatch_size = config.training.num_envs * config.training.num_steps // config.training.num_minibatches
    if minibatch_size * config.training.num_minibatches != config.training.num_steps * config.training.num_envs:
        raise ValueError("`batch_size` must be equal to `num_steps * num_envs`")

    keep_hist = bool(config.training.keep_metrics_history)
    hist_bytes_per_update = config.training.num_steps * config.training.num_envs * (4 * 5 + 1)
    if keep_hist and num_updates * hist_bytes_per_update > _METRICS_HISTORY_MAX_BYTES:
        raise ValueError(f"training.keep_metrics_history: {num_updates} updates x [{config.training.num_steps}, {config.training.num_envs}] x 6 fields = "
                         f"{num_updates * hist_bytes_per_update / 2**30:.1f} GiB of history (limit {_METRICS_HISTORY_MAX_BYTES / 2**30:.0f} GiB): the reference-shaped "
                         "metrics are for small runs; the default returns per-update reductions")

    def train(rng: Any, max_updates: Optional[int] = None, log_every: int = 0) -> TrainOutput:
        tr = Trainer(config, seed=_seed_from_rng(rng), **trainer_kwargs)
        tr.init_comm()
        tr.keep_metrics_history(keep_hist)
        hist: list = []
        tc = config.training
        ckpt = (tc.checkpoint_path + (f".rank{tr.rank}" if tr.world_size > 1 else "")) if tc.checkpoint_path else ""
        if tc.resume_from:
            tr.load_checkpoint(tc.resume_from + (f".rank{tr.rank}" if tr.world_size > 1 else ""))
            logger.info("resumed from %s at update %d", tc.resume_from, tr.updates_done)
        else:
            tr.reset()
        first = tr.updates_done
        n = num_updates if max_updates is None else min(num_updates, first + max_updates)
        metrics = {"mean_reward": [], "done_fraction": [], "episodes": [], "mean_episode_return": [], "mean_episode_length": [], "total_loss": [], "value_loss": [],
                   "actor_loss": [], "entropy": []}
        t0 = time.time()
        peers = tr.world_size > 1 and tr.comm_mode() == "peer"
        for u in range(first, n):
            tr.update()
            if keep_hist:
                hist.append(tr.metrics_history())
            if ckpt and tc.checkpoint_every > 0 and (u + 1) % tc.checkpoint_every == 0 and u + 1 < n:
                if peers:
                    tr.check_peers()  # collective: never checkpoint parameters that ANY rank updated with a timed-out exchange
                if tr.world_size > 1:
                    tr.check_replicas()  # collective: the replicas hold the same bits (a corrupted exchange shows here, not only at the end of the run)
                tr.check_status()
                tr.save_checkpoint(ckpt)
                tr.barrier()  # the ranks' files take different times to write: nobody starts the next exchange seconds ahead of a peer
            if log_every and ((u + 1) % log_every == 0 or u + 1 == n):
                if peers:
                    tr.check_peers()  # (the statistics below synchronise anyway) a dead peer ends the job here, not as silent garbage
                st, lo = tr.rollout_stats(reduce=True), tr.losses(reduce=True).reshape(-1, 4).mean(0)
                for k, v in st.items():
                    metrics[k].append(v)
                for k, v in zip(("total_loss", "value_loss", "actor_loss", "entropy"), lo):
                    metrics[k].append(float(v))
                sps = (u + 1 - first) * tr.T * tr.N * tr.world_size / (time.time() - t0)
                logger.info("update %d/%d  reward %.3f  done %.4f  episode return %.2f length %.1f  loss %.4f  %.0f env-steps/s", u + 1, n, st["mean_reward"],
                            st["done_fraction"], st["mean_episode_return"], st["mean_episode_length"], lo[0], sps)
        if peers:
            tr.check_peers()
        if tr.world_size > 1:
            tr.check_replicas()
        tr.check_status()
        if ckpt:
            tr.save_checkpoint(ckpt)
            tr.barrier()
        params = tr.params
        count = int(tr._to_host(tr.region("count"))[0]) if n else 0
        state = TrainState(step=count, params=params, opt_state={"mu": flat_to_tree(tr._to_host(tr.region("adam_m")), tr.O, tr.A, tr.H, tr.L),
                                                                   "nu": flat_to_tree(tr._to_host(tr.region("adam_v")), tr.O, tr.A, tr.H, tr.L), "count": count})
        last_obs = tr._to_host(tr.region("obs", (tr.T + 1, tr.N, tr.OP))[0, :, :tr.O])
        rs = RunnerState(train_state=state, env_state=tr._to_host(tr.region("state", (tr.N, tr.dims.rec_dim))), last_obs=last_obs, rng=rng)
        if keep_hist:
            # the reference's `metric` (train.py:283,287-289): EnvMetrics of [num_updates, T, N] arrays (this rank's environments: with several
            # ranks N is the rank's share); the per-update reductions are then not returned
            from minppo_amd.env import EnvMetrics

            stacked = EnvMetrics(*(np.stack([getattr(h, f) for h in hist]) if hist else np.zeros((0, tr.T, tr.N), dt) for f, dt in Trainer._METRIC_FIELDS))
            out = TrainOutput(runner_state=rs, metrics=stacked)
        else:
            out = TrainOutput(runner_state=rs, metrics={k: np.asarray(v) for k, v in metrics.items()})
        tr.close()
        return out

    return train


def _dist_env() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) as torch.distributed.run / bench.py export them; (0, 1, 0) for a plain run."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def main(args: Sequence[str] | None = None) -> None:
    """`minppo train <config> [overrides]` (`minppo/train.py:294-315`).  Under `python -m torch.distributed.run
    --nproc-per-node G -m minppo_amd.train ...` every process is one rank of the env-sharded run:
    rank r drives GPU LOCAL_RANK with num_envs / G environments, the engine all-reduces gradients over RCCL, and
    only rank 0 writes the model file."""
    logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(levelname)s - %(message)s")
    if args is None:
        args = sys.argv[1:]
    config = load_config_from_cli(args)
    logger.info("Configuration loaded")
    rank, world, local_rank = _dist_env()
    kwargs: Dict[str, Any] = {}
    if world > 1:
        import torch
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        kwargs = dict(rank=rank, world_size=world, device=f"cuda:{local_rank}")
        logger.info("rank %d of %d on cuda:%d (%d environments per rank)", rank, world, local_rank, config.training.num_envs // world)
    rng = config.training.seed
    logger.info(f"Random seed set to {config.training.seed}")
    train = make_train(config, **kwargs)
    logger.info("Training function bound to the MI355X engine")
    logger.info("Starting training...")
    log_every = max(1, (config.training.total_timesteps // config.training.num_steps // config.training.num_envs) // 100)
    out = train(rng, log_every=log_every if rank == 0 or world > 1 else 0)
    logger.info("Training completed")
    if rank == 0:
        logger.info(f"Saving model to {config.training.model_save_path}")
        save_model(out.runner_state.train_state.params, config.training.model_save_path)
        logger.info("Model saved successfully")
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
