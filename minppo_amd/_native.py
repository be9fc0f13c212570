"""ctypes binding of the C ABI in include/minppo_hip.h.

The product path loads exactly one library, `minppo_amd/libminppo_hip.so` (hand-written HIP for
gfx950, built by `__graft_entry__.build()` / `minppo_amd/build.py`), and raises if it is missing:
there is no CPU fallback.  (cffi, which the north star names, is not installed on the target
image; ctypes ABI mode is the same boundary.)

`Lib(path)` is also what the test-suite uses to bind the same ABI to the CPU SIMT-emulator
build of the kernel sources (tests/emu/), with NumPy arrays standing in for device memory.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path
from typing import Optional

HERE = Path(__file__).resolve().parent
HIP_LIB_PATH = HERE / "libminppo_hip.so"

c_f = C.c_float
c_i32 = C.c_int32
c_u64 = C.c_uint64
c_sz = C.c_size_t
c_vp = C.c_void_p


class NativeError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"[mppo error {code}] {msg}")
        self.code = code


class ModelDims(C.Structure):
    _fields_ = [(n, c_i32) for n in ("nq", "nv", "nu", "nbody", "njnt", "ncon", "nlimit", "nefc", "obs_dim", "obs_pad",
                                     "rec_dim", "lds_bytes")] + [("timestep", c_f)]


class RewardCfg(C.Structure):
    _fields_ = [(n, c_f) for n in ("height_min_z", "height_max_z", "exp_coefficient", "subtraction_factor", "max_diff_norm",
                                   "w_ctrl_cost", "w_original_pos", "w_is_healthy", "w_velocity")]


class EnvMetrics(C.Structure):
    _fields_ = [(n, c_vp) for n in ("episode_returns", "episode_lengths", "returned_episode_returns",
                                    "returned_episode_lengths", "timestep", "returned_episode")]


class ForwardProbe(C.Structure):
    _fields_ = [(n, c_vp) for n in ("qM", "qfrc_bias", "qfrc_passive", "qfrc_actuator", "qacc_smooth", "efc_J", "efc_D",
                                    "efc_aref", "qacc", "cinert", "cvel", "subtree_com1", "xpos", "qacc_euler",
                                    "solver_niter")]


class Net(C.Structure):
    _fields_ = [(n, c_i32) for n in ("O", "OP", "A", "H", "use_tanh", "bf16", "num_layers")]  # num_layers 0 = 2 (the reference default)


class Batch(C.Structure):
    _fields_ = [("obs", c_vp), ("obs_ld", c_i32), ("action", c_vp), ("act_ld", c_i32), ("value", c_vp), ("log_prob", c_vp),
                ("adv", c_vp), ("target", c_vp)]


class GemmDesc(C.Structure):
    _fields_ = [(n, c_vp) for n in ("A", "B", "C", "bias", "aux", "gather", "bias_out")] + [(n, c_i32) for n in ("M", "N", "K", "lda", "ldb", "ldc", "ldaux", "act")]


class LossCfg(C.Structure):
    _fields_ = [("clip_eps", c_f), ("vf_coef", c_f), ("ent_coef", c_f)]


class AdamCfg(C.Structure):
    _fields_ = [("lr", c_f), ("max_grad_norm", c_f), ("b1", c_f), ("b2", c_f), ("eps", c_f), ("anneal", c_i32),
                ("sched_div", c_i32), ("num_updates", c_i32)]


class EngineCfg(C.Structure):
    _fields_ = [("num_envs", c_i32), ("num_steps", c_i32), ("num_minibatches", c_i32), ("update_epochs", c_i32),
                ("n_frames", c_i32), ("num_updates", c_i32), ("world_size", c_i32), ("rank", c_i32), ("gamma", c_f),
                ("gae_lambda", c_f), ("loss", LossCfg), ("adam", AdamCfg), ("reward", RewardCfg), ("net", Net),
                ("seed", c_u64), ("use_graph", c_i32), ("external_random", c_i32), ("rng_impl", c_i32), ("reserved0", c_i32)]


P = C.POINTER
ABI_VERSION = 7  # include/minppo_hip.h: MPPO_ABI_VERSION

# name -> (restype, argtypes); restype c_i32 functions are checked and raise NativeError
SIGNATURES = {
    "mppo_last_error": (C.c_char_p, []),
    "mppo_abi_version": (c_i32, []),
    "mppo_model_open": (c_i32, [c_vp, c_sz, c_vp, P(c_vp)]),
    "mppo_model_close": (c_i32, [c_vp]),
    "mppo_model_get_dims": (c_i32, [c_vp, P(ModelDims)]),
    "mppo_model_is_specialized": (c_i32, [c_vp, P(c_i32)]),
    "mppo_model_scratch_bytes": (c_i32, [c_vp, c_i32, P(c_sz)]),
    "mppo_model_attach_kernel": (c_i32, [c_vp, c_vp, c_sz, P(C.c_char_p), c_i32, P(c_i32)]),
    "mppo_env_reset": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, P(EnvMetrics), c_vp]),
    "mppo_env_step": (c_i32, [c_vp, c_i32, c_i32, P(RewardCfg), c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_vp, c_vp,
                              P(EnvMetrics), c_vp]),
    "mppo_physics_forward": (c_i32, [c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, P(ForwardProbe), c_vp]),
    "mppo_param_count": (c_sz, [P(Net)]),
    "mppo_policy_ws_bytes": (c_sz, [P(Net), c_i32]),
    "mppo_policy_forward": (c_i32, [P(Net), c_vp, c_i32, c_vp, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mppo_gemm_batch": (c_i32, [P(GemmDesc), c_i32, c_i32, c_i32, c_sz, c_i32, c_vp]),
    "mppo_gae": (c_i32, [c_i32, c_i32, c_f, c_f, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "mppo_grad_ws_bytes": (c_sz, [P(Net), c_i32]),
    "mppo_minibatch_path": (c_i32, [P(Net), P(Batch), P(c_i32)]),
    "mppo_minibatch_rows_per_workgroup": (c_i32, [P(Net), c_i32, c_i32, P(c_i32)]),
    "mppo_minibatch_grad": (c_i32, [P(Net), c_vp, P(Batch), c_vp, c_i32, c_vp, c_f, P(LossCfg), c_vp, c_vp, c_vp, c_sz,
                                    c_vp]),
    "mppo_minibatch_rowpass": (c_i32, [P(Net), c_vp, P(Batch), c_vp, c_i32, c_vp, c_f, P(LossCfg), c_vp, c_sz, c_vp]),
    "mppo_adv_sums": (c_i32, [c_vp, c_vp, c_i32, c_i32, c_vp, c_vp]),
    "mppo_adv_stats_finalize": (c_i32, [c_vp, c_i32, C.c_double, c_vp, c_vp]),
    "mppo_adam_ws_bytes": (c_sz, [c_sz]),
    "mppo_clip_adam": (c_i32, [c_sz, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, P(AdamCfg), c_vp, c_sz, c_vp]),
    "mppo_shadow_refresh": (c_i32, [P(Net), c_vp, c_i32, c_vp, c_sz, c_vp]),
    "mppo_minibatch_rowpass_shadow": (c_i32, [P(Net), c_vp, P(Batch), c_vp, c_i32, c_vp, c_f, P(LossCfg), c_vp, c_sz, c_vp]),
    "mppo_minibatch_grad_shadow": (c_i32, [P(Net), c_vp, P(Batch), c_vp, c_i32, c_vp, c_f, P(LossCfg), c_vp, c_vp, c_vp, c_sz, c_vp]),
    "mppo_clip_adam_shadow": (c_i32, [P(Net), c_i32, c_vp, c_sz, c_sz, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, P(AdamCfg), c_vp, c_sz, c_vp]),
    "mppo_gather_rows": (c_i32, [P(Net), P(Batch), c_vp, c_i32, c_vp, c_sz, c_i32, c_vp]),
    "mppo_minibatch_rowpass_pre": (c_i32, [P(Net), c_vp, P(Batch), c_vp, c_vp, c_i32, c_vp, c_f, P(LossCfg), c_vp, c_sz, c_i32, c_vp]),
    "mppo_minibatch_grad_pre": (c_i32, [P(Net), c_vp, P(Batch), c_vp, c_vp, c_i32, c_vp, c_f, P(LossCfg), c_vp, c_vp, c_vp, c_sz, c_i32, c_vp]),
    "mppo_normal_fill": (c_i32, [c_u64, c_u64, c_sz, c_vp, c_vp]),
    "mppo_permutation_ws_bytes": (c_sz, [c_i32]),
    "mppo_permutation": (c_i32, [c_u64, c_u64, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "mppo_threefry_normal": (c_i32, [c_vp, c_sz, c_vp, c_vp]),
    "mppo_threefry_bits": (c_i32, [c_vp, c_sz, c_vp, c_vp]),
    "mppo_threefry_update_keys": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "mppo_threefry_permutation": (c_i32, [c_vp, c_i32, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "mppo_engine_arena_bytes": (c_i32, [c_vp, P(EngineCfg), P(c_sz)]),
    "mppo_engine_create": (c_i32, [c_vp, P(EngineCfg), c_vp, c_sz, P(c_vp)]),
    "mppo_engine_destroy": (c_i32, [c_vp]),
    "mppo_engine_region": (c_i32, [c_vp, C.c_char_p, P(c_sz), P(c_sz)]),
    "mppo_comm_unique_id": (c_i32, [c_vp]),
    "mppo_engine_comm_init": (c_i32, [c_vp, c_vp]),
    "mppo_engine_peer_export": (c_i32, [c_vp, c_vp]),
    "mppo_engine_peer_connect": (c_i32, [c_vp, c_vp, c_i32]),
    "mppo_engine_comm_mode": (c_i32, [c_vp, P(c_i32)]),
    "mppo_engine_peer_status": (c_i32, [c_vp, P(c_i32), P(c_i32)]),
    "mppo_engine_peer_selftest": (c_i32, [c_vp, c_vp, P(c_i32)]),
    "mppo_engine_peer_latency": (c_i32, [c_vp, c_i32, c_i32, c_i32, c_vp, P(C.c_double)]),
    "mppo_engine_peer_disable": (c_i32, [c_vp]),
    "mppo_engine_reset": (c_i32, [c_vp, c_vp]),
    "mppo_engine_update": (c_i32, [c_vp, c_vp]),
    "mppo_engine_prepare": (c_i32, [c_vp, c_vp]),
    "mppo_engine_graph_active": (c_i32, [c_vp, P(c_i32)]),
    "mppo_engine_rollout": (c_i32, [c_vp, c_vp]),
    "mppo_engine_learn": (c_i32, [c_vp, c_vp]),
}

UNCHECKED = {"mppo_last_error", "mppo_abi_version", "mppo_param_count", "mppo_policy_ws_bytes", "mppo_grad_ws_bytes",
             "mppo_adam_ws_bytes", "mppo_permutation_ws_bytes"}


class Lib:
    """One loaded instance of the C ABI."""

    def __init__(self, path: os.PathLike | str):
        self.path = str(path)
        # RTLD_LOCAL (the default): the test-suite's emulator build exports the same C++ symbols; with global visibility whichever
        # library came first would serve the other's internal calls
        self._dll = C.CDLL(self.path)
        self._fn = {}
        for name, (res, args) in SIGNATURES.items():
            try:
                f = getattr(self._dll, name)
            except AttributeError:
                continue
            f.restype = res
            f.argtypes = args
            self._fn[name] = f

    def has(self, name: str) -> bool:
        return name in self._fn

    def exported(self):
        return sorted(self._fn)

    def last_error(self) -> str:
        return self._fn["mppo_last_error"]().decode("utf-8", "replace")

    def __getattr__(self, name: str):
        key = "mppo_" + name
        fn = self.__dict__.get("_fn", {}).get(key)
        if fn is None:
            raise AttributeError(f"{self.path} does not export {key}")
        if key in UNCHECKED:
            return fn

        def checked(*a):
            rc = fn(*a)
            if rc != 0:
                raise NativeError(rc, self.last_error())
            return rc

        return checked


_LIB: Optional[Lib] = None


def load() -> Lib:
    """Loads the HIP engine.  Raises if it has not been built: no CPU fallback exists."""
    global _LIB
    if _LIB is None:
        if not HIP_LIB_PATH.exists():
            raise ImportError(
                f"{HIP_LIB_PATH} is missing: build the gfx950 engine first (python -c 'import __graft_entry__ as g; "
                f"g.build()' or python -m minppo_amd.build). minppo_amd has no CPU fallback."
            )
        # torch ships its own HIP runtime / RCCL with the same SONAMEs; import it first so that the
        # engine binds to the runtime instance that owns torch's streams and allocations.
        import torch  # noqa: F401

        _LIB = Lib(HIP_LIB_PATH)
        if _LIB.abi_version() != ABI_VERSION:
            raise ImportError(f"{HIP_LIB_PATH}: ABI version {_LIB.abi_version()} != {ABI_VERSION} (rebuild: python -m minppo_amd.build)")
    return _LIB


def ptr(x) -> int:
    """Device (torch) or host (numpy) array -> raw address."""
    if x is None:
        return 0
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return x.ctypes.data
