/*
 * minppo_hip.h — C ABI of libminppo_hip.so, the MI355X (gfx950) engine behind the
 * minppo train / env surface.
 *
 * The reference (kscalelabs/minppo @ 2024-10-16) exposes NO FFI / plugin interface: its hot
 * path is one jitted JAX program reached through plain Python callables.  The
 * entry points below are therefore the boundary a maintainer would bind (ctypes / cffi ABI
 * mode) to replace the stages of that program; each one cites the
 * reference lines it replaces.  All paths are relative to /root/reference/.
 *
 * Conventions
 *   - every function returns 0 (MPPO_OK) or a negative MPPO_E* code; the message of the last
 *     failure on the calling thread is returned by mppo_last_error().
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  No function
 *     allocates device memory, synchronises the host with the device or touches the default
 *     stream behind the caller's back, unless stated.  Device pointers are owned by the caller
 *     (torch tensors) and must outlive the enqueued work.
 *   - all device arrays are float32 unless stated; matrices are row-major.
 *   - flat parameter vector layout (P = 2(O*H + H + H*H + H) + H*A + 2A + H + 1 parameters):
 *       actor : W1[O,H] b1[H] W2[H,H] b2[H] W3[H,A] b3[A]  log_std[A]
 *       critic: W1[O,H] b1[H] W2[H,H] b2[H] W3[H,1] b3[1]
 *     (`kernel [in,out]`, `bias [out]`: the Flax layout of train.py:63,68; two hidden layers,
 *     config.py:53.)  Every tensor starts at the next multiple of 4 floats (16 bytes), whatever
 *     A is; the alignment words in between hold zeros in the parameter, gradient and moment
 *     vectors.  mppo_param_count() is the length including them.
 *   - trajectories are time-major [T,N,...]; flat sample index = t*N + n (train.py:260).
 *   - observation rows are padded to OP = round_up(O, 4) floats (pad = 0) so that every row
 *     starts 16-byte aligned.
 */
#ifndef MINPPO_HIP_H
#define MINPPO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPO_OK 0
#define MPPO_EINVAL (-1)   /* bad argument / shape the kernels do not support            */
#define MPPO_EHIP (-2)     /* a HIP runtime call failed (message has hipGetErrorString)   */
#define MPPO_EMODEL (-3)   /* malformed model blob                                        */
#define MPPO_ESTATE (-4)   /* call order violated (e.g. update before reset)              */
#define MPPO_ENCCL (-5)    /* an RCCL call failed                                         */
#define MPPO_ENOMEM (-6)   /* caller-provided workspace too small                         */

#define MPPO_ABI_VERSION 7  /* 7: mppo_model_attach_kernel (an environment kernel compiled for the robot at run time); 6: mppo_model_scratch_bytes (a large robot's matrices in global memory); 5: mppo_minibatch_rows_per_workgroup; a bf16 network's fragment copies are tile-major (4: mppo_engine_peer_selftest runs on the caller's stream) */

const char* mppo_last_error(void);
int32_t mppo_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Robot model.  Replaces `load_mjcf_model` + `mjcf.load_model` + the solver override
 * (minppo/env.py:27-50, 94-100): the host compiles an MJCF-like description into one
 * little-endian blob of 4-byte words (minppo_amd/model.py: _to_blob documents the layout);
 * the same bytes must be resident in device memory at `dev_blob` for the model's lifetime.
 * ---------------------------------------------------------------------------------------- */
typedef struct mppo_model mppo_model_t;

typedef struct mppo_model_dims {
  int32_t nq, nv, nu, nbody, njnt, ncon, nlimit, nefc;
  int32_t obs_dim;      /* O  = nq + 2 nv + 16 (nbody-1)     (env.py:245-253, include_c_vals) */
  int32_t obs_pad;      /* OP = round_up(O, 4)                                                */
  int32_t rec_dim;      /* floats per env in the persistent state record                      */
  int32_t lds_bytes;    /* dynamic LDS of one env_step workgroup                              */
  float timestep;
} mppo_model_dims_t;

int32_t mppo_model_open(const void* host_blob, size_t nbytes, const void* dev_blob, mppo_model_t** out);
int32_t mppo_model_close(mppo_model_t* m);
int32_t mppo_model_get_dims(const mppo_model_t* m, mppo_model_dims_t* out);
/* *out = 1 when this model's dimensions match one of the environment-kernel instantiations compiled for fixed dimensions
 * (csrc/spec_dims.inc, written by `python -m minppo_amd.build`; MPPO_SPECIALIZE=robot.xml[,...] adds models), 0 when it runs the
 * run-time-sized kernel.  Same results either way; the fixed-size kernel is faster (env_kernel). */
int32_t mppo_model_is_specialized(const mppo_model_t* m, int32_t* out);
/* Attaches a gfx950 code object that holds the environment kernel compiled for exactly this robot's dimensions - the build-time
 * MPPO_SPECIALIZE at run time, for a robot the library was not built for (`minppo_amd/jit.py` makes one: hipcc, device side only, of
 * csrc/k_physics.hip with a one-robot list, cached by the hash of the kernel sources and the dimensions).  What it replaces in the
 * reference: `jax.jit` of the environment step (env.py:123,147 `@partial(jax.jit, ...)` on reset / step, train.py:133,138,306: XLA
 * compiles them for the loaded robot's shapes at start-up).  `names[0..2]`: the symbols of the kernel's three modes (reset, step, probe) - their mangled names must
 * spell this model's dimensions; `regchol_max_nv`: the MPPO_REGCHOL_MAX_NV the object was compiled with (48 by default: its LDS layout
 * follows from it).  The object must carry the library's `mppo_env_kernel_tag` (argument-struct sizes, blob version).  Before it is
 * used, a reset and four steps of 24 environments must equal the run-time-sized kernel's bit for bit on this device.  *used = 1: the
 * model now runs the attached kernel (mppo_model_is_specialized reports 2); *used = 0 with MPPO_OK: it does not - the library has an
 * instantiation for this robot already, MPPO_ENV_GENERIC / MPPO_ENV_SPILL are set, the robot is too large for four environments per
 * wave, or the kernel failed the comparison (a message on stderr) - and the model is as it was.  Call before the model is handed to an
 * engine (mppo_model_get_dims().lds_bytes and mppo_model_scratch_bytes change with the kernel). */
int32_t mppo_model_attach_kernel(mppo_model_t* m, const void* image, size_t nbytes, const char* const* names, int32_t regchol_max_nv, int32_t* used);
/* Bytes of global memory the environment kernel uses beside the state for N environments: 0 for a robot whose per-environment working set
 * fits LDS four waves to a CU; a larger robot (about 30 dofs / 20 contact slots and up) keeps its mass matrix and contact Jacobian in
 * per-environment records there (L2 / Infinity-Cache resident).  mppo_env_reset / mppo_env_step / mppo_physics_forward
 * allocate them on demand inside the handle (hipMalloc, grown when N grows, freed by mppo_model_close; not during a stream capture, and
 * one stream at a time may launch through such a handle); an engine takes them from its arena (mppo_engine_arena_bytes includes them). */
int32_t mppo_model_scratch_bytes(const mppo_model_t* m, int32_t N, size_t* out);

/* Reward / termination constants read by compute_reward and is_done (env.py:199-242,
 * config.py:36-48). */
typedef struct mppo_reward_cfg {
  float height_min_z, height_max_z;
  float exp_coefficient, subtraction_factor, max_diff_norm;
  float w_ctrl_cost, w_original_pos, w_is_healthy, w_velocity;
} mppo_reward_cfg_t;

/* Per-env episode bookkeeping, `EnvMetrics` (env.py:53-59), structure-of-arrays, each [N]. */
typedef struct mppo_env_metrics {
  float* episode_returns;
  int32_t* episode_lengths;
  float* returned_episode_returns;
  int32_t* returned_episode_lengths;
  int32_t* timestep;
  uint8_t* returned_episode;
} mppo_env_metrics_t;

/* Persistent per-env record, [N, rec_dim] floats:
 *   [0,O)        the observation of this state: qpos, qvel, cinert[1:], cvel[1:], qfrc_actuator
 *                (derived fields are those of the forward pass that produced the state, env.py:245-261)
 *   [OP,OP+nv)   qacc_warmstart
 *   OP+nv        subtree_com[1].x  (env.py:222-223)     OP+nv+1  time
 * `reset_rec` [rec_dim] is the constant reset state (reset_noise_scale = 0, env.py:87,115-121).
 *
 * mppo_env_reset : `HumanoidEnv.reset` vmapped over N (env.py:124-145; train.py:133-143):
 *                  pipeline_init at qpos0 / zero velocity / zero ctrl, zero metrics, obs [N,OP].
 * mppo_env_step  : `HumanoidEnv.step` vmapped over N (env.py:148-196; train.py:138-140,165):
 *                  n_frames x mjx.step, pre-step observation, reward, done (height or NaN),
 *                  auto-reset, metrics.  action is [N, act_ld] with nu valid columns. */
int32_t mppo_env_reset(const mppo_model_t* m, int32_t N, float* state, float* reset_rec, float* obs, int32_t obs_ld,
                       float* reward, uint8_t* done, const mppo_env_metrics_t* metrics, void* stream);
int32_t mppo_env_step(const mppo_model_t* m, int32_t N, int32_t n_frames, const mppo_reward_cfg_t* rc, float* state,
                      const float* reset_rec, const float* action, int32_t act_ld, float* obs, int32_t obs_ld,
                      float* reward, uint8_t* done, const mppo_env_metrics_t* metrics, void* stream);
/* Debug / parity probe: one mjx.forward on caller-given (qpos,qvel,ctrl,qacc_warmstart) [N,*]
 * with every intermediate the parity tests compare exported.  Any output pointer may be NULL.
 * M [N,nv,nv]; efc_* [N,nefc]; J [N,nefc,nv]; cinert [N,nbody,10]; cvel [N,nbody,6]. */
typedef struct mppo_forward_probe {
  float *qM, *qfrc_bias, *qfrc_passive, *qfrc_actuator, *qacc_smooth, *efc_J, *efc_D, *efc_aref, *qacc, *cinert, *cvel,
      *subtree_com1, *xpos, *qacc_euler;
  int32_t* solver_niter;
} mppo_forward_probe_t;
int32_t mppo_physics_forward(const mppo_model_t* m, int32_t N, const float* qpos, const float* qvel, const float* ctrl,
                             const float* qacc_warmstart, const mppo_forward_probe_t* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * PPO stages
 * ---------------------------------------------------------------------------------------- */

/* Network geometry + activation choice (train.py:56-83; config.py:52-55). */
typedef struct mppo_net {
  int32_t O, OP, A, H;
  int32_t use_tanh;  /* actor activation; the critic is always ReLU (train.py:82) */
  int32_t bf16;      /* 0: exact f32 MFMA; 1: bf16-in/f32-accumulate MFMA for the hidden GEMMs */
  int32_t num_layers; /* hidden layers of each MLP (`model.num_layers`, config.py:53, train.py:79,82): 1 .. 4; 0 means 2, the
                         reference's default.  Two hidden layers take the fused kernels, other depths the layer-wise path. */
} mppo_net_t;

/* Length (floats) of the flat parameter / gradient / Adam-moment vectors: the model's parameters with every tensor starting on
 * a 16-byte boundary (250 140 for O = 225, A = 10, H = 256: 250 133 parameters + 7 alignment words, which hold zeros). */
size_t mppo_param_count(const mppo_net_t* net);

/* `ActorCritic.__call__` + sample + log_prob on n rows (train.py:157-160, 79-83):
 *   mean = actor(obs); value = critic(obs); action = mean + exp(log_std)*noise;
 *   log_prob = MVNDiag.log_prob(action).  noise may be NULL (then action/log_prob are not
 *   written: the bootstrap-value call, train.py:182).  ws: workspace, >= mppo_policy_ws_bytes. */
size_t mppo_policy_ws_bytes(const mppo_net_t* net, int32_t n);
int32_t mppo_policy_forward(const mppo_net_t* net, const float* params, int32_t n, const float* obs, int32_t obs_ld,
                            const float* noise, float* action, float* log_prob, float* value, float* mean_out,
                            void* ws, size_t ws_bytes, void* stream);

/* The dense contraction every MLP layer goes through (`nn.Dense` forward, train.py:63,68, and the
 * two products its backward needs), as a batched launch of up to 6 problems on the f32 matrix cores.
 *   variant 0: C = act(A.B + bias)                A [M,K] (rows optionally gathered), B [K,N]
 *   variant 1: C = (A.B^T) * act'(aux)            B stored [N,K]
 *   variant 2: C = A^T.B  (split-K slabs)         A stored [K,M] (rows = samples, optionally gathered);
 *              bias_out (optional) receives the column sums of B (the bias gradient), one slab per split
 * act: 0 none, 1 tanh, 2 relu.  Exposed so that benchmarks can time exactly the kernel the engine runs. */
typedef struct mppo_gemm_desc {
  const float* A; const float* B; float* C; const float* bias; const float* aux; const int32_t* gather; float* bias_out;
  int32_t M, N, K, lda, ldb, ldc, ldaux, act;
} mppo_gemm_desc_t;
int32_t mppo_gemm_batch(const mppo_gemm_desc_t* probs, int32_t count, int32_t variant, int32_t ksplit, size_t slab_stride,
                        int32_t bf16, void* stream);

/* `_calculate_gae` (train.py:185-205): reverse scan over T, independent per env. */
int32_t mppo_gae(int32_t T, int32_t N, float gamma, float lam, const float* reward, const float* value,
                 const uint8_t* done, const float* last_val, float* adv, float* target, void* stream);

/* One `_update_minibatch` gradient (train.py:213-247): gathers rows idx[0..mb) of the flat
 * [B,...] batch, forward, clipped-PPO loss with the given advantage statistics
 * (adv_stat = {mean, 1/(std+1e-8)}; per-minibatch population statistics, train.py:235),
 * backward.  grad [P] receives d(total_loss)/d(params) with every row weighted by inv_count
 * (1/mb on one GPU, 1/(mb*world) when ranks are summed afterwards).  loss4 = {total, value,
 * actor, entropy} partial sums weighted the same way. */
typedef struct mppo_batch {
  const float* obs;       int32_t obs_ld;   /* [B,OP]        */
  const float* action;    int32_t act_ld;   /* [B,act_ld]    */
  const float* value;                        /* [B] old value */
  const float* log_prob;                     /* [B] old logp  */
  const float* adv;                          /* [B]           */
  const float* target;                       /* [B]           */
} mppo_batch_t;
typedef struct mppo_loss_cfg { float clip_eps, vf_coef, ent_coef; } mppo_loss_cfg_t;
/* *fused = 1 if mppo_minibatch_grad takes the fused row pass + single-launch weight gradients for this geometry
 * (H a multiple of 32 up to 256, A <= 32, 16-byte aligned observation rows), 0 if it runs the layer-wise kernels. */
int32_t mppo_minibatch_path(const mppo_net_t* net, const mppo_batch_t* batch, int32_t* fused);
/* *rows = minibatch rows per workgroup of the fused row pass for a minibatch of mb rows: 16, or 32 where the engine's minibatch loop
 * (pre_gathered = 1) runs a float network's row pass on two 16-row tiles per workgroup because the 16-row tiling would need more
 * workgroups than the chip has CUs (BASELINE configs[4]: 2560-row minibatches).  One row of loss partials is written per workgroup. */
int32_t mppo_minibatch_rows_per_workgroup(const mppo_net_t* net, int32_t mb, int32_t pre_gathered, int32_t* rows);
size_t mppo_grad_ws_bytes(const mppo_net_t* net, int32_t mb);
int32_t mppo_minibatch_grad(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx,
                            int32_t mb, const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, float* grad,
                            float* loss4, void* ws, size_t ws_bytes, void* stream);

/* The row-local half of mppo_minibatch_grad on its own (forward of both networks, loss terms, d loss / d outputs and the
 * activation gradients dZ2, dZ1; one fused launch when H % 32 == 0 and A <= 16).  Same arguments; leaves its results in
 * `ws` for the weight-gradient product.  Exposed so that benchmarks can time exactly this kernel. */
int32_t mppo_minibatch_rowpass(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx,
                               int32_t mb, const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, void* ws,
                               size_t ws_bytes, void* stream);

/* Per-minibatch advantage statistics for all E*M minibatches of an update in one launch:
 * sums[k] = {sum adv, sum adv^2} (float64) over rows idx[k*mb .. (k+1)*mb); then
 * mppo_adv_stats_finalize turns (possibly all-reduced) sums into {mean, 1/(std+1e-8)}. */
int32_t mppo_adv_sums(const float* adv, const int32_t* idx, int32_t num_minibatches_total, int32_t mb, double* sums,
                      void* stream);
int32_t mppo_adv_stats_finalize(const double* sums, int32_t num_minibatches_total, double count, float* stats,
                                void* stream);

/* `optax.chain(clip_by_global_norm(max_norm), adam(lr_schedule, eps))` + apply (train.py:98-101,
 * 115-124, 248).  The step index is count_base[0] + step_offset (device int32 + host int, so
 * that the call can be captured in a hipGraph); lr = anneal ? lr*(1 - (count // sched_div)/num_updates)
 * : lr  with sched_div = minibatch_size*update_epochs as written in the reference (quirk C-2).
 * ws: >= mppo_adam_ws_bytes(P). */
typedef struct mppo_adam_cfg {
  float lr, max_grad_norm, b1, b2, eps;
  int32_t anneal, sched_div, num_updates;
} mppo_adam_cfg_t;
size_t mppo_adam_ws_bytes(size_t P);
int32_t mppo_clip_adam(size_t P, float* params, float* m, float* v, const float* grad, const int32_t* count_base,
                       int32_t step_offset, const mppo_adam_cfg_t* cfg, void* ws, size_t ws_bytes, void* stream);

/* Shadow copies.  The gradient workspace also holds W2^T of both networks (2 x H x H floats): with them the backward row pass
 * streams the second-layer weights like a forward layer (23.9 -> 21.9 us for the headline minibatch).  The workspace carries no
 * state, so the caller says when the copies match `params`:
 *   mppo_shadow_refresh          builds them from `params` (one small launch; after an upload, a restore, an external write);
 *   mppo_clip_adam_shadow        = mppo_clip_adam that also writes the updated W2 entries into the copies of `grad_ws`;
 *   mppo_minibatch_*_shadow      = mppo_minibatch_rowpass / _grad reading the copies (results identical to the plain forms).
 * The engine (mppo_engine_update / _learn) refreshes once per update and uses the _shadow forms throughout. */
int32_t mppo_shadow_refresh(const mppo_net_t* net, const float* params, int32_t mb, void* grad_ws, size_t grad_ws_bytes, void* stream);
int32_t mppo_minibatch_rowpass_shadow(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx,
                                      int32_t mb, const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, void* ws,
                                      size_t ws_bytes, void* stream);
int32_t mppo_minibatch_grad_shadow(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx,
                                   int32_t mb, const float* adv_stat, float inv_count, const mppo_loss_cfg_t* lc, float* grad,
                                   float* loss4, void* ws, size_t ws_bytes, void* stream);
int32_t mppo_clip_adam_shadow(const mppo_net_t* net, int32_t mb, void* grad_ws, size_t grad_ws_bytes, size_t P, float* params,
                              float* m, float* v, const float* grad, const int32_t* count_base, int32_t step_offset,
                              const mppo_adam_cfg_t* cfg, void* ws, size_t ws_bytes, void* stream);

/* Pre-gathered rows (the engine's minibatch loop).  The observation rows of a minibatch depend on the permutation only, not on
 * the parameters, so the row pass of optimizer step s gathers the rows of step s + 1 on extra workgroups of its own launch into
 * the other of two k-quad buffers of the gradient workspace (`parity` 0 / 1 names the buffer that holds the CURRENT step's rows);
 * step s + 1 then starts from a contiguous tile instead of the index -> row chain (reference train.py:261-265, the gather).
 * The same workgroups gather the rows' scalars of the loss (action, old log_prob, advantage, old value, target) behind the observation
 * rows of the buffer, four rows of a column per float4: the next row pass reads them without an index -> row chain either.
 *   mppo_gather_rows               rows idx[0 .. mb) -> buffer `parity` (the first step of an update)
 *   mppo_minibatch_rowpass_pre     = mppo_minibatch_rowpass_shadow reading buffer `parity` and gathering idx_next (may be NULL: the
 *                                    last step) into the other buffer
 *   mppo_minibatch_grad_pre        the same, followed by the weight-gradient launch: results identical to mppo_minibatch_grad. */
int32_t mppo_gather_rows(const mppo_net_t* net, const mppo_batch_t* batch, const int32_t* idx, int32_t mb, void* grad_ws,
                         size_t grad_ws_bytes, int32_t parity, void* stream);
int32_t mppo_minibatch_rowpass_pre(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx,
                                   const int32_t* idx_next, int32_t mb, const float* adv_stat, float inv_count,
                                   const mppo_loss_cfg_t* lc, void* ws, size_t ws_bytes, int32_t parity, void* stream);
int32_t mppo_minibatch_grad_pre(const mppo_net_t* net, const float* params, const mppo_batch_t* batch, const int32_t* idx,
                                const int32_t* idx_next, int32_t mb, const float* adv_stat, float inv_count,
                                const mppo_loss_cfg_t* lc, float* grad, float* loss4, void* ws, size_t ws_bytes, int32_t parity,
                                void* stream);

/* Counter-based RNG (Philox4x32-10), the engine's own stream (not JAX threefry).
 * mppo_normal_fill: out[i] ~ N(0,1), i in [0,n), a pure function of (seed, stream_id, i).
 * mppo_permutation: idx = a uniformly random permutation of [0,B) (sort of random keys, as
 * jax.random.permutation does; train.py:258).  ws >= mppo_permutation_ws_bytes(B). */
int32_t mppo_normal_fill(uint64_t seed, uint64_t stream_id, size_t n, float* out, void* stream);
size_t mppo_permutation_ws_bytes(int32_t B);
int32_t mppo_permutation(uint64_t seed, uint64_t stream_id, int32_t B, int32_t* idx, void* ws, size_t ws_bytes,
                         void* stream);
/* jax.random-compatible streams (threefry2x32, conventions of jax 0.4.3x; host restatement: minppo_amd/jaxrng.py).
 * key2 / rng2 / *_keys are DEVICE pointers to uint32 pairs.
 *   mppo_threefry_normal      jax.random.normal(key, (n,))  (what `pi.sample(seed=key)` draws, train.py:158-159)
 *   mppo_threefry_bits        jax.random.bits(key, (n,))
 *   mppo_threefry_update_keys the split tree of one _update_step: rng2 (in/out), act_keys [T][2], sort_keys [E][rounds][2]
 *   mppo_threefry_permutation jax.random.permutation(key, B) given its `rounds` sort keys (ws as for mppo_permutation) */
int32_t mppo_threefry_normal(const uint32_t* key2, size_t n, float* out, void* stream);
int32_t mppo_threefry_bits(const uint32_t* key2, size_t n, uint32_t* out, void* stream);
int32_t mppo_threefry_update_keys(uint32_t* rng2, int32_t T, int32_t E, int32_t rounds, uint32_t* act_keys, uint32_t* sort_keys, void* stream);
int32_t mppo_threefry_permutation(const uint32_t* sort_keys, int32_t rounds, int32_t B, int32_t* idx, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Engine: the whole `_update_step` (train.py:146-283) as one call, launches enqueued from
 * C++ (optionally replayed from a hipGraph), gradients summed over ranks with RCCL.
 * ---------------------------------------------------------------------------------------- */
typedef struct mppo_engine mppo_engine_t;

typedef struct mppo_engine_cfg {
  int32_t num_envs;          /* N on THIS rank                                     */
  int32_t num_steps;         /* T  (rl.num_env_steps == training.num_steps)        */
  int32_t num_minibatches;   /* M                                                   */
  int32_t update_epochs;     /* E                                                   */
  int32_t n_frames;
  int32_t num_updates;       /* for the LR schedule (train.py:93)                  */
  int32_t world_size, rank;
  float gamma, gae_lambda;
  mppo_loss_cfg_t loss;
  mppo_adam_cfg_t adam;      /* sched_div / num_updates are filled by the engine   */
  mppo_reward_cfg_t reward;
  mppo_net_t net;
  uint64_t seed;
  int32_t use_graph;         /* capture the update in a hipGraph and replay it      */
  int32_t external_random;   /* 1: noise / permutations are written by the caller into the arena (parity tests) */
  int32_t rng_impl;          /* 0: the engine's Philox streams; 1: jax.random-compatible threefry2x32 streams following the
                              * reference's split tree (train.py:158,163,252,258); the carried key is arena region "jax_rng" */
  int32_t reserved0;
} mppo_engine_cfg_t;

/* Arena: ONE device allocation owned by the caller (a torch uint8 tensor); the engine lays
 * every buffer out inside it.  mppo_engine_arena_bytes gives the size; named regions can be
 * located with mppo_engine_region (offset in bytes, size in bytes) to view them as tensors:
 * "params" "adam_m" "adam_v" "grad" "count" "state" "reset_rec" "obs" "action" "value" "reward"
 * "log_prob" "done" "last_val" "adv" "target" "noise" "perm" "adv_stats" "losses" "rollout_stats"
 * (8 floats: sum reward, episodes ended, sum of their returns, sum of their lengths, the two means, 0, 0) ... */
int32_t mppo_engine_arena_bytes(const mppo_model_t* m, const mppo_engine_cfg_t* cfg, size_t* out);
int32_t mppo_engine_create(const mppo_model_t* m, const mppo_engine_cfg_t* cfg, void* arena, size_t arena_bytes,
                           mppo_engine_t** out);
int32_t mppo_engine_destroy(mppo_engine_t* e);
int32_t mppo_engine_region(const mppo_engine_t* e, const char* name, size_t* offset, size_t* nbytes);
/* RCCL: rank 0 makes an id (128 bytes), the caller broadcasts it, every rank calls comm_init.
 * Collective on the engine's stream: one sum all-reduce of the [P] gradient per optimizer step
 * and one of the [E*M*2] float64 advantage sums per update. */
int32_t mppo_comm_unique_id(void* id128);
int32_t mppo_engine_comm_init(mppo_engine_t* e, const void* id128);
/* Peer-to-peer gradient exchange between the ranks of one node, the alternative to RCCL (csrc/peer.h; the reference has no data
 * parallelism: train.py:136,140).  Every rank allocates one exchange buffer (fine-grained device memory; this call ALLOCATES and
 * synchronises the device) and exports it as a 64-byte hipIpc handle; the caller gathers the handles of all ranks in rank order
 * (64 * world_size bytes), hands them to every rank, and places a host-side barrier between the last connect and the first update.
 * From then on the weight-gradient launch of every optimizer step publishes the local gradient, and the Adam launch reduces this
 * rank's slice, broadcasts it and waits for the others' (two hops, no extra launch, part of the hipGraph); the advantage sums take one
 * small kernel per update.  Takes precedence over an RCCL communicator.  Requires HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool.
 * shared_device != 0: several of the ranks run on ONE GPU (how the path is exercised on a one-GPU box): a kernel that waits for a peer
 * must then leave room for that peer's kernels on every CU, so the two waits of a step become one-wave kernels of their own and
 * nothing else waits (five launches per optimizer step instead of three).
 *   mppo_engine_comm_mode    *out = 0 no exchange, 1 RCCL, 2 peer-to-peer (fused), 3 peer-to-peer in two launches, 4 peer-to-peer,
 *                            shared-GPU form
 *   mppo_engine_peer_status  synchronises the device; *timed_out != 0: a rank waited longer than MPPO_PEER_TIMEOUT_MS (default
 *                            60000) for a peer - the kernels ran to their end, the results are invalid.  info8 (optional, 8 words):
 *                            the first such wait {kind: 1 a peer's local gradient, 2 a reduced piece, 3 a peer's advantage sums;
 *                            index; epoch waited for; value seen}, then {optimizer steps, updates, workgroups counted into an unfinished gradient slice (0 between
 *                            launches), pieces per slice}
 *   mppo_engine_peer_selftest  collective (every rank, after the barrier that follows connect), on `stream`: one all-reduce of a
 *                            known vector through the mapped buffers, then ONE full optimizer step's exchange on a known gradient and
 *                            scratch parameters in the form the engine will launch it (publish, reduce + broadcast of this rank's
 *                            slice, wait for the others, Adam); *ok = 0 when a wait timed out or a value is wrong on THIS rank.  The caller
 *                            agrees on the outcome across ranks and, if any rank failed, calls
 *   mppo_engine_peer_disable   on every rank (frees the exchange; mppo_engine_comm_init may follow) */
int32_t mppo_engine_peer_export(mppo_engine_t* e, void* handle64);
int32_t mppo_engine_peer_connect(mppo_engine_t* e, const void* handles, int32_t shared_device);
int32_t mppo_engine_comm_mode(const mppo_engine_t* e, int32_t* out);
int32_t mppo_engine_peer_status(const mppo_engine_t* e, int32_t* timed_out, int32_t* info8);
int32_t mppo_engine_peer_selftest(mppo_engine_t* e, void* stream, int32_t* ok);
/* One-way latency, in microseconds, of a system-scope flag between this rank and `other_rank`: `iters` round trips of one word through the two
 * ranks' exchange buffers, timed on the device (100 MHz clock) - what one dependent trip of the fused exchange costs on this machine (on one GPU
 * about 0.65 us; over an xGMI hop: the t_link of DESIGN.md 7.2).  BOTH ranks call it at the same time, one with initiator = 1; synchronises.
 * `bench.py --gpus N` reports it per peer of rank 0 (config.t_link_us). */
int32_t mppo_engine_peer_latency(mppo_engine_t* e, int32_t other_rank, int32_t iters, int32_t initiator, void* stream, double* one_way_us);
int32_t mppo_engine_peer_disable(mppo_engine_t* e);
/* env reset (train.py:142-144) */
int32_t mppo_engine_reset(mppo_engine_t* e, void* stream);
/* one full update: T rollout steps, bootstrap value, GAE, E epochs x M minibatches */
int32_t mppo_engine_update(mppo_engine_t* e, void* stream);
/* what mppo_engine_update does before it enqueues anything: the one-time capture of the update into a hipGraph (no device work).
 * Several ranks: call it on every rank, then a host-side barrier, then the first update - the ranks then start their first gradient
 * exchange together, not a capture time apart (the peer-to-peer exchange bounds every wait by MPPO_PEER_TIMEOUT_MS) */
int32_t mppo_engine_prepare(mppo_engine_t* e, void* stream);
/* *out = 1 once mppo_engine_update replays a captured hipGraph, 0 while it launches eagerly (use_graph = 0, null stream, a failed
 * capture, or an RCCL communicator without MPPO_GRAPH_COMM=1: RCCL calls are captured into the graph only on request, and the
 * ranks then agree through one eager all-reduce that every capture succeeded) */
int32_t mppo_engine_graph_active(const mppo_engine_t* e, int32_t* out);
/* pieces of the update, for stage-wise parity tests */
int32_t mppo_engine_rollout(mppo_engine_t* e, void* stream);
int32_t mppo_engine_learn(mppo_engine_t* e, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MINPPO_HIP_H */
