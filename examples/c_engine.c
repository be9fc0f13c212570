/* c_engine.c — the whole PPO update driven from plain C through the C ABI (no Python, no torch):
 * what a maintainer of a compiled host would write instead of minppo/train.py:103-289.
 *
 *   python -c "from minppo_amd.model import load_model; from minppo_amd.train import init_flat_params; import numpy as np; \
 *              cm = load_model('synth_stompy_pro'); open('/tmp/model.blob','wb').write(cm.to_blob()); \
 *              init_flat_params(1337, cm.obs_size(), cm.nu, 256).tofile('/tmp/params.f32')"
 *   gcc -std=c99 -D_POSIX_C_SOURCE=199309L -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_engine.c \
 *       -Lminppo_amd -lminppo_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/minppo_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_engine
 *   /tmp/c_engine /tmp/model.blob /tmp/params.f32 4096 20
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "minppo_hip.h"

#define CK(x) do { int32_t rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, mppo_last_error()); return 1; } } while (0)
#define HK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  *n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(*n);
  if (fread(p, 1, *n, f) != *n) { fclose(f); free(p); return NULL; }
  fclose(f);
  return p;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s model.blob params.f32 num_envs updates\n", argv[0]); return 2; }
  size_t nblob = 0, nparam = 0;
  void* blob = slurp(argv[1], &nblob);
  float* params = (float*)slurp(argv[2], &nparam);
  if (!blob || !params) { fprintf(stderr, "cannot read inputs\n"); return 2; }
  const int N = atoi(argv[3]), updates = atoi(argv[4]);

  void* dblob = NULL;
  HK(hipMalloc(&dblob, nblob));
  HK(hipMemcpy(dblob, blob, nblob, hipMemcpyHostToDevice));
  mppo_model_t* model = NULL;
  CK(mppo_model_open(blob, nblob, dblob, &model));
  mppo_model_dims_t d;
  CK(mppo_model_get_dims(model, &d));

  mppo_engine_cfg_t c;
  memset(&c, 0, sizeof c);
  c.num_envs = N; c.num_steps = 10; c.num_minibatches = 32; c.update_epochs = 4; c.n_frames = 1; c.num_updates = 24414;
  c.world_size = 1; c.rank = 0; c.gamma = 0.99f; c.gae_lambda = 0.95f;
  c.loss.clip_eps = 0.2f; c.loss.vf_coef = 0.5f; c.loss.ent_coef = 0.0f;                          /* config.py:66-71 */
  c.adam.lr = 3e-4f; c.adam.max_grad_norm = 0.5f; c.adam.b1 = 0.9f; c.adam.b2 = 0.999f; c.adam.eps = 1e-5f; c.adam.anneal = 1;
  c.reward.height_min_z = -0.2f; c.reward.height_max_z = 2.0f; c.reward.exp_coefficient = 2.0f; c.reward.subtraction_factor = 0.2f;
  c.reward.max_diff_norm = 0.5f; c.reward.w_ctrl_cost = 0.1f; c.reward.w_original_pos = 4.0f; c.reward.w_is_healthy = 1.0f;
  c.reward.w_velocity = 1.25f;                                                                    /* config.py:36-48 */
  c.net.O = d.obs_dim; c.net.OP = d.obs_pad; c.net.A = d.nu; c.net.H = 256; c.net.use_tanh = 1; c.net.bf16 = 0;
  c.seed = 1337; c.use_graph = 1; c.external_random = 0;
  if (nparam != mppo_param_count(&c.net) * sizeof(float)) { fprintf(stderr, "params file has %zu bytes, expected %zu\n", nparam, mppo_param_count(&c.net) * 4); return 2; }

  size_t bytes = 0, off = 0, nb = 0;
  CK(mppo_engine_arena_bytes(model, &c, &bytes));
  void* arena = NULL;
  HK(hipMalloc(&arena, bytes));            /* the caller owns ALL device memory */
  HK(hipMemset(arena, 0, bytes));
  mppo_engine_t* e = NULL;
  CK(mppo_engine_create(model, &c, arena, bytes, &e));
  CK(mppo_engine_region(e, "params", &off, &nb));
  HK(hipMemcpy((char*)arena + off, params, nparam, hipMemcpyHostToDevice));
  hipStream_t s;
  HK(hipStreamCreate(&s));
  CK(mppo_engine_reset(e, s));
  for (int u = 0; u < 3; ++u) CK(mppo_engine_update(e, s));  /* warm-up: graph capture */
  HK(hipStreamSynchronize(s));
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int u = 0; u < updates; ++u) CK(mppo_engine_update(e, s));
  HK(hipStreamSynchronize(s));
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);

  float losses[4], stats[4];
  CK(mppo_engine_region(e, "losses", &off, &nb));
  HK(hipMemcpy(losses, (char*)arena + off + nb - sizeof losses, sizeof losses, hipMemcpyDeviceToHost));  /* last optimizer step */
  CK(mppo_engine_region(e, "rollout_stats", &off, &nb));
  HK(hipMemcpy(stats, (char*)arena + off, sizeof stats, hipMemcpyDeviceToHost));
  int32_t count[4];
  CK(mppo_engine_region(e, "count", &off, &nb));
  HK(hipMemcpy(count, (char*)arena + off, sizeof count, hipMemcpyDeviceToHost));
  printf("{\"env_steps_per_s\": %.1f, \"ms_per_update\": %.3f, \"optimizer_steps\": %d, \"updates\": %d, \"total_loss\": %.6f, \"value_loss\": %.6f, \"stats0\": %.6f}\n",
         (double)N * 10.0 * updates / sec, 1e3 * sec / updates, count[0], count[1], losses[0], losses[1], stats[0]);
  CK(mppo_engine_destroy(e));
  CK(mppo_model_close(model));
  return 0;
}
