// R1: policy query kernel (aog_actor_act).
#include "host_common.h"
#include "k_actor.h"

using namespace aog_host;

extern "C" {

int aog_actor_act(const aog_actor* n, int device, const void* obs_dev, int obs_is_f16, float* mean_dev, float* action_dev, float* log_prob_dev,
                  void* stream) {
  if (!n || !obs_dev) return fail(AOG_ERR_INVALID, "aog_actor_act: null argument");
  if (n->batch < 0 || n->state_dim < 1 || n->hidden_dim < 1 || n->act_dim < 1 || n->hidden_dim > 1024 || n->state_dim > 1024 || n->act_dim > 4096)
    return fail(AOG_ERR_INVALID, "aog_actor_act: bad dimensions (batch %d, state %d, hidden %d, act %d)", n->batch, n->state_dim, n->hidden_dim, n->act_dim);
  if (!n->w1 || !n->b1 || !n->w2 || !n->b2 || !n->w3 || !n->b3 || !n->wo || !n->bo) return fail(AOG_ERR_INVALID, "aog_actor_act: null weight pointer");
  if (((uintptr_t)n->w1 | (uintptr_t)n->w2 | (uintptr_t)n->w3 | (uintptr_t)n->wo) & 15) return fail(AOG_ERR_INVALID, "aog_actor_act: weight matrices must be 16-byte aligned");
  if (!(n->dropout_p >= 0.f && n->dropout_p < 1.f) || !(n->cov_var > 0.f)) return fail(AOG_ERR_INVALID, "aog_actor_act: dropout_p must be in [0,1), cov_var > 0");
  if (n->batch == 0) return AOG_OK;
  HIP_TRY(hipSetDevice(device));
  aog::ActorArgs a{};
  a.obs = obs_dev;
  a.obs_f16 = obs_is_f16 ? 1 : 0;
  a.w1 = n->w1; a.b1 = n->b1; a.w2 = n->w2; a.b2 = n->b2; a.w3 = n->w3; a.b3 = n->b3; a.wo = n->wo; a.bo = n->bo;
  a.mean = mean_dev; a.action = action_dev; a.log_prob = log_prob_dev;
  a.B = n->batch; a.S = n->state_dim; a.H = n->hidden_dim; a.A = n->act_dim;
  a.kpad = round_up(std::max(n->state_dim, n->hidden_dim), 16);
  a.p_drop = n->dropout_p;
  a.keep_scale = 1.0f / (1.0f - n->dropout_p);
  a.std = std::sqrt(n->cov_var);
  a.logp_const = 0.5f * (float)n->act_dim * std::log(2.0f * (float)M_PI * n->cov_var);
  a.seed = n->seed;
  a.call_lo = (uint32_t)n->call_index;
  a.call_hi = (uint32_t)(n->call_index >> 32);
  a.env_base = n->env_id_base;
  const size_t lds = ((size_t)2 * a.kpad * 16 + 16 + (size_t)aog::kActorWFloats) * sizeof(float);
  if (int rc = aog_host::ensure_dynamic_lds(reinterpret_cast<const void*>(aog::k_actor_act), lds, device)) return rc;
  hipLaunchKernelGGL(aog::k_actor_act, dim3((n->batch + 15) / 16), dim3(aog::kActorThreads), lds, static_cast<hipStream_t>(stream), a);
  HIP_TRY(hipGetLastError());
  return AOG_OK;
}

}  // extern "C"
