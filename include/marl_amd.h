/*
 * marl_amd.h — C ABI of libmarl_amd.so (gfx950 / MI355X).
 *
 * This is the drop-in boundary for the one hot path this repository replaces in
 * BMDACMER/marl: the batched envs/edge_computing step + observation, the GAE
 * reverse scan, and the discriminator reward evaluation.  All citations are
 * path:line inside the reference tree.
 *
 * Conventions
 *   - every function returns 0 on success or a negative EC_E* code; nothing
 *     throws across the ABI; ec_last_error() returns a thread-local message.
 *   - unless a parameter says "host", array arguments are DEVICE pointers owned
 *     by the caller (contiguous, dtype exactly as declared).
 *   - work is enqueued on the caller's HIP stream (void* = hipStream_t, NULL =
 *     the null stream); no hidden synchronisation except where documented.
 *   - a handle is not thread-safe; distinct handles are independent.
 */
#ifndef MARL_AMD_H
#define MARL_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EC_OK 0
#define EC_EINVAL (-1)  /* bad argument / shape */
#define EC_EHIP (-2)    /* HIP runtime error */
#define EC_ENOMEM (-3)
#define EC_ESTATE (-4)  /* device status word reports a fault (see ec_status) */

/* Bits of the device status word returned by ec_status(). */
#define EC_ST_BAD_ACTION 1u      /* action is neither a neighbour, self nor idle: the reference divides by a
                                    zero transmission rate (edge_node.py:227) -> ZeroDivisionError */
#define EC_ST_BUFFER_OVERFLOW 2u /* buffer_queue ring capacity exceeded */
#define EC_ST_RECV_OVERFLOW 4u   /* a receiving_queues[src] ring capacity exceeded */
#define EC_ST_WAIT_OVERFLOW 8u   /* execution_queue ring capacity exceeded */
#define EC_ST_TICK_OVERFLOW 16u  /* a timer exceeded the tick LUT (cfg.max_ticks) */
#define EC_ST_HOP_OVERFLOW 32u   /* hop >= hop_hist_len on success; counted in the last bin */

/* Environment constants: the fields of the reference's argparse namespace that the step reads
 * (envs/edge_computing/multihop/edge_computing_config.py:1-49). */
typedef struct ec_config {
    int32_t n_nodes;        /* edge_node_num (= n_agents); n_actions = n_nodes + 1 */
    int32_t episode_limit;  /* episode_limit */
    int32_t mini_slots;     /* mini_time_slot_num */
    int32_t k;              /* k: tasks executing in parallel per node */
    double mini_time_slot;  /* mini_time_slot */
    double deadline;        /* deadline */
    double task_size_min, task_size_max;             /* KB, task.py:11 */
    double task_complexity_min, task_complexity_max; /* cycles/bit, task.py:12 */
    double task_drop_penalty, task_failure_penalty, task_success_reward, task_hop_penalty;
    double execution_failure_rate_max; /* obs[0] normaliser, edge_node.py:112 */
    double task_probability_max;       /* obs[2] normaliser, edge_node.py:116 */
    double core_norm;                  /* cpu_core_list[-1], obs[1] normaliser, edge_node.py:114 */
    double rl_queue_coeff;             /* edge_node.py:186-187 */
    int64_t task_seed;    /* env e, node i seeds RandomState(task_seed + env_id_base + e + i)     */
    int64_t failure_seed; /* (parallel_episode_runner.py:19-20 + edge_node.py:28-29)              */
    int32_t hop_hist_len; /* bins of the cumulative max_hop_dict histogram (edge_node.py:287-290) */
    int32_t cap_buffer;   /* ring capacities per node (records); 0 = derive a bound from the config */
    int32_t cap_recv;
    int32_t cap_wait;
    int32_t max_ticks;    /* tick LUT length; 0 = mini_slots * (episode_limit + 2) */
    int32_t reserved;
} ec_config;

/* Static per-node tables (HOST pointers, row-major), built once per process exactly as
 * EdgeNode.__init__ does (edge_node.py:22-85) from generate_graph (network_graph.py:219-264). */
typedef struct ec_topology {
    const int32_t* cores;       /* [N]   cpu_core_num, edge_node.py:31 */
    const double* task_p;       /* [N]   task_probability, edge_node.py:34 */
    const double* efr;          /* [N]   execution_failure_rate, edge_node.py:57 */
    const double* cpu_capacity; /* [N]   edge_node.py:33 */
    const double* queue_len;    /* [N]   execution_queue_len, edge_node.py:64 */
    const double* rate;         /* [N,N] transmission_rates, 0 where not adjacent, edge_node.py:43-52 */
    const double* tfr;          /* [N,N] transmission_failure_rates, edge_node.py:43-52,60-61 */
    const uint8_t* adj;         /* [N,N+1] 1 where j in node.actions (neighbours + self), edge_node.py:54-56 */
} ec_topology;

typedef struct ec_env* ec_handle;

/* --- environment: replaces EdgeComputingEnv for a batch of n_envs instances ------------------- */

/* EdgeComputingEnv.__init__ (edge_computing_env.py:8-30) for envs [env_id_base, env_id_base+n_envs):
 * seeds both MT19937 streams of every node and performs the constructor's generate_task() pass
 * (edge_computing_env.py:17-18).  Synchronous. */
int ec_create(const ec_config* cfg, const ec_topology* topo, int64_t n_envs, int64_t env_id_base,
              int device, ec_handle* out);
int ec_destroy(ec_handle h);

/* EdgeComputingEnv.reset (edge_computing_env.py:145-148, edge_node.py:171-177): clears queues and
 * current_step only; RNG cursors, max_hop_dict and node parameters persist. */
int ec_reset(ec_handle h, void* stream);

/* get_obs + get_avail_actions (edge_computing_env.py:32-37,73-77; edge_node.py:109-136,179-197).
 * obs [B,N,9] f32 (fp64 arithmetic, rounded once), avail [B,N,N+1] i32. Either may be NULL. */
int ec_observe(ec_handle h, float* obs, int32_t* avail, void* stream);

/* EdgeComputingEnv.step (edge_computing_env.py:84-143) for all envs.
 *   actions     [B,N] i64   (dtype of episode_buffer.py:9)
 *   reward      [B]   f32   fp64 sum in the reference's order, rounded once
 *   terminated  [B]   u8
 *   counters    [B,4] i32   finish, success, drop, failure (info dict, edge_computing_env.py:130-135)
 *   completion_time [B] f32 info["task_completion_time"]
 *   hop_hist    [B,H] i32 or NULL: cumulative max_hop_dict (never cleared, edge_node.py:82)
 *   next_obs / next_avail: optional fused observation of the post-step state (NULL to skip) */
int ec_step(ec_handle h, const int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
            float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail, void* stream);

/* Uniform choice over the availability mask (stands in for action_selectors.py:6-8 with a
 * counter-based device RNG; NOT stream-compatible with torch's generator). avail [rows,A] i32,
 * actions [rows] i64.  counter_dev: NULL, or a zero-initialised device u64[2] whose first word is added
 * to `counter` and incremented once per launch (lets a captured hipGraph draw fresh numbers per replay). */
int ec_random_policy(const int32_t* avail, int64_t rows, int32_t n_actions, uint64_t seed, uint64_t counter,
                     uint64_t* counter_dev, int64_t* actions, void* stream);

/* ec_step followed, in the same launch, by the uniform-random policy on the availability mask of the NEW state
 * (requires next_avail): next_actions [B,N] i64 receives exactly what
 *   ec_random_policy(next_avail, B*N, N+1, seed, counter + s, NULL, next_actions, stream)
 * would write, where s is the number of steps this handle has taken since ec_create, this one included (a device-side
 * launch counter without atomics: all envs of a handle step together).  next_actions may alias actions (each lane
 * reads its action before it writes the next one).  The env-only benchmark loop (BASELINE config 2). */
int ec_step_random_policy(ec_handle h, const int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
                          float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail,
                          uint64_t seed, uint64_t counter, int64_t* next_actions, void* stream);

/* n_steps consecutive ec_step_random_policy calls in ONE launch: the env loop `while not done: a = random over
 * avail; env.step(a)` of the reference (the uniform-random policy of action_selectors.py:6-8 driving
 * edge_computing_env.py:84-143), e.g. one call per episode.  Results are identical to calling
 * ec_step_random_policy n_steps times with next_actions fed back as actions (the parity tests compare the two);
 * what it saves is the launch boundary: every block carries its envs through all steps, so there is no gap
 * between steps and the blocks do not hit memory in lock step.
 *   keep_history = 0: every step overwrites the same [B,...] outputs (what the n_steps calls would leave behind);
 *                     next_actions may alias actions.
 *   keep_history = 1: reward [n_steps,B], terminated [n_steps,B], counters [n_steps,B,4], completion_time
 *                     [n_steps,B], next_obs [n_steps,B,N,9], next_avail [n_steps,B,N,N+1] and next_actions
 *                     [n_steps,B,N] receive one slice per step (slice t = the outputs of step t; next_actions[t] is
 *                     the action taken at step t+1).  NULL outputs are skipped; next_avail and next_actions are
 *                     required.  hop_hist (cumulative) is copied once, after the last step.
 * The caller keeps n_steps within the episode (the envs are not reset inside). */
int ec_rollout_random_policy(ec_handle h, int32_t n_steps, int32_t keep_history, const int64_t* actions, float* reward,
                             uint8_t* terminated, int32_t* counters, float* completion_time, int32_t* hop_hist,
                             float* next_obs, int32_t* next_avail, uint64_t seed, uint64_t counter,
                             int64_t* next_actions, void* stream);

/* n_steps environment steps in ONE launch with the Actor acting inside the kernel: per step the rollout of
 * ParallelRunner.run (runners/parallel_episode_runner.py:58-104) — agent.select_actions on the observation / mask
 * (ppo_agent.py:60-74: Linear 9+N -> 64 -> 64 -> N+1 on [obs | one-hot agent], mask to -1e10, softmax, draw) followed by
 * env.step and the next observation — without leaving the launch.  The actions equal ec_policy_forward's on the
 * same weights, observation, mask and uniforms bit for bit, the env outputs ec_step's (the parity tests compare
 * the per-step launches with this call).
 *   w1 [64, 9+N], b1 [64], w2 [64,64], b2 [64], w3 [N+1, 64], b3 [N+1]   row-major fp32, as ec_policy_forward
 *   uniforms [n_steps, B*N] in [0,1) for the inverse-CDF draw of step t, or NULL: greedy (first maximum)
 *   obs0 [B,N,9], avail0 [B,N,N+1]   the observation the first action is chosen on (ec_observe)
 *   actions [n_steps,B,N] i64 (out: the action taken at step t), reward [n_steps,B], terminated [n_steps,B],
 *   counters [n_steps,B,4], completion_time [n_steps,B], next_obs [n_steps,B,N,9], next_avail [n_steps,B,N,N+1]:
 *   slice t = outputs of step t; actions, next_obs, next_avail are required, the others may be NULL.  hop_hist
 *   (cumulative) is copied once, after the last step.
 * Compiled for k <= 4 parallel execution slots and either in-degree <= 8 with N+1 <= 32 actions or in-degree <= 4 with
 * N+1 <= 64 actions (the 40-node topology of BASELINE configs[4]); EC_EINVAL otherwise (callers
 * keep the per-step launches for those shapes). */
int ec_rollout_mlp_policy(ec_handle h, int32_t n_steps, const float* w1, const float* b1, const float* w2, const float* b2,
                          const float* w3, const float* b3, const float* uniforms, const float* obs0,
                          const int32_t* avail0, int64_t* actions, float* reward, uint8_t* terminated, int32_t* counters,
                          float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail, void* stream);

/* ec_rollout_mlp_policy for the expert-guided rollouts of GAILRunner / DGAILRunner (runners/gail_runner.py:72-98,
 * runners/dgail_runner.py:46-94): at step t env b executes the expert's decision (ec_expert_policy's, evaluated inside the
 * launch on the state step t-1 left) iff coin[t][b] < ratios[t], else the Actor's draw.
 *   coin [n_steps,B] f32 uniform draws, ratios [n_steps] f32 (device);
 *   actions [n_steps,B,N] <- the AGENT's draw of every step (what the runners file into the PPO buffer, :73-77 / :81),
 *   expert_actions [n_steps,B,N] <- the expert's decision of every step (ec_expert_append_episode files the expert-driven
 *   ones); everything else as ec_rollout_mlp_policy.  Equals n_steps rounds of ec_policy_forward, ec_expert_policy,
 *   select, ec_step bit for bit (tests/test_rollout_gpu.py). */
int ec_rollout_mlp_expert_policy(ec_handle h, int32_t n_steps, const float* w1, const float* b1, const float* w2,
                                 const float* b2, const float* w3, const float* b3, const float* uniforms, const float* obs0,
                                 const int32_t* avail0, const float* coin, const float* ratios, int64_t* actions,
                                 int64_t* expert_actions, float* reward, uint8_t* terminated, int32_t* counters,
                                 float* completion_time, int32_t* hop_hist, float* next_obs, int32_t* next_avail,
                                 void* stream);

/* The expert of the GAIL / DGAIL runners for every env of the handle: OptimalAgent.get_current_optimal_action
 * (optimal/optimal_agent.py:102-120) = _conservative_action_selection at conservativeness 1.0 (:166-209) with the
 * reliability / time-margin / load / efficiency scores of :211-375, evaluated on the device state instead of walking
 * env.edge_nodes[i] Python objects.  avail [B,N,N+1] i32 is the mask of the same state (ec_observe / ec_step);
 * actions [B,N] i64. */
int ec_expert_policy(ec_handle h, const int32_t* avail, int64_t* actions, void* stream);

/* Device status word (EC_ST_* bits, sticky until cleared). Synchronises the stream. */
int ec_status(ec_handle h, uint32_t* status_host, int clear, void* stream);

/* Parity-test helpers. Synchronous.
 * ec_debug_dump: every live task of env `env` as rows of 11 doubles
 *   [node, qtype, pos, size, cycles, hop, ewt, cet, twt, ctt, bwt]  (timers in mini-slot ticks)
 *   qtype: 0 new_task, 1 buffer_queue, 2 execution_queue, 3 executing_queue, 4+src receiving_queues[src]
 * rows_host holds max_rows rows; *n_rows_host receives the number written (or needed, if larger). */
int ec_debug_dump(ec_handle h, int64_t env, double* rows_host, int64_t max_rows, int64_t* n_rows_host);
/* words consumed so far from each MT19937 stream: cursors_host [B,N,2] (task, failure) u64 */
int ec_debug_rng_cursors(ec_handle h, uint64_t* cursors_host);
/* total live tasks over all envs (for the roofline's L-bar) */
int ec_debug_live_tasks(ec_handle h, int64_t* total_host);
/* geometry chosen by ec_create: {B, N, k, D(max in-degree), cap_buffer, cap_recv, cap_wait, max_ticks,
 * threads_per_block, envs_per_block, state_bytes_lo, state_bytes_hi, lds_bytes_per_block, envs_per_wave,
 * main_waves, rng_producer_waves, rollout_kernel_resident, rollout_mlp_policy_available}; the first n (<= 18)
 * are written */
int ec_info(ec_handle h, int64_t* info_host, int32_t n);

/* --- GAE: replaces utils/advantage_utils.py:4-9 ------------------------------------------------ */
/* r, v, nv, m, adv: [B,T,N] f32 contiguous (the reference's trailing unit dim is a view).
 * Evaluation order is the reference's, without FMA contraction, so results are bit-identical to
 * the fp32 torch loop. */
int ec_gae(const float* r, const float* v, const float* nv, const float* m, float* adv, int64_t B, int64_t T,
           int64_t N, float gamma, float gamma_lambda, void* stream);

/* --- small fp32 MLPs of the rollout / reward path ----------------------------------------------------------
 * Weights are torch nn.Linear layout [out][in] f32 contiguous on the device; hidden must be 64
 * (ppo_config.py:6 / dgail_config.py hidden_dim).  Floating point: results agree with torch fp32 to 1e-5
 * relative (different summation order), not bitwise. */

/* Actor forward for `rows` = B*n_agents rows (row r is agent r % n_agents): replaces
 * rl/policy_gradient_rl/ppo/ppo_network.py:21-25 + ppo_agent.py:61-74 (and dgail_network.py:6-19 +
 * dgail_agent.py:114-126): input = (obs[9] | one-hot agent id), masked logits (-1e10 where avail == 0),
 * softmax, and the action: inverse-CDF draw with the caller's uniforms[rows] in [0,1) (multinomial of
 * action_selectors.py:11-16 with the randomness injected) or, uniforms == NULL, the first arg-max
 * (action_selectors.py:18-20).  avail may be NULL (no mask); logits/probs [rows,n_actions] and
 * actions [rows] i64 are each optional (NULL to skip).  n_actions <= 64. */
int ec_policy_forward(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                      const float* b3, const float* obs, const int32_t* avail, const float* uniforms, int64_t rows,
                      int32_t n_agents, int32_t n_actions, int32_t hidden, float* logits, float* probs,
                      int64_t* actions, void* stream);

/* GAIL discriminator reward: replaces dgail_network.py:37-50 + gail_agent.py / dgail_agent.py:482-485,501-518.
 *   p = sigmoid(MLP(obs | one-hot action)); out = -log(1 - clamp(p, 1e-8, 1-1e-8)) * weight
 *       (+ env_coef * env_reward[r] when env_reward != NULL).  prob [rows] optional (unclamped p). */
int ec_gail_reward(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                   const float* obs, const int64_t* actions, const float* env_reward, int64_t rows, int32_t n_actions,
                   int32_t hidden, float weight, float env_coef, float* prob, float* out, void* stream);

/* DGAIL diffusion-discriminator reward: replaces DiffusionDiscriminator.compute_loss for c = expert and
 * c = agent (dgail_network.py:71-108) and dgail_agent.py:487-499.  The reference draws (t, noise) inside
 * compute_loss; here the caller passes them (t_* u8 [rows] in {0,1}, noise_* f32 [rows, 9+n_actions]) so that
 * the result is a function of its inputs.  w1 [64][9+n_actions+128] and w2 are the EFFECTIVE (spectrally
 * normalised) weights, w3 [9+n_actions][64]; time_embed / cond_embed [2][64].  9+n_actions <= 64.
 *   loss_c = mean_d (eps - noise)^2 + r1_lambda * |d sum(eps) / d x_t|^2      (analytic input gradient)
 *   d_phi = e^-Le / (e^-Le + e^-La + 1e-8);  reward = -log(1 - d_phi + 1e-8)
 * loss_expert, loss_agent, d_phi, reward: [rows] f32, each optional. */
int ec_diffusion_reward(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                        const float* b3, const float* time_embed, const float* cond_embed, const float* obs,
                        const int64_t* actions, const uint8_t* t_expert, const float* noise_expert,
                        const uint8_t* t_agent, const float* noise_agent, int64_t rows, int32_t n_actions,
                        int32_t hidden, float r1_lambda, float* loss_expert, float* loss_agent, float* d_phi,
                        float* reward, void* stream);

/* DGAIL diffusion-discriminator update, one compute_loss call (one condition c) with its backward: replaces
 * DiffusionDiscriminator.compute_loss (dgail_network.py:82-108; the R1 term is an autograd.grad with create_graph=True
 * there) and, through `coef`, the loss.backward() of DGAILAgent._update_discriminator (dgail_agent.py:203-243).
 *   w1 [64][ld1] (effective weight; columns [0, 9+n_actions) are the x part), bias_t [2][64] = b1 + W1t E_time[t] +
 *   W1c E_cond[c] for t = 0, 1, w2 [64][64] (effective), w3 [9+n_actions][64], w3sum [64] = column sums of w3;
 *   t u8 [rows] and noise [rows, 9+n_actions] are the draws of dgail_network.py:90-91.
 *   loss [rows] (optional) <- mean_d (eps - noise)^2 + r1_lambda |d sum(eps) / d x_t|^2
 *   grads (optional, ACCUMULATED into; needs coef [rows] = d objective / d loss[r]):
 *       [dW1x 64*(9+n_actions) | dbias_t[0]+dbias_t[1] 64 | dbias_t[1] 64 | dW2 64*64 | db2 64 | dW3 (9+n_actions)*64 |
 *        db3 9+n_actions | dw3sum 64]
 * The chain rule to the module's own parameters (spectral norm, embeddings) is the caller's: a few 64x64 operations.
 * 9+n_actions <= 64 (with 32 or 64 the first layer's gradient tile has no spare input column for the per-t bias sums; the
 * t = 1 share is then summed from the kernel's delta tile). */
int ec_diffusion_grad(const float* w1, int32_t ld1, const float* bias_t, const float* w2, const float* b2, const float* w3,
                      const float* b3, const float* w3sum, const float* obs, const int64_t* actions, const uint8_t* t,
                      const float* noise, const float* coef, int64_t rows, int32_t n_actions, int32_t hidden,
                      float r1_lambda, float* loss, float* grads, void* stream);

/* ec_diffusion_grad with the per-row coefficient taken in the kernel: the derivative of the discriminator objective of
 * dgail_agent.py:209-221 -- D_phi = exp(-lp) / (exp(-lp) + exp(-ln) + 1e-8), F.binary_cross_entropy(D_phi, label), mean over
 * rows -- with respect to this call's per-row loss, from that loss and the other condition's (`other_loss` [rows]).
 *   side 0: this call is lp (condition 1), 1: ln (condition 0);  label: 1 expert rows, 0 agent rows;  scale: 1 / rows.
 * loss (optional) <- this call's per-row loss.  Lets a data set's update run as forward(c=1), forward+backward(c=0),
 * forward+backward(c=1): three launches instead of four. */
int ec_diffusion_grad_bce(const float* w1, int32_t ld1, const float* bias_t, const float* w2, const float* b2,
                          const float* w3, const float* b3, const float* w3sum, const float* obs, const int64_t* actions,
                          const uint8_t* t, const float* noise, const float* other_loss, int32_t side, int32_t label,
                          float scale, int64_t rows, int32_t n_actions, int32_t hidden, float r1_lambda, float* loss,
                          float* grads, void* stream);

/* DGAIL reward from the diffusion discriminator, second half: replaces the agent-condition compute_loss call and the D_phi /
 * reward arithmetic of DGAILAgent._compute_rewards (dgail_agent.py:486-499, 514-516).  Arguments as ec_diffusion_grad for
 * condition c = 0 with the weights THAT call reads (the reference runs one spectral-norm power iteration per call);
 * loss_expert [rows] is the condition-1 loss of the same rows (ec_diffusion_grad, loss only, with that call's weights).
 *   loss_agent [rows] (optional) <- this call's per-row loss ln
 *   reward [rows] <- -log(1 - D + 1e-8) * weight + env_coef * env_reward[r],  D = e^-lp / (e^-lp + e^-ln + 1e-8)
 *   (env_reward may be NULL).  One evaluation per condition: ec_diffusion_reward evaluates both conditions per call. */
int ec_diffusion_reward_from_loss(const float* w1, int32_t ld1, const float* bias_t, const float* w2, const float* b2,
                                  const float* w3, const float* b3, const float* w3sum, const float* obs,
                                  const int64_t* actions, const uint8_t* t, const float* noise, const float* loss_expert,
                                  const float* env_reward, float weight, float env_coef, int64_t rows, int32_t n_actions,
                                  int32_t hidden, float r1_lambda, float* loss_agent, float* reward, void* stream);

/* --- rollout bookkeeping ----------------------------------------------------------------------------------------
 * One launch per env step instead of ~20 torch launches: replaces EpisodeBuffer.insert (buffer/episode_buffer.py:21-28)
 * with the team-reward broadcast and mask = 1 - terminated of parallel_episode_runner.py:101-102, and the running
 * episode statistics of :94-99 (same pattern in gail_runner.py:59-95, dgail_runner.py:66-91).
 *   obs / avail: the env's observation AFTER step t (ec_step's next_obs / next_avail); it is stored as next_obs[:, t]
 *                and, when t + 1 < episode_limit, as obs[:, t+1] / avail_actions[:, t+1].  t = -1 stores the initial
 *                observation as obs[:, 0] / avail_actions[:, 0] and touches nothing else.
 *   actions [B,N] i64 is what the buffer keeps for step t (the agent's action, also under expert guidance).
 *   buf_*: the [B,T,N,.] tensors of the buffer (all NULL: statistics only, test mode).
 *   totals f64[6] += {sum reward, sum completion time, finish, success, drop, failure} over the batch, summed in a
 *   fixed order (deterministic). */
int ec_store_transition(const float* obs, const int32_t* avail, const int64_t* actions, const float* reward,
                        const uint8_t* terminated, const int32_t* counters, const float* completion_time,
                        int64_t n_envs, int32_t n_agents, int32_t obs_dim, int32_t n_actions, int64_t t,
                        int64_t episode_limit, float* buf_obs, int32_t* buf_avail, int64_t* buf_actions,
                        float* buf_rewards, float* buf_masks, float* buf_next_obs, double* totals, void* stream);

/* ec_store_transition for a whole episode at once, from the time-major histories ec_rollout_mlp_policy (or
 * ec_rollout_random_policy with keep_history) wrote: obs0 / avail0 = the observation before the first step; hist_*
 * [T, ...] as documented there (hist_actions[t] = the action taken at step t).  Fills the [B,T,N,.] EpisodeBuffer
 * tensors (buffer/episode_buffer.py:6-28; obs[:,0] = obs0, obs[:,t+1] = next_obs[:,t] = hist_obs[t]) and adds the
 * episode's statistics to totals[6] in the order T ec_store_transition calls would (same bits).  buf_* all NULL:
 * statistics only; totals NULL: buffer only.  partial [T,6] f64 and done (u32, zero on entry, left zero) are
 * caller-provided scratch. */
int ec_store_episode(const float* obs0, const int32_t* avail0, const float* hist_obs, const int32_t* hist_avail,
                     const int64_t* hist_actions, const float* hist_reward, const uint8_t* hist_terminated,
                     const int32_t* hist_counters, const float* hist_completion_time, int64_t n_envs, int32_t n_agents,
                     int32_t obs_dim, int32_t n_actions, int64_t episode_limit, float* buf_obs, int32_t* buf_avail,
                     int64_t* buf_actions, float* buf_rewards, float* buf_masks, float* buf_next_obs, double* totals,
                     double* partial, uint32_t* done, void* stream);

/* ec_store_episode for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE: it is the same launch as ec_store_episode with
 * the `partial` / `done` workspace taken from a static device array inside the library instead of from the caller.
 * It calls no hipMalloc / hipFree or any allocating routine; the caller still owns every input and output.
 *   limit: episode_limit <= EC_STORE_EPISODE_STATIC_MAX_T (EC_EINVAL beyond it; the array is 256 B + MAX_T * 48 B
 *          = 197 KB per GPU the library runs on, held until the process ends)
 *   Every call on one GPU shares that GPU's array (each GPU has its own copy, found once per device with
 *   hipGetSymbolAddress on the first call there): calls that overlap on one GPU — two streams, two threads — race;
 *   calls on one stream are ordered and safe.  The arrival counter is cleared on the stream before each launch.
 *   Outputs are bit-identical to ec_store_episode's. */
#define EC_STORE_EPISODE_STATIC_MAX_T 4096
int ec_store_episode_static(const float* obs0, const int32_t* avail0, const float* hist_obs, const int32_t* hist_avail,
                            const int64_t* hist_actions, const float* hist_reward, const uint8_t* hist_terminated,
                            const int32_t* hist_counters, const float* hist_completion_time, int64_t n_envs,
                            int32_t n_agents, int32_t obs_dim, int32_t n_actions, int64_t episode_limit, float* buf_obs,
                            int32_t* buf_avail, int64_t* buf_actions, float* buf_rewards, float* buf_masks,
                            float* buf_next_obs, double* totals, void* stream);
/* Test hook of ec_store_episode_static (no counterpart in the reference either): the current device's copy of the static
 * scratch as the library looked it up -- address and size -- after filling it with `fill_byte` on `stream` when
 * fill_byte >= 0 (a poisoned arrival counter and stale partial sums for the next call to start from).  Allocates nothing. */
int ec_debug_store_episode_static_scratch(int32_t fill_byte, void** ptr_out, int64_t* bytes_out, void* stream);


/* --- fused PPO loss + gradients ----------------------------------------------------------------------------------
 * One launch per network does the forward pass, the loss and the whole backward pass of the (obs | one-hot agent id)
 * -> 64 -> 64 -> {n_actions, 1} MLP over `rows` = B*T*n_agents rows (row r is agent r % n_agents) and ACCUMULATES
 *   grads  flat f32 [dW1 | db1 | dW2 | db2 | dW3 | db3] in torch's parameter order (zero it first)
 *   loss   f32 [1]: the rank's share of the loss (numerator over the global denominator)
 * Agreement with torch autograd: 1e-4 relative on the gradient (fp32, different summation order, atomics).
 *
 * Actor: replaces evaluate_actions + the clipped-surrogate/entropy objective + backward of
 * rl/policy_gradient_rl/ppo/ppo_agent.py:76-83,123,135-147 (identical in gail_agent.py:109-116,199-226):
 *   logits masked to -1e10 where avail == 0, softmax, Categorical(probs) log-prob / entropy (probabilities clamped to
 *   [eps, 1-eps] as torch does), ratio = exp(logp - old_logp), objective = min(ratio*adv, clip(ratio)*adv)
 *   + entropy_coef * entropy, loss = -sum(objective * mask) * inv_denom.
 *   old_logp == NULL: forward only (logp / entropy out: the no-grad evaluate_actions of :113); grads == NULL: loss only.
 * Critic: replaces ppo_agent.py:124-131: loss = sum(((v - target) * mask)^2) * inv_denom.
 *   target == NULL: forward only (values out: the no-grad critic passes of :114-115). */
int ec_ppo_actor_grad(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                      const float* b3, const float* obs, const int32_t* avail, const int64_t* actions,
                      const float* old_logp, const float* adv, const float* mask, int64_t rows, int32_t n_agents,
                      int32_t n_actions, int32_t hidden, float eps_clip, float entropy_coef, float inv_denom,
                      float* grads, float* loss, float* logp, float* entropy, void* stream);
int ec_ppo_critic_grad(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                       const float* b3, const float* obs, const float* target, const float* mask, int64_t rows,
                       int32_t n_agents, int32_t hidden, float inv_denom, float* grads, float* loss, float* values,
                       void* stream);

/* The same two launches for the minibatch updates of DGAILAgent._update_policy (rl/policy_gradient_rl/dgail/
 * dgail_agent.py:283-325), which differ from PPOAgent.train in four ways the extra arguments cover:
 *   row_index [rows] i64 or NULL: row r of the pass is row row_index[r] of the per-row tables (obs, avail, actions,
 *       old_logp, adv, target, mask) -- a minibatch of the DataLoader's shuffle without gathering it first; the agent
 *       id of a row is (table row) % n_agents; logp / entropy / values are indexed by r.
 *   avail NULL: no action mask (dgail_agent.py:297 evaluates unmasked logits);  mask NULL: all ones.
 *   flags & EC_PPO_LOGITS: torch's Categorical(logits=...) (:298-300): the exact log-softmax for log-prob and
 *       entropy, without Categorical(probs=...)'s [eps, 1-eps] clamp.
 *   critic n_agents = 0: the network reads the raw 9 observation inputs (ValueNet, dgail_network.py:22-34), w1 [64,9].
 * With row_index NULL, flags 0 and avail / mask given they are ec_ppo_actor_grad / ec_ppo_critic_grad. */
/* Backward passes (grads != NULL) of ec_ppo_*_grad[_ex] and ec_gail_disc_grad sum per-block partial gradients in a fixed
 * order through a per-device workspace the library allocates at the first such call (13 MB, kept until the process
 * ends): the same inputs give the same bits.  Passes on one device share the workspace, so they must not overlap
 * (different streams); MARL_AMD_PPO_ATOMICS=1 in the environment switches to an atomic flush without a workspace. */
#define EC_PPO_LOGITS 1
/* flags & EC_PPO_GATHERED: the per-row tables are already in pass order (the caller gathered the minibatch: coalesced
 * reads instead of one random row per lane); row_index then only names each row's agent, (row_index[r]) % n_agents. */
#define EC_PPO_GATHERED 2
/* flags & EC_PPO_NLL: the objective is the log-probability of `actions` itself, i.e. loss = -sum mask * logp * inv_denom
 * (F.cross_entropy of the behaviour-cloning term, dgail_agent.py:255-262); old_logp / adv / eps_clip / entropy_coef
 * are not read. */
#define EC_PPO_NLL 4
/* flags & EC_PPO_FIRST_EPOCH: epoch 0 of ppo_agent.py:96-141 -- the parameters are still the ones the rollout acted with,
 * so the "old" log-prob of a row is this pass's own (ratio = exp(0) = 1 exactly, as the reference's first epoch computes
 * it): old_logp must be NULL, adv and logp are required, and logp [rows] receives the log-probs the later epochs pass
 * back as old_logp.  Saves the forward-only launch that would produce them. */
#define EC_PPO_FIRST_EPOCH 8
int ec_ppo_actor_grad_ex(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                         const float* b3, const float* obs, const int32_t* avail, const int64_t* actions,
                         const float* old_logp, const float* adv, const float* mask, int64_t rows, int32_t n_agents,
                         int32_t n_actions, int32_t hidden, float eps_clip, float entropy_coef, float inv_denom,
                         float* grads, float* loss, float* logp, float* entropy, const int64_t* row_index, int32_t flags,
                         void* stream);
int ec_ppo_critic_grad_ex(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3,
                          const float* b3, const float* obs, const float* target, const float* mask, int64_t rows,
                          int32_t n_agents, int32_t hidden, float inv_denom, float* grads, float* loss, float* values,
                          const int64_t* row_index, void* stream);

/* GAIL discriminator step of DGAIL, one data set per call: replaces GAILDiscriminator.forward (dgail_network.py:37-50) on
 * (obs | one-hot action), the loss terms of dgail_agent.py:425-431 and their autograd backward.
 *   expert != 0: loss += -inv_denom * sum log(p + 1e-8);   expert == 0: loss += -inv_denom * sum log(1 - p + 1e-8)
 * grads (optional, accumulated into): flat [dW1 | db1 | dW2 | db2 | dW3 | db3] in torch parameter order, w1 [64, 9+n_actions];
 * prob (optional) [rows] <- p = sigmoid(MLP).  9 + n_actions <= 64. */
int ec_gail_disc_grad(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                      const float* obs, const int64_t* actions, int64_t rows, int32_t n_actions, int32_t hidden,
                      int32_t expert, float inv_denom, float* grads, float* loss, float* prob, void* stream);

/* Expert mixing of the guided runners, one env step: replaces the per-env coin test, the action select and the append of
 * expert-driven envs' (observation, expert action) to the FIFO of expert transitions (gail_runner.py:72-98,
 * dgail_runner.py:46-94, gail_agent.py:77-83 / dgail_agent.py:135-149).
 *   env b is expert-driven iff coin[b] < *ratio (ratio: device scalar);  exec_actions[b] = expert or agent actions;
 *   selected envs, in env order, go to ring slots (*written + rank) % capacity of ring_obs [capacity(+),N,O] /
 *   ring_actions [capacity(+),N]; when one call selects more than `capacity` envs the newest `capacity` are kept;
 *   *written += number selected;  *n_expert (optional) += the same.  Two launches, no host synchronisation. */
int ec_expert_mix(const float* coin, const float* ratio, const int64_t* expert_actions, const int64_t* agent_actions,
                  const float* obs, int64_t n_envs, int32_t n_agents, int32_t obs_dim, int64_t capacity,
                  int64_t* exec_actions, float* ring_obs, int64_t* ring_actions, int64_t* written, int64_t* n_expert,
                  void* stream);

/* The FIFO append of ec_expert_mix for a whole episode of ec_rollout_mlp_expert_policy: for t = 0 .. n_steps-1, in env
 * order, every env with coin[t][b] < ratios[t] appends (the observation step t acted on: obs0 for t = 0, else
 * obs_hist[t-1]; expert_actions[t][b]) -- the ring ends up as n_steps ec_expert_mix calls (or store_expert_data calls,
 * gail_agent.py:77-83) would leave it.  *written / *n_expert advance by the number appended.
 *   workspace: ec_expert_append_episode_workspace(n_steps, n_envs) bytes of device memory (contents irrelevant).
 * Three launches, no host synchronisation. */
int64_t ec_expert_append_episode_workspace(int32_t n_steps, int64_t n_envs);
int ec_expert_append_episode(const float* coin, const float* ratios, const int64_t* expert_actions, const float* obs0,
                             const float* obs_hist, int32_t n_steps, int64_t n_envs, int32_t n_agents, int32_t obs_dim,
                             int64_t capacity, float* ring_obs, int64_t* ring_actions, int64_t* written, int64_t* n_expert,
                             void* workspace, void* stream);

/* torch.nn.utils.clip_grad_norm_(params, max_norm) followed by torch.optim.Adam.step() (no weight decay, no amsgrad) for
 * up to 4 small networks of up to 16 fp32 tensors each in one launch, one block per network: the end of every optimiser
 * step of the learners (dgail_agent.py:319-325, ppo_agent.py:128-141, gail_agent.py:222-231), ~40 torch launches otherwise.
 *   n_tensors [n_nets];  params / grads / exp_avg / exp_avg_sq / steps: device pointers, the networks' tensors one after
 *   the other (sum n_tensors entries), numel likewise;  steps: the optimiser's device step counters (capturable Adam;
 *   incremented by the launch), or NULL entries for a network whose step count the host keeps: step_host [n_nets] then holds
 *   the NEW step count;  lr, beta1, beta2, eps, max_norm [n_nets] (host doubles, rounded to fp32 where torch rounds
 *   them: 1 - beta in double first; max_norm <= 0: no clipping).
 * grads are left scaled by the clip coefficient, as clip_grad_norm_ leaves them. */
int ec_adam_clip_step(int32_t n_nets, const int32_t* n_tensors, float* const* params, float* const* grads,
                      float* const* exp_avg, float* const* exp_avg_sq, float* const* steps, const int64_t* numel,
                      const double* step_host, const double* lr, const double* beta1, const double* beta2,
                      const double* eps, const double* max_norm, void* stream);

/* The tail of a GAIL-discriminator layer -- BatchNorm1d in training mode, ReLU, Dropout(p) (gail_network.py:97-125:
 * fc -> bn -> activation -> dropout) -- over x [rows, C] fp32, C a multiple of 4 with C/4 dividing 256, with the batch
 * statistics folded into scale = invstd * weight and shift = bias - mean * scale by the caller:
 *   fwd:      out = relu(x * scale + shift) * keep / (1 - p)
 *   bwd_sums: partials [n_partials][2][C] <- per-block sums over rows of dy and of dy * x,
 *             dy = dout * keep / (1 - p) * [x * scale + shift > 0]   (the caller adds the partials up)
 *   bwd_dx:   dx = a * dy + b * x + c   with per-column a, b, c (the training-mode BatchNorm backward, assembled by the
 *             caller from the sums and the statistics)
 * keep: `mask` [rows, C] u8 when given, else a counter-based Bernoulli(1 - p) draw from (seed, element index) that the
 * backward recomputes instead of reading.  Replaces F.batch_norm / relu / F.dropout and their autograd backward. */
/* y = x W^T + b for a tall batch (the same layers' forward, and with w transposed by the caller their input gradient):
 * x [rows, k_in <= 128], w [c_out <= 128, k_in], b [c_out] or NULL, fp32; one read of x, y written in whole 128-byte lines
 * (rows as the MFMA's M dimension).  Replaces F.linear for these shapes. */
int ec_linear_fwd(const float* x, const float* w, const float* b, int64_t rows, int32_t k_in, int32_t c_out, float* y,
                  void* stream);
/* ec_linear_fwd that also leaves the batch statistics of y for a BatchNorm that follows (gail_network.py:97-125: fc -> bn):
 * partials [n_partials <= 1024][2][c_out] <- per-block column sums of (y - pivot) and (y - pivot)^2, the input of
 * ec_bn_finalize with x_row0 = pivot; pivot [c_out] = a row of y (the caller's one-row call).  Saves the statistics pass.
 * y = NULL: the statistics only, nothing of y written (for a layer whose output ec_linear2_fwd_bn recomputes). */
int ec_linear_fwd_stats(const float* x, const float* w, const float* b, int64_t rows, int32_t k_in, int32_t c_out, float* y,
                        const float* pivot, int32_t n_partials, float* partials, void* stream);
/* The same with the previous layer's BatchNorm -> ReLU -> Dropout tail applied on the way in:
 * y = (relu(x * scale + shift) * keep / (1 - p)) W^T + b, keep = ec_bn_relu_dropout_fwd's counter-based draw for `seed`
 * (element index row * k_in + col), 0 <= p < 1.  The activated [rows, k_in] tensor is never written: for passes that
 * keep nothing for a backward (gail_agent.py:165-167, the reward pass under no_grad with the network in training mode).
 * partials: NULL, or [n_partials <= 1024][2][c_out] <- per-block column sums of (y - b) and (y - b)^2: the input of
 * ec_bn_finalize for the next layer's BatchNorm with x_row0 = b (its statistics pass, without reading y again); needs b. */
int ec_linear_fwd_bn(const float* x, const float* scale, const float* shift, uint64_t seed, float p, const float* w,
                     const float* b, int64_t rows, int32_t k_in, int32_t c_out, float* y, int32_t n_partials, float* partials,
                     void* stream);
/* Layers 1 and 2 of the GAIL discriminator (gail_network.py:97-125: fc1 -> bn1 -> relu -> dropout1 -> fc2) in one kernel,
 * for the reward pass (gail_agent.py:165-170: training mode under no_grad, nothing kept for a backward):
 *   y [rows, c_out = 64] = (relu((x W1^T + b1) * scale + shift) * keep / (1 - p)) W2^T + b2
 * x [rows, k_in <= 96], w1 [c_mid = 128, k_in], w2 [64, 128], b1 / b2 or NULL; scale, shift [128] of bn1 from a
 * statistics-only pass (ec_linear_fwd_stats with y = NULL, ec_bn_finalize); keep as in ec_linear_fwd_bn (element index
 * row * 128 + col).  Layer 1's [rows, 128] output -- two thirds of the two layers' memory traffic -- is neither written
 * nor read back.  y equals ec_linear_fwd followed by ec_linear_fwd_bn bit for bit; partials as in ec_linear_fwd_bn
 * ([n_partials <= 1024][2][64] about b2).  Other widths: EC_EINVAL (the caller takes the layer-by-layer kernels). */
int ec_linear2_fwd_bn(const float* x, const float* w1, const float* b1, const float* scale, const float* shift, uint64_t seed,
                      float p, const float* w2, const float* b2, int64_t rows, int32_t k_in, int32_t c_mid, int32_t c_out,
                      float* y, int32_t n_partials, float* partials, void* stream);

/* First and second moments of the rows of x [rows, k_in <= 96] about its first row, the row index as the MFMA's K
 * dimension: partials [n_partials <= 4096][KP * KP + KP], KP = 32 ceil(k_in / 32) <- per block, sum over its rows of d d^T
 * (KP x KP row-major, zero beyond k_in), then sum of d, with d = x[row] - x[0].  The batch statistics of fc1's output for
 * bn1 (gail_network.py:97-125) follow from them without computing that output: sum(y - y0) = W sum(d) and
 * sum((y - y0)^2)[o] = w_o^T (sum d d^T) w_o.  Replaces the statistics pass over fc1(x) in the reward pass. */
int ec_row_moments(const float* x, int64_t rows, int32_t k_in, int32_t n_partials, float* partials, void* stream);

/* The GAIL discriminator's input rows (gail_agent.py:131-141, gail_network.py:127-131: obs with the agent's one-hot id,
 * concatenated with the one-hot action): out [rows, obs_dim + n_agents + n_actions] fp32 from obs [rows, obs_dim] fp32 and
 * actions [rows] int64, rows in [.., agent] order (row r belongs to agent r % n_agents).  One write of the rows instead of
 * two torch.cat passes and a one_hot. */
int ec_disc_rows(const float* obs, const int64_t* actions, int64_t rows, int32_t obs_dim, int32_t n_agents, int32_t n_actions,
                 float* out, void* stream);

/* Weight and bias gradient of a Linear layer over a tall batch (gail_network.py:97-125, the discriminator's layers on
 * 4.1 M .. 65.5 M rows): dW [c_out, k_in] = dy^T x, db [c_out] = sum_rows dy for dy [rows, c_out], x [rows, k_in] fp32,
 * c_out in {32, 64, 96, 128}, k_in <= 128.  partials [n_partials][c_out * k_in + c_out] <- per-block sums (dW row-major,
 * then db); the caller adds them up.  One read of dy and x, the row index as the MFMA's K dimension.  Replaces
 * dy.t() @ x and dy.sum(0) of torch's Linear backward. */
int ec_linear_wgrad(const float* dy, const float* x, int64_t rows, int32_t c_out, int32_t k_in, int32_t n_partials,
                    float* partials, void* stream);

/* The batch statistics of the same layer in one streaming pass: partials [n_partials][2][C] <- per-block sums over rows of
 * (x - x[0]) and (x - x[0])^2 per column (row 0 as the pivot); the caller adds the partials up (in float64) and takes
 * mean = x[0] + S1/n, biased var = S2/n - (S1/n)^2.  Replaces torch.var_mean(x, dim=0). */
int ec_bn_col_stats(const float* x, int64_t rows, int32_t C, int32_t n_partials, float* partials, void* stream);
/* The per-column arithmetic between those passes, one launch each (float64 sums over the partials):
 *   ec_bn_finalize: partials of ec_bn_col_stats (+ x_row0 = x[0], its pivot) -> mean, biased var, invstd = rsqrt(var + eps),
 *     scale = invstd * weight, shift = bias - mean * scale [C]; running_mean / running_var (optional) updated as
 *     nn.BatchNorm1d does in training mode (momentum < 0: cumulative average with weight 1 / *num_batches_tracked, which the
 *     caller has advanced for this batch);
 *   ec_bn_bwd_coeffs: partials of ec_bn_relu_dropout_bwd_sums -> a, b, c of ec_bn_relu_dropout_bwd_dx and
 *     dweight = sum dy * xhat, dbias = sum dy. */
int ec_bn_finalize(const float* partials, int32_t n_partials, const float* x_row0, int64_t rows, int32_t C, const float* weight,
                   const float* bias, float eps, float momentum, float* running_mean, float* running_var,
                   const int64_t* num_batches_tracked, float* mean, float* var, float* invstd, float* scale, float* shift,
                   void* stream);
int ec_bn_bwd_coeffs(const float* partials, int32_t n_partials, const float* mean, const float* invstd, const float* weight,
                     int64_t rows, int32_t C, float* a, float* b, float* c, float* dweight, float* dbias, void* stream);
int ec_bn_relu_dropout_fwd(const float* x, const float* scale, const float* shift, const uint8_t* mask, uint64_t seed, float p,
                           int64_t rows, int32_t C, float* out, void* stream);
int ec_bn_relu_dropout_bwd_sums(const float* dout, const float* x, const float* scale, const float* shift, const uint8_t* mask,
                                uint64_t seed, float p, int64_t rows, int32_t C, int32_t n_partials, float* partials,
                                void* stream);
int ec_bn_relu_dropout_bwd_dx(const float* dout, const float* x, const float* scale, const float* shift, const uint8_t* mask,
                              uint64_t seed, float p, const float* a, const float* b, const float* c, int64_t rows, int32_t C,
                              float* dx, void* stream);

const char* ec_last_error(void);
const char* ec_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MARL_AMD_H */
