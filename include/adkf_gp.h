/*
 * adkf_gp.h - C ABI of libadkf_gp.so: the MI355X (gfx950) batched exact-GP / IFT-hypergradient path of
 * ADKF-IFT's inner loop.
 *
 * The reference (Wenlin-Chen/ADKF-IFT) is pure Python and has NO foreign-function interface for this
 * path: every GP operation is a GPyTorch/BoTorch call made from fs_mol/models/adaptive_dkt.py and
 * fs_mol/utils/adaptive_dkt_utils.py.  Each entry point below names the reference call site(s) it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions:
 *   - plain `extern "C"`, no exceptions, no ownership transfer: the caller allocates every input, output
 *     and the workspace (device memory, e.g. torch tensors' data_ptr()) and passes raw pointers + sizes;
 *   - all matrices are contiguous row-major float32: Z_s [T, ns_max, d], Z_q [T, nq_max, d],
 *     y_s [T, ns_max], y_q [T, nq_max], phi [T, 3], priors [T, 4];  per-task true sizes n_s[T], n_q[T]
 *     (int32, device; NULL = every task uses ns_max / nq_max).  Rows >= n are padding and are ignored on
 *     input and written as 0 on output;
 *   - phi[t] = (raw_noise, raw_outputscale, raw_lengthscale) in the reference's gp_params() order
 *     (fs_mol/models/adaptive_dkt.py:81-86); noise = softplus(raw)+1e-4, outputscale / lengthscale =
 *     softplus(raw);
 *   - priors[t] = (noise_loc, noise_scale, ls_loc, ls_scale): LogNormal priors of
 *     fs_mol/models/adaptive_dkt.py:94-100,112-119; a scale <= 0 disables that prior (DKLModel has no
 *     noise prior, fs_mol/models/dkl.py:86; use_lengthscale_prior=False);
 *   - every function is ASYNCHRONOUS on `stream` (a hipStream_t passed as void*), re-entrant across
 *     streams, and performs no allocation; no synchronisation either, with ONE documented exception: the
 *     multi-launch fits (more than 128 points; ARD) poll a convergence counter every few evaluations in
 *     convergence mode - never in exact-evals mode and never while the stream is being captured - so every
 *     sequence may be captured into a hipGraph.  State kept across calls: a thread-local last-HIP-error
 *     (adkf_last_hip_error), the outcome of the dynamic-LDS opt-ins (asked once per process; see
 *     adkf_path_info) and read-once environment switches for experiments (ADKF_* in DESIGN.md section 1);
 *     nothing a call leaves behind changes what a later call computes;
 *   - return value: 0 = enqueued, < 0 = rejected argument (ADKF_E_*).  Numerical failure is reported
 *     per task in the device array info[T]: 0 = ok, k > 0 = the k-th pivot of a Cholesky factorisation was
 *     not positive (the reference's NotPSDError after jitter retries; this library never adds jitter),
 *     with ADKF_INFO_OUTER_BASE added when it was the N_q x N_q predictive covariance.
 *     adkf_check_info() synchronises and folds info[] into one status (index+1 of the first bad task).
 */
#ifndef ADKF_GP_H
#define ADKF_GP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADKF_KERNEL_RBF 0      /* gpytorch ScaleKernel(RBFKernel)            fs_mol/utils/gp_utils.py:26-27 */
#define ADKF_KERNEL_MATERN52 1 /* gpytorch ScaleKernel(MaternKernel nu=2.5)  fs_mol/utils/gp_utils.py:29-30 */

#define ADKF_E_BADARG (-1)
#define ADKF_E_SIZE (-2)      /* a dimension is outside what this build supports (see adkf_max_points) */
#define ADKF_E_WORKSPACE (-3) /* workspace too small */
#define ADKF_E_LAUNCH (-4)    /* the HIP runtime refused a launch */

#define ADKF_INFO_OUTER_BASE 100000

/* adkf_batch_t.flags: promises that let consecutive calls on the SAME batch (same pointers, shape, kernel) and the
 * SAME workspace skip work that is already there.  The usual meta-step sequence is
 *   adkf_init_params (flags 0) -> adkf_fit (REUSE_DIST) -> adkf_ift_hypergrad (REUSE_DIST | REUSE_INNER). */
#define ADKF_BATCH_REUSE_DIST 1  /* squared distances (incl. query blocks when Z_q was given) are in the workspace */
#define ADKF_BATCH_REUSE_INNER 2 /* A^-1, alpha and the scalars of exactly this phi are in the workspace (adkf_fit
                                    leaves them for its result; adkf_mll_value_grad for its argument) */

#define ADKF_BATCH_DEFER_REFINE 8 /* adkf_fit only: skip the float64 re-evaluation of ill-conditioned tasks at phi* (csrc/refine64.h) - the
                                    caller goes on to adkf_ift_hypergrad / adkf_outer_nll_value_grad / adkf_predict with REUSE_INNER on
                                    this workspace, which redo those tasks in float64 themselves; f_final / gnorm of such tasks are
                                    then the float32 values */

#define ADKF_BATCH_LG_UNFUSED 16 /* more than 128 points only, A/B runs and tests: the blocked sweep as three launches per block step
                                    (k_lg_diag, panel, update: csrc/large.h) instead of the update of step k and the diagonal sweep of
                                    step k + 1 in one launch (csrc/large_fused.h).  Both give bit-identical results. */
#define ADKF_BATCH_LG_FUSED 32   /* ... and the fused form whatever the size (without either flag: fused from 512 points on, where it is faster) */

/* ARD kernel (``use_ard``: fs_mol/models/adaptive_dkt.py:107-108 -> gpytorch ``ard_num_dims``): one lengthscale per
 * feature dimension.  With this flag EVERY phi / g_phi / v argument has h = 2 + d entries per task, laid out
 * (raw_noise, raw_outputscale, raw_lengthscale[0..d)); priors stay [T,4] (the lengthscale prior applies to each
 * dimension and is summed, as gpytorch does); the workspace must have adkf_workspace_bytes_ard() bytes; the h x h
 * Hessian is never formed (pass H = NULL): adkf_ift_hypergrad solves H v = grad f_out by conjugate gradients on
 * closed-form Hessian-vector products.  REUSE_DIST is ignored (distances depend on the lengthscales). */
#define ADKF_BATCH_ARD 4
#define ADKF_CG_DEFAULT_MAXITER 48
#define ADKF_CG_DEFAULT_TOL 1e-6f
#define ADKF_INFO_CG_BASE 200000 /* info = 200000 + k: CG met non-positive curvature at iteration k (H not PD) */

/* flags of adkf_ift_hypergrad: fs_mol/utils/cauchy_hypergradient.py:11-13 */
#define ADKF_IGNORE_GRAD_CORRECTION 1
#define ADKF_IGNORE_DIRECT_GRAD 2

typedef struct adkf_batch {
    int32_t T;          /* tasks */
    int32_t ns_max;     /* padded support rows */
    int32_t nq_max;     /* padded query rows (0 when no query set is involved) */
    int32_t d;          /* feature dimension */
    int32_t kernel;     /* ADKF_KERNEL_* */
    int32_t flags;      /* ADKF_BATCH_* reuse promises (0 = recompute everything) */
    const int32_t* n_s; /* [T] or NULL */
    const int32_t* n_q; /* [T] or NULL */
    const float* Z_s;   /* [T, ns_max, d] */
    const float* y_s;   /* [T, ns_max] */
    const float* Z_q;   /* [T, nq_max, d] or NULL */
    const float* y_q;   /* [T, nq_max] or NULL */
    const float* priors; /* [T, 4] */
} adkf_batch_t;

typedef struct adkf_fit_options {
    int32_t max_evals; /* hard cap on MLL value+gradient evaluations per task (SciPy maxfun) */
    int32_t exact_evals; /* != 0: spend exactly max_evals evaluations (benchmark mode, deterministic work): same stopping
                          * rules, but a converged task re-evaluates at its optimum until the budget is used up */
    float gtol;        /* stop when max|grad| <= gtol          (SciPy L-BFGS-B pgtol, default 1e-5) */
    float ftol;        /* stop when rel. decrease <= ftol      (SciPy factr*eps: 2.22e-9 is the reference setting) */
    void* ev_start;    /* optional hipEvent_t recorded on `stream` immediately before the optimiser kernel */
    void* ev_stop;     /* optional hipEvent_t recorded immediately after it (roofline timing in bench.py) */
} adkf_fit_options_t;

const char* adkf_version(void);

/* Diagnostics: the HIP runtime's description of the error behind the calling thread's last ADKF_E_LAUNCH. */
const char* adkf_last_hip_error(void);

/* Diagnostics: which kernels a batch padded to (ns_max, nq_max) takes ON THIS DEVICE (needs a GPU: it asks the runtime for the
 * dynamic-LDS opt-ins exactly as the entry points do), as a mask of ADKF_PATH_* bits, or ADKF_E_SIZE.  The library never computes
 * anything else than what the header promises, but two of its fast paths depend on an opt-in the runtime may refuse, and then the
 * slower pipeline runs silently: this is where a caller (bench.py prints it) sees which one is live. */
#define ADKF_PATH_FUSED_OUTER 1    /* 64 < max(ns, nq) <= 128: the outer / hypergradient stage of a task as ONE kernel (csrc/hyper.h, 159 KB of
                                      dynamic LDS); clear: the sixteen-launch pipeline */
#define ADKF_PATH_BLOCKED 2        /* max(ns, nq) > 128: blocked sweep through L2 / HBM (csrc/large.h) */
#define ADKF_PATH_BLOCKED_FUSED 4  /* ... with update k + sweep k + 1 in one launch (csrc/large_fused.h: from 512 points on); clear: the three launches */
#define ADKF_PATH_R64_REGION 8     /* the workspace of this shape carries the float64 region of the ill-conditioned-task path (csrc/refine64.h) */
#define ADKF_PATH_R64_LDS 16       /* that path's inverses run in 128 KB of dynamic LDS; clear: in global memory */
int adkf_path_info(int32_t ns_max, int32_t nq_max);

/* Largest support/query set this build handles in its LDS-resident factorisation. */
int adkf_max_points(void);

/* Bytes of device workspace every call below needs for a batch of this shape: per task 4 (4 ns^2 + 4 nq ns + 3 nq^2) bytes of
 * float32 matrices (+ three 128-row panels beyond 128 points) and, up to 1024 points, the float64 region of the ill-conditioned-task
 * path (csrc/refine64.h; about 1.7 x the float32 part: 1.3 MB per task at 128 points, 82 MB at 1024 - every task has one, any of
 * them may be flagged).  Environment ADKF_R64_MAXN=<points> (read once) lowers the batch size up to which that region is carved;
 * larger batches then stay in float32 whatever their conditioning. */
size_t adkf_workspace_bytes(int32_t T, int32_t ns_max, int32_t nq_max, int32_t d);

/* a3: ADKTModel.compute_median_lengthscale_init (fs_mol/models/adaptive_dkt.py:128-131):
 * l0[t] = sqrt(0.5 * lower_median{ |z_i - z_j|^2 : i<j, > 0 }). */
int adkf_median_lengthscale(const adkf_batch_t* b, float* l0, void* ws, size_t ws_bytes, void* stream);

/* a4: ADKTModel.reinit_gp_params / __create_tail_GP (fs_mol/models/adaptive_dkt.py:88-126) and
 * ExactGPLayer.__init__ (fs_mol/utils/gp_utils.py:14-17): fresh per-task phi and priors.
 * noise0 = 0.1 (classification) or 0.01 (numeric labels); writes phi [T,3], priors [T,4], l0 [T] (nullable). */
int adkf_init_params(const adkf_batch_t* b, int32_t use_numeric_labels, int32_t use_lengthscale_prior,
                     float* phi, float* priors, float* l0, void* ws, size_t ws_bytes, void* stream);

/* a5+a6: -ExactMarginalLogLikelihood of the support set = f_inner (fs_mol/models/adaptive_dkt.py:173-176;
 * DKLModel.compute_loss fs_mol/models/dkl.py:155-157): f_in [T], optional d f_in/d phi [T,3] and
 * d f_in/d Z_s [T, ns_max, d]. */
int adkf_mll_value_grad(const adkf_batch_t* b, const float* phi, float* f_in, float* g_phi, float* dZ_s,
                        int32_t* info, void* ws, size_t ws_bytes, void* stream);

/* a7: botorch.optim.fit.fit_gpytorch_scipy(model.mll) (fs_mol/utils/adaptive_dkt_utils.py:91): minimise
 * f_inner over phi, all tasks at once, on the device (quasi-Newton with the exact analytic gradient).
 * phi [T,3] in/out ([T, 2 + d] for ARD batches); f_final [T], gnorm [T] (max|grad| at the result), n_evals [T] are
 * nullable.  Up to 128 support points the whole optimisation is ONE kernel launch.  Larger sets and ARD batches are a
 * sequence of launches per evaluation; there, in convergence mode (exact_evals == 0, max_evals > 16) and outside
 * stream capture, the call synchronises the stream every 8 evaluations to stop enqueueing once every task has
 * finished - with exact_evals != 0 nothing is ever synchronised. */
int adkf_fit(const adkf_batch_t* b, float* phi, const adkf_fit_options_t* opt, float* f_final, float* gnorm,
             int32_t* n_evals, int32_t* info, void* ws, size_t ws_bytes, void* stream);

/* a8 (eval branch): gp_likelihood(gp_model(x_q)) (fs_mol/models/adaptive_dkt.py:198-203;
 * fs_mol/models/dkl.py:143-151): posterior mean [T, nq_max], variance diag incl. noise [T, nq_max]
 * (nullable) and full covariance incl. noise [T, nq_max, nq_max] (nullable). */
int adkf_predict(const adkf_batch_t* b, const float* phi, float* mean, float* var, float* cov, int32_t* info,
                 void* ws, size_t ws_bytes, void* stream);

/* a8 (training branch) = f_outer (fs_mol/models/adaptive_dkt.py:183-191): joint predictive NLL of the query
 * set, with gradients: f_out [T], g_phi [T,3] (nullable), dZ_s, dZ_q (nullable). */
int adkf_outer_nll_value_grad(const adkf_batch_t* b, const float* phi, float* f_out, float* g_phi, float* dZ_s,
                              float* dZ_q, int32_t* info, void* ws, size_t ws_bytes, void* stream);

/* a9/a10: cauchy_hypergradient / cauchy_hypergradient_jvp (fs_mol/utils/cauchy_hypergradient.py:5-163,
 * cauchy_hypergradient_jvp.py:5-156) at the feature-matrix level, all tasks at once:
 *   dL/dZ = d f_out/dZ - d( v^T grad_phi f_in )/dZ,  v = H^-1 grad_phi f_out,  H = d2 f_in / d phi2.
 * Outputs: f_out [T]; dZ_s [T,ns_max,d]; dZ_q [T,nq_max,d]; g_phi_out [T,3] (what the reference leaves in
 * phi.grad); v [T,3] and H [T,9] (nullable, diagnostics).  One backward pass of the caller's feature
 * extractor with (dZ_s, dZ_q) as cotangents then yields exactly theta.grad of the reference. */
int adkf_ift_hypergrad(const adkf_batch_t* b, const float* phi, int32_t flags, float* f_out, float* dZ_s,
                       float* dZ_q, float* g_phi_out, float* v, float* H, int32_t* info, void* ws,
                       size_t ws_bytes, void* stream);

/* Diagnostic: flagged[t] = 1 if task t took the float64 path (ill-conditioned: pivot ratio of the sweep of A or of
 * Sigma_q above the threshold, csrc/refine64.h) in the last adkf_ift_hypergrad / adkf_outer_nll_value_grad on this
 * workspace, else 0.  Non-ARD batches.  No reference counterpart (GPyTorch has one precision); bench.py --regression
 * reports the fraction (fs_mol/utils/gp_utils.py:17: noise 0.01 for numeric labels is where such tasks come from). */
int adkf_double_path_tasks(const adkf_batch_t* b, int32_t* flagged, void* ws, size_t ws_bytes, void* stream);

/* Workspace size for batches that carry ADKF_BATCH_ARD (a superset of adkf_workspace_bytes). */
size_t adkf_workspace_bytes_ard(int32_t T, int32_t ns_max, int32_t nq_max, int32_t d);

/* adkf_ift_hypergrad for ARD batches with explicit conjugate-gradient controls (the north-star's "HVP + CG"):
 * at most cg_maxiter iterations, stop at |r| <= cg_tol |grad f_out|; cg_iters [T] (nullable) receives the iterations
 * each task used.  Converged tasks drop out of the products; outside stream capture (and for cg_maxiter > 16) the call
 * synchronises the stream every 8 iterations to stop enqueueing once every task has converged. */
int adkf_ift_hypergrad_cg(const adkf_batch_t* b, const float* phi, int32_t flags, int32_t cg_maxiter, float cg_tol,
                          float* f_out, float* dZ_s, float* dZ_q, float* g_phi_out, float* v, int32_t* cg_iters,
                          int32_t* info, void* ws, size_t ws_bytes, void* stream);

/* a1 (message functions of RelationalMP, fs_mol/modules/gnn.py:95-148, for the default depth-1 message MLP): all towers of ALL
 * edge types in one batched GEMM with the source / target node states gathered on the fly:
 *   msgs[e_off(t) + e, h, :] = relu(cat(x[src_e, h, :], x[tgt_e, h, :]) W_t[h] + bias_t[h])   for edge e of edge type t,
 * x [V, H, in], W_t [H, 2 in, out], bias_t [H, out], msgs [E_all, H, out]; the edge types' rows follow one another in msgs in
 * the order of `ets` (e_off = number of edges of the types before); at most 4 edge types.
 * The backward is reproducible to the bit - no floating-point atomics (the reference's scatter ops and PyTorch's index_add_
 * are, on a GPU): d cat[e, h, :] = (d msgs . [msgs > 0]) W[h]^T is written once per edge into dcat [E_all, H, 2 in]; dx [V, H, in]
 * is then the sum over each node's outgoing edges of the first half of d cat and over its incoming edges of the second half, in
 * the order of the two CSR lists (perm_*: edge ids of the concatenated edge list sorted stably by source / target node,
 * rowptr_* [V + 1]); every edge type's dW [H, 2 in, out] and db [H, out] are sums over fixed chunks of its edges (partials in
 * `scratch`, at least adkf_msg_backward_scratch_bytes() bytes), added in a fixed order.  Nothing needs initialising; an edge type
 * without edges gets exact zeros.  msgs = NULL in the backward: d_msgs is already the gradient in front of the ReLU
 * (adkf_pna_aggregate_backward_relu below) and no mask is applied. */
typedef struct adkf_msg_et {
    const int64_t* src; /* [E] source node of every edge */
    const int64_t* tgt; /* [E] target node */
    const float* W;     /* [H, 2 in, out] */
    const float* bias;  /* [H, out] (forward) */
    float* dW;          /* [H, 2 in, out] (backward: output) */
    float* db;          /* [H, out]       (backward: output) */
    int32_t E;
} adkf_msg_et_t;
int adkf_msg_forward(const float* x, const adkf_msg_et_t* ets, int32_t n_et, int32_t H, int32_t in, int32_t out, float* msgs,
                     void* stream);
size_t adkf_msg_backward_scratch_bytes(const adkf_msg_et_t* ets, int32_t n_et, int32_t H, int32_t in, int32_t out);
int adkf_msg_backward(const float* x, const adkf_msg_et_t* ets, int32_t n_et, int32_t H, int32_t in, int32_t out, const float* msgs,
                      const float* d_msgs, const int64_t* perm_src, const int64_t* rowptr_src, const int64_t* perm_tgt,
                      const int64_t* rowptr_tgt, int32_t V, float* dcat, float* dx, void* scratch, size_t scratch_bytes,
                      void* stream);

/* a1 (per-graph pooling of CombinedGraphReadout, fs_mol/modules/graph_readout.py:119-177: the weighted-mean head's
 * scatter_softmax + index_add_ :238-252, the weighted-sum head's sigmoid weights :236, the max pooler's scatter :289) between
 * the node-level MLPs and the combination layers, every per-graph sum in the fixed order of the graph's node list:
 *   w_mean = segment-softmax(s_mean), g_mean[g] = sum_v w_mean[v] v_mean[v];  w_sum = sigmoid(s_sum), g_sum likewise;
 *   g_max[g] = max_v emb[v] (0 and argmax -1 for a graph without nodes; first maximum in list order).
 * s_* [V, nh], v_* [V, nh, hd], emb [V, D]; perm [V] node ids sorted stably by graph, rowptr [G + 1]; nh <= 64.  Outputs w_*
 * [V, nh] (kept for the backward), g_mean / g_sum [G, nh hd], g_max / argmax [G, D].  The backward writes every element of
 * d_s_* [V, nh], d_v_* [V, nh hd], d_emb [V, D] exactly once (no scatter). */
int adkf_readout_pool(const float* s_mean, const float* v_mean, const float* s_sum, const float* v_sum, const float* emb,
                      const int64_t* perm, const int64_t* rowptr, int32_t V, int32_t G, int32_t nh, int32_t hd, int32_t D,
                      float* w_mean, float* w_sum, float* g_mean, float* g_sum, float* g_max, int32_t* argmax, void* stream);
int adkf_readout_pool_backward(const float* v_mean, const float* v_sum, const float* w_mean, const float* w_sum,
                               const float* g_mean, const int32_t* argmax, const int64_t* node_to_graph, const float* dg_mean,
                               const float* dg_sum, const float* dg_max, int32_t V, int32_t G, int32_t nh, int32_t hd, int32_t D,
                               float* d_s_mean, float* d_v_mean, float* d_s_sum, float* d_v_sum, float* d_emb, void* stream);

/* The same pooling taken BEFORE the last layer of the two value MLPs (graph_readout.py:219-223: _transformation_mlp = Linear . ReLU .
 * Linear; :242-252 pools its output).  Pooling is linear in that last layer, so
 *     g[g, h, :] = W2[h] p[h, g, :] + b2[h] wtot[g, h],   p[h, g, :] = sum_v w[v, h] r_v,   wtot[g, h] = sum_v w[v, h],
 * with r_v [K] the hidden activations: the caller multiplies nh small [G, K] x [K, hd] products instead of [V, K] x [K, nh hd].
 * h_mean / h_sum: rows of K floats with row stride ldh (column blocks of one activation tensor), K <= 1024; p_* [nh, G, K];
 * wtot_mean (1, or 0 for an empty graph) and wtot_sum [G, nh]; everything else as adkf_readout_pool.  The backward takes dp_*
 * [nh, G, K], dwtot_sum [G, nh], dg_max and writes d_s_* [V, nh], d_h_* [V, K] (contiguous), d_emb [V, D], every element once. */
int adkf_readout_pool_hidden(const float* s_mean, const float* h_mean, const float* s_sum, const float* h_sum, int32_t ldh,
                             const float* emb, const int64_t* perm, const int64_t* rowptr, int32_t V, int32_t G, int32_t nh,
                             int32_t K, int32_t D, float* w_mean, float* w_sum, float* p_mean, float* p_sum, float* wtot_mean,
                             float* wtot_sum, float* g_max, int32_t* argmax, void* stream);
int adkf_readout_pool_hidden_backward(const float* h_mean, const float* h_sum, int32_t ldh, const float* w_mean, const float* w_sum,
                                      const int32_t* argmax, const int64_t* perm, const int64_t* rowptr, const float* dp_mean,
                                      const float* dp_sum, const float* dwtot_sum, const float* dg_max, int32_t V, int32_t G,
                                      int32_t nh, int32_t K, int32_t D, float* d_s_mean, float* d_h_mean, float* d_s_sum,
                                      float* d_h_sum, float* d_emb, void* stream);

/* a1 (aggregation inside RelationalMultiAggrMP._aggregate_messages, fs_mol/modules/gnn.py:197-265; torch_scatter's
 * scatter_sum / scatter_mean / scatter_max there): SUM | MEAN | STD | MAX of the incoming messages of every target
 * node in one pass.  msgs [E, H, 3m] post-ReLU messages (per tower: sum-part | mean/std-part | max-part), perm [E] the
 * message ids sorted by target, rowptr [V+1]; agg [V, H, 4m], argmax [V, H, m] (message id of the maximum, -1 for an
 * empty segment, whose aggregates are 0).  The backward writes every element of d_msgs [E, H, 3m] exactly once. */
int adkf_pna_aggregate(const float* msgs, const int64_t* perm, const int64_t* rowptr, int32_t V, int32_t H, int32_t m,
                       float* agg, int32_t* argmax, void* stream);
int adkf_pna_aggregate_backward(const float* msgs, const int64_t* perm, const int64_t* rowptr, const float* agg,
                                const int32_t* argmax, const float* d_agg, int32_t V, int32_t H, int32_t m,
                                float* d_msgs, void* stream);
/* The same with the ReLU in front of the messages folded in (msgs are the ReLU OUTPUTS of the message functions, gnn.py:141):
 * d_pre = d_msgs . [msgs > 0], the gradient in front of that ReLU.  adkf_msg_backward takes it with msgs = NULL and then reads one
 * [E, H, 3m] tensor per product instead of gradient + mask (its three products are bound by exactly that traffic). */
int adkf_pna_aggregate_backward_relu(const float* msgs, const int64_t* perm, const int64_t* rowptr, const float* agg,
                                     const int32_t* argmax, const float* d_agg, int32_t V, int32_t H, int32_t m, float* d_pre,
                                     void* stream);

/* a1 (the element-wise middle of GNNBlock.forward, fs_mol/modules/gnn.py:477-515, between the output projection of the message
 * passing and the BOOM MLP):   new = p0 + amp[v] p1 + att[v] p2 + bias  (p = [p0 | p1 | p2] [V, 3 hid]: the projected unscaled
 * aggregates; amp / att [V]: the PNA scalers),   x1 = x + alpha new  (ReZero),   h = LayerNorm(x1; gamma, beta, eps).
 * Outputs x1, h [V, hid] and the row statistics mu, rstd [V] the backward needs.  hid a multiple of 64, at most 256 (else
 * ADKF_E_SIZE: the caller keeps its own path).  The backward takes the gradients arriving at x1 (g_x1) and h (g_h) and writes d_p
 * [V, 3 hid], d_x [V, hid] and the parameter gradients d_bias, d_gamma, d_beta [hid], d_alpha [1], reduced over the nodes in a fixed
 * order (per-workgroup partials in `scratch`, adkf_block_combine_scratch_bytes(V, hid) bytes): bit-reproducible. */
int adkf_block_combine(const float* p, const float* x, const float* amp, const float* att, const float* bias, const float* alpha,
                       const float* gamma, const float* beta, float eps, int32_t V, int32_t hid, float* x1, float* h, float* mu,
                       float* rstd, void* stream);
size_t adkf_block_combine_scratch_bytes(int32_t V, int32_t hid);
int adkf_block_combine_backward(const float* p, const float* x1, const float* amp, const float* att, const float* bias,
                                const float* alpha, const float* gamma, const float* mu, const float* rstd, const float* g_x1,
                                const float* g_h, int32_t V, int32_t hid, float* d_p, float* d_x, float* d_bias, float* d_alpha,
                                float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes, void* stream);

/* H (outer update of ADKTModelTrainer.train_loop, fs_mol/utils/adaptive_dkt_utils.py:402-413: task-mean of the
 * accumulated gradients, torch.nn.utils.clip_grad_norm_, torch.optim.Adam.step) for a handful of parameter tensors.
 *   adkf_grad_sumsq      partials[0 .. ADKF_SUMSQ_PARTS) = per-workgroup sums of g^2 (fixed partition: deterministic).
 *   adkf_clip_adam_step  with |g| = sqrt(sum of the n_partials partials, i.e. those of ALL tensors laid side by side):
 *                        g <- g * scale * min(1, clip / (scale |g| + 1e-6));  then Adam step number `step` (1-based;
 *                        exp_avg m, exp_avg_sq v, no amsgrad, weight_decay added to g as torch does).  clip = +inf
 *                        disables clipping.  Hyper-parameters are doubles: 1 - beta and lr / (1 - beta1^step) are formed in double
 *                        and rounded once, as torch does.  All pointers 16-byte aligned, n elements each.
 *   adkf_clip_adam_step_one  the two calls above as ONE launch for ONE tensor of at most ADKF_CLIP_ADAM_ONE_MAX elements (every
 *                        workgroup re-adds the whole gradient's squares in the same fixed order: deterministic, identical on every rank;
 *                        ADKF_E_SIZE beyond the limit).  planes_t (or NULL): the tensor is a row-major weight w[K][N] (n = K * N, K and N
 *                        multiples of 64) and the three bfloat16 pieces of every UPDATED weight are also written as planes[q][n][k] - what
 *                        adkf_split_planes_t(w) would give after the step, bit for bit - so that the next adkf_dense_forward of y = x w
 *                        needs no split launch. */
#define ADKF_SUMSQ_PARTS 256
#define ADKF_CLIP_ADAM_ONE_MAX 131072
int adkf_grad_sumsq(const float* g, int64_t n, float* partials, void* stream);
int adkf_clip_adam_step(float* p, float* g, float* m, float* v, int64_t n, const float* partials, int32_t n_partials,
                        float scale, float clip, double lr, double beta1, double beta2, double eps, double weight_decay,
                        int32_t step, void* stream);
int adkf_clip_adam_step_one(float* p, float* g, float* m, float* v, int64_t n, float scale, float clip, double lr, double beta1,
                            double beta2, double eps, double weight_decay, int32_t step, uint16_t* planes_t, int32_t K, int32_t N,
                            void* stream);

/* a1 / a2 (the dense layers of the feature extractor and of the fc head: torch.nn.Linear, fs_mol/modules/gnn.py:477-515,
 * fs_mol/modules/graph_readout.py, fs_mol/models/adaptive_dkt.py:50-65) with FP32 products on the BF16 matrix pipe (csrc/dense_x3.h,
 * csrc/gemm_x3.h: a float is the exact sum of three bfloat16 values; six BF16 MFMAs per product block; errors below the FP32 GEMM's).
 *   adkf_split_planes         planes[q][r][k], q = 0..2: the three bfloat16 pieces of x[r][k] (x0 = bf16(x), x1 = bf16(x - x0), x2 = the rest:
 *                       their sum is x exactly).  `planes`: 3 * rows * K uint16, 16-byte aligned; K even.
 *   adkf_split_planes_t       the same from the transpose: w[K][N] (a weight stored input-major, as torch.matmul(x, w) takes it) ->
 *                       planes[q][n][k], i.e. the planes adkf_dense_forward wants for y = x w.  K even; N * K a multiple of 8.
 *   adkf_dense_forward  y[M, N] = x[M, K] w[N, K]^T (+ bias[N]): torch's F.linear, with w given as the planes adkf_split_planes wrote
 *                       (weights are split once per update, activations on the fly).  K a multiple of 32; ldx, ldy the row strides
 *                       of x, y in floats (ldx a multiple of 4); x, y, w_planes 16-byte aligned.  The backward product with respect to x is
 *                       the same call on the planes of w^T.  Returns ADKF_E_LAUNCH when the device refuses the kernel's 120 KB of
 *                       dynamic LDS.  Short contractions over many rows (K = 64, 128 or 256 and at least one 128-row tile per CU) take a
 *                       persistent form of the kernel (k_dense3_sk: a row tile's whole K extent in registers); the results are the same bits. */
int adkf_split_planes(const float* x, uint16_t* planes, int64_t rows, int64_t K, void* stream);
int adkf_split_planes_t(const float* w, uint16_t* planes, int64_t K, int64_t N, void* stream);
int adkf_dense_forward(const float* x, int32_t ldx, const uint16_t* w_planes, const float* bias, float* y, int32_t ldy, int32_t M,
                       int32_t N, int32_t K, void* stream);
 /*   adkf_dense_weight_grad  dw[N, K] = g[M, N]^T x[M, K] (torch: g.t() @ x), the contraction over the rows cut into row ranges whose
 *                       partial products are added in a fixed order (bit-reproducible); `scratch`: adkf_dense_weight_grad_scratch_bytes(M, N, K)
 *                       bytes.  Any M, N, K; ldg, ldx the row strides of g, x in floats. */
size_t adkf_dense_weight_grad_scratch_bytes(int32_t M, int32_t N, int32_t K);
int adkf_dense_weight_grad(const float* g, int32_t ldg, const float* x, int32_t ldx, float* dw, int32_t M, int32_t N, int32_t K,
                           void* scratch, size_t scratch_bytes, void* stream);

/* Synchronises `stream`, then returns 0 or (index+1) of the first task with info != 0. */
int adkf_check_info(const int32_t* info, int32_t T, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADKF_GP_H */
