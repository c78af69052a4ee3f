/*
 * phoenix_hip.h -- C ABI of the MI355X-native PHOENIX NeuralODE engine (libphoenix_hip.so).
 *
 * The reference (QuackenbushLab/phoenix) is pure Python and has NO plugin / FFI interface; its
 * boundary for this path is the Python pair
 *     odeint(func, y0, t, rtol, atol, method, options)           torchdiffeq/_impl/odeint.py:30
 *     odeint_adjoint(...)                                          torchdiffeq/_impl/adjoint.py:165
 * plus the `func(t, y)` protocol implemented by ODENet (odenet.py:85-98).  This header is the
 * native boundary a maintainer binds *underneath* those two functions (INTEGRATION.md shows the
 * ctypes stub); phoenix_amd/ is exactly such a binding.  Each entry point names the reference
 * code it replaces.  All paths are relative to /root/reference/ode_net/code/.
 *
 * Conventions
 *  - plain pointers + sizes; every data pointer is a DEVICE pointer unless marked HOST.
 *  - the caller owns all memory (parameters, states, outputs, workspace); the library never
 *    allocates or frees device memory and keeps no pointer past the call.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it; no call synchronises.
 *  - every function returns a phx_status (0 = enqueued OK).  Solver-level failures (the
 *    reference's AssertionErrors) are reported per trajectory in the `status` output array.
 *  - fp32 state, fp64 time scalars, exactly like the reference (rk_common.py:115-131).
 */
#ifndef PHOENIX_HIP_H
#define PHOENIX_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PHX_ABI_VERSION 7

/* ODENet parameters (odenet.py:42-82), gene-contiguous: every matrix is [rows, N] row-major.
 *   Ws  [H, N]   net_sums.linear_out.weight            (reference layout as is)
 *   Wp  [H, N]   net_prods.linear_out.weight           (reference layout as is)
 *   WaT [2H, N]  net_alpha_combine.linear_out.weight TRANSPOSED (reference stores [N, 2H])
 *   bs, bp [H]   biases;   g [N] gene_multipliers ([1,N] in the reference)                  */
typedef struct phx_params {
    const float *Ws, *bs, *Wp, *bp, *WaT, *g;
    int N, H;
    const void *wimg; /* optional: LDS weight images of THESE parameter values, made by phx_pack_weight_images; the solve
                         entry points then skip their per-launch packing kernel.  NULL: packed per launch. */
} phx_params;

/* Gradient buffers with the same shapes/layouts; the engine ACCUMULATES (+=) into them, or -- overwrite != 0 -- writes
 * them (=): the buffers may then be uninitialised (saves the caller a zero fill). */
typedef struct phx_grads {
    float *Ws, *bs, *Wp, *bp, *WaT, *g;
    int overwrite;
    float *Wa; /* optional (ABI 6): non-NULL = the gradient of net_alpha_combine.linear_out.weight is written HERE in the
                  reference's own layout [N, 2H] (what autograd hands the optimizer, odenet.py:57-60) and WaT is not
                  touched (may be NULL): the caller needs no transposed copy of the 2HN floats afterwards. */
} phx_grads;

/* torchdiffeq SOLVERS entries on the BASELINE path (odeint.py:14-27) */
enum phx_method { PHX_EULER = 0, PHX_MIDPOINT = 1, PHX_RK4 = 2 /* 3/8 rule */, PHX_DOPRI5 = 3 };

/* step control of a batch */
enum phx_control {
    PHX_CTRL_SHARED = 0,        /* reference odeint on y0[B,1,N]: one controller, rms over B*N */
    PHX_CTRL_PER_TRAJECTORY = 1 /* reference training loop: B independent odeint calls        */
};

enum phx_status {
    PHX_OK = 0,
    PHX_ERR_MAX_STEPS = 1,    /* AssertionError 'max_num_steps exceeded'        rk_common.py:154 */
    PHX_ERR_DT_UNDERFLOW = 2, /* AssertionError 'underflow in dt'               rk_common.py:175 */
    PHX_ERR_NONFINITE = 3,    /* AssertionError 'non-finite values in state y'  rk_common.py:176 */
    PHX_ERR_BAD_ARG = 4,
    PHX_ERR_WORKSPACE = 5,    /* workspace too small */
    PHX_ERR_LAUNCH = 6,       /* HIP launch failure */
    PHX_ERR_SYNC_TIMEOUT = 7  /* in-kernel grid barrier timed out (kernel aborted itself) */
};

typedef struct phx_solve_opts {
    int method;         /* phx_method */
    int control;        /* phx_control */
    double rtol, atol;  /* odeint defaults 1e-7 / 1e-9 (odeint.py:30) */
    int t_per_sample;   /* 0: t is [T] shared;  1: t is [B, T] (training loop: t[b] = (t_i, t_{i+1})) */
    int t_is_f32;       /* 1: the caller's t tensor was float32 (fixed-grid dt is then formed in fp32);
                           2: ... and the buffer passed as `t` still holds float32 values (no conversion) */
    long long max_num_steps; /* <=0: 2^31-1 like the reference (rk_common.py:110) */
    int calls;          /* phx_odeint with PHX_CTRL_SHARED only; > 1: the B rows are `calls` independent odeint calls of
                           B/calls rows each (rows of a call contiguous), every call under its own shared step
                           controller -- the loop of find_gene_influences.py:64-77 in one launch.  0/1: one call.
                           The adjoint entry point ignores it. */
    int ws_keep;        /* ABI 6.  0: the workspace may hold anything -- the call zero-fills its exchange buffers first (6-24 MB
                           at breast scale, on `stream`, in front of the kernel).  1: the caller vouches that the PREVIOUS
                           call that used this workspace was this same entry point with the same shape and options, is
                           ordered before this one on `stream`, returned PHX_OK, and that nothing else wrote to the
                           workspace since: kernels that keep two alternating sets of exchange buffers (the third-generation
                           solve kernels) then skip the fill and clean the idle set themselves; every other kernel fills
                           as with 0.  A first call on a workspace, or one after a shape change, must pass 0. */
} phx_solve_opts;

int phx_abi_version(void);
const char *phx_status_string(int status);
/* number of CUs the engine sizes its persistent grids for (0 if no device) */
int phx_device_cus(void);

/* Workspace size (bytes) for one call of entry point `op` with this shape. */
enum phx_op { PHX_OP_RHS_FORWARD = 0, PHX_OP_RHS_VJP = 1, PHX_OP_ODEINT = 2, PHX_OP_ADJOINT = 3 };
size_t phx_workspace_bytes(int op, int N, int H, int B, int T);
/* Weight images for phx_params.wimg: the per-gene-block LDS layout of the MFMA solve kernels depends only on (N, H), so
 * a caller that runs several solves with the same parameter values (forward + backward of a training step, a validation
 * loop, an analysis scan) packs once.  phx_weight_image_bytes: buffer size (0: this shape has no MFMA plan). */
size_t phx_weight_image_bytes(int N, int H);
int phx_pack_weight_images(const phx_params *p, void *wimg, void *stream);
/* ABI 6: the whole engine layout of one parameter version in ONE kernel, straight from the tensors the reference's
 * ODENet holds (odenet.py:42-82): Ws, Wp [H, N], Wa = net_alpha_combine.linear_out.weight as PyTorch stores it [N, 2H],
 * g [N].  Writes WaT_out [2H, N] (phx_params.WaT) and, when `wimg_out` is non-NULL (phx_weight_image_bytes(N, H) bytes),
 * the packed weight images (phx_params.wimg).  Replaces a transposed copy + phx_pack_weight_images per optimizer step. */
int phx_layout_params(const float *Ws, const float *Wp, const float *Wa, const float *g, int N, int H, float *WaT_out,
                      void *wimg_out, void *stream);
/* ... for phx_odeint with opts->calls = calls (0: this batch of calls cannot be planned, solve the calls one by one) */
size_t phx_odeint_calls_workspace_bytes(int N, int H, int B, int T, int calls);

/* Replaces ODENet.forward / prior_only_forward (odenet.py:85-98): out[B,N] = f(y[B,N]). */
int phx_rhs_forward(const phx_params *p, const float *y, float *out, int B, int prior_only,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Replaces torch.autograd.grad through ODENet.forward (adjoint.py:116-119; and the autograd
 * backward of prior_only_forward, train_insilico.py:134-138):
 *   vjp_y[B,N] = cot^T df/dy   (overwritten; may be NULL)
 *   grads     += cot^T df/dtheta summed over B  (may be NULL)
 *   f_out[B,N] = f(y)          (may be NULL)                                               */
int phx_rhs_vjp(const phx_params *p, const float *y, const float *cot, float *vjp_y,
                const phx_grads *grads, float *f_out, int B, int prior_only, void *workspace,
                size_t workspace_bytes, void *stream);

/* Replaces odeint(ODENet, y0, t, rtol, atol, method) (odeint.py:30-74; solvers.py:23-30,77-95;
 * rk_common.py:39-228; fixed_grid.py:6-38; dopri5.py; misc.py:47-103; interp.py).
 *   y0 [B,N]; t (double) [T] or [B,T] increasing or decreasing; sol [T,B,N] (sol[0] = y0).
 *   status [B] (int, phx_status per trajectory), nfe [B] (int, RHS evaluations), nsteps [B].
 *   Outputs a failed trajectory (status != 0) never reached are set to NaN.                    */
int phx_odeint(const phx_params *p, const float *y0, const double *t, int B, int T,
               const phx_solve_opts *opts, float *sol, int *status, int *nfe, int *nsteps,
               void *workspace, size_t workspace_bytes, void *stream);

/* Replaces OdeintAdjointMethod.backward (adjoint.py:32-162):
 *   y_saved [T,B,N] forward outputs, grad_y [T,B,N] cotangents of every output time,
 *   adj_y0 [B,N] (overwritten) = dL/dy0,   grads += dL/dtheta (summed over B).
 * The parameter gradient is integrated as a quadrature with the solver's own weights (identical
 * mathematics to the reference's P-sized augmented state, adjoint.py:86-87,127); the adaptive
 * controller's norm covers the [t, y, adj_y] blocks (the reference also includes theta).      */
int phx_odeint_adjoint_backward(const phx_params *p, const double *t, int B, int T,
                                const phx_solve_opts *opts, const float *y_saved,
                                const float *grad_y, float *adj_y0, const phx_grads *grads,
                                int *status, int *nfe, int *nsteps, void *workspace,
                                size_t workspace_bytes, void *stream);

/* prior_grad = X[K,N] @ P[N,N] with the prior matrix P in CSC form
 * (colptr [N+1], rowidx/vals [nnz], rows ascending inside a column).  Replaces the reference's dense
 * `torch.matmul(batch_for_prior, prior_mat)` (train_insilico.py:207-211) and the 0.5 GB dense matrix that
 * `read_prior_matrix(..., sparse=True)` materialises (train_insilico.py:64-73).  out [K,N] is overwritten. */
int phx_prior_targets(const int *colptr, const int *rowidx, const float *vals, const float *X, float *out,
                      int K, int N, void *stream);
/* The same product with P in sliced-ELL form (the fast path; phoenix_amd.prior builds it once per prior matrix):
 * columns in slices of 64; width[s] = longest column of slice s; entry i of column 64 s + l at sptr[s] + i*64 + l in
 * ridx / vals (rows ascending inside a column, padding row 0 / value 0).  X rows are staged in LDS, so N*4 bytes must
 * fit it (PHX_ERR_BAD_ARG otherwise: use phx_prior_targets). */
int phx_prior_targets_sell(const long long *sptr, const int *width, const int *ridx, const float *vals, const float *X,
                           float *out, int K, int N, void *stream);

/* Fused head of the prior branch of training_step (train_insilico.py:134-135):
 *   loss[0] = mean((prior_only_forward(X) - target)^2)  over B*N elements   (device float)
 *   cot[B,N] = d loss / d prior_only_forward(X) = 2 (pred - target) / (B N)
 * without materialising the prediction; `cot` is what phx_rhs_vjp(prior_only = 1) takes for the backward.
 * Workspace: phx_workspace_bytes(PHX_OP_RHS_FORWARD, ...) for B >= 1024 rows; PHX_ERR_BAD_ARG when the batch chain
 * cannot be planned (the caller then evaluates the unfused formula).                                              */
int phx_prior_mse(const phx_params *p, const float *X, const float *target, int B, float *cot, float *loss,
                  void *workspace, size_t workspace_bytes, void *stream);

/* The same step with the hidden rows kept for the backward (ABI 5; what is kept changed in ABI 7).  The forward chain of
 * phx_prior_mse reduces the hidden rows z = [Ws a(X) + bs ; exp(Wp l(X) + bp)] of every row of X anyway, and its last
 * kernel holds the loss cotangent in registers.  phx_prior_mse_save also contracts that cotangent with the rows of Wa
 * there and writes the four hidden sections  du | dv | z_u | z_p  of every row ([4][16 HT][16 ceil(B / 16)] floats, the
 * operand layout of the gradient contraction) to `z_save` (phx_prior_z_bytes(N, H, B) bytes, device, caller-owned; 0 =
 * this shape has no such path: H > 128 or the batch chain cannot be planned).  phx_prior_vjp_saved then computes the
 * parameter gradients of sum(cot * prior_only_forward(X)) -- what phx_rhs_vjp(prior_only = 1, vjp_y = NULL) computes --
 * with the contraction kernel alone: `cot` must be the cotangent phx_prior_mse_save wrote (the caller scales the
 * gradients by d loss afterwards; the saved sections are those of d loss = 1).  Same workspaces as phx_prior_mse /
 * phx_rhs_vjp.  Replaces nothing new in the reference: it is train_insilico.py:134-138 again, with autograd's saved
 * activations made explicit.                                                                                          */
size_t phx_prior_z_bytes(int N, int H, int B);
int phx_prior_mse_save(const phx_params *p, const float *X, const float *target, int B, float *cot, float *loss,
                       float *z_save, void *workspace, size_t workspace_bytes, void *stream);
int phx_prior_vjp_saved(const phx_params *p, const float *X, const float *cot, const float *z_saved,
                        const phx_grads *grads, int B, void *workspace, size_t workspace_bytes, void *stream);

/* the ground-truth Hill-kinetics simulator behind the reference's in-silico data
 * (GraphGRN_core.R:425-486 emits one rate expression per gene -- ode_system_functions_*.csv -- and
 * SimulationGRN_core_init_var.R:218-247 integrates them per sample).  The caller compiles the expressions to
 * postfix programs: code [L][2] = (op, arg) with op 0 PUSHC consts[arg], 1 PUSHX x[arg], 2 ADD, 3 SUB, 4 MUL,
 * 5 DIV, 6 NEG, 7 FACT with consts[arg..arg+2] = (B, K_n, n) of fAct (GraphGRN_core.R:431-436); off/len [N] give each
 * gene's slice (len 0 = "input gene", rate 0).  phx_hill_rhs: out[B,N] = rates at x[B,N].  phx_hill_simulate:
 * out[T,B,N] = states at times[T] (out[0] = x0), classical RK4 with equal sub-steps <= dt_max per interval.       */
int phx_hill_rhs(const int *code, const int *off, const int *len, const float *consts, const float *x,
                 float *out, int B, int N, void *stream);
int phx_hill_simulate(const int *code, const int *off, const int *len, const float *consts, const float *x0,
                      const double *times, int T, double dt_max, float *out, int B, int N, void *stream);

/* Diagnostic only (not part of the drop-in surface): with PHX_PROF=1 in the environment the v1 kernels
 * write 16 per-workgroup segment timers (100 MHz ticks) into the workspace; this returns where. */
/* Diagnostic only: the next phx_odeint / phx_odeint_adjoint_backward call on this thread records these two
 * hipEvent_t immediately before and after its solve kernel (not around its memset nodes / reduce kernel). */
void phx_debug_set_kernel_events(void *ev_start, void *ev_stop);
/* Diagnostic only: a process-wide FIFO of such pairs; every following phx_odeint / phx_odeint_adjoint_backward launch
 * (from any thread) takes the next pair.  (NULL, NULL) empties the queue.  Lets a bench time the solve kernels of
 * ordinary back-to-back training steps (forward on the caller's thread, backward on an autograd worker thread). */
void phx_debug_queue_kernel_events(void *ev_start, void *ev_stop);
int phx_debug_profile_region(int op, int N, int H, int B, int T, int control, size_t *offset,
                             int *n_workgroups, int *plan);
/* Diagnostic only: which backward-solve kernel phx_odeint_adjoint_backward launches for this shape:
 * 0 = k_solve_adj (VALU, grid barriers), 1 = k1_solve_adj (MFMA, one wave per trajectory tile),
 * 2 = k1_solve_adj2 (MFMA, wave pairs, fused sweeps), 3 = k1_solve_adj3 (MFMA, dopri5 with H <= 48: fused sweeps over
 * a 16-vector private state).  bench.py keys its profile lookups with it.  The first form assumes dopri5. */
int phx_debug_adjoint_kernel(int N, int H, int B, int T, int control);
int phx_debug_adjoint_kernel_m(int N, int H, int B, int T, int control, int method);
/* ... and which forward-solve kernel phx_odeint launches: 0 = k_solve_fwd (VALU), 1 = k1_solve_fwd (MFMA),
 * 3 = k1_solve_fwd3 (MFMA, dopri5 with H <= 48), 4 = k1_solve_fwd3c (its hidden-chunked form, 48 < H <= 256); the
 * adjoint query above likewise returns 4 for k1_solve_adj3c. */
int phx_debug_forward_kernel_m(int N, int H, int B, int T, int control, int method);
/* Diagnostic only: how many launches of that solve kernel one call with this batch makes (a batch that does not fit one
 * residency is walked in chunks: e.g. 256 B-cell trajectories = two launches of k1_solve_adj3c).  op = PHX_OP_ODEINT or
 * PHX_OP_ADJOINT; 0 when no plan exists.  bench.py multiplies per-launch profile figures with it. */
int phx_debug_solve_launches(int op, int N, int H, int B, int T, int control, int method);

#ifdef __cplusplus
}
#endif
#endif
