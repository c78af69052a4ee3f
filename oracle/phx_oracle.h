/*
 * phx_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, fp32 state / fp64 time scalars) of the PHOENIX
 * NeuralODE hot path of the reference (QuackenbushLab/phoenix):
 *   ode_net/code/odenet.py                       (RHS model)
 *   ode_net/code/torchdiffeq/_impl/ (*.py)          (vendored torchdiffeq 0.1.1)
 * Each function in phx_oracle.c cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library -- as the checker / the timed CPU baseline, never as the
 * product path.  Parity pin: tests/golden/ *.npz captured from the reference
 * itself by tests/golden/make_goldens.py (see tests/test_oracle_vs_golden.py).
 */
#ifndef PHX_ORACLE_H
#define PHX_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* Parameter tensors in the reference's own (PyTorch) layouts:
 *   Ws [H,N] net_sums.linear_out.weight      bs [H]
 *   Wp [H,N] net_prods.linear_out.weight     bp [H]
 *   Wa [N,2H] net_alpha_combine.linear_out.weight (no bias)
 *   g  [N]   gene_multipliers ([1,N])
 * (odenet.py:42-82) */
typedef struct {
    int N, H;
    const float *Ws, *bs, *Wp, *bp, *Wa, *g;
} phxo_net;

/* Gradient buffers, same shapes/layouts; routines ACCUMULATE (+=) into them. */
typedef struct {
    float *Ws, *bs, *Wp, *bp, *Wa, *g;
} phxo_grads;

enum { PHXO_EULER = 0, PHXO_MIDPOINT = 1, PHXO_RK4 = 2, PHXO_DOPRI5 = 3 };

/* status codes (map onto the reference's AssertionErrors) */
enum {
    PHXO_OK = 0,
    PHXO_ERR_MAX_STEPS = 1, /* rk_common.py:154 */
    PHXO_ERR_DT_UNDERFLOW = 2, /* rk_common.py:175 */
    PHXO_ERR_NONFINITE = 3, /* rk_common.py:176 */
    PHXO_ERR_BAD_ARG = 4
};

/* f = ODENet.forward (prior_only=0) or prior_only_forward (=1); y,f: [B,N] */
int phxo_rhs(const phxo_net *net, const float *y, float *f, int B, int prior_only);

/* VJP of ODENet.forward: cot [B,N] is the cotangent on f.
 * vjp_y [B,N] is overwritten; grads (may be NULL) are accumulated, summed over B.
 * f_out (may be NULL) receives f(y). prior_only selects prior_only_forward. */
int phxo_rhs_vjp(const phxo_net *net, const float *y, const float *cot, int B,
                 float *vjp_y, phxo_grads *grads, float *f_out, int prior_only);

/* odeint(func=ODENet, y0[B,1,N], t[T]) with the reference's semantics: ONE solve of
 * the flattened [B*N] state (step control shared by the batch, rms norm over B*N).
 * t may be decreasing.  sol: [T,B,N].  nfe (may be NULL): number of RHS evals.
 * t_is_f32: the caller's t tensor was float32 (affects fixed-grid dt rounding). */
int phxo_odeint(const phxo_net *net, const float *y0, int B, const double *t, int T,
                int t_is_f32, int method, double rtol, double atol, float *sol,
                long long *nfe, long long *nsteps);

/* OdeintAdjointMethod.backward for one odeint_adjoint call with y0 [B,1,N]:
 * inputs y_saved [T,B,N] (forward outputs) and grad_y [T,B,N]; outputs
 * adj_y0 [B,N] (overwritten) and grads (accumulated).
 * theta_in_norm: 1 = reference mixed norm over [t, y, adj, theta] blocks
 *                0 = blocks [t, y, adj] only (what the HIP engine implements). */
int phxo_adjoint_backward(const phxo_net *net, int B, const double *t, int T,
                          int t_is_f32, int method, double rtol, double atol,
                          const float *y_saved, const float *grad_y,
                          int theta_in_norm, float *adj_y0, phxo_grads *grads,
                          long long *nfe, long long *nsteps);

/* Reference-shaped "python loop over samples" (train_insilico.py:128-130):
 * B independent solves of y0[b] over t[b,0..T) (t: [B,T]), OpenMP across samples.
 * sol: [B,T,N]. */
int phxo_odeint_per_sample(const phxo_net *net, const float *y0, int B, const double *t,
                           int T, int t_is_f32, int method, double rtol, double atol,
                           float *sol, long long *nfe, long long *nsteps, int nthreads);

/* and their backward: y_saved/grad_y [B,T,N]; adj_y0 [B,N]; grads summed over samples. */
int phxo_adjoint_backward_per_sample(const phxo_net *net, int B, const double *t, int T,
                                     int t_is_f32, int method, double rtol, double atol,
                                     const float *y_saved, const float *grad_y,
                                     int theta_in_norm, float *adj_y0, phxo_grads *grads,
                                     long long *nfe, long long *nsteps, int nthreads);

/* Controller unit functions (misc.py / interp.py), exposed for golden G6. */
float phxo_rms_norm(const float *x, long long n);
float phxo_mixed_norm(const float *x, const long long *block_sizes, int nblocks);
double phxo_optimal_step_size(double last_step, float error_ratio, double safety,
                              double ifactor, double dfactor, int order);
void phxo_interp_fit(const float *y0, const float *y1, const float *y_mid, const float *f0,
                     const float *f1, float dt, long long n, float *coef /*[5,n] e,d,c,b,a*/);
void phxo_interp_eval(const float *coef, long long n, double t0, double t1, double t, float *out);

int phxo_version(void);

#ifdef __cplusplus
}
#endif
#endif
