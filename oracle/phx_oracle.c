/*
 * phx_oracle.c -- TEST INFRASTRUCTURE ONLY (see phx_oracle.h).
 *
 * Plain-C restatement of the reference's NeuralODE hot path.  fp32 state,
 * fp64 "time-like" scalars, exactly as the reference keeps them
 * (torchdiffeq/_impl/rk_common.py:115-131).  All file:line citations are
 * relative to /root/reference/ode_net/code/.
 *
 * Parity status: PINNED against goldens captured from the reference itself
 * (tests/golden/make_goldens.py -> tests/golden/ *.npz, checked by
 * tests/test_oracle_vs_golden.py).  The reference ships no tests of its own.
 */
#include "phx_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int phxo_version(void) { return 1; }

/* ------------------------------------------------------------------ */
/* RHS model: odenet.py:16-35 (activations), :85-98 (forward)          */
/* ------------------------------------------------------------------ */

/* SoftsignMod.forward (odenet.py:21-25) and LogShiftedSoftSignMod.forward
 * (odenet.py:31-35), computed the way the reference computes them. */
static inline void act_pair(float y, float *a, float *l)
{
    float s = y - 0.5f;
    float as = fabsf(s);
    float v = s / (1.0f + as);
    *a = v;
    *l = log1pf(v);
}

/* derivatives da/dy, dl/dy in closed form (verified there
 * against torch.autograd in fp64 to 9e-16) */
static inline void act_grad(float y, float *da, float *dl)
{
    float s = y - 0.5f;
    float as = fabsf(s);
    float d = 1.0f + as;
    *da = 1.0f / (d * d);
    if (s < 0.0f)
        *dl = 1.0f / d;
    else
        *dl = 1.0f / ((1.0f + s) * (1.0f + 2.0f * s));
}

static inline float dotf(const float *x, const float *w, int n)
{
    float acc = 0.0f;
#pragma omp simd reduction(+ : acc)
    for (int i = 0; i < n; ++i) acc += x[i] * w[i];
    return acc;
}

/* one row of ODENet.forward / prior_only_forward.
 * scratch: a[N], l[N], z[2H]. j_out may be NULL. */
static void rhs_row(const phxo_net *net, const float *y, float *f, int prior_only,
                    float *a, float *l, float *z, float *j_out)
{
    const int N = net->N, H = net->H;
    for (int n = 0; n < N; ++n) act_pair(y[n], &a[n], &l[n]);
    /* sums = net_sums(y) : Linear(N->H) with bias  (odenet.py:86) */
    for (int h = 0; h < H; ++h) z[h] = dotf(a, net->Ws + (size_t)h * N, N) + net->bs[h];
    /* prods = exp(net_prods(y))                     (odenet.py:87) */
    for (int h = 0; h < H; ++h) z[H + h] = expf(dotf(l, net->Wp + (size_t)h * N, N) + net->bp[h]);
    /* joint = net_alpha_combine(cat(sums, prods))   (odenet.py:88-89) */
    const int K = 2 * H;
    for (int n = 0; n < N; ++n) {
        float jn = dotf(z, net->Wa + (size_t)n * K, K);
        if (j_out) j_out[n] = jn;
        if (prior_only) {
            f[n] = jn; /* odenet.py:93-98 */
        } else {
            float r = net->g[n] > 0.0f ? net->g[n] : 0.0f; /* relu(gene_multipliers) */
            f[n] = r * (jn - y[n]);                        /* odenet.py:90 */
        }
    }
}

int phxo_rhs(const phxo_net *net, const float *y, float *f, int B, int prior_only)
{
    const int N = net->N, H = net->H;
    int par = 0;
#ifdef _OPENMP
    par = (B >= 4) && !omp_in_parallel();
#endif
#pragma omp parallel if (par)
    {
        float *a = (float *)malloc(sizeof(float) * (2 * (size_t)N + 2 * H));
        float *l = a + N, *z = l + N;
#pragma omp for schedule(static)
        for (int b = 0; b < B; ++b)
            rhs_row(net, y + (size_t)b * N, f + (size_t)b * N, prior_only, a, l, z, NULL);
        free(a);
    }
    return PHXO_OK;
}

/* VJP through ODENet.forward for one row; what torch.autograd.grad computes in
 * augmented_dynamics (torchdiffeq/_impl/adjoint.py:101-119). */
static void vjp_row(const phxo_net *net, const float *y, const float *cot, float *vjp_y,
                    phxo_grads *gr, float *f_out, int prior_only, float *a, float *l, float *z,
                    float *j, float *q, float *dz, float *tmp)
{
    const int N = net->N, H = net->H, K = 2 * H;
    float *ftmp = f_out ? f_out : tmp;
    rhs_row(net, y, ftmp, prior_only, a, l, z, j);
    for (int n = 0; n < N; ++n) {
        float r = prior_only ? 1.0f : (net->g[n] > 0.0f ? net->g[n] : 0.0f);
        q[n] = cot[n] * r;
    }
    /* dz = q . Wa */
    for (int k = 0; k < K; ++k) dz[k] = 0.0f;
    for (int n = 0; n < N; ++n) {
        const float qn = q[n];
        const float *w = net->Wa + (size_t)n * K;
#pragma omp simd
        for (int k = 0; k < K; ++k) dz[k] += qn * w[k];
    }
    /* through exp: dv = dz_p * p */
    for (int h = 0; h < H; ++h) dz[H + h] *= z[H + h];
    /* vjp_y = -q + (du.Ws) a' + (dv.Wp) l' */
    for (int n = 0; n < N; ++n) tmp[n] = 0.0f;
    float *tmp2 = tmp + N;
    for (int n = 0; n < N; ++n) tmp2[n] = 0.0f;
    for (int h = 0; h < H; ++h) {
        const float du = dz[h], dv = dz[H + h];
        const float *ws = net->Ws + (size_t)h * N, *wp = net->Wp + (size_t)h * N;
#pragma omp simd
        for (int n = 0; n < N; ++n) {
            tmp[n] += du * ws[n];
            tmp2[n] += dv * wp[n];
        }
    }
    for (int n = 0; n < N; ++n) {
        float da, dl;
        act_grad(y[n], &da, &dl);
        float v = tmp[n] * da + tmp2[n] * dl;
        if (!prior_only) v -= q[n];
        vjp_y[n] = v;
    }
    if (gr) {
        for (int n = 0; n < N; ++n) {
            const float qn = q[n];
            float *w = gr->Wa + (size_t)n * K;
#pragma omp simd
            for (int k = 0; k < K; ++k) w[k] += qn * z[k];
        }
        for (int h = 0; h < H; ++h) {
            const float du = dz[h], dv = dz[H + h];
            float *ws = gr->Ws + (size_t)h * N, *wp = gr->Wp + (size_t)h * N;
#pragma omp simd
            for (int n = 0; n < N; ++n) {
                ws[n] += du * a[n];
                wp[n] += dv * l[n];
            }
            gr->bs[h] += du;
            gr->bp[h] += dv;
        }
        if (!prior_only)
            for (int n = 0; n < N; ++n)
                if (net->g[n] > 0.0f) gr->g[n] += cot[n] * (j[n] - y[n]);
    }
}

int phxo_rhs_vjp(const phxo_net *net, const float *y, const float *cot, int B, float *vjp_y,
                 phxo_grads *grads, float *f_out, int prior_only)
{
    const int N = net->N, H = net->H;
    float *buf = (float *)malloc(sizeof(float) * (6 * (size_t)N + 4 * H));
    float *a = buf, *l = a + N, *j = l + N, *q = j + N, *tmp = q + N /* 2N */, *z = tmp + 2 * N,
          *dz = z + 2 * H;
    for (int b = 0; b < B; ++b)
        vjp_row(net, y + (size_t)b * N, cot + (size_t)b * N, vjp_y + (size_t)b * N, grads,
                f_out ? f_out + (size_t)b * N : NULL, prior_only, a, l, z, j, q, dz, tmp);
    free(buf);
    return PHXO_OK;
}

/* ------------------------------------------------------------------ */
/* norms: torchdiffeq/_impl/misc.py:10-24                              */
/* ------------------------------------------------------------------ */
float phxo_rms_norm(const float *x, long long n)
{
    /* tensor.pow(2).mean().sqrt() in fp32; torch's CPU sum is a cascade sum, so
     * accumulate wide and round once. */
    double acc = 0.0;
    for (long long i = 0; i < n; ++i) {
        float sq = x[i] * x[i];
        acc += (double)sq;
    }
    float mean = (float)(acc / (double)n);
    return sqrtf(mean);
}

float phxo_mixed_norm(const float *x, const long long *bs, int nb)
{
    /* _mixed_linf_rms_norm: max over blocks of the block rms (misc.py:14-24) */
    float best = 0.0f;
    long long off = 0;
    for (int i = 0; i < nb; ++i) {
        if (bs[i] > 0) {
            float r = phxo_rms_norm(x + off, bs[i]);
            if (i == 0 || r > best || isnan(r)) best = r;
        }
        off += bs[i];
    }
    return best;
}

/* ------------------------------------------------------------------ */
/* generic explicit-RK machinery over a flat fp32 state                */
/* ------------------------------------------------------------------ */
typedef struct {
    void (*f)(void *ctx, const float *y, float *dy); /* autonomous RHS */
    void *ctx;
    long long len;
    int nblocks; /* norm blocks (1 => plain rms over len) */
    long long blocks[4];
    long long nfe, nsteps;
    long long max_steps;
} sys_t;

static float sys_norm(const sys_t *s, const float *x)
{
    if (s->nblocks <= 1) return phxo_rms_norm(x, s->len);
    return phxo_mixed_norm(x, s->blocks, s->nblocks);
}

static void sys_f(sys_t *s, const float *y, float *dy)
{
    s->f(s->ctx, y, dy);
    s->nfe++;
}

/* --- fixed grid: solvers.py:77-95 (grid == t), fixed_grid.py:6-38,
 *     rk_common.py:96-103 (rk4_alt_step_func = 3/8 rule) --- */
static void fixed_step(sys_t *s, int method, const float *y, float dt, float *y1, float *w)
{
    const long long n = s->len;
    float *k1 = w, *k2 = w + n, *k3 = w + 2 * n, *k4 = w + 3 * n, *yt = w + 4 * n;
    if (method == PHXO_EULER) {
        /* dt * func(t, y)                      fixed_grid.py:13-14 */
        sys_f(s, y, k1);
        for (long long i = 0; i < n; ++i) y1[i] = y[i] + dt * k1[i];
    } else if (method == PHXO_MIDPOINT) {
        /* half_dt = 0.5*dt; y_mid = y + f*half_dt; dt*func(y_mid)  fixed_grid.py:24-27 */
        const float half_dt = 0.5f * dt;
        sys_f(s, y, k1);
        for (long long i = 0; i < n; ++i) yt[i] = y[i] + k1[i] * half_dt;
        sys_f(s, yt, k2);
        for (long long i = 0; i < n; ++i) y1[i] = y[i] + dt * k2[i];
    } else {
        /* rk4_alt_step_func             rk_common.py:96-103 */
        const float one_third = (float)(1.0 / 3.0);
        sys_f(s, y, k1);
        for (long long i = 0; i < n; ++i) yt[i] = y[i] + (dt * k1[i]) * one_third;
        sys_f(s, yt, k2);
        for (long long i = 0; i < n; ++i) yt[i] = y[i] + dt * (k2[i] - k1[i] * one_third);
        sys_f(s, yt, k3);
        for (long long i = 0; i < n; ++i) yt[i] = y[i] + dt * ((k1[i] - k2[i]) + k3[i]);
        sys_f(s, yt, k4);
        for (long long i = 0; i < n; ++i) {
            float dy = (((k1[i] + 3.0f * (k2[i] + k3[i])) + k4[i]) * dt) * 0.125f;
            y1[i] = y[i] + dy;
        }
    }
    s->nsteps++;
}

static float grid_dt(const double *t, int i, int t_is_f32)
{
    if (t_is_f32) return (float)t[i + 1] - (float)t[i]; /* subtraction done in fp32 */
    return (float)(t[i + 1] - t[i]);
}

/* t must be increasing here (caller normalises). sol [T,len]. */
static int solve_fixed(sys_t *s, int method, const float *y0, const double *t, int T, int t_is_f32,
                       float *sol)
{
    const long long n = s->len;
    float *w = (float *)malloc(sizeof(float) * 5 * (size_t)n);
    memcpy(sol, y0, sizeof(float) * n);
    for (int i = 0; i + 1 < T; ++i)
        fixed_step(s, method, sol + (size_t)i * n, grid_dt(t, i, t_is_f32), sol + (size_t)(i + 1) * n, w);
    free(w);
    return PHXO_OK;
}

/* --- Dormand-Prince: dopri5.py:5-30 --- */
static const double DP_ALPHA[6] = {1.0 / 5, 3.0 / 10, 4.0 / 5, 8.0 / 9, 1.0, 1.0};
static const double DP_BETA[6][6] = {
    {1.0 / 5},
    {3.0 / 40, 9.0 / 40},
    {44.0 / 45, -56.0 / 15, 32.0 / 9},
    {19372.0 / 6561, -25360.0 / 2187, 64448.0 / 6561, -212.0 / 729},
    {9017.0 / 3168, -355.0 / 33, 46732.0 / 5247, 49.0 / 176, -5103.0 / 18656},
    {35.0 / 384, 0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84},
};
static const double DP_CERR[7] = {
    35.0 / 384 - 1951.0 / 21600,
    0,
    500.0 / 1113 - 22642.0 / 50085,
    125.0 / 192 - 451.0 / 720,
    -2187.0 / 6784 - -12231.0 / 42400,
    11.0 / 84 - 649.0 / 6300,
    -1.0 / 60.0,
};
static const double DP_CMID[7] = {
    6025192743.0 / 30085553152.0 / 2, 0, 51252292925.0 / 65400821598.0 / 2,
    -2691868925.0 / 45128329728.0 / 2, 187940372067.0 / 1594534317056.0 / 2,
    -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2};

/* _optimal_step_size: misc.py:94-103 */
double phxo_optimal_step_size(double last_step, float error_ratio, double safety, double ifactor,
                              double dfactor, int order)
{
    if (error_ratio == 0.0f) return last_step * ifactor;
    if (error_ratio < 1.0f) dfactor = 1.0;
    double er = (double)error_ratio;
    double exponent = 1.0 / (double)order;
    double factor = safety / pow(er, exponent);
    /* torch.max / torch.min propagate NaN (a NaN error ratio makes dt NaN => 'underflow in dt') */
    if (factor == factor && !(factor > dfactor)) factor = dfactor;
    if (factor == factor && !(factor < ifactor)) factor = ifactor;
    return last_step * factor;
}

/* _interp_fit: interp.py:1-22 ; coef rows: e, d, c, b, a */
void phxo_interp_fit(const float *y0, const float *y1, const float *ym, const float *f0,
                     const float *f1, float dt, long long n, float *coef)
{
    float *e = coef, *d = coef + n, *c = coef + 2 * n, *b = coef + 3 * n, *a = coef + 4 * n;
    const float two_dt = 2.0f * dt;
    for (long long i = 0; i < n; ++i) {
        a[i] = (two_dt * (f1[i] - f0[i]) - 8.0f * (y1[i] + y0[i])) + 16.0f * ym[i];
        b[i] = ((dt * (5.0f * f0[i] - 3.0f * f1[i]) + 18.0f * y0[i]) + 14.0f * y1[i]) - 32.0f * ym[i];
        c[i] = ((dt * (f1[i] - 4.0f * f0[i]) - 11.0f * y0[i]) - 5.0f * y1[i]) + 16.0f * ym[i];
        d[i] = dt * f0[i];
        e[i] = y0[i];
    }
}

/* _interp_evaluate: interp.py:25-47 */
void phxo_interp_eval(const float *coef, long long n, double t0, double t1, double t, float *out)
{
    double x = (t - t0) / (t1 - t0);
    double xp = x;
    const float x1 = (float)x;
    for (long long i = 0; i < n; ++i) out[i] = coef[i] + x1 * coef[n + i];
    for (int c = 2; c < 5; ++c) {
        xp = xp * x;
        const float xpf = (float)xp;
        const float *cc = coef + (size_t)c * n;
        for (long long i = 0; i < n; ++i) out[i] = out[i] + xpf * cc[i];
    }
}

/* _select_initial_step: misc.py:47-86 (order = solver.order - 1 = 4, rk_common.py:143) */
static double select_initial_step(sys_t *s, const float *y0, const float *f0, double rtol,
                                  double atol, float *w /* 3*len */)
{
    const long long n = s->len;
    const float rt = (float)rtol, at = (float)atol;
    float *scale = w, *t1 = w + n, *t2 = w + 2 * n;
    for (long long i = 0; i < n; ++i) scale[i] = at + fabsf(y0[i]) * rt;
    for (long long i = 0; i < n; ++i) t1[i] = y0[i] / scale[i];
    float d0 = sys_norm(s, t1);
    for (long long i = 0; i < n; ++i) t1[i] = f0[i] / scale[i];
    float d1 = sys_norm(s, t1);
    float h0;
    if (d0 < 1e-5f || d1 < 1e-5f)
        h0 = 1e-6f;
    else
        h0 = (0.01f * d0) / d1;
    for (long long i = 0; i < n; ++i) t1[i] = y0[i] + h0 * f0[i];
    sys_f(s, t1, t2); /* f1 = func(t0 + h0, y1) */
    for (long long i = 0; i < n; ++i) t2[i] = (t2[i] - f0[i]) / scale[i];
    float d2 = sys_norm(s, t2) / h0;
    float h1;
    if (d1 <= 1e-15f && d2 <= 1e-15f) {
        h1 = h0 * 1e-3f;
        if (h1 < 1e-6f) h1 = 1e-6f;
    } else {
        float m = d1 > d2 ? d1 : d2;
        h1 = powf(0.01f / m, (float)(1.0 / 5.0));
    }
    float r = 100.0f * h0;
    if (h1 < r) r = h1;
    return (double)r;
}

/* AdaptiveStepsizeODESolver.integrate (solvers.py:23-30) with
 * RKAdaptiveStepsizeODESolver (rk_common.py:140-228).  t increasing. */
static int solve_dopri5(sys_t *s, const float *y0_in, const double *t, int T, double rtol,
                        double atol, float *sol)
{
    const long long n = s->len;
    const double safety = 0.9, ifactor = 10.0, dfactor = 0.2;
    const int order = 5;
    float *k = (float *)malloc(sizeof(float) * (size_t)n * (7 + 1 + 1 + 1 + 5 + 3));
    if (!k) return PHXO_ERR_BAD_ARG;
    float *y0 = k + 7 * n, *y1 = y0 + n, *tmp = y1 + n, *coef = tmp + n, *w = coef + 5 * n;
    float *f0 = k; /* k[0] */
    int status = PHXO_OK;

    memcpy(sol, y0_in, sizeof(float) * n);
    memcpy(y0, y0_in, sizeof(float) * n);
    /* _before_integrate  rk_common.py:140-148 */
    sys_f(s, y0, f0);
    double dt = select_initial_step(s, y0, f0, rtol, atol, w);
    double rk_t0 = t[0], rk_t1 = t[0];
    for (int c = 0; c < 5; ++c) memcpy(coef + (size_t)c * n, y0, sizeof(float) * n);

    const float rt = (float)rtol, at = (float)atol;
    float alpha32[6], beta32[6][6], cerr32[7], cmid32[7];
    for (int i = 0; i < 6; ++i) {
        alpha32[i] = (float)DP_ALPHA[i];
        for (int j = 0; j <= i; ++j) beta32[i][j] = (float)DP_BETA[i][j];
    }
    for (int i = 0; i < 7; ++i) {
        cerr32[i] = (float)DP_CERR[i];
        cmid32[i] = (float)DP_CMID[i];
    }
    (void)alpha32;

    for (int it = 1; it < T && status == PHXO_OK; ++it) {
        const double next_t = t[it];
        long long n_steps = 0;
        /* _advance  rk_common.py:150-157 */
        while (next_t > rk_t1) {
            if (n_steps >= s->max_steps) { status = PHXO_ERR_MAX_STEPS; break; }
            /* _adaptive_step  rk_common.py:159-220 ; (y0, f0, t0) <- (y1, f1, t1) of state */
            const double t0 = rk_t1;
            if (!(t0 + dt > t0)) { status = PHXO_ERR_DT_UNDERFLOW; break; }
            int finite = 1;
            for (long long i = 0; i < n; ++i)
                if (!isfinite(y0[i])) { finite = 0; break; }
            if (!finite) { status = PHXO_ERR_NONFINITE; break; }

            /* _runge_kutta_step  rk_common.py:39-77 */
            const float dtf = (float)dt;
            for (int i = 0; i < 6; ++i) {
                float c[6];
                for (int j = 0; j <= i; ++j) c[j] = beta32[i][j] * dtf; /* beta_i * dt */
                for (long long e = 0; e < n; ++e) {
                    float acc = 0.0f;
                    for (int j = 0; j <= i; ++j) acc += k[(size_t)j * n + e] * c[j];
                    y1[e] = y0[e] + acc;
                }
                sys_f(s, y1, k + (size_t)(i + 1) * n);
            }
            /* y1 = last yi (c_sol == beta[-1]); f1 = k[6]; y1_error = k . (dt*c_error) */
            float ce[7];
            for (int j = 0; j < 7; ++j) ce[j] = dtf * cerr32[j];
            for (long long e = 0; e < n; ++e) {
                float acc = 0.0f;
                for (int j = 0; j < 7; ++j) acc += k[(size_t)j * n + e] * ce[j];
                float ay0 = fabsf(y0[e]), ay1 = fabsf(y1[e]);
                float tol = at + rt * (ay0 > ay1 ? ay0 : ay1); /* misc.py:89-91 */
                tmp[e] = acc / tol;
            }
            const float error_ratio = sys_norm(s, tmp);
            const int accept = error_ratio <= 1.0f;
            s->nsteps++;
            if (accept) {
                /* _interp_fit (rk_common.py:222-228) */
                float cm[7];
                for (int j = 0; j < 7; ++j) cm[j] = dtf * cmid32[j];
                for (long long e = 0; e < n; ++e) {
                    float acc = 0.0f;
                    for (int j = 0; j < 7; ++j) acc += k[(size_t)j * n + e] * cm[j];
                    tmp[e] = y0[e] + acc; /* y_mid */
                }
                phxo_interp_fit(y0, y1, tmp, k, k + 6 * (size_t)n, dtf, n, coef);
                rk_t0 = t0;
                rk_t1 = t0 + dt;
                memcpy(y0, y1, sizeof(float) * n);
                memcpy(k, k + 6 * (size_t)n, sizeof(float) * n); /* f_next = f1 (FSAL) */
            } else {
                rk_t0 = t0;
                rk_t1 = t0;
            }
            dt = phxo_optimal_step_size(dt, error_ratio, safety, ifactor, dfactor, order);
            n_steps++;
        }
        if (status != PHXO_OK) break;
        phxo_interp_eval(coef, n, rk_t0, rk_t1, next_t, sol + (size_t)it * n);
    }
    free(k);
    return status;
}

/* ------------------------------------------------------------------ */
/* odeint (odeint.py:30-74, misc.py:165-241)                           */
/* ------------------------------------------------------------------ */
typedef struct {
    const phxo_net *net;
    int B;
    float sign; /* -1 => _ReverseFunc (misc.py:156-162) */
} fwd_ctx;

static void fwd_f(void *c, const float *y, float *dy)
{
    fwd_ctx *x = (fwd_ctx *)c;
    phxo_rhs(x->net, y, dy, x->B, 0);
    if (x->sign < 0) {
        const long long n = (long long)x->B * x->net->N;
        for (long long i = 0; i < n; ++i) dy[i] = -dy[i];
    }
}

static int t_decreasing(const double *t, int T)
{
    if (T < 2) return 0;
    for (int i = 0; i + 1 < T; ++i)
        if (!(t[i + 1] < t[i])) return 0;
    return 1;
}

int phxo_odeint(const phxo_net *net, const float *y0, int B, const double *t, int T, int t_is_f32,
                int method, double rtol, double atol, float *sol, long long *nfe,
                long long *nsteps)
{
    if (T < 1 || B < 1) return PHXO_ERR_BAD_ARG;
    fwd_ctx ctx = {net, B, 1.0f};
    double *tt = (double *)malloc(sizeof(double) * T);
    if (t_decreasing(t, T)) { /* misc.py:210-212: t = -t, func = _ReverseFunc(func) */
        for (int i = 0; i < T; ++i) tt[i] = -t[i];
        ctx.sign = -1.0f;
    } else {
        memcpy(tt, t, sizeof(double) * T);
    }
    for (int i = 0; i + 1 < T; ++i)
        if (!(tt[i + 1] > tt[i])) { free(tt); return PHXO_ERR_BAD_ARG; } /* misc.py:114-115 */
    sys_t s;
    memset(&s, 0, sizeof s);
    s.f = fwd_f;
    s.ctx = &ctx;
    s.len = (long long)B * net->N;
    s.nblocks = 1;
    s.max_steps = 2147483647LL;
    int st;
    if (method == PHXO_DOPRI5)
        st = solve_dopri5(&s, y0, tt, T, rtol, atol, sol);
    else
        st = solve_fixed(&s, method, y0, tt, T, t_is_f32, sol);
    if (nfe) *nfe = s.nfe;
    if (nsteps) *nsteps = s.nsteps;
    free(tt);
    return st;
}

/* ------------------------------------------------------------------ */
/* adjoint backward: torchdiffeq/_impl/adjoint.py:32-162               */
/* ------------------------------------------------------------------ */
typedef struct {
    const phxo_net *net;
    int B;
    float base_sign; /* forward problem was time-reversed by _check_inputs */
    float *cot;      /* scratch [B*N] */
} aug_ctx;

/* flat augmented state: [vjp_t (1)] [y (B*N)] [adj_y (B*N)] [theta (P)],
 * theta stored as g[N], Wp[H*N], bp[H], Ws[H*N], bs[H], Wa[N*2H]
 * (= module.parameters() order, adjoint.py:207-220; only the block as a whole
 *  matters to the norm). */
static void theta_views(float *th, int N, int H, phxo_grads *v)
{
    v->g = th;
    v->Wp = v->g + N;
    v->bp = v->Wp + (size_t)H * N;
    v->Ws = v->bp + H;
    v->bs = v->Ws + (size_t)H * N;
    v->Wa = v->bs + H;
}

/* _ReverseFunc(_TupleFunc(augmented_dynamics))  (adjoint.py:94-127; misc.py:145-162):
 * the backward solves always run on a decreasing time pair, so the result is negated. */
static void aug_f_rev(void *c, const float *st, float *d)
{
    aug_ctx *x = (aug_ctx *)c;
    const int N = x->net->N, H = x->net->H;
    const long long BN = (long long)x->B * N;
    const long long P = 4LL * H * N + 2 * H + N;
    const float *y = st + 1, *adj = st + 1 + BN;
    float *d_y = d + 1, *d_adj = d + 1 + BN, *d_th = d + 1 + 2 * BN;
    phxo_grads gv;
    theta_views(d_th, N, H, &gv);
    memset(d_th, 0, sizeof(float) * P);
    /* cotangent = -adj_y (adjoint.py:117); base func possibly reversed => f_base = s*f */
    const float cs = -x->base_sign;
    for (long long i = 0; i < BN; ++i) x->cot[i] = cs * adj[i];
    phxo_rhs_vjp(x->net, y, x->cot, x->B, d_adj, &gv, d_y, 0);
    /* func_eval = base_sign * f ; then _ReverseFunc negates everything */
    const float fs = -x->base_sign;
    for (long long i = 0; i < BN; ++i) d_y[i] = fs * d_y[i];
    for (long long i = 0; i < BN; ++i) d_adj[i] = -d_adj[i];
    for (long long i = 0; i < P; ++i) d_th[i] = -d_th[i];
    d[0] = -0.0f; /* vjp_t = zeros (t does not require grad) */
}

int phxo_adjoint_backward(const phxo_net *net, int B, const double *t_in, int T, int t_is_f32,
                          int method, double rtol, double atol, const float *y_saved,
                          const float *grad_y, int theta_in_norm, float *adj_y0,
                          phxo_grads *grads, long long *nfe, long long *nsteps)
{
    const int N = net->N, H = net->H;
    const long long BN = (long long)B * N;
    const long long P = 4LL * H * N + 2 * H + N;
    const long long len = 1 + 2 * BN + P;
    if (T < 1) return PHXO_ERR_BAD_ARG;
    double *t = (double *)malloc(sizeof(double) * T);
    aug_ctx ctx;
    ctx.net = net;
    ctx.B = B;
    ctx.base_sign = 1.0f;
    if (t_decreasing(t_in, T)) {
        for (int i = 0; i < T; ++i) t[i] = -t_in[i];
        ctx.base_sign = -1.0f;
    } else {
        memcpy(t, t_in, sizeof(double) * T);
    }
    ctx.cot = (float *)malloc(sizeof(float) * BN);
    float *state = (float *)calloc((size_t)len, sizeof(float));
    float *sol = (float *)malloc(sizeof(float) * 2 * (size_t)len);
    /* aug_state = [0, y[-1], grad_y[-1], zeros_like(params)]   adjoint.py:86-87 */
    memcpy(state + 1, y_saved + (size_t)(T - 1) * BN, sizeof(float) * BN);
    memcpy(state + 1 + BN, grad_y + (size_t)(T - 1) * BN, sizeof(float) * BN);

    sys_t s;
    memset(&s, 0, sizeof s);
    s.f = aug_f_rev;
    s.ctx = &ctx;
    s.len = len;
    s.max_steps = 2147483647LL;
    /* _mixed_linf_rms_norm over [t, y, adj_y, params]   adjoint.py:72-78 */
    s.nblocks = 4;
    s.blocks[0] = 1;
    s.blocks[1] = BN;
    s.blocks[2] = BN;
    s.blocks[3] = theta_in_norm ? P : 0;
    if (!theta_in_norm) {
        /* engine variant: the theta block is still integrated by the same RK algebra but is
         * left out of the norm (phxo_mixed_norm then only walks the [t, y, adj] prefix) */
        s.nblocks = 3;
    }
    int st = PHXO_OK;
    for (int i = T - 1; i >= 1 && st == PHXO_OK; --i) {
        /* odeint(aug, state, t[i-1:i+1].flip(0)) -> reversed pair (-t[i], -t[i-1]) */
        double tp[2] = {-t[i], -t[i - 1]};
        if (method == PHXO_DOPRI5) {
            st = solve_dopri5(&s, state, tp, 2, rtol, atol, sol);
        } else {
            st = solve_fixed(&s, method, state, tp, 2, t_is_f32, sol);
        }
        memcpy(state, sol + len, sizeof(float) * len); /* a[1] */
        /* aug_state[1] = y[i-1]; aug_state[2] += grad_y[i-1]    adjoint.py:152-154 */
        memcpy(state + 1, y_saved + (size_t)(i - 1) * BN, sizeof(float) * BN);
        const float *gy = grad_y + (size_t)(i - 1) * BN;
        float *a = state + 1 + BN;
        for (long long e = 0; e < BN; ++e) a[e] += gy[e];
    }
    memcpy(adj_y0, state + 1 + BN, sizeof(float) * BN);
    if (grads) {
        phxo_grads gv;
        theta_views(state + 1 + 2 * BN, N, H, &gv);
        for (int n = 0; n < N; ++n) grads->g[n] += gv.g[n];
        for (long long e = 0; e < (long long)H * N; ++e) {
            grads->Wp[e] += gv.Wp[e];
            grads->Ws[e] += gv.Ws[e];
        }
        for (int h = 0; h < H; ++h) {
            grads->bp[h] += gv.bp[h];
            grads->bs[h] += gv.bs[h];
        }
        for (long long e = 0; e < (long long)N * 2 * H; ++e) grads->Wa[e] += gv.Wa[e];
    }
    if (nfe) *nfe = s.nfe;
    if (nsteps) *nsteps = s.nsteps;
    free(sol);
    free(state);
    free(ctx.cot);
    free(t);
    return st;
}

/* ------------------------------------------------------------------ */
/* reference-shaped per-sample loops (train_insilico.py:128-130)       */
/* ------------------------------------------------------------------ */
int phxo_odeint_per_sample(const phxo_net *net, const float *y0, int B, const double *t, int T,
                           int t_is_f32, int method, double rtol, double atol, float *sol,
                           long long *nfe, long long *nsteps, int nthreads)
{
    const int N = net->N;
    int status = PHXO_OK;
    long long tot_nfe = 0, tot_steps = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic) num_threads(nthreads) reduction(+ : tot_nfe, tot_steps)
    for (int b = 0; b < B; ++b) {
        long long f = 0, s = 0;
        int st = phxo_odeint(net, y0 + (size_t)b * N, 1, t + (size_t)b * T, T, t_is_f32, method,
                             rtol, atol, sol + (size_t)b * T * N, &f, &s);
        tot_nfe += f;
        tot_steps += s;
        if (st != PHXO_OK) {
#pragma omp critical
            status = st;
        }
    }
    if (nfe) *nfe = tot_nfe;
    if (nsteps) *nsteps = tot_steps;
    return status;
}

int phxo_adjoint_backward_per_sample(const phxo_net *net, int B, const double *t, int T,
                                     int t_is_f32, int method, double rtol, double atol,
                                     const float *y_saved, const float *grad_y,
                                     int theta_in_norm, float *adj_y0, phxo_grads *grads,
                                     long long *nfe, long long *nsteps, int nthreads)
{
    const int N = net->N, H = net->H;
    const long long P = 4LL * H * N + 2 * H + N;
    int status = PHXO_OK;
    long long tot_nfe = 0, tot_steps = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    (void)nthreads;
#endif
#pragma omp parallel num_threads(nthreads) reduction(+ : tot_nfe, tot_steps)
    {
        /* per-thread gradient accumulator, summed in thread order at the end */
        float *loc = (float *)calloc((size_t)P, sizeof(float));
        phxo_grads lg;
        theta_views(loc, N, H, &lg);
#pragma omp for schedule(dynamic)
        for (int b = 0; b < B; ++b) {
            long long f = 0, s = 0;
            int st = phxo_adjoint_backward(net, 1, t + (size_t)b * T, T, t_is_f32, method, rtol,
                                           atol, y_saved + (size_t)b * T * N,
                                           grad_y + (size_t)b * T * N, theta_in_norm,
                                           adj_y0 + (size_t)b * N, grads ? &lg : NULL, &f, &s);
            tot_nfe += f;
            tot_steps += s;
            if (st != PHXO_OK) {
#pragma omp critical
                status = st;
            }
        }
        if (grads) {
#pragma omp critical
            {
                for (int n = 0; n < N; ++n) grads->g[n] += lg.g[n];
                for (long long e = 0; e < (long long)H * N; ++e) {
                    grads->Wp[e] += lg.Wp[e];
                    grads->Ws[e] += lg.Ws[e];
                }
                for (int h = 0; h < H; ++h) {
                    grads->bp[h] += lg.bp[h];
                    grads->bs[h] += lg.bs[h];
                }
                for (long long e = 0; e < (long long)N * 2 * H; ++e) grads->Wa[e] += lg.Wa[e];
            }
        }
        free(loc);
    }
    if (nfe) *nfe = tot_nfe;
    if (nsteps) *nsteps = tot_steps;
    return status;
}
