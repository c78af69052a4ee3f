// phx_host.hpp -- host-side helpers shared by the translation units of libphoenix_hip.so.
// Inline variables (C++17): one instance per shared library, whichever TU references them.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>
#include <utility>
#include <vector>

#include "phx_solver.hpp"

namespace phxh {

// CU count of the CURRENT device (cached per device ordinal)
inline std::mutex g_mu;
inline std::map<int, int> g_cus;
inline int num_cus()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cus.find(dev);
    if (it != g_cus.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    g_cus[dev] = n;
    return n;
}

// Raises the dynamic-LDS limit of a kernel.  hipFuncAttributeMaxDynamicSharedMemorySize is a per-device, per-function
// attribute: the cache is keyed by (device, function pointer) and guarded by a mutex; the driver call is made only
// when a launch needs more than was granted before.
inline std::map<std::pair<int, const void *>, size_t> g_lds_granted;
inline bool set_lds_fn(const void *fn, size_t bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lk(g_mu);
    auto key = std::make_pair(dev, fn);
    auto it = g_lds_granted.find(key);
    if (it != g_lds_granted.end() && it->second >= bytes) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
    g_lds_granted[key] = bytes;
    return true;
}
template <typename K>
inline bool set_lds(K kernel, size_t bytes) { return set_lds_fn(reinterpret_cast<const void *>(kernel), bytes); }

// Co-residency guard of the persistent kernels (they spin on each other's rows): the launch is refused unless the
// occupancy query says that `grid` workgroups of this shape fit the device at once.  Cached per (device, fn, threads, lds).
inline std::map<std::tuple<int, const void *, int, size_t>, int> g_occ;
inline bool fits_resident(const void *fn, int threads, size_t lds, int grid)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    int per_cu = -1;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_occ.find(std::make_tuple(dev, fn, threads, lds));
        if (it != g_occ.end()) per_cu = it->second;
    }
    if (per_cu < 0) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, lds) != hipSuccess) return false;
        per_cu = n;
        std::lock_guard<std::mutex> lk(g_mu);
        g_occ[std::make_tuple(dev, fn, threads, lds)] = per_cu;
    }
    return (long long)per_cu * num_cus() >= grid;
}

// persistent kernels: LDS limit + co-residency check in one call (PHX_ERR_LAUNCH instead of a grid that would spin)
template <typename K>
inline bool set_lds_resident(K kernel, size_t bytes, int threads, int grid)
{
    const void *fn = reinterpret_cast<const void *>(kernel);
    return set_lds_fn(fn, bytes) && fits_resident(fn, threads, bytes, grid);
}

// Persistent kernels whose workgroups wait for each other's rows need the whole grid resident at once.  Two ways to get
// that: (default) a plain launch behind the occupancy check of fits_resident -- the library refuses a grid the device
// cannot hold, and the engine's stream must not share the device with another long-running kernel (launch
// planning); or PHX_COOP=1: hipLaunchCooperativeKernel, where the runtime itself guarantees co-residency.  The
// cooperative path is NOT the default because it costs 18-19 us per launch on this stack (C4, same box, round 4:
// forward 0.286 vs 0.268 ms, backward 0.616 vs 0.598 ms, step 0.939 vs 0.892 ms); it is there for deployments that
// cannot keep other long kernels off the device.
inline bool use_coop()
{
    const char *e = getenv("PHX_COOP");
    return e && e[0] == '1';
}
template <typename... Args>
inline hipError_t launch_persistent(const void *fn, dim3 grid, dim3 block, size_t lds, hipStream_t st, Args... args)
{
    void *argv[] = {(void *)&args...};
    if (use_coop()) return hipLaunchCooperativeKernel(fn, grid, block, argv, (unsigned int)lds, st);
    return hipLaunchKernel(fn, grid, block, argv, lds, st);
}

// diagnostic: optional HIP events recorded immediately around the next solve kernel (bench.py roofline timing)
inline thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;
// ... and a process-wide FIFO of such pairs (phx_debug_queue_kernel_events): successive solve launches take one pair each,
// whichever thread issues them (the backward solve of a training step runs on an autograd worker thread), so a bench can
// time the kernels of ordinary back-to-back steps instead of isolated launches
inline std::mutex g_evq_mu;
inline std::vector<std::pair<hipEvent_t, hipEvent_t>> g_evq;
inline void ev_begin(hipStream_t st)
{
    if (!g_ev_start) {
        std::lock_guard<std::mutex> lk(g_evq_mu);
        if (!g_evq.empty()) {
            g_ev_start = g_evq.front().first;
            g_ev_stop = g_evq.front().second;
            g_evq.erase(g_evq.begin());
        }
    }
    if (g_ev_start) (void)hipEventRecord(g_ev_start, st);
}
inline void ev_end(hipStream_t st)
{
    if (g_ev_stop) (void)hipEventRecord(g_ev_stop, st);
    g_ev_start = nullptr;
    g_ev_stop = nullptr;
}

inline Net to_net(const phx_params *p) { return Net{p->Ws, p->bs, p->Wp, p->bp, p->WaT, p->g, p->N, p->H}; }

constexpr size_t LDS_BUDGET = 163840 - 1024;

inline bool force_v0()
{
    const char *e = getenv("PHX_ENGINE");
    return e && strcmp(e, "v0") == 0;
}

}  // namespace phxh

// ---- second-generation adjoint kernel (phx_adj2.hip): plain-type entry points used by the C ABI in phx_engine.hip
namespace phxh {
// trajectories per launch for this shape (B when one launch takes the whole batch), 0: no plan -> caller falls back
int adj2_chunk(int N, int H, int B, int T, int control);
size_t adj2_workspace_bytes(int N, int H, int B, int T);   // max over the control modes (0 when unplanned)
int adj2_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6);
// runs the whole batch (in chunks when needed); returns a PHX_* status
int adj2_run(const phx_params *p, const double *t_all, int B, int T, const phx_solve_opts *o, const float *y_saved_all,
             const float *grad_y_all, float *adj_y0_all, const phx_grads *grads, int *status_all, int *nfe_all,
             int *nsteps_all, void *workspace, size_t workspace_bytes, hipStream_t st);
}  // namespace phxh

// ---- third backward kernel (phx_adj3.hip: dopri5, H <= 48, fused sweeps over a 16-vector private state)
namespace phxh {
int adj3_chunk(int N, int H, int B, int T, int control, int method);
size_t adj3_workspace_bytes(int N, int H, int B, int T);   // max over the control modes (0 when unplanned)
int adj3_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6);
int adj3_run(const phx_params *p, const double *t_all, int B, int T, const phx_solve_opts *o, const float *y_saved_all,
             const float *grad_y_all, float *adj_y0_all, const phx_grads *grads, int *status_all, int *nfe_all,
             int *nsteps_all, void *workspace, size_t workspace_bytes, hipStream_t st);
}  // namespace phxh

// ---- third-generation forward solve (phx_fwd3.hip: dopri5, H <= 48)
namespace phxh {
int fwd3_chunk(int N, int H, int B, int T, int control, int method);
size_t fwd3_workspace_bytes(int N, int H, int B, int T);
int fwd3_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6);
int fwd3_run(const phx_params *p, const float *y0_all, const double *t_all, int B, int T, const phx_solve_opts *o,
             float *sol_all, int *status_all, int *nfe_all, int *nsteps_all, void *workspace, size_t workspace_bytes,
             hipStream_t st);
}  // namespace phxh

// ---- third-generation kernels for wide hidden layers (phx_fwd3c.hip / phx_adj3c.hip: dopri5, 48 < H <= 256, hidden chunks
// of <= 48 rows)
namespace phxh {
int fwd3c_chunk(int N, int H, int B, int T, int control, int method);
size_t fwd3c_workspace_bytes(int N, int H, int B, int T);
int fwd3c_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6);
int fwd3c_run(const phx_params *p, const float *y0_all, const double *t_all, int B, int T, const phx_solve_opts *o,
              float *sol_all, int *status_all, int *nfe_all, int *nsteps_all, void *workspace, size_t workspace_bytes,
              hipStream_t st);
int adj3c_chunk(int N, int H, int B, int T, int control, int method);
size_t adj3c_workspace_bytes(int N, int H, int B, int T);
int adj3c_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6);
int adj3c_run(const phx_params *p, const double *t_all, int B, int T, const phx_solve_opts *o, const float *y_saved_all,
              const float *grad_y_all, float *adj_y0_all, const phx_grads *grads, int *status_all, int *nfe_all,
              int *nsteps_all, void *workspace, size_t workspace_bytes, hipStream_t st);
}  // namespace phxh
