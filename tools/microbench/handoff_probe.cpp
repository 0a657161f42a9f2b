// Round 6 probe ("measure the agent-scope release / acquire cost first with a 50-line probe; if it is > 3 us per
// hand-off, stop and record it"): what ONE producer -> consumer hand-off between two workgroups of a launch costs -- the step a
// single-launch pyramid chain would take per strip and level (rows of level s made by one workgroup, consumed by another behind a
// completion flag).  Workgroup 0 and workgroup 1 (different XCDs: workgroups go round-robin over them) play ping-pong N times:
// write `payload` bytes, raise a flag; the other side waits for the flag, reads the payload, answers.  Time per ONE-WAY hand-off =
// total / 2N (10 ns ticks of the constant-frequency counter).
//   fences   payload plain stores / loads; __threadfence() (agent-scope release: L2 write-back) before the flag store, an acquire
//            fence (L2 / L1 invalidate) behind the flag load -- the portable form
//   relaxed  every access that crosses the workgroups is a relaxed device-scope atomic access (sc1: written through / read around
//            the non-coherent caches), no fence -- the form select_kernel uses
// Build: hipcc -O3 --offload-arch=gfx950 handoff_probe.cpp -o handoff_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
template <bool FENCES>
__global__ __launch_bounds__(256) void pingpong(int* flag, uint32_t* buf, int words_per_thread, int n, unsigned long long* ticks, uint32_t* sink)
{
    const int me = blockIdx.x, tid = threadIdx.x;
    if (me > 1) return;
    uint32_t acc = 0;
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= n; it++) {
        for (int side = 0; side < 2; side++) {
            uint32_t* mine = buf + (size_t)side * 256 * words_per_thread;
            if (me == side) {                                   // producer of this half-step
                for (int w = 0; w < words_per_thread; w++) {
                    if (FENCES) mine[w * 256 + tid] = (uint32_t)(it + w);
                    else __hip_atomic_store(&mine[w * 256 + tid], (uint32_t)(it + w), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (FENCES) __threadfence(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(&flag[side * 32], it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {                                            // consumer
                if (tid == 0) for (int spin = 0; spin < (1 << 20) && __hip_atomic_load(&flag[side * 32], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < it; spin++) __builtin_amdgcn_s_sleep(1);      // (bounded: a broken build must not hang the GPU)
                __syncthreads();
                if (FENCES) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                for (int w = 0; w < words_per_thread; w++)
                    acc += FENCES ? mine[w * 256 + tid] : __hip_atomic_load(&mine[w * 256 + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (tid == 0 && me == 0) *ticks = wall_clock64() - t0;
    sink[me * 256 + tid] = acc;
}
int main()
{
    int* flag; uint32_t* buf; unsigned long long* ticks; uint32_t* sink;
    (void)hipMalloc(&flag, 256); (void)hipMalloc(&buf, 2 * 256 * 64 * 4); (void)hipMalloc(&ticks, 8); (void)hipMalloc(&sink, 2048);
    const int n = 2000;
    for (int wpt : { 0, 1, 16, 64 }) {
        double us[2];
        for (int f = 0; f < 2; f++) {
            (void)hipMemset(flag, 0, 256);
            if (f) hipLaunchKernelGGL(pingpong<true>, dim3(2), dim3(256), 0, 0, flag, buf, wpt, n, ticks, sink);
            else hipLaunchKernelGGL(pingpong<false>, dim3(2), dim3(256), 0, 0, flag, buf, wpt, n, ticks, sink);
            unsigned long long t = 0;
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
            us[f] = (double)t * 0.01 / (2.0 * n);
        }
        printf("payload %5d bytes per hand-off: relaxed device-scope accesses %.2f us | release / acquire fences %.2f us\n", wpt * 256 * 4, us[0], us[1]);
    }
    return 0;
}
