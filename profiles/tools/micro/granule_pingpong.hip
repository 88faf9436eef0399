// Micro-benchmark (diagnostic, not the product): what does one {tag, value} granule cost to cross from one work-group to another
// on gfx950, by cache policy of the store / the polling load and by placement (same XCD / different XCDs)?
// The tree, multi-CU speculative and packed-multi schedules exchange such granules every round (ptnn_device.hpp: granule_store /
// granule_wait, relaxed agent-scope atomics): the Iris tree's PMC passes show 1.0 MB fetched + 0.4 MB written per swap interval
// for records that are 64 B per node and round.
//   hipcc --offload-arch=gfx950 -O3 -o granule_pingpong granule_pingpong.hip && ./granule_pingpong [reps]
// Work-groups are dispatched round-robin over the 8 XCDs, so with a grid of 16 blocks, blocks 0 and 8 share an XCD (and its L2) and
// blocks 0 and 1 do not.  Block A stores tag k, block B waits for it and answers, A waits: `reps` round trips, timed with
// s_memrealtime (100 MHz).  Every spin is bounded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
constexpr unsigned SPIN_LIMIT = 1u << 20;

enum { V_AGENT = 0, V_WORKGROUP, V_SYSTEM, V_ASM_SC0, V_ASM_SC1, V_ASM_SC0SC1, V_ASM_NT, V_ASM_PLAIN_ST_SC0_LD, V_PLAIN_ST_NT_LD, V_SC0_ST_NT_LD, V_NT_ST_SC1_LD, V_COUNT };
static const char* NAMES[V_COUNT] = {"atomic relaxed, agent scope (the product's)", "atomic relaxed, workgroup scope", "atomic relaxed, system scope",
                                     "asm store sc0 / load sc0", "asm store sc1 / load sc1", "asm store sc0 sc1 / load sc0 sc1", "asm store nt / load nt",
                                     "asm store (no bits) / load sc0", "asm store (no bits) / load nt", "asm store sc0 / load nt", "asm store nt / load sc1"};

template <int V>
__device__ __forceinline__ void g_store(u64* p, u64 x) {
    if (V == V_AGENT) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (V == V_WORKGROUP) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (V == V_SYSTEM) __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (V == V_ASM_SC0 || V == V_SC0_ST_NT_LD) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(x) : "memory");
    else if (V == V_ASM_SC1) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
    else if (V == V_ASM_SC0SC1) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(x) : "memory");
    else if (V == V_ASM_NT || V == V_NT_ST_SC1_LD) asm volatile("global_store_dwordx2 %0, %1, off nt" ::"v"(p), "v"(x) : "memory");
    else asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(x) : "memory");
}
template <int V>
__device__ __forceinline__ u64 g_load(const u64* p) {
    u64 x;
    if (V == V_AGENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (V == V_WORKGROUP) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (V == V_SYSTEM) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (V == V_ASM_SC0 || V == V_ASM_PLAIN_ST_SC0_LD) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    else if (V == V_ASM_SC1 || V == V_NT_ST_SC1_LD) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    else if (V == V_ASM_SC0SC1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx2 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(p) : "memory");
    return x;
}
template <int V>
__device__ __forceinline__ bool g_wait(const u64* p, unsigned tag) {
    for (unsigned s = 0; s < SPIN_LIMIT; ++s) {
        if ((unsigned)(g_load<V>(p) >> 32) == tag) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// ga: written by A, polled by B; gb: written by B, polled by A (separate 64-byte lines)
template <int V>
__global__ void pingpong(u64* ga, u64* gb, int blockA, int blockB, int reps, unsigned base, u64* out) {
    if (threadIdx.x != 0) return;
    const int b = blockIdx.x;
    if (b == blockA) {
        bool ok = true;
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        for (int k = 1; k <= reps && ok; ++k) {
            g_store<V>(ga, ((u64)(base + k) << 32) | (u64)k);
            ok = g_wait<V>(gb, base + k);
        }
        const u64 t1 = __builtin_amdgcn_s_memrealtime();
        out[0] = t1 - t0;
        out[1] = ok ? 1 : 0;
    } else if (b == blockB) {
        bool ok = true;
        for (int k = 1; k <= reps && ok; ++k) {
            ok = g_wait<V>(ga, base + k);
            g_store<V>(gb, ((u64)(base + k) << 32) | (u64)k);
        }
        out[2] = ok ? 1 : 0;
    }
}

template <int V>
static void run(u64* d, u64* d_out, int reps, unsigned& base) {
    for (int placement = 0; placement < 2; ++placement) {
        const int A = 0, B = placement == 0 ? 8 : 1;
        u64 h[3] = {0, 0, 0};
        (void)hipMemset(d_out, 0, 3 * sizeof(u64));
        hipLaunchKernelGGL(pingpong<V>, dim3(16), dim3(64), 0, 0, d, d + 64, A, B, reps, base, d_out);
        const hipError_t e = hipDeviceSynchronize();
        (void)hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
        base += (unsigned)reps + 16;
        printf("%-46s blocks 0 <-> %d (%s): %s  round trip %.0f ns (one way %.0f ns)\n", NAMES[V], B, placement == 0 ? "same XCD" : "other XCD",
               (e == hipSuccess && h[1] && h[2]) ? "ok     " : "TIMEOUT", 10.0 * (double)h[0] / reps, 5.0 * (double)h[0] / reps);
        fflush(stdout);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 2000;
    const int only = argc > 2 ? atoi(argv[2]) : -1;      // one variant only (for a rocprofv3 --pmc pass)
    u64 *d = nullptr, *d_out = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess || hipMalloc(&d_out, 64) != hipSuccess) return 1;
    (void)hipMemset(d, 0, 4096);
    unsigned base = 16;
#define RUN(V) if (only < 0 || only == V) run<V>(d, d_out, reps, base);
    RUN(V_AGENT) RUN(V_WORKGROUP) RUN(V_SYSTEM) RUN(V_ASM_SC0) RUN(V_ASM_SC1) RUN(V_ASM_SC0SC1) RUN(V_ASM_NT) RUN(V_ASM_PLAIN_ST_SC0_LD)
    RUN(V_PLAIN_ST_NT_LD) RUN(V_SC0_ST_NT_LD) RUN(V_NT_ST_SC1_LD)
    return 0;
}
