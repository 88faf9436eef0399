// Micro-benchmark (diagnostic, not the product): do fp32 matrix instructions (v_mfma_f32_32x32x2_f32) and VALU / transcendental
// instructions overlap on one SIMD of gfx950 -- inside one wave, and between the two waves that share a SIMD?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__device__ __forceinline__ float mfma_work(int n, float a, float b) {
    f32x16 acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][15];
    return s;
}
// the epilogue's instruction mix: per element sub, mul, exp, add, rcp, 2 fma (16 independent elements)
__device__ __forceinline__ float valu_work(int n, float x) {
    float v[16], s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = x + (float)r;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z = (v[r] - 0.25f) * -1.442695f;
            const float h = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
            s0 = fmaf(h, 0.5f, s0); s1 = fmaf(h, 0.25f, s1);
            v[r] = h;
        }
    }
    return s0 + s1;
}
// both in one wave, interleaved by the scheduler hints: one matrix instruction, then 8 VALU
template <int NACC>
__device__ __forceinline__ float both_work(int n, float a, float b, float x) {
    f32x16 acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.0f;
    float v[16], s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = x + (float)r;
    for (int i = 0; i < n; ++i) {                               // per iteration: 16 MFMA (4 rounds of NACC... scaled below) + 16 elements
#pragma unroll
        for (int k = 0; k < 16 / NACC; ++k)
#pragma unroll
            for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z = (v[r] - 0.25f) * -1.442695f;
            const float h = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
            s0 = fmaf(h, 0.5f, s0); s1 = fmaf(h, 0.25f, s1);
            v[r] = h;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
        }
    }
    float s = s0 + s1;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c][0];
    return s;
}

typedef short bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bmfma_work(int n, bf16x8 a, bf16x8 b) {      // ONE accumulator: a dependent chain
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    }
    return acc[0] + acc[15];
}
// the split-operand forward pass's mix: 18 bf16 matrix instructions per 16 sigmoid elements, interleaved
template <bool HINT>
__device__ __forceinline__ float bboth_work(int n, bf16x8 a, bf16x8 b, float x) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float v[16], s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = x + (float)r;
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int k = 0; k < 18; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float z = (v[r] - 0.25f) * -1.442695f;
            const float h = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z));
            s0 = fmaf(h, 0.5f, s0); s1 = fmaf(h, 0.25f, s1);
            v[r] = h;
        }
        if (HINT) {
#pragma unroll
            for (int k = 0; k < 18; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
            }
        }
    }
    return s0 + s1 + acc[0];
}

// role per wave: 0 idle, 1 matrix (4 accumulators), 2 VALU, 3 both interleaved, 4 matrix with ONE accumulator (dependent chain)
__global__ void __launch_bounds__(512) k(const int* roles, int n, float* out, unsigned long long* cyc, unsigned* hwid) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int role = roles[wave];
    const float a = 1.0f + lane * 1e-3f, b = 0.5f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float r = 0.0f;
    if (role == 1) r = mfma_work<4>(n, a, b);
    else if (role == 2) r = valu_work(n, a);
    else if (role == 3) r = both_work<4>(n, a, b, a);
    else if (role == 4) r = mfma_work<1>(4 * n, a, b);
    else if (role == 5 || role == 6 || role == 7) {
        bf16x8 ba, bb;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ba[j] = (short)(0x3f80 + lane + j); bb[j] = (short)0x3f00; }
        if (role == 5) r = bmfma_work(n, ba, bb);                // 4 n bf16 matrix instructions
        else if (role == 6) r = bboth_work<true>(n, ba, bb, a);  // 18 n matrix instructions + 16 n elements
        else r = bboth_work<false>(n, ba, bb, a);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = r;
    if (lane == 0) { cyc[wave] = t1 - t0; hwid[wave] = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11)); }
}

int main() {
    int* d_roles; float* d_out; unsigned long long* d_cyc; unsigned* d_hw;
    hipMalloc(&d_roles, 8 * sizeof(int)); hipMalloc(&d_out, 512 * sizeof(float)); hipMalloc(&d_cyc, 8 * 8); hipMalloc(&d_hw, 8 * 4);
    const int n = 256;                                          // role 1: 4 n = 1024 MFMA; role 2: 16 n = 4096 elements; role 3: 16 n MFMA + 16 n elements
    struct Case { const char* name; int roles[8]; };
    const Case cases[] = {
        {"matrix alone (4 acc), wave 0", {1, 0, 0, 0, 0, 0, 0, 0}},
        {"matrix alone (1 acc chain), wave 0", {4, 0, 0, 0, 0, 0, 0, 0}},
        {"VALU alone, wave 0", {2, 0, 0, 0, 0, 0, 0, 0}},
        {"matrix wave 0 + VALU wave 4 (same SIMD?)", {1, 0, 0, 0, 2, 0, 0, 0}},
        {"matrix wave 0 + VALU wave 1 (other SIMD?)", {1, 2, 0, 0, 0, 0, 0, 0}},
        {"matrix wave 0 + matrix wave 4", {1, 0, 0, 0, 1, 0, 0, 0}},
        {"matrix wave 0 + matrix wave 1", {1, 1, 0, 0, 0, 0, 0, 0}},
        {"VALU wave 0 + VALU wave 4", {2, 0, 0, 0, 2, 0, 0, 0}},
        {"VALU wave 0 + VALU wave 1", {2, 2, 0, 0, 0, 0, 0, 0}},
        {"both interleaved in wave 0 (4x the matrix work of 'alone')", {3, 0, 0, 0, 0, 0, 0, 0}},
        {"both interleaved in waves 0 and 4", {3, 0, 0, 0, 3, 0, 0, 0}},
        {"bf16 matrix alone (1 acc chain, 1024), wave 0", {5, 0, 0, 0, 0, 0, 0, 0}},
        {"bf16 matrix wave 0 + VALU wave 4", {5, 0, 0, 0, 2, 0, 0, 0}},
        {"bf16 matrix waves 0 and 4", {5, 0, 0, 0, 5, 0, 0, 0}},
        {"bf16 18 n + 16 n elements, hinted, wave 0", {6, 0, 0, 0, 0, 0, 0, 0}},
        {"bf16 18 n + 16 n elements, compiler order, wave 0", {7, 0, 0, 0, 0, 0, 0, 0}},
        {"bf16 18 n + 16 n elements, hinted, waves 0 and 4", {6, 0, 0, 0, 6, 0, 0, 0}},
        {"bf16 18 n + 16 n elements, compiler order, waves 0 and 4", {7, 0, 0, 0, 7, 0, 0, 0}},
    };
    for (const Case& c : cases) {
        hipMemcpy(d_roles, c.roles, sizeof c.roles, hipMemcpyHostToDevice);
        unsigned long long cyc[8]; unsigned hw[8];
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, d_roles, n, d_out, d_cyc, d_hw);
            hipDeviceSynchronize();
        }
        hipMemcpy(cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost);
        hipMemcpy(hw, d_hw, sizeof hw, hipMemcpyDeviceToHost);
        printf("%-62s", c.name);
        for (int w = 0; w < 8; ++w) if (c.roles[w]) printf("  w%d(simd %u): %7llu", w, (hw[w] >> 4) & 3, cyc[w]);
        printf("\n");
    }
    printf("bf16 rows: role 5 = 1024 v_mfma_f32_32x32x16_bf16 (32 pipe cycles each = 32768); roles 6/7 = 4608 of them + 4096 sigmoid elements (143 k alone)\n");
    printf("units: s_memtime ticks; role 1 = 1024 MFMA 32x32x2 f32 (64 pipe cycles each = 65536), role 2 = 4096 sigmoid elements\n");
    return 0;
}
