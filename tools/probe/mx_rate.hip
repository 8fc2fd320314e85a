// Issue-rate probe: f16 32x32x16 MFMA vs block-scaled fp8 32x32x64 MFMA (gfx950), one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    h8 ha, hb; v8i ia, ib;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.001f + i); hb[i] = (_Float16)(1.0f + i * 0.01f); ia[i] = 0x38383838 + threadIdx.x; ib[i] = 0x3a3a3a3a; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c3, 0, 0, 0);
        } else if (MODE == 1) {
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, c0, 0, 0, 0, 127, 0, 127);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, c1, 0, 0, 0, 127, 0, 127);
            c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, c2, 0, 0, 0, 127, 0, 127);
            c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, c3, 0, 0, 0, 127, 0, 127);
        } else { // the fc0 mix per K=64: 4 f16 + 2 scaled fp8 on one accumulator, x2 accumulators
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, c1, 0, 0, 0);
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, c0, 0, 0, 0, 110, 0, 120);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ia, ib, c1, 0, 0, 0, 110, 0, 120);
            c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ib, ia, c0, 0, 0, 0, 110, 0, 120);
            c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ib, ia, c1, 0, 0, 0, 110, 0, 120);
        }
    }
    float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 1024 * 256 * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        if (mode == 0) rate<0><<<1024, 256>>>(d, iters); else if (mode == 1) rate<1><<<1024, 256>>>(d, iters); else rate<2><<<1024, 256>>>(d, iters);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        // per SIMD: 1024 WG * 4 waves / 1024 SIMDs = 4 waves per SIMD sequential-ish (one resident per SIMD at a time? no: 4 WGs/CU)
        const double insts = mode == 2 ? 12.0 : 4.0;
        printf("mode %d: %.3f ms, %.1f ns per loop iteration per wave-slot\n", mode, ms, ms * 1e6 / iters);
        (void)insts;
    }
    return 0;
}
