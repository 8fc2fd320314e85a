// overlap_probe.hip — do VALU instructions issue underneath MFMAs on gfx950?  One workgroup per CU, W waves per SIMD;
// each loop iteration issues 4 independent 32x32x16 f16 MFMAs (accumulators in VGPRs or AGPRs) and NV independent
// v_fma_f32.  If the two overlap, time stays flat until 4*NV clk of VALU exceeds the 128 clk of MFMA pipe time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
template <int NV, bool AGPR, bool DEP, int KIND = 0> // KIND 0: v_fma_f32, 1: s_add_u32, 2: ds_read_b128 (+ one lgkmcnt(0) per iteration)
__global__ __launch_bounds__(512) void probe(float* out, int iters, const float4* gsrc_in = nullptr) {
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    h8 ha, hb;
    for (int i = 0; i < 8; ++i) { ha[i] = (_Float16)(threadIdx.x * 0.001f + i); hb[i] = (_Float16)(1.0f + i * 0.01f); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.5f + i;
    const float m = 1.0001f, a = 0.5f;
    __shared__ float4 lds[2048];
    lds[threadIdx.x] = float4{1, 2, 3, 4};
    __syncthreads();
    const uint32_t ldsaddr = threadIdx.x * 16;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float4* gsrc = gsrc_in + wv * 64;
    const uint32_t goff = (threadIdx.x & 63) * 16;
    const uint32_t ldsbase = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)lds + wv * 4096;
    uint32_t sc[4] = {1, 2, 3, 4};
    float4 q[4] = {};
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v gq[4] = {};
    for (int it = 0; it < iters; ++it) {
        if (KIND == 5 || KIND == 6) {
#define MF(cx) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(cx) : "v"(ha), "v"(hb))
#define LD(i) do { if (KIND == 5) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(gq[i]) : "v"(goff), "s"(gsrc) : "memory"); \
                   else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(goff), "s"(gsrc), "s"(ldsbase + (i) * 1024) : "memory", "m0"); } while (0)
            MF(c0); if (NV > 0) LD(0);
            MF(c1); if (NV > 1) LD(1);
            MF(c2); if (NV > 2) LD(2);
            MF(c3); if (NV > 3) LD(3);
            if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            continue;
        }
        if (KIND == 3) {
#define VF(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(k) & 7]) : "v"(m), "v"(a))
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(ha), "v"(hb));
#pragma unroll
            for (int k = 0; k < NV / 4; ++k) VF(k);
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c1) : "v"(ha), "v"(hb));
#pragma unroll
            for (int k = 0; k < NV / 4; ++k) VF(k + 2);
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c2) : "v"(ha), "v"(hb));
#pragma unroll
            for (int k = 0; k < NV / 4; ++k) VF(k + 4);
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c3) : "v"(ha), "v"(hb));
#pragma unroll
            for (int k = 0; k < NV / 4; ++k) VF(k + 6);
            continue;
        }
        if (AGPR) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c0) : "v"(ha), "v"(hb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(DEP ? c0 : c1) : "v"(ha), "v"(hb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(DEP ? c0 : c2) : "v"(ha), "v"(hb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(DEP ? c0 : c3) : "v"(ha), "v"(hb));
        } else {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c0) : "v"(ha), "v"(hb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(DEP ? c0 : c1) : "v"(ha), "v"(hb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(DEP ? c0 : c2) : "v"(ha), "v"(hb));
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(DEP ? c0 : c3) : "v"(ha), "v"(hb));
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(m), "v"(a));
            else if (KIND == 1) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sc[k & 3]) : : "scc");
            else asm volatile("ds_read_b128 %0, %1" : "=v"(q[k & 3]) : "v"(ldsaddr));
        }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += sc[i] + q[i].x + q[i].w + gq[i][0] + gq[i][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NV, bool AGPR, bool DEP, int KIND = 0>
void run(float* d, int wpg, const char* tag) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 20000;
    probe<NV, AGPR, DEP, KIND><<<256, wpg * 64>>>(d, 100, (const float4*)d);
    hipEventRecord(a);
    probe<NV, AGPR, DEP, KIND><<<256, wpg * 64>>>(d, iters, (const float4*)d);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-28s waves/SIMD %d  NV %2d: %7.1f ns/iter/wave-slot  (x waves/SIMD = SIMD time per 4 MFMA + NV VALU per wave)\n", tag, wpg / 4, NV, ms * 1e6 / iters);
}
int main() {
    float* d; hipMalloc(&d, 256 * 512 * 4);
    run<0, true, false, 5>(d, 4, "1 global_load_dwordx4 per gap"); run<1, true, false, 5>(d, 4, "1 global_load_dwordx4 per gap"); run<2, true, false, 5>(d, 4, "1 global_load_dwordx4 per gap"); run<4, true, false, 5>(d, 4, "1 global_load_dwordx4 per gap");
    run<1, true, false, 6>(d, 4, "1 LDS-DMA per gap"); run<2, true, false, 6>(d, 4, "1 LDS-DMA per gap"); run<4, true, false, 6>(d, 4, "1 LDS-DMA per gap");
    run<8, true, false, 3>(d, 4, "AGPR interleaved v_fma"); run<16, true, false, 3>(d, 4, "AGPR interleaved v_fma"); run<24, true, false, 3>(d, 4, "AGPR interleaved v_fma"); run<32, true, false, 3>(d, 4, "AGPR interleaved v_fma"); run<64, true, false, 3>(d, 4, "AGPR interleaved v_fma");
    run<0, true, false, 1>(d, 4, "AGPR + s_add"); run<32, true, false, 1>(d, 4, "AGPR + s_add"); run<64, true, false, 1>(d, 4, "AGPR + s_add");
    run<0, true, false, 2>(d, 4, "AGPR + ds_read_b128"); run<8, true, false, 2>(d, 4, "AGPR + ds_read_b128"); run<16, true, false, 2>(d, 4, "AGPR + ds_read_b128");
    for (int wpg = 4; wpg <= 4; wpg += 4) {
        run<0, false, false>(d, wpg, "acc VGPR indep"); run<8, false, false>(d, wpg, "acc VGPR indep"); run<16, false, false>(d, wpg, "acc VGPR indep"); run<32, false, false>(d, wpg, "acc VGPR indep"); run<64, false, false>(d, wpg, "acc VGPR indep");
        run<0, true, false>(d, wpg, "acc AGPR indep"); run<16, true, false>(d, wpg, "acc AGPR indep"); run<32, true, false>(d, wpg, "acc AGPR indep"); run<64, true, false>(d, wpg, "acc AGPR indep");
        run<0, false, true>(d, wpg, "acc VGPR dependent chain"); run<16, false, true>(d, wpg, "acc VGPR dependent chain"); run<32, false, true>(d, wpg, "acc VGPR dependent chain");
        run<0, true, true>(d, wpg, "acc AGPR dependent chain"); run<32, true, true>(d, wpg, "acc AGPR dependent chain");
    }
    return 0;
}
