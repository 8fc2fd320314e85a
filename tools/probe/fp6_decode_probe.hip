// fp6_decode_probe.hip — v_cvt_scalef32_pk32_f16_fp6 / ..._f32_fp6 (gfx950): which element of the result comes from which 6-bit
// field, and the direction of the scale.  Needed by k_sib_children (net_kernels.hip), which reads the base position's fc0 operand
// entries back (f16 hi + block-scaled fp6 residual) to form the child's difference row.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef float f32v __attribute__((ext_vector_type(32)));
typedef unsigned int u6 __attribute__((ext_vector_type(6)));
static float dec6(unsigned e) { // e2m3
    const int s = (e >> 5) & 1, ex = (e >> 3) & 3, m = e & 7;
    const float v = ex == 0 ? m * 0.125f : (1.0f + m * 0.125f) * (float)(1 << (ex - 1));
    return s ? -v : v;
}
__global__ void k(const unsigned* in, float scale, float* o16, float* o32) {
    u6 a;
    for (int i = 0; i < 6; ++i) a[i] = in[i];
    const h32 r = __builtin_amdgcn_cvt_scalef32_pk32_f16_fp6(a, scale);
    const f32v q = __builtin_amdgcn_cvt_scalef32_pk32_f32_fp6(a, scale);
    if (threadIdx.x == 0) for (int i = 0; i < 32; ++i) { o16[i] = (float)r[i]; o32[i] = q[i]; }
}
int main() {
    unsigned code[32], pk[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 32; ++i) { // a permutation of codes with both signs: field i holds code (i * 7 + 3) & 63
        code[i] = (unsigned)(i * 7 + 3) & 63u;
        const int bit = 6 * i;
        pk[bit >> 5] |= code[i] << (bit & 31);
        if ((bit & 31) > 26) pk[(bit >> 5) + 1] |= code[i] >> (32 - (bit & 31));
    }
    unsigned* din; float *d16, *d32;
    hipMalloc(&din, 24); hipMalloc(&d16, 128); hipMalloc(&d32, 128);
    hipMemcpy(din, pk, 24, hipMemcpyHostToDevice);
    for (float scale : {1.0f, 0.25f, 0.000244140625f /* 2^-12 */, 3.814697265625e-06f /* 2^-18 */}) {
        k<<<1, 64>>>(din, scale, d16, d32);
        float o16[32], o32[32];
        hipMemcpy(o16, d16, 128, hipMemcpyDeviceToHost); hipMemcpy(o32, d32, 128, hipMemcpyDeviceToHost);
        int ok16 = 1, ok32 = 1, inv16 = 1;
        for (int i = 0; i < 32; ++i) {
            const float want = dec6(code[i]) * scale;
            if (o16[i] != (float)(_Float16)want) ok16 = 0;
            if (o32[i] != want) ok32 = 0;
            if (o16[i] != (float)(_Float16)(dec6(code[i]) / scale)) inv16 = 0;
        }
        printf("scale %g: pk32_f16_fp6 element i = field i * scale: %s (field i / scale: %s)   pk32_f32_fp6: %s\n", scale, ok16 ? "YES" : "no",
               inv16 ? "YES" : "no", ok32 ? "YES" : "no");
        if (!ok16 || !ok32) { printf("  f16:"); for (int i = 0; i < 32; ++i) printf(" %g", o16[i]); printf("\n  want:"); for (int i = 0; i < 32; ++i) printf(" %g", dec6(code[i]) * scale); printf("\n"); }
    }
    return 0;
}
