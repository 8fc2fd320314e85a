// cvt_probe.hip — semantics of v_cvt_scalef32_pk_fp8_f16 on gfx950 (scale direction, rounding, saturation)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short short2v __attribute__((ext_vector_type(2)));
__global__ void k(const _Float16* x, float scale, uint32_t* out_new, uint32_t* out_old, float mul, int n) {
    int i = threadIdx.x + blockIdx.x * blockDim.x;
    if (i * 2 >= (n < 0 ? -n : n)) return;
    if (n < 0) { n = -n; asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 23, 1), 1"); }
    half2v s = {x[2 * i], x[2 * i + 1]};
    short2v o = {0, 0};
    short2v r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(o, s, scale, false);
    out_new[i] = (uint32_t)(uint16_t)r[0];
    float f0 = (float)s[0] * mul, f1 = (float)s[1] * mul;
    out_old[i] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false) & 0xFFFFu;
}
int main() {
    const int n = 4096;
    _Float16 h[n];
    for (int i = 0; i < n; ++i) {
        float v = (i & 1 ? -1.f : 1.f) * ldexpf(1.0f + (i % 97) / 97.0f, (i / 64) % 24 - 14);
        h[i] = (_Float16)v;
    }
    h[0] = (_Float16)60000.f; h[1] = (_Float16)-60000.f; h[2] = (_Float16)0.f; h[3] = (_Float16)1e-7f;
    _Float16* dx; uint32_t *dn, *dd;
    hipMalloc(&dx, sizeof(h)); hipMalloc(&dn, n * 2); hipMalloc(&dd, n * 2);
    hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
    static uint32_t a[n / 2], b[n / 2];
    for (int e = -3; e <= 3; e += 3) {
        float scale = ldexpf(1.f, e);
        for (int dir = 0; dir < 2; ++dir) {
            float mul = dir ? scale : 1.f / scale;
            k<<<n / 2 / 64, 64>>>(dx, scale, dn, dd, mul, n);
            hipMemcpy(a, dn, n * 2, hipMemcpyDeviceToHost); hipMemcpy(b, dd, n * 2, hipMemcpyDeviceToHost);
            int diff = 0, first = -1;
            for (int i = 0; i < n / 2; ++i) if (a[i] != b[i]) { if (first < 0) first = i; ++diff; }
            printf("scale=2^%d  vs fp8(x %s scale): %d/%d differ", e, dir ? "*" : "/", diff, n / 2);
            if (first >= 0) printf("  first i=%d x=(%g,%g) new=%04x old=%04x", first, (float)h[2 * first], (float)h[2 * first + 1], a[first], b[first]);
            printf("\n");
        }
        printf("   saturation: x=+-60000 -> new=%04x (old path %04x)\n", a[0], b[0]);
    }
    k<<<n / 2 / 64, 64>>>(dx, 1.0f, dn, dd, 1.0f, -n);
    hipMemcpy(a, dn, n * 2, hipMemcpyDeviceToHost); hipMemcpy(b, dd, n * 2, hipMemcpyDeviceToHost);
    printf("FP16_OVFL=1: x=+-60000 -> new=%04x old=%04x ; x=(0,1e-7) -> %04x\n", a[0], b[0], a[1]);
    return 0;
}
