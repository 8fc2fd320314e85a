// fp6_probe.hip — semantics needed for fp6 (e2m3) correction terms in fc0 (gfx950):
//  1. v_cvt_scalef32_pk32_fp6_f16 / v_cvt_scalef32_2xpk16_fp6_f32: which 6-bit field gets which source element, scale direction
//  2. v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 2: operand bit layout (assumed: lane l = row/col l&31, k = 32*(l>>5)+i
//     in bits [6i, 6i+5] of the lane's 192 bits) and PER-LANE scale bytes (assumed: a lane's byte scales its own 32-k block)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned int u6 __attribute__((ext_vector_type(6)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
static float dec6(unsigned e) { // e2m3
    const int s = (e >> 5) & 1, ex = (e >> 3) & 3, m = e & 7;
    const float v = ex == 0 ? m * 0.125f : (1.0f + m * 0.125f) * (float)(1 << (ex - 1));
    return s ? -v : v;
}
__global__ void k_cvt(const _Float16* in16, const float* in32, float scale, unsigned* out) {
    h32 a;
    for (int i = 0; i < 32; ++i) a[i] = in16[i];
    u6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(a, scale);
    f16v x, y;
    for (int i = 0; i < 16; ++i) { x[i] = in32[i]; y[i] = in32[16 + i]; }
    u6 q = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x, y, scale);
    if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) { out[i] = r[i]; out[6 + i] = q[i]; }
}
__global__ void k_mfma(const unsigned* a6, const unsigned* b6, const unsigned* sa, const unsigned* sb, float* out) {
    const int l = threadIdx.x;
    v8i a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) { a[i] = (int)a6[l * 6 + i]; b[i] = (int)b6[l * 6 + i]; }
    v16f c = {0};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, (int)sa[l], 0, (int)sb[l]);
    for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
int main() {
    // ---- 1. conversions
    _Float16 h16[32]; float h32v[32];
    for (int i = 0; i < 32; ++i) { h32v[i] = dec6(i) * 4.0f; h16[i] = (_Float16)h32v[i]; } // all 32 non-negative codes, pre-multiplied by the scale
    _Float16* d16; float* d32; unsigned* dout;
    hipMalloc(&d16, sizeof(h16)); hipMalloc(&d32, sizeof(h32v)); hipMalloc(&dout, 64);
    hipMemcpy(d16, h16, sizeof(h16), hipMemcpyHostToDevice); hipMemcpy(d32, h32v, sizeof(h32v), hipMemcpyHostToDevice);
    k_cvt<<<1, 64>>>(d16, d32, 4.0f, dout);
    unsigned o[12]; hipMemcpy(o, dout, 48, hipMemcpyDeviceToHost);
    for (int which = 0; which < 2; ++which) {
        unsigned long long lo = 0; unsigned char bytes[24]; memcpy(bytes, o + 6 * which, 24); (void)lo;
        printf("%s: field i holds code:", which ? "2xpk16_fp6_f32(x = codes 0..15, y = 16..31)" : "pk32_fp6_f16(codes 0..31)     ");
        int ok = 1;
        for (int i = 0; i < 32; ++i) {
            const int bit = 6 * i; unsigned v = 0;
            for (int b = 0; b < 6; ++b) v |= ((bytes[(bit + b) >> 3] >> ((bit + b) & 7)) & 1u) << b;
            printf(" %u", v); if ((int)v != i) ok = 0;
        }
        printf("  -> %s (value / scale, field i = element i)\n", ok ? "IDENTITY" : "permuted");
    }
    // ---- 2. MFMA layout with per-lane scales
    srand(3);
    static float A[32][64], B[64][32]; static unsigned ca[32][64], cb[64][32];
    for (int r = 0; r < 32; ++r) for (int k = 0; k < 64; ++k) { ca[r][k] = rand() & 63; A[r][k] = dec6(ca[r][k]); }
    for (int k = 0; k < 64; ++k) for (int c = 0; c < 32; ++c) { cb[k][c] = rand() & 63; B[k][c] = dec6(cb[k][c]); }
    unsigned a6[64 * 6] = {0}, b6[64 * 6] = {0}, sa[64], sb[64];
    auto put = [](unsigned* w, int i, unsigned code) { const int bit = 6 * i; for (int b = 0; b < 6; ++b) if ((code >> b) & 1) w[(bit + b) >> 5] |= 1u << ((bit + b) & 31); };
    for (int l = 0; l < 64; ++l) {
        for (int i = 0; i < 32; ++i) { put(a6 + l * 6, i, ca[l & 31][32 * (l >> 5) + i]); put(b6 + l * 6, i, cb[32 * (l >> 5) + i][l & 31]); }
        sa[l] = 127 + (l % 3); sb[l] = 126 + ((l >> 2) % 3);
    }
    unsigned *da, *db, *dsa, *dsb; float* dc;
    hipMalloc(&da, sizeof(a6)); hipMalloc(&db, sizeof(b6)); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dc, 64 * 16 * 4);
    hipMemcpy(da, a6, sizeof(a6), hipMemcpyHostToDevice); hipMemcpy(db, b6, sizeof(b6), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(da, db, dsa, dsb, dc);
    static float C[64 * 16]; hipMemcpy(C, dc, sizeof(C), hipMemcpyDeviceToHost);
    double maxerr = 0; int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 16; ++r) {
        const int col = l & 31, row = (r >> 2) * 8 + (l >> 5) * 4 + (r & 3);
        double ref = 0;
        for (int k = 0; k < 64; ++k) {
            const int la = row + 32 * (k >> 5), lb = col + 32 * (k >> 5); // the lane that supplied this (row, k-block) / (col, k-block)
            ref += (double)A[row][k] * ldexp(1.0, (int)sa[la] - 127) * (double)B[k][col] * ldexp(1.0, (int)sb[lb] - 127);
        }
        const double e = fabs(ref - C[l * 16 + r]); if (e > maxerr) maxerr = e; if (e > 1e-3 * (1 + fabs(ref))) ++bad;
    }
    printf("mfma fp6 x fp6, per-lane scales: %d / 1024 outputs off, max abs err %.3g  -> %s\n", bad, maxerr, bad ? "layout / scale assumption WRONG" : "layout and per-lane block scales as assumed");
    return 0;
}
