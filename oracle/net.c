/* ORACLE (test infrastructure only) — fp32 forward of the policy/value net.
 *
 * Restates the inference graph of alpha-zero/src/network.rs:51-262 built from
 * network-utils/src/lib.rs:95-170 (conv2d + BiasAdd), :172-262 (depthwise 3x3 + pointwise + bias),
 * :285-330 (fc), :386-461 (bottleneck residual: add BEFORE the activation, network.rs:108-111).
 * LeakyRelu alpha is TF's default 0.2 (no attr set, network.rs:77).  NHWC, stride 1, SAME.
 *
 * The arithmetic itself lives in libtensorflow (tensorflow 0.21.0 / tensorflow-sys 0.24.0,
 * Cargo.lock:3530-3559), absent from the reference tree, and the reference has no tests at
 * this boundary: PARITY UNPINNED by the reference.  This is the published op semantics in
 * plain fp32 (k-ascending sums, bias added after the sum), cross-checked against an
 * independent torch implementation by tools/make_golden.py (tests/golden/net_*.npz).
 *
 * Tensor order = reference variable order (network.rs:78-79,113-122,149-150,162-163,201-202,
 * 240-241): conv_w[1,1,3,128] conv_b | x3 { w0[1,1,128,32] b0 dw[3,3,32,1] pw[1,1,32,32] b1
 * w2[1,1,32,128] b2 } | fc0_w[128*HW,512] fc0_b fc1_w[512,512] fc1_b v_w[512,1] v_b p_w[512,HW] p_b
 */
#include "omok_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NT 31
#define C 128
#define M 32
#define F 512

struct orc_net {
    int n, hw;
    float* t[NT];
    int64_t size[NT];
};

int orc_net_num_tensors(void) { return NT; }

orc_net* orc_net_create(int n) {
    orc_net* net = (orc_net*)calloc(1, sizeof(orc_net));
    net->n = n;
    net->hw = n * n;
    const int64_t hw = net->hw;
    int64_t* s = net->size;
    s[0] = 3 * C; s[1] = C;
    for (int b = 0; b < 3; ++b) {
        int64_t* q = s + 2 + 7 * b;
        q[0] = C * M; q[1] = M; q[2] = 9 * M; q[3] = M * M; q[4] = M; q[5] = M * C; q[6] = C;
    }
    s[23] = C * hw * F; s[24] = F; s[25] = F * F; s[26] = F; s[27] = F; s[28] = 1; s[29] = F * hw; s[30] = hw;
    for (int i = 0; i < NT; ++i) net->t[i] = (float*)calloc((size_t)s[i], sizeof(float));
    return net;
}

void orc_net_destroy(orc_net* net) {
    if (!net) return;
    for (int i = 0; i < NT; ++i) free(net->t[i]);
    free(net);
}

int64_t orc_net_tensor_size(const orc_net* net, int idx) { return idx < 0 || idx >= NT ? -1 : net->size[idx]; }

int orc_net_load(orc_net* net, int idx, const float* data, int64_t count) {
    if (idx < 0 || idx >= NT || count != net->size[idx]) return -1;
    memcpy(net->t[idx], data, sizeof(float) * (size_t)count);
    return 0;
}

static inline float lrelu(float x) { return x > 0.0f ? x : 0.2f * x; }

/* trunk for one sample: in [3*HW] -> x [HW][128] */
static inline __attribute__((always_inline)) void trunk(const orc_net* net, const float* in, float* x, float* h, float* d, float* g) {
    const int n = net->n, hw = net->hw;
    const float* cw = net->t[0];
    const float* cb = net->t[1];
    for (int i = 0; i < hw; ++i) {
        float* xo = x + (size_t)i * C;
        for (int o = 0; o < C; ++o) xo[o] = 0.0f;
        for (int c = 0; c < 3; ++c) {
            const float a = in[3 * i + c]; /* NHWC view of the flat buffer */
            for (int o = 0; o < C; ++o) xo[o] += a * cw[c * C + o];
        }
        for (int o = 0; o < C; ++o) xo[o] = lrelu(xo[o] + cb[o]);
    }
    for (int b = 0; b < 3; ++b) {
        float* const* t = (float* const*)(net->t + 2 + 7 * b);
        const float *w0 = t[0], *b0 = t[1], *dw = t[2], *pw = t[3], *b1 = t[4], *w2 = t[5], *b2 = t[6];
        for (int i = 0; i < hw; ++i) { /* 1x1 128->32 + bias + lrelu */
            float acc[M];
            for (int o = 0; o < M; ++o) acc[o] = 0.0f;
            const float* xi = x + (size_t)i * C;
            for (int k = 0; k < C; ++k) {
                const float a = xi[k];
                for (int o = 0; o < M; ++o) acc[o] += a * w0[k * M + o];
            }
            for (int o = 0; o < M; ++o) h[(size_t)i * M + o] = lrelu(acc[o] + b0[o]);
        }
        for (int y = 0; y < n; ++y) /* depthwise 3x3 SAME, no bias */
            for (int xx = 0; xx < n; ++xx) {
                float acc[M];
                for (int o = 0; o < M; ++o) acc[o] = 0.0f;
                for (int dy = 0; dy < 3; ++dy)
                    for (int dx = 0; dx < 3; ++dx) {
                        const int yy = y + dy - 1, xq = xx + dx - 1;
                        if (yy < 0 || yy >= n || xq < 0 || xq >= n) continue;
                        const float* hi = h + (size_t)(yy * n + xq) * M;
                        const float* wk = dw + (dy * 3 + dx) * M;
                        for (int o = 0; o < M; ++o) acc[o] += hi[o] * wk[o];
                    }
                for (int o = 0; o < M; ++o) d[(size_t)(y * n + xx) * M + o] = acc[o];
            }
        for (int i = 0; i < hw; ++i) { /* pointwise 32->32 + bias + lrelu */
            float acc[M];
            for (int o = 0; o < M; ++o) acc[o] = 0.0f;
            const float* di = d + (size_t)i * M;
            for (int k = 0; k < M; ++k) {
                const float a = di[k];
                for (int o = 0; o < M; ++o) acc[o] += a * pw[k * M + o];
            }
            for (int o = 0; o < M; ++o) g[(size_t)i * M + o] = lrelu(acc[o] + b1[o]);
        }
        for (int i = 0; i < hw; ++i) { /* 1x1 32->128 + bias, + x, lrelu */
            float acc[C];
            for (int o = 0; o < C; ++o) acc[o] = 0.0f;
            const float* gi = g + (size_t)i * M;
            for (int k = 0; k < M; ++k) {
                const float a = gi[k];
                for (int o = 0; o < C; ++o) acc[o] += a * w2[k * C + o];
            }
            float* xi = x + (size_t)i * C;
            for (int o = 0; o < C; ++o) xi[o] = lrelu((acc[o] + b2[o]) + xi[o]);
        }
    }
}

#define SB 8 /* samples per weight pass */

/* lg / vpre (optional): the policy head's output in front of the softmax (network.rs:227-247) and the value head's in front of tanh */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
__attribute__((target_clones("avx512f", "avx2", "default")))
#endif
void orc_net_forward_impl_(const orc_net* net, const float* in, int B, float* p, float* v, float* lg, float* vpre, int threads) {
    const int hw = net->hw;
    const int64_t K0 = (int64_t)C * hw;
    if (threads < 1) threads = 1;
    const int nblk = (B + SB - 1) / SB;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int blk = 0; blk < nblk; ++blk) {
        const int s0 = blk * SB, ns = (B - s0) < SB ? (B - s0) : SB;
        float* x = (float*)malloc(sizeof(float) * (size_t)K0 * SB);
        float* h = (float*)malloc(sizeof(float) * (size_t)hw * M * 3);
        float* h0 = (float*)malloc(sizeof(float) * F * SB * 2);
        float* h1 = h0 + F * SB;
        for (int s = 0; s < ns; ++s)
            trunk(net, in + (size_t)(s0 + s) * 3 * (size_t)hw, x + (size_t)s * (size_t)K0, h, h + (size_t)hw * M, h + (size_t)hw * M * 2);
        /* fc0 */
        for (int i = 0; i < F * SB; ++i) h0[i] = 0.0f;
        const float* w = net->t[23];
        for (int64_t k = 0; k < K0; ++k) {
            const float* wk = w + k * F;
            for (int s = 0; s < ns; ++s) {
                const float a = x[(size_t)s * (size_t)K0 + (size_t)k];
                float* acc = h0 + s * F;
                for (int o = 0; o < F; ++o) acc[o] += a * wk[o];
            }
        }
        for (int s = 0; s < ns; ++s)
            for (int o = 0; o < F; ++o) h0[s * F + o] = lrelu(h0[s * F + o] + net->t[24][o]);
        /* fc1 */
        for (int s = 0; s < ns; ++s) {
            float* acc = h1 + s * F;
            for (int o = 0; o < F; ++o) acc[o] = 0.0f;
            for (int k = 0; k < F; ++k) {
                const float a = h0[s * F + k];
                const float* wk = net->t[25] + (size_t)k * F;
                for (int o = 0; o < F; ++o) acc[o] += a * wk[o];
            }
            for (int o = 0; o < F; ++o) acc[o] = lrelu(acc[o] + net->t[26][o]);
        }
        /* heads */
        for (int s = 0; s < ns; ++s) {
            const float* a1 = h1 + s * F;
            float vv = 0.0f;
            for (int k = 0; k < F; ++k) vv += a1[k] * net->t[27][k];
            v[s0 + s] = tanhf(vv + net->t[28][0]);
            if (vpre) vpre[s0 + s] = vv + net->t[28][0];
            float logits[ORC_MAX_HW];
            for (int o = 0; o < hw; ++o) logits[o] = 0.0f;
            for (int k = 0; k < F; ++k) {
                const float a = a1[k];
                const float* wk = net->t[29] + (size_t)k * (size_t)hw;
                for (int o = 0; o < hw; ++o) logits[o] += a * wk[o];
            }
            float mx = -INFINITY;
            for (int o = 0; o < hw; ++o) { logits[o] += net->t[30][o]; if (logits[o] > mx) mx = logits[o]; }
            if (lg) for (int o = 0; o < hw; ++o) lg[(size_t)(s0 + s) * (size_t)hw + o] = logits[o];
            float sum = 0.0f;
            float* po = p + (size_t)(s0 + s) * (size_t)hw;
            for (int o = 0; o < hw; ++o) { po[o] = expf(logits[o] - mx); sum += po[o]; }
            for (int o = 0; o < hw; ++o) po[o] = po[o] / sum;
        }
        free(x); free(h); free(h0);
    }
}

void orc_net_forward(const orc_net* net, const float* in, int B, float* p, float* v, int threads) { orc_net_forward_impl_(net, in, B, p, v, NULL, NULL, threads); }
void orc_net_forward_logits(const orc_net* net, const float* in, int B, float* p, float* v, float* logits, float* vpre, int threads) {
    orc_net_forward_impl_(net, in, B, p, v, logits, vpre, threads);
}
