/* ORACLE (test infrastructure only) — the build-defined RNG contract.
 *
 * The reference draws from rand::thread_rng() (unseeded ChaCha12) at
 * alpha-zero/src/parallel_mcts_executor.rs:45,49-53,117 and alpha-zero/src/agent.rs:130-132,
 * so it is not reproducible; "fixed seeds" therefore means the stream defined here, consumed
 * identically by this oracle and by the HIP kernels:
 *
 *   Philox4x32-10, key = seed + episode * 0x9E3779B97F4A7C15 (mod 2^64; orc_stream_key) split into (low, high) words:
 *   the reference draws fresh thread_rng values in every trainer iteration (src/trainer.rs:74-93), so every
 *   reset (= iteration) of a self-play object advances `episode` and with it the whole stream,
 *   counter = (c0 = draw index, c1 = ply, c2 = tree_global = 2*game_global + side, c3 = purpose)
 *     purpose 1 EXPAND: c0 = simulation index inside the execute() call (round*K + i)
 *             -> untried-action index = mulhi(out[0], |A|)
 *     purpose 2 NOISE : c0 = cell*256 + attempt -> Gamma(alpha) draw of that cell
 *     purpose 3 SAMPLE: c0 = 0 -> u = (out[0] >> 8) * 2^-24 for the Boltzmann categorical draw
 *
 * Transcendentals (log/exp) are evaluated by fixed-order f64 polynomial code (no libm, no FMA
 * contraction) so CPU and GPU agree bit for bit.
 */
#include "omok_oracle.h"
#include <string.h>

uint64_t orc_stream_key(uint64_t seed, uint64_t episode) { return seed + episode * 0x9E3779B97F4A7C15ULL; }

void orc_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t out[4]) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double bits_to_double(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static uint64_t double_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* natural log of a positive, normal double. log(m*2^e) = e*ln2 + 2*atanh((m-1)/(m+1)) */
double orc_det_log(double x) {
    uint64_t b = double_to_bits(x);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    double m = bits_to_double((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL); /* [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }
    const double s = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double poly = 1.0 / 23.0;
    poly = poly * s2 + 1.0 / 21.0;
    poly = poly * s2 + 1.0 / 19.0;
    poly = poly * s2 + 1.0 / 17.0;
    poly = poly * s2 + 1.0 / 15.0;
    poly = poly * s2 + 1.0 / 13.0;
    poly = poly * s2 + 1.0 / 11.0;
    poly = poly * s2 + 1.0 / 9.0;
    poly = poly * s2 + 1.0 / 7.0;
    poly = poly * s2 + 1.0 / 5.0;
    poly = poly * s2 + 1.0 / 3.0;
    poly = poly * s2 + 1.0;
    const double lm = 2.0 * s * poly;
    return (double)e * 0.6931471805599453 + lm;
}

double orc_det_exp(double x) {
    if (x > 709.0) return bits_to_double(0x7ff0000000000000ULL);
    if (x < -745.0) return 0.0;
    /* k = nearest integer to x / ln2 */
    double t = x * 1.4426950408889634 + 0.5;
    /* floor via truncation (|t| < 2^31) */
    long long ki = (long long)t;
    if ((double)ki > t) ki = ki - 1;
    const double k = (double)ki;
    const double r = (x - k * 0.6931471803691238) - k * 1.9082149292705877e-10;
    double p = 1.0 / 6227020800.0; /* 1/13! */
    p = p * r + 1.0 / 479001600.0;
    p = p * r + 1.0 / 39916800.0;
    p = p * r + 1.0 / 3628800.0;
    p = p * r + 1.0 / 362880.0;
    p = p * r + 1.0 / 40320.0;
    p = p * r + 1.0 / 5040.0;
    p = p * r + 1.0 / 720.0;
    p = p * r + 1.0 / 120.0;
    p = p * r + 1.0 / 24.0;
    p = p * r + 1.0 / 6.0;
    p = p * r + 0.5;
    p = p * r + 1.0;
    p = p * r + 1.0;
    int kk = (int)ki;
    if (kk < -1000) {
        p = p * bits_to_double((uint64_t)(1023 - 1000) << 52); /* 2^-1000 */
        kk = kk + 1000;
    }
    return p * bits_to_double((uint64_t)(1023 + kk) << 52);
}

float orc_det_expf(float x) { return (float)orc_det_exp((double)x); }

static double u01(uint32_t x) { return ((double)x + 0.5) * 2.3283064365386963e-10; /* 2^-32 */ }

/* Gamma(alpha, 1) draw for one cell.  alpha = k + f, k integer >= 0, f in [0,1):
 *   Gamma(f) by Ahrens-Dieter GS rejection (needs only log/exp), attempts 0..199 use counter
 *   c0 = cell*256 + attempt;  the k unit exponentials use c0 = cell*256 + 255 - i (k <= 32).
 * Draws below 1e-30 are flushed to 0 so no f32 subnormals enter the policy mix. */
float orc_gamma(float alpha_f, uint64_t seed, uint32_t cell, uint32_t ply, uint32_t tree_global) {
    const double alpha = (double)alpha_f;
    int k = (int)alpha;
    if (k > 32) k = 32;
    const double f = alpha - (double)k;
    double x = 0.0;
    uint32_t o[4];
    if (f > 0.0) {
        const double b = 1.0 + f * 0.36787944117144233; /* (e + f) / e */
        double g = 0.0;
        for (uint32_t attempt = 0; attempt < 200; ++attempt) {
            orc_philox(seed, cell * 256u + attempt, ply, tree_global, ORC_RNG_NOISE, o);
            const double u1 = u01(o[0]), u2 = u01(o[1]);
            const double p = b * u1;
            if (p <= 1.0) {
                const double c = orc_det_exp(orc_det_log(p) / f);
                if (u2 <= orc_det_exp(-c)) { g = c; break; }
            } else {
                const double c = -orc_det_log((b - p) / f);
                if (u2 <= orc_det_exp((f - 1.0) * orc_det_log(c))) { g = c; break; }
            }
        }
        x = g;
    }
    for (int i = 0; i < k; ++i) {
        orc_philox(seed, cell * 256u + 255u - (uint32_t)i, ply, tree_global, ORC_RNG_NOISE, o);
        x = x + (-orc_det_log(u01(o[0])));
    }
    if (x < 1e-30) x = 0.0;
    return (float)x;
}
