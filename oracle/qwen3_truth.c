/*
 * qwen3_truth.c — float64, NO-intermediate-rounding forward of one KV-cached Qwen3 W4A16 decode step.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/).  This is the GROUND TRUTH the tolerance of every model-level parity test is
 * derived from: the bf16 oracle (qwen3_decode.c / tiny_oracle.OracleQwen3) and the HIP engine both round activations to
 * bfloat16 at every reference op boundary, in different fp32 summation orders, so they differ from each other by
 * accumulated one-ulp flips.  Neither is "right"; both approximate the real-valued function computed here.  A test then
 * asserts  |HIP - truth| <= c * |oracle - truth|  instead of an arbitrary band between the two rounded paths.
 *
 * Same wiring and the same constants as qwen3_decode.c (reference src/tiny_llm_ref/qwen3_week2.py:96-146,236-247,357-392;
 * dequantisation q*s + beta with the stored bf16 scale / bias, quantize.py:103-121; RMSNorm week2_kernels.metal:41-47;
 * RoPE positional_encoding.py:4-66; attention attention.py:30-66; SwiGLU week2_kernels.metal:115-116), evaluated in
 * double precision with no rounding anywhere.  Weights and norm vectors are the checkpoint's stored bf16 values (exact).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint32_t *w;
    const uint16_t *s;
    const uint16_t *b;
    int rows, cols;
} ot_w4;

typedef struct {
    ot_w4 q, k, v, o, gate, up, down;
    const uint16_t *input_norm, *post_norm, *q_norm, *k_norm;
} ot_layer;

typedef struct {
    int hidden, layers, heads, kv_heads, head_dim, inter, vocab, max_ctx;
    float rope_theta, eps;
} ot_config;

typedef struct {
    ot_config c;
    ot_layer *L;
    ot_w4 embed, head;
    const uint16_t *final_norm;
    double *kc, *vc; /* [layers][max_ctx][kv_heads*head_dim] */
    int ctx;
    double *x, *h, *xn, *q, *k, *v, *att, *gate, *up, *act, *tmp, *scores;
} ot_model;

static inline double bf16_to_d(uint16_t v) {
    uint32_t u = ((uint32_t)v) << 16;
    float f;
    memcpy(&f, &u, 4);
    return (double)f;
}

static void w4_matvec(const ot_w4 *W, const double *a, double *out) {
    const int groups = W->cols / 128, words = W->cols / 8;
#pragma omp parallel for schedule(static)
    for (int r = 0; r < W->rows; ++r) {
        const uint32_t *wr = W->w + (size_t)r * words;
        double acc = 0.0;
        for (int g = 0; g < groups; ++g) {
            const double s = bf16_to_d(W->s[(size_t)r * groups + g]);
            const double z = bf16_to_d(W->b[(size_t)r * groups + g]);
            const double *ag = a + g * 128;
            double part = 0.0;
            for (int j = 0; j < 16; ++j) {
                const uint32_t p = wr[g * 16 + j];
                for (int e = 0; e < 8; ++e) part += ag[j * 8 + e] * ((double)((p >> (4 * e)) & 0xfu) * s + z);
            }
            acc += part;
        }
        out[r] = acc;
    }
}

static void rms_norm(const double *x, const uint16_t *w, double *out, int n, double eps) {
    double ss = 0.0;
    for (int i = 0; i < n; ++i) ss += x[i] * x[i];
    const double inv = 1.0 / sqrt(ss / (double)n + eps);
    for (int i = 0; i < n; ++i) out[i] = x[i] * inv * bf16_to_d(w[i]);
}

static void rope_inplace(double *x, int D, int pos, double base) {
    const int half = D / 2;
    for (int d = 0; d < half; ++d) {
        const double angle = (double)pos * pow(base, -(double)d / (double)half);
        const double c = cos(angle), s = sin(angle);
        const double re = x[d], im = x[d + half];
        x[d] = re * c - im * s;
        x[d + half] = im * c + re * s;
    }
}

ot_model *ot_create(const ot_config *c, const ot_layer *layers, const ot_w4 *embed, const ot_w4 *head,
                    const uint16_t *final_norm) {
    ot_model *m = (ot_model *)calloc(1, sizeof(ot_model));
    m->c = *c;
    m->L = (ot_layer *)malloc(sizeof(ot_layer) * (size_t)c->layers);
    memcpy(m->L, layers, sizeof(ot_layer) * (size_t)c->layers);
    m->embed = *embed;
    m->head = head ? *head : *embed;
    m->final_norm = final_norm;
    const size_t kvw = (size_t)c->kv_heads * c->head_dim;
    m->kc = (double *)calloc((size_t)c->layers * c->max_ctx * kvw, sizeof(double));
    m->vc = (double *)calloc((size_t)c->layers * c->max_ctx * kvw, sizeof(double));
    const int qd = c->heads * c->head_dim;
    m->x = (double *)malloc(sizeof(double) * c->hidden);
    m->h = (double *)malloc(sizeof(double) * c->hidden);
    m->xn = (double *)malloc(sizeof(double) * c->hidden);
    m->tmp = (double *)malloc(sizeof(double) * c->hidden);
    m->q = (double *)malloc(sizeof(double) * qd);
    m->att = (double *)malloc(sizeof(double) * qd);
    m->k = (double *)malloc(sizeof(double) * kvw);
    m->v = (double *)malloc(sizeof(double) * kvw);
    m->gate = (double *)malloc(sizeof(double) * c->inter);
    m->up = (double *)malloc(sizeof(double) * c->inter);
    m->act = (double *)malloc(sizeof(double) * c->inter);
    m->scores = (double *)malloc(sizeof(double) * (size_t)c->heads * c->max_ctx);
    return m;
}

void ot_destroy(ot_model *m) {
    if (!m) return;
    free(m->L); free(m->kc); free(m->vc); free(m->x); free(m->h); free(m->xn); free(m->tmp); free(m->q);
    free(m->att); free(m->k); free(m->v); free(m->gate); free(m->up); free(m->act); free(m->scores);
    free(m);
}

void ot_reset(ot_model *m) { m->ctx = 0; }
int ot_context(const ot_model *m) { return m->ctx; }

/* Feed `token` at position ctx; logits_out[vocab] receives the unrounded float64 logits.  Returns the argmax id, or -1
 * when the cache is full. */
int ot_decode_step(ot_model *m, int token, double *logits_out) {
    const ot_config *c = &m->c;
    if (m->ctx >= c->max_ctx || token < 0 || token >= c->vocab) return -1;
    const int D = c->head_dim, Hq = c->heads, Hkv = c->kv_heads, rep = Hq / Hkv, pos = m->ctx;
    const size_t kvw = (size_t)Hkv * D;
    const double scale = 1.0 / sqrt((double)D);
    {
        const int groups = c->hidden / 128, words = c->hidden / 8;
        for (int j = 0; j < words; ++j) {
            const uint32_t p = m->embed.w[(size_t)token * words + j];
            const double s = bf16_to_d(m->embed.s[(size_t)token * groups + j / 16]);
            const double b = bf16_to_d(m->embed.b[(size_t)token * groups + j / 16]);
            for (int e = 0; e < 8; ++e) m->x[j * 8 + e] = (double)((p >> (4 * e)) & 0xfu) * s + b;
        }
    }
    for (int l = 0; l < c->layers; ++l) {
        const ot_layer *W = &m->L[l];
        rms_norm(m->x, W->input_norm, m->xn, c->hidden, (double)c->eps);
        w4_matvec(&W->q, m->xn, m->q);
        w4_matvec(&W->k, m->xn, m->k);
        w4_matvec(&W->v, m->xn, m->v);
        for (int hq = 0; hq < Hq; ++hq) {
            rms_norm(m->q + hq * D, W->q_norm, m->q + hq * D, D, (double)c->eps);
            rope_inplace(m->q + hq * D, D, pos, (double)c->rope_theta);
        }
        for (int hk = 0; hk < Hkv; ++hk) {
            rms_norm(m->k + hk * D, W->k_norm, m->k + hk * D, D, (double)c->eps);
            rope_inplace(m->k + hk * D, D, pos, (double)c->rope_theta);
        }
        double *kc = m->kc + ((size_t)l * c->max_ctx) * kvw;
        double *vc = m->vc + ((size_t)l * c->max_ctx) * kvw;
        memcpy(kc + (size_t)pos * kvw, m->k, sizeof(double) * kvw);
        memcpy(vc + (size_t)pos * kvw, m->v, sizeof(double) * kvw);
        const int S = pos + 1;
#pragma omp parallel for schedule(static)
        for (int hq = 0; hq < Hq; ++hq) {
            const int hk = hq / rep;
            const double *qv = m->q + hq * D;
            double *sc = m->scores + (size_t)hq * c->max_ctx;
            double mx = -INFINITY;
            for (int t = 0; t < S; ++t) {
                const double *kr = kc + (size_t)t * kvw + hk * D;
                double d = 0.0;
                for (int i = 0; i < D; ++i) d += qv[i] * kr[i];
                sc[t] = d * scale;
                if (sc[t] > mx) mx = sc[t];
            }
            double den = 0.0;
            for (int t = 0; t < S; ++t) {
                sc[t] = exp(sc[t] - mx);
                den += sc[t];
            }
            double *o = m->att + hq * D;
            for (int i = 0; i < D; ++i) o[i] = 0.0;
            for (int t = 0; t < S; ++t) {
                const double *vr = vc + (size_t)t * kvw + hk * D;
                const double p = sc[t] / den;
                for (int i = 0; i < D; ++i) o[i] += p * vr[i];
            }
        }
        w4_matvec(&W->o, m->att, m->tmp);
        for (int i = 0; i < c->hidden; ++i) m->h[i] = m->x[i] + m->tmp[i];
        rms_norm(m->h, W->post_norm, m->xn, c->hidden, (double)c->eps);
        w4_matvec(&W->gate, m->xn, m->gate);
        w4_matvec(&W->up, m->xn, m->up);
        for (int i = 0; i < c->inter; ++i) {
            const double g = m->gate[i];
            m->act[i] = (g / (1.0 + exp(-g))) * m->up[i];
        }
        w4_matvec(&W->down, m->act, m->tmp);
        for (int i = 0; i < c->hidden; ++i) m->x[i] = m->h[i] + m->tmp[i];
    }
    m->ctx = pos + 1;
    rms_norm(m->x, m->final_norm, m->xn, c->hidden, (double)c->eps);
    w4_matvec(&m->head, m->xn, logits_out);
    int best = 0;
    for (int i = 1; i < c->vocab; ++i)
        if (logits_out[i] > logits_out[best]) best = i;
    return best;
}
