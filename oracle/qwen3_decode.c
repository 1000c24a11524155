/*
 * qwen3_decode.c — plain-C CPU restatement of ONE KV-cached Qwen3 W4A16 decode step.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/): used by tests/ as a second, independent checker of the numpy oracle
 * (oracle/tiny_oracle.py) and by bench.py's `cpu_baseline` leg ("port" kind: the reference's own CPU path
 * needs MLX, which is not installable here, and its custom primitives throw on CPU anyway,
 * src/extensions_ref/src/quantized_matmul.cpp:103-109).  The product never links or calls this.
 *
 * Follows, op by op (paths relative to /root/reference):
 *   embedding dequant      src/tiny_llm_ref/embedding.py:38-54, quantize.py:103-121 (nibble i of word j = element 8j+i)
 *   RMSNorm (fast order)   src/extensions_ref/src/week2_kernels.metal:41-47  T(x * rsqrt(mean(x^2)+eps) * w)
 *   W4A16 matvec           src/extensions_ref/src/quantized_matmul.metal:510-521  sum_g (s_g * sum a*q + beta_g * sum a), fp32
 *   RoPE (non-traditional) src/extensions_ref/src/week2_kernels.metal:86-104, positional_encoding.py:4-66
 *   causal GQA attention   src/tiny_llm_ref/attention.py:30-66 (fp32 softmax over the cached context), output T
 *   SwiGLU                 src/extensions_ref/src/week2_kernels.metal:115-116
 *   layer wiring           src/tiny_llm_ref/qwen3_week2.py:96-146,236-247,357-392
 * with activations rounded to bfloat16 after every reference op (values held in float).
 *
 * PARITY: pinned only against oracle/tiny_oracle.py (tests/test_oracle_c.py) — see that file's header for what
 * pins the numpy oracle itself.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint32_t *w; /* [rows, cols/8] */
    const uint16_t *s; /* [rows, cols/128] bf16 bits */
    const uint16_t *b; /* [rows, cols/128] bf16 bits */
    int rows, cols;
} oq_w4;

typedef struct {
    oq_w4 q, k, v, o, gate, up, down;
    const uint16_t *input_norm, *post_norm, *q_norm, *k_norm; /* bf16 bits */
} oq_layer;

typedef struct {
    int hidden, layers, heads, kv_heads, head_dim, inter, vocab, max_ctx;
    float rope_theta, eps;
} oq_config;

typedef struct {
    oq_config c;
    oq_layer *L;
    oq_w4 embed, head;
    const uint16_t *final_norm;
    float *kc, *vc; /* [layers][max_ctx][kv_heads*head_dim] (bf16-rounded values) */
    int ctx;
    float *x, *h, *xn, *q, *k, *v, *att, *gate, *up, *act, *tmp, *scores;
} oq_model;

static inline float bf16_to_f(uint16_t v) {
    uint32_t u = ((uint32_t)v) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
/* round-to-nearest-even, like static_cast<bfloat16_t>(float) */
static inline float bf16_round(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return f; /* NaN */
    u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
    memcpy(&f, &u, 4);
    return f;
}

/* out[r] = bf16( sum_g ( s*sum(a*q) + beta*sum(a) ) )  for one activation row a[cols] */
static void w4_matvec(const oq_w4 *W, const float *a, float *out) {
    const int groups = W->cols / 128, words = W->cols / 8;
    float *asum = (float *)malloc(sizeof(float) * (size_t)groups);
    for (int g = 0; g < groups; ++g) {
        float t = 0.f;
        for (int i = 0; i < 128; ++i) t += a[g * 128 + i];
        asum[g] = t;
    }
#pragma omp parallel for schedule(static)
    for (int r = 0; r < W->rows; ++r) {
        const uint32_t *wr = W->w + (size_t)r * words;
        float acc = 0.f;
        for (int g = 0; g < groups; ++g) {
            float lane[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const float *ag = a + g * 128;
            for (int j = 0; j < 16; ++j) {
                const uint32_t p = wr[g * 16 + j];
                for (int e = 0; e < 8; ++e) lane[e] += ag[j * 8 + e] * (float)((p >> (4 * e)) & 0xfu);
            }
            const float qdot = ((lane[0] + lane[1]) + (lane[2] + lane[3])) + ((lane[4] + lane[5]) + (lane[6] + lane[7]));
            acc += bf16_to_f(W->s[(size_t)r * groups + g]) * qdot + bf16_to_f(W->b[(size_t)r * groups + g]) * asum[g];
        }
        out[r] = bf16_round(acc);
    }
    free(asum);
}

static void rms_norm(const float *x, const uint16_t *w, float *out, int n, float eps) {
    float ss = 0.f;
    for (int i = 0; i < n; ++i) ss += x[i] * x[i];
    const float inv = 1.0f / sqrtf(ss / (float)n + eps);
    for (int i = 0; i < n; ++i) out[i] = bf16_round(x[i] * inv * bf16_to_f(w[i]));
}

static void rope_inplace(float *x, int D, int pos, float base) {
    const int half = D / 2;
    for (int d = 0; d < half; ++d) {
        const float angle = (float)pos * powf(base, -(float)d / (float)half);
        const float c = cosf(angle), s = sinf(angle);
        const float re = x[d], im = x[d + half];
        x[d] = bf16_round(re * c - im * s);
        x[d + half] = bf16_round(im * c + re * s);
    }
}

oq_model *oq_create(const oq_config *c, const oq_layer *layers, const oq_w4 *embed, const oq_w4 *head,
                    const uint16_t *final_norm) {
    oq_model *m = (oq_model *)calloc(1, sizeof(oq_model));
    m->c = *c;
    m->L = (oq_layer *)malloc(sizeof(oq_layer) * (size_t)c->layers);
    memcpy(m->L, layers, sizeof(oq_layer) * (size_t)c->layers);
    m->embed = *embed;
    m->head = head ? *head : *embed;
    m->final_norm = final_norm;
    const size_t kvw = (size_t)c->kv_heads * c->head_dim;
    m->kc = (float *)calloc((size_t)c->layers * c->max_ctx * kvw, sizeof(float));
    m->vc = (float *)calloc((size_t)c->layers * c->max_ctx * kvw, sizeof(float));
    const int qd = c->heads * c->head_dim;
    m->x = (float *)malloc(sizeof(float) * c->hidden);
    m->h = (float *)malloc(sizeof(float) * c->hidden);
    m->xn = (float *)malloc(sizeof(float) * c->hidden);
    m->tmp = (float *)malloc(sizeof(float) * c->hidden);
    m->q = (float *)malloc(sizeof(float) * qd);
    m->att = (float *)malloc(sizeof(float) * qd);
    m->k = (float *)malloc(sizeof(float) * kvw);
    m->v = (float *)malloc(sizeof(float) * kvw);
    m->gate = (float *)malloc(sizeof(float) * c->inter);
    m->up = (float *)malloc(sizeof(float) * c->inter);
    m->act = (float *)malloc(sizeof(float) * c->inter);
    m->scores = (float *)malloc(sizeof(float) * (size_t)c->heads * c->max_ctx);
    return m;
}

void oq_destroy(oq_model *m) {
    if (!m) return;
    free(m->L); free(m->kc); free(m->vc); free(m->x); free(m->h); free(m->xn); free(m->tmp); free(m->q);
    free(m->att); free(m->k); free(m->v); free(m->gate); free(m->up); free(m->act); free(m->scores);
    free(m);
}

void oq_reset(oq_model *m) { m->ctx = 0; }
int oq_context(const oq_model *m) { return m->ctx; }

/* Feed `token` at position ctx; logits_out[vocab] receives bf16-rounded logits. Returns the argmax id, or -1
 * when the cache is full. */
int oq_decode_step(oq_model *m, int token, float *logits_out) {
    const oq_config *c = &m->c;
    if (m->ctx >= c->max_ctx || token < 0 || token >= c->vocab) return -1;
    const int D = c->head_dim, Hq = c->heads, Hkv = c->kv_heads, rep = Hq / Hkv, pos = m->ctx;
    const size_t kvw = (size_t)Hkv * D;
    const float scale = 1.0f / sqrtf((float)D);
    { /* embedding row */
        const int groups = c->hidden / 128, words = c->hidden / 8;
        for (int j = 0; j < words; ++j) {
            const uint32_t p = m->embed.w[(size_t)token * words + j];
            const float s = bf16_to_f(m->embed.s[(size_t)token * groups + j / 16]);
            const float b = bf16_to_f(m->embed.b[(size_t)token * groups + j / 16]);
            for (int e = 0; e < 8; ++e) m->x[j * 8 + e] = bf16_round((float)((p >> (4 * e)) & 0xfu) * s + b);
        }
    }
    for (int l = 0; l < c->layers; ++l) {
        const oq_layer *W = &m->L[l];
        rms_norm(m->x, W->input_norm, m->xn, c->hidden, c->eps);
        w4_matvec(&W->q, m->xn, m->q);
        w4_matvec(&W->k, m->xn, m->k);
        w4_matvec(&W->v, m->xn, m->v);
        for (int hq = 0; hq < Hq; ++hq) {
            rms_norm(m->q + hq * D, W->q_norm, m->q + hq * D, D, c->eps);
            rope_inplace(m->q + hq * D, D, pos, c->rope_theta);
        }
        for (int hk = 0; hk < Hkv; ++hk) {
            rms_norm(m->k + hk * D, W->k_norm, m->k + hk * D, D, c->eps);
            rope_inplace(m->k + hk * D, D, pos, c->rope_theta);
        }
        float *kc = m->kc + ((size_t)l * c->max_ctx) * kvw;
        float *vc = m->vc + ((size_t)l * c->max_ctx) * kvw;
        memcpy(kc + (size_t)pos * kvw, m->k, sizeof(float) * kvw);
        memcpy(vc + (size_t)pos * kvw, m->v, sizeof(float) * kvw);
        const int S = pos + 1;
#pragma omp parallel for schedule(static)
        for (int hq = 0; hq < Hq; ++hq) {
            const int hk = hq / rep;
            const float *qv = m->q + hq * D;
            float *sc = m->scores + (size_t)hq * c->max_ctx;
            float mx = -INFINITY;
            for (int t = 0; t < S; ++t) {
                const float *kr = kc + (size_t)t * kvw + hk * D;
                float d = 0.f;
                for (int i = 0; i < D; ++i) d += qv[i] * kr[i];
                sc[t] = d * scale;
                if (sc[t] > mx) mx = sc[t];
            }
            float den = 0.f;
            for (int t = 0; t < S; ++t) {
                sc[t] = expf(sc[t] - mx);
                den += sc[t];
            }
            float *o = m->att + hq * D;
            for (int i = 0; i < D; ++i) o[i] = 0.f;
            for (int t = 0; t < S; ++t) {
                const float *vr = vc + (size_t)t * kvw + hk * D;
                const float p = sc[t] / den;
                for (int i = 0; i < D; ++i) o[i] += p * vr[i];
            }
            for (int i = 0; i < D; ++i) o[i] = bf16_round(o[i]);
        }
        w4_matvec(&W->o, m->att, m->tmp);
        for (int i = 0; i < c->hidden; ++i) m->h[i] = bf16_round(m->x[i] + m->tmp[i]);
        rms_norm(m->h, W->post_norm, m->xn, c->hidden, c->eps);
        w4_matvec(&W->gate, m->xn, m->gate);
        w4_matvec(&W->up, m->xn, m->up);
        for (int i = 0; i < c->inter; ++i) {
            const float g = m->gate[i];
            m->act[i] = bf16_round((g / (1.0f + expf(-g))) * m->up[i]);
        }
        w4_matvec(&W->down, m->act, m->tmp);
        for (int i = 0; i < c->hidden; ++i) m->x[i] = bf16_round(m->h[i] + m->tmp[i]);
    }
    m->ctx = pos + 1;
    rms_norm(m->x, m->final_norm, m->xn, c->hidden, c->eps);
    w4_matvec(&m->head, m->xn, logits_out);
    int best = 0;
    for (int i = 1; i < c->vocab; ++i)
        if (logits_out[i] > logits_out[best]) best = i;
    return best;
}

/* Bytes of weights one decode step streams (same formula as SURVEY.md §8d: 0.53125 B per weight). */
double oq_weight_bytes(const oq_model *m) {
    double t = 0;
    for (int l = 0; l < m->c.layers; ++l) {
        const oq_w4 *ws[7] = {&m->L[l].q, &m->L[l].k, &m->L[l].v, &m->L[l].o, &m->L[l].gate, &m->L[l].up, &m->L[l].down};
        for (int i = 0; i < 7; ++i) t += (double)ws[i]->rows * ws[i]->cols * 0.53125;
    }
    return t + (double)m->head.rows * m->head.cols * 0.53125;
}
