// Measurement driver (not part of the product): a Qwen3-4B-shaped engine built through the C ABI alone (include/tinyllm_engine.h),
// random W4 weights filled on the device, one prompt prefilled, then decode steps -- no Python, no torch in the process, so that
// `rocprofv3 --pmc` (which crashes under the full Python bench on this image) can count HBM bytes on the ENGINE's own decode step
// instead of on tools/lab/gemv_lab's replay of its GEMV kernels.  A pure stream kernel of known size runs first: its FETCH_SIZE
// calibrates the counter's unit (guides/MI355X_MICROARCH.md, HBM section).
//   usage: engine_step_lab [context tokens = 128] [decode steps = 20] [batch = 1] [kv pages: 0 = bf16, 1 = FP8 E4M3]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/tinyllm_engine.h"
#include "../../include/tinyllm_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define TL(x) do { int rc_ = (x); if (rc_ != 0) { printf("engine error %d at line %d: %s\n", rc_, __LINE__, tl_last_error()); exit(1); } } while (0)

// words: random nibbles; bf16 tables: base + a few random mantissa bits (scales ~0.008-0.012, biases ~ -0.06 .. -0.09, norms ~1)
__global__ void fill_words_kernel(uint32_t *p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = x;
    }
}
__global__ void fill_bf16_kernel(uint16_t *p, size_t n, uint16_t base, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13;
        p[i] = (uint16_t)(base + (x & 0x3f));
    }
}
// the calibration stream: every thread reads 16 bytes per iteration, non-temporal, and keeps an XOR so that nothing is dropped
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
__global__ void stream_read_kernel(const u32x4_t *p, size_t n16, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4_t v = __builtin_nontemporal_load(p + i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

static std::vector<void *> g_allocs;
static uint32_t g_seed = 1;
static tl_w4 make_w4(int rows, int cols) {
    tl_w4 w{};
    uint32_t *wd; uint16_t *s, *b;
    const size_t words = (size_t)rows * cols / 8, groups = (size_t)rows * (cols / 128);
    CK(hipMalloc(&wd, words * 4)); CK(hipMalloc(&s, groups * 2)); CK(hipMalloc(&b, groups * 2));
    fill_words_kernel<<<2048, 256>>>(wd, words, g_seed++ * 7919u);
    fill_bf16_kernel<<<512, 256>>>(s, groups, 0x3c00, g_seed++ * 104729u);   // bf16 0.0078 ..
    fill_bf16_kernel<<<512, 256>>>(b, groups, 0xbd80, g_seed++ * 1299709u);  // bf16 -0.0625 ..
    g_allocs.push_back(wd); g_allocs.push_back(s); g_allocs.push_back(b);
    w.weight_dev = wd; w.scales_dev = s; w.biases_dev = b; w.rows = rows; w.cols = cols;
    return w;
}
static const void *make_norm(int n) {
    uint16_t *p; CK(hipMalloc(&p, (size_t)n * 2));
    fill_bf16_kernel<<<8, 256>>>(p, n, 0x3f80, g_seed++ * 15485863u);  // bf16 1.0 ..
    g_allocs.push_back(p);
    return p;
}

int main(int argc, char **argv) {
    const int context = argc > 1 ? atoi(argv[1]) : 128;
    const int steps = argc > 2 ? atoi(argv[2]) : 20;
    const int batch = argc > 3 ? atoi(argv[3]) : 1;
    const int kv_format = argc > 4 && atoi(argv[4]) == 1 ? TL_KV_FP8_E4M3 : TL_KV_BF16;
    const int H = 2560, L = 36, HQ = 32, HKV = 8, D = 128, I = 9728, V = 151936, page = 128;

    // calibration stream first: 1 GiB read 4 times (far beyond the 256 MiB infinity cache)
    {
        const size_t bytes = (size_t)1 << 30;
        u32x4_t *buf; uint32_t *sink;
        CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4));
        fill_words_kernel<<<2048, 256>>>((uint32_t *)buf, bytes / 4, 99u);
        CK(hipDeviceSynchronize());
        for (int i = 0; i < 4; ++i) stream_read_kernel<<<4096, 256>>>(buf, bytes / 16, sink);
        CK(hipDeviceSynchronize());
        printf("calibration: stream_read_kernel x 4, %zu bytes each\n", bytes);
        CK(hipFree(buf)); CK(hipFree(sink));
    }

    std::vector<tl_layer_weights> layers(L);
    for (int l = 0; l < L; ++l) {
        layers[l].wqkv = make_w4((HQ + 2 * HKV) * D, H);
        layers[l].wo = make_w4(H, HQ * D);
        layers[l].wgu = make_w4(2 * I, H);
        layers[l].wdown = make_w4(H, I);
        layers[l].input_norm_dev = make_norm(H);
        layers[l].post_norm_dev = make_norm(H);
        layers[l].q_norm_dev = make_norm(D);
        layers[l].k_norm_dev = make_norm(D);
    }
    const tl_w4 embed = make_w4(V, H);
    const void *final_norm = make_norm(H);
    CK(hipDeviceSynchronize());

    const int total = context + 16 + steps + 64;
    const int per_seq = (total + page - 1) / page + 1;
    tl_engine_config cfg{};
    cfg.hidden_size = H; cfg.num_layers = L; cfg.num_heads = HQ; cfg.num_kv_heads = HKV; cfg.head_dim = D; cfg.intermediate_size = I; cfg.vocab_size = V;
    cfg.rope_theta = 1000000.f; cfg.rms_norm_eps = 1e-6f;
    cfg.page_size = page; cfg.num_pages = per_seq * batch + 2; cfg.max_batch = batch; cfg.max_pages_per_seq = per_seq;
    cfg.max_prefill_rows = context < 2048 ? (context < 8 ? 8 : context) : 2048;
    tl_engine *e = nullptr;
    TL(tl_engine_create_kv(&cfg, layers.data(), &embed, final_norm, nullptr, nullptr, kv_format, &e));

    std::vector<int32_t> prompt(context);
    for (int b = 0; b < batch; ++b) {
        for (int i = 0; i < context; ++i) prompt[i] = 256 + (int)(((unsigned)i * 2654435761u + (unsigned)b * 40503u) % (unsigned)(V - 256));
        TL(tl_engine_begin(e, b));
        for (int at = 0; at < context; at += cfg.max_prefill_rows) {
            const int n = context - at < cfg.max_prefill_rows ? context - at : cfg.max_prefill_rows;
            TL(tl_engine_prefill(e, b, prompt.data() + at, n, at + n == context));
        }
    }
    TL(tl_engine_decode(e, batch, 16, 1));  // capture + warm steps
    TL(tl_engine_synchronize(e));
    const auto t0 = std::chrono::steady_clock::now();  // (the engine owns its stream: the host clock around enqueue + synchronise)
    TL(tl_engine_decode(e, batch, steps, 1));
    TL(tl_engine_synchronize(e));
    const float ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    tl_engine_stats st{}; TL(tl_engine_get_stats(e, &st));
    printf("engine: qwen3-4b shape, %s pages, context %d, batch %d, %d timed steps, route %s, %.4f ms per step (host-bracketed), algorithmic bytes per step %zu\n",
           kv_format == TL_KV_FP8_E4M3 ? "FP8 E4M3" : "bf16", context, batch, steps, tl_engine_replay_route(e), ms / steps, tl_engine_step_bytes(e, batch));
    std::vector<int32_t> ids(steps);
    TL(tl_engine_read_tokens(e, 0, steps < 8 ? steps : 8, ids.data()));
    printf("first ids of slot 0:");
    for (int i = 0; i < (steps < 8 ? steps : 8); ++i) printf(" %d", ids[i]);
    printf("\n");
    tl_engine_destroy(e);
    for (void *p : g_allocs) CK(hipFree(p));
    return 0;
}
