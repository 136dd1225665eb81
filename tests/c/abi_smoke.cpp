// Pure C/HIP consumer of the C ABI (include/gcfr.h): no Python, no torch.  Allocates with hipMalloc, runs the
// one-call forward and the three-call forward, checks them against each other (bit-equal) and against the C
// oracle (oracle/gcfr_oracle.c, linked as a shared library) within the north_star gates.
// Build + run: see tests/test_gpu_c_abi.py.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/gcfr.h"

extern "C" {
void gcfr_oracle_sample_table(double t0, double dt, int n, double *out);
void gcfr_oracle_light_prep(const float *light_raw, int B, int clamp_z, float clamp_min, float light_distance,
                            float *unit_out, float *light_pt_out);
void gcfr_oracle_shadow_min_distance(const float *depth, const uint8_t *mask_u8, int mask_batch,
                                     const float *light_pt, int B, int L, int H, int W, int N,
                                     const double *t_table, float bonus, float bx_lo, float bx_hi, float by_lo,
                                     float by_hi, float *min_dist, int32_t *argmin);
void gcfr_oracle_shade(const double *normals, const float *depth, const float *albedo, const float *light_pt,
                       const float *ambient, const float *min_dist, int B, int L, int H, int W, float intensity,
                       float *shadow_w, double *full_shading, double *final_shading, float *rendered,
                       double *normals_out);
}

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                          \
        }                                                                      \
    } while (0)
#define GK(x)                                                                  \
    do {                                                                       \
        int s_ = (x);                                                          \
        if (s_ != GCFR_OK) {                                                   \
            std::fprintf(stderr, "gcfr status %d at %s:%d\n", s_, __FILE__, __LINE__); \
            return 3;                                                          \
        }                                                                      \
    } while (0)

template <class T>
static T *dev_copy(const std::vector<T> &h)
{
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess)
        return nullptr;
    (void)hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return d;
}
template <class T>
static T *dev_alloc(size_t n)
{
    T *d = nullptr;
    return hipMalloc(&d, n * sizeof(T)) == hipSuccess ? d : nullptr;
}
template <class T>
static std::vector<T> host_copy(const T *d, size_t n)
{
    std::vector<T> h(n);
    (void)hipMemcpy(h.data(), d, n * sizeof(T), hipMemcpyDeviceToHost);
    return h;
}

int main()
{
    const int B = 3, L = 2, H = 96, W = 128, N = 80;
    const size_t P = (size_t)H * W;
    std::vector<float> depth(B * P), albedo(B * 3 * P), normals(B * 3 * P), light(B * L * 3), ambient(B * L);
    std::vector<uint8_t> mask(B * P);
    uint32_t seed = 12345;
    auto rnd = [&]() {
        seed = seed * 1664525u + 1013904223u;
        return (seed >> 8) * (1.0f / 16777216.0f);
    };
    for (int b = 0; b < B; ++b)
        for (int r = 0; r < H; ++r)
            for (int c = 0; c < W; ++c) {
                const float dx = (c - 0.5f * W) / (0.3f * W), dy = (r - 0.5f * H) / (0.35f * H);
                depth[b * P + (size_t)r * W + c] = 25.0f * std::exp(-(dx * dx + dy * dy)) + rnd();
                mask[b * P + (size_t)r * W + c] = (dx * dx + dy * dy < 1.2f + 0.3f * b) ? 1 : 0;
            }
    for (auto &v : albedo) v = 0.1f + 0.8f * rnd();
    for (auto &v : normals) v = rnd() - 0.5f;
    const float lights[6][3] = {{0.3f, 0.5f, 0.8f}, {-0.9f, 0.1f, 0.2f}, {0.01f, -0.02f, 1.0f},
                                {0.7f, -0.7f, 0.05f}, {0.0f, 0.9f, 0.3f}, {0.5f, -0.8f, -0.3f}};
    for (int i = 0; i < B * L; ++i)
        for (int k = 0; k < 3; ++k)
            light[3 * i + k] = lights[i][k];
    for (auto &v : ambient) v = 0.3f + 0.4f * rnd();

    std::vector<double> tt(N);
    GK(gcfr_sample_table(0.025, 0.01, N, tt.data()));

    float *d_depth = dev_copy(depth), *d_albedo = dev_copy(albedo), *d_normals = dev_copy(normals);
    float *d_light = dev_copy(light), *d_amb = dev_copy(ambient);
    uint8_t *d_mask = dev_copy(mask);
    double *d_tt = dev_copy(tt);
    const size_t BL = (size_t)B * L;
    float *unit = dev_alloc<float>(BL * 3), *pt = dev_alloc<float>(BL * 3);
    float *md = dev_alloc<float>(BL * P), *w = dev_alloc<float>(BL * P), *full = dev_alloc<float>(BL * P);
    float *fin = dev_alloc<float>(BL * P), *ren = dev_alloc<float>(BL * 3 * P);
    int32_t *am = dev_alloc<int32_t>(BL * P);
    const size_t ws_bytes = gcfr_shadow_workspace_bytes(B, H, W);
    void *ws = nullptr;
    CK(hipMalloc(&ws, ws_bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));

    // (0) the library implements the ABI revision this consumer was compiled against (a stale .so keeps every symbol NAME)
    if (gcfr_abi_version() != GCFR_ABI_VERSION) {
        std::fprintf(stderr, "ABI revision %d, header %d\n", (int)gcfr_abi_version(), (int)GCFR_ABI_VERSION);
        return 1;
    }
    // (1) one call
    // per-call options: explicit knobs (LDS staging of the mask / bounds records; the NULL-options calls
    // below use the defaults) and an event pair the library records around the march kernel
    gcfr_options opt;
    gcfr_options_default(&opt);
    hipEvent_t ev0, ev1;
    CK(hipEventCreate(&ev0));
    CK(hipEventCreate(&ev1));
    opt.lds_stage = 1;   // (bit-identical by contract: the three-call forward below runs with the defaults and must agree)
    opt.event_start = ev0;
    opt.event_stop = ev1;
    GK(gcfr_render_fwd(d_light, 1, 0.0f, 4013.0f, d_depth, d_mask, B, d_normals, d_albedo, d_amb, B, L, H, W, N,
                       d_tt, 0.0f, nullptr, 0.5f, unit, pt, md, am, w, full, fin, ren, ws, ws_bytes, st, &opt));
    CK(hipStreamSynchronize(st));
    float march_ms = -1.0f;
    CK(hipEventElapsedTime(&march_ms, ev0, ev1));
    if (!(march_ms > 0.0f)) {
        std::fprintf(stderr, "event hook: march kernel time %g ms\n", march_ms);
        return 1;
    }
    auto h_md = host_copy(md, BL * P), h_w = host_copy(w, BL * P), h_ren = host_copy(ren, BL * 3 * P);
    auto h_am = host_copy(am, BL * P);
    auto h_pt = host_copy(pt, BL * 3);

    // (2) three calls, no workspace (plain kernel): must give the same bits
    float *md2 = dev_alloc<float>(BL * P), *ren2 = dev_alloc<float>(BL * 3 * P), *w2 = dev_alloc<float>(BL * P);
    int32_t *am2 = dev_alloc<int32_t>(BL * P);
    GK(gcfr_light_prep(d_light, (int)BL, 1, 0.0f, 4013.0f, unit, pt, st));
    GK(gcfr_shadow_fwd(d_depth, d_mask, B, pt, B, L, H, W, N, d_tt, 0.0f, nullptr, md2, am2, nullptr, 0, st, nullptr));
    GK(gcfr_shade_fwd(d_normals, d_depth, d_albedo, pt, d_amb, md2, B, L, H, W, 0.5f, w2, nullptr, nullptr, ren2, st));
    CK(hipStreamSynchronize(st));
    auto h_md2 = host_copy(md2, BL * P), h_ren2 = host_copy(ren2, BL * 3 * P);
    auto h_am2 = host_copy(am2, BL * P);
    size_t diff = 0;
    for (size_t i = 0; i < BL * P; ++i)
        diff += (h_md[i] != h_md2[i]) + (h_am[i] != h_am2[i]);
    for (size_t i = 0; i < BL * 3 * P; ++i)
        diff += (h_ren[i] != h_ren2[i]);
    std::printf("one-call vs three-call (plain kernel): %zu differing values\n", diff);

    // (3) the C oracle
    std::vector<double> ott(N);
    gcfr_oracle_sample_table(0.025, 0.01, N, ott.data());
    std::vector<float> o_unit(BL * 3), o_pt(BL * 3), o_md(BL * P), o_w(BL * P), o_ren(BL * 3 * P);
    std::vector<int32_t> o_am(BL * P);
    std::vector<double> o_full(BL * P), o_fin(BL * P), n64(normals.begin(), normals.end()), o_n(B * 3 * P);
    gcfr_oracle_light_prep(light.data(), (int)BL, 1, 0.0f, 4013.0f, o_unit.data(), o_pt.data());
    gcfr_oracle_shadow_min_distance(depth.data(), mask.data(), B, o_pt.data(), B, L, H, W, N, ott.data(), 0.0f, 0.0f,
                                    -1.0f, 0.0f, -1.0f, o_md.data(), o_am.data());
    gcfr_oracle_shade(n64.data(), depth.data(), albedo.data(), o_pt.data(), ambient.data(), o_md.data(), B, L, H, W,
                      0.5f, o_w.data(), o_full.data(), o_fin.data(), o_ren.data(), o_n.data());
    double e_w = 0, e_rgb = 0, e_pt = 0;
    for (size_t i = 0; i < BL * P; ++i)
        e_w = std::fmax(e_w, std::fabs((double)h_w[i] - o_w[i]));
    for (size_t i = 0; i < BL * 3 * P; ++i)
        e_rgb = std::fmax(e_rgb, std::fabs((double)h_ren[i] - o_ren[i]));
    for (size_t i = 0; i < BL * 3; ++i)
        e_pt = std::fmax(e_pt, std::fabs((double)h_pt[i] - o_pt[i]));
    std::printf("vs C oracle: max|dw| = %.3g  max|dRGB| = %.3g  max|dLightPt| = %.3g\n", e_w, e_rgb, e_pt);
    std::printf("%s\n", gcfr_version());
    const bool ok = diff == 0 && e_w <= 1e-4 && e_rgb <= 1e-3 && e_pt == 0.0;
    std::printf(ok ? "C-ABI SMOKE OK\n" : "C-ABI SMOKE FAILED\n");
    return ok ? 0 : 1;
}
