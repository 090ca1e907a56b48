// Standalone check of one fused backward step (cor_asv_ann_amd/csrc/gemm_bwd.hip: cell backward + data GEMM in one launch)
// against the same step computed on the host in double precision.  Built and run by tests/test_gpu_train.py:
//   hipcc -O3 --offload-arch=gfx950 -I include tests/native/bwd_step_check.hip cor_asv_ann_amd/csrc/gemm_bwd.hip -o check
//   ./check ROWS WIDTH N [time]      prints the largest absolute errors of dZ, dL/dc and of the GEMM's output
#include "../../cor_asv_ann_amd/csrc/common.h"
#include "../../cor_asv_ann_amd/csrc/train_kernels.h"
#include <vector>
#include <random>
#include <cstdio>
#include <cmath>
using namespace casv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 4, W = argc > 2 ? atoi(argv[2]) : 32, N = argc > 3 ? atoi(argv[3]) : 32;
    const int K = 4 * W;
    std::mt19937 rng(1);
    std::uniform_real_distribution<float> U(-1.f, 1.f), P(0.05f, 0.95f);
    std::vector<float> a(B * W), mk(W), b(B * W), gates(B * K), cell(B * W), cp(B * W), dc(B * W), Bt((size_t)N * K);
    for (auto& x : a) x = U(rng); for (auto& x : mk) x = P(rng) * 1.25f; for (auto& x : b) x = U(rng);
    for (auto& x : gates) x = P(rng); for (auto& x : cell) x = U(rng) * 2; for (auto& x : cp) x = U(rng) * 2; for (auto& x : dc) x = U(rng);
    for (auto& x : Bt) x = U(rng) * 0.1f;
    // gates g column is tanh-range
    for (int r = 0; r < B; ++r) for (int u = 0; u < W; ++u) gates[(size_t)r * K + (u >> 5) * 128 + 64 + (u & 31)] = U(rng);
    std::vector<double> dz((size_t)B * K), dcn(B * W), out((size_t)B * N, 0.0);
    for (int r = 0; r < B; ++r) for (int u = 0; u < W; ++u) {
        const size_t gi = (size_t)r * K + (u >> 5) * 128 + (u & 31);
        double dh = (double)a[r * W + u] * mk[u] + b[r * W + u];
        double ig = gates[gi], fg = gates[gi + 32], gg = gates[gi + 64], og = gates[gi + 96];
        double tc = tanh((double)cell[r * W + u]);
        double dov = dh * tc, dct = dh * og * (1 - tc * tc) + dc[r * W + u];
        dz[gi] = dct * gg * ig * (1 - ig); dz[gi + 32] = dct * cp[r * W + u] * fg * (1 - fg); dz[gi + 64] = dct * ig * (1 - gg * gg); dz[gi + 96] = dov * og * (1 - og);
        dcn[r * W + u] = dct * fg;
    }
    for (int r = 0; r < B; ++r) for (int n = 0; n < N; ++n) { double s = 0; for (int k = 0; k < K; ++k) s += dz[(size_t)r * K + k] * Bt[(size_t)n * K + k]; out[(size_t)r * N + n] = s; }
    float *d_a, *d_mk, *d_b, *d_g, *d_cell, *d_cp, *d_dc, *d_dc2, *d_dz, *d_Bt, *d_out;
#define UP(D, H) CK(hipMalloc(&D, H.size() * 4)); CK(hipMemcpy(D, H.data(), H.size() * 4, hipMemcpyHostToDevice));
    UP(d_a, a) UP(d_mk, mk) UP(d_b, b) UP(d_g, gates) UP(d_cell, cell) UP(d_cp, cp) UP(d_dc, dc) UP(d_Bt, Bt)
    CK(hipMalloc(&d_dc2, B * W * 4)); CK(hipMalloc(&d_dz, (size_t)B * K * 4)); CK(hipMalloc(&d_out, (size_t)B * N * 4));
    CK(hipMemset(d_dc2, 0xff, B * W * 4)); CK(hipMemset(d_dz, 0xff, (size_t)B * K * 4)); CK(hipMemset(d_out, 0, (size_t)B * N * 4));
    BwdStepBatch fb{};
    fb.count = 1;
    BwdStepJob& q = fb.j[0];
    q.p.a = d_a; q.p.lda = W; q.p.mask_a = d_mk; q.p.b = d_b; q.p.ldb = W; q.p.gates = d_g; q.p.cell = d_cell; q.p.c_prev = d_cp; q.p.ld_cprev = W;
    q.p.dc = d_dc2; q.p.dz = d_dz; q.p.rows = B; q.p.W = W; q.dc_in = d_dc; q.Bt = d_Bt; q.out = d_out; q.ld_out = N; q.N = N;
    launch_lstm_bwd_gemm(fb, nullptr);
    CK(hipGetLastError());
    CK(hipDeviceSynchronize());
    std::vector<float> h_dz((size_t)B * K), h_dc(B * W), h_out((size_t)B * N);
    CK(hipMemcpy(h_dz.data(), d_dz, h_dz.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_dc.data(), d_dc2, h_dc.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h_out.data(), d_out, h_out.size() * 4, hipMemcpyDeviceToHost));
    double e1 = 0, e2 = 0, e3 = 0, s3 = 0;
    for (size_t i = 0; i < h_dz.size(); ++i) e1 = std::max(e1, std::fabs(h_dz[i] - dz[i]));
    for (size_t i = 0; i < h_dc.size(); ++i) e2 = std::max(e2, std::fabs(h_dc[i] - dcn[i]));
    for (size_t i = 0; i < h_out.size(); ++i) { e3 = std::max(e3, std::fabs(h_out[i] - out[i])); s3 = std::max(s3, std::fabs(out[i])); }
    if (argc > 4) {       // timing: two jobs per launch as in the train step
        fb.count = 2; fb.j[1] = fb.j[0];
        float *o2, *z2, *c2; CK(hipMalloc(&o2, (size_t)B * N * 4)); CK(hipMalloc(&z2, (size_t)B * K * 4)); CK(hipMalloc(&c2, B * W * 4));
        fb.j[1].out = o2; fb.j[1].p.dz = z2; fb.j[1].p.dc = c2;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int i = 0; i < 20; ++i) launch_lstm_bwd_gemm(fb, nullptr);
        CK(hipEventRecord(e0, nullptr));
        const int reps = 300;
        for (int i = 0; i < reps; ++i) launch_lstm_bwd_gemm(fb, nullptr);
        CK(hipEventRecord(e1, nullptr)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %.2f us per launch (back to back, incl. gaps)\n", 1e3 * ms / reps);
    }
    printf("B %d W %d N %d: dz err %.3g  dc err %.3g  out err %.3g (scale %.3g)\n", B, W, N, e1, e2, e3, s3);
    return 0;
}
