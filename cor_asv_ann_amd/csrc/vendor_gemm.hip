// CALIBRATION path (off by default, option "vendor_gemm" = 1): the train step's PLAIN whole-sequence contractions
// C[M][N] (+)= A[M][K] . Bt[N][K]^T (+ bias) -- the input projections of all time steps and their data gradients, nothing fused,
// ~2.2 TFLOP per train step -- through hipBLASLt, so that this library's own kernel can be timed against the vendor's on the same
// step.  hipBLASLt is loaded at run time (dlopen: no link-time dependency; its header is needed at build time only -- without it
// this file compiles to a stub that takes no contraction); a shape the heuristic has no solution for, a missing library or
// CASV_VENDOR_GEMM=0 fall back to gemm.hip.  One workspace per (device, stream): two models training on one GPU never share one.
// Host code only.
#include "common.h"
#if __has_include(<hipblaslt/hipblaslt.h>)
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>
#include <map>
#include <mutex>
#include <tuple>
#include <cstdlib>

namespace casv {

namespace {
struct Api {
    void* lib = nullptr;
    hipblasStatus_t (*Create)(hipblasLtHandle_t*) = nullptr;
    hipblasStatus_t (*MatmulDescCreate)(hipblasLtMatmulDesc_t*, hipblasComputeType_t, hipDataType) = nullptr;
    hipblasStatus_t (*MatmulDescSetAttribute)(hipblasLtMatmulDesc_t, hipblasLtMatmulDescAttributes_t, const void*, size_t) = nullptr;
    hipblasStatus_t (*MatrixLayoutCreate)(hipblasLtMatrixLayout_t*, hipDataType, uint64_t, uint64_t, int64_t) = nullptr;
    hipblasStatus_t (*PreferenceCreate)(hipblasLtMatmulPreference_t*) = nullptr;
    hipblasStatus_t (*PreferenceSetAttribute)(hipblasLtMatmulPreference_t, hipblasLtMatmulPreferenceAttributes_t, const void*, size_t) = nullptr;
    hipblasStatus_t (*AlgoGetHeuristic)(hipblasLtHandle_t, hipblasLtMatmulDesc_t, hipblasLtMatrixLayout_t, hipblasLtMatrixLayout_t,
                                        hipblasLtMatrixLayout_t, hipblasLtMatrixLayout_t, hipblasLtMatmulPreference_t, int,
                                        hipblasLtMatmulHeuristicResult_t*, int*) = nullptr;
    hipblasStatus_t (*Matmul)(hipblasLtHandle_t, hipblasLtMatmulDesc_t, const void*, const void*, hipblasLtMatrixLayout_t, const void*,
                              hipblasLtMatrixLayout_t, const void*, const void*, hipblasLtMatrixLayout_t, void*, hipblasLtMatrixLayout_t,
                              const hipblasLtMatmulAlgo_t*, void*, size_t, hipStream_t) = nullptr;
    hipblasLtHandle_t handle = nullptr;
    size_t workspace_bytes = 0;
    std::map<hipStream_t, void*> workspace;               // per stream: solutions with a split K keep partial sums there
    bool ok = false;
};

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo{};
    bool usable = false;
};

std::mutex g_mu;
std::map<int, Api> g_api;                                 // per device
using Key = std::tuple<int, long long, int, int, long long, long long, long long, int, int>;
std::map<Key, Plan> g_plans;

Api& api_for(int dev) {
    Api& a = g_api[dev];
    if (a.lib || a.ok) return a;
    const char* env = getenv("CASV_VENDOR_GEMM");
    if (env && env[0] == '0') { a.lib = (void*)1; return a; }
    a.lib = dlopen("libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) a.lib = dlopen("/opt/rocm/lib/libhipblaslt.so", RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) { a.lib = (void*)1; return a; }
#define CASV_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.lib, name)); if (!a.field) return a;
    CASV_SYM(Create, "hipblasLtCreate")
    CASV_SYM(MatmulDescCreate, "hipblasLtMatmulDescCreate")
    CASV_SYM(MatmulDescSetAttribute, "hipblasLtMatmulDescSetAttribute")
    CASV_SYM(MatrixLayoutCreate, "hipblasLtMatrixLayoutCreate")
    CASV_SYM(PreferenceCreate, "hipblasLtMatmulPreferenceCreate")
    CASV_SYM(PreferenceSetAttribute, "hipblasLtMatmulPreferenceSetAttribute")
    CASV_SYM(AlgoGetHeuristic, "hipblasLtMatmulAlgoGetHeuristic")
    CASV_SYM(Matmul, "hipblasLtMatmul")
#undef CASV_SYM
    if (a.Create(&a.handle) != HIPBLAS_STATUS_SUCCESS) return a;
    a.workspace_bytes = 64u << 20;
    a.ok = true;
    return a;
}
}  // namespace

// the model that owned `stream` is being destroyed: its workspace goes with it
void vendor_gemm_release(hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_api.find(dev);
    if (it == g_api.end()) return;
    auto ws = it->second.workspace.find(stream);
    if (ws == it->second.workspace.end()) return;
    (void)hipFree(ws->second);
    it->second.workspace.erase(ws);
}

// true = the contraction was enqueued on `stream`; false = not taken (the caller runs its own kernel)
bool vendor_gemm_nt(const float* A, long long lda, long long M, int K, const float* Bt, int N, const float* bias, float* C, long long ldc,
                    int accumulate, hipStream_t stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_mu);
    Api& a = api_for(dev);
    if (!a.ok) return false;
    auto ws = a.workspace.find(stream);
    if (ws == a.workspace.end()) {
        void* w = nullptr;
        if (hipMalloc(&w, a.workspace_bytes) != hipSuccess) return false;
        ws = a.workspace.emplace(stream, w).first;
    }
    const Key key{dev, M, N, K, lda, ldc, (long long)K, bias != nullptr, accumulate};
    auto it = g_plans.find(key);
    if (it == g_plans.end()) {
        Plan p;
        // column-major view: C^T (N x M, ld ldc) = Bt_cm^T (N x K) . A_cm (K x M)
        const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
        bool good = a.MatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS &&
                    a.MatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)) == HIPBLAS_STATUS_SUCCESS &&
                    a.MatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)) == HIPBLAS_STATUS_SUCCESS &&
                    a.MatrixLayoutCreate(&p.la, HIP_R_32F, K, N, K) == HIPBLAS_STATUS_SUCCESS &&
                    a.MatrixLayoutCreate(&p.lb, HIP_R_32F, K, M, lda) == HIPBLAS_STATUS_SUCCESS &&
                    a.MatrixLayoutCreate(&p.lc, HIP_R_32F, N, M, ldc) == HIPBLAS_STATUS_SUCCESS;
        if (good && bias) {
            const hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_BIAS;
            good = a.MatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS;
        }
        hipblasLtMatmulPreference_t pref = nullptr;
        if (good) good = a.PreferenceCreate(&pref) == HIPBLAS_STATUS_SUCCESS &&
                         a.PreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &a.workspace_bytes, sizeof(a.workspace_bytes)) == HIPBLAS_STATUS_SUCCESS;
        if (good) {
            if (bias) {   // (the heuristic wants to see a bias pointer)
                const void* bp = bias;
                good = a.MatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) == HIPBLAS_STATUS_SUCCESS;
            }
            hipblasLtMatmulHeuristicResult_t res[1];
            int n = 0;
            good = good && a.AlgoGetHeuristic(a.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &n) == HIPBLAS_STATUS_SUCCESS && n > 0;
            if (good) p.algo = res[0].algo;
        }
        p.usable = good;
        it = g_plans.emplace(key, p).first;
    }
    Plan& p = it->second;
    if (!p.usable) return false;
    if (bias) {
        const void* bp = bias;
        if (a.MatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) != HIPBLAS_STATUS_SUCCESS) return false;
    }
    const float alpha = 1.0f, beta = accumulate ? 1.0f : 0.0f;
    return a.Matmul(a.handle, p.desc, &alpha, Bt, p.la, A, p.lb, &beta, C, p.lc, C, p.lc, &p.algo, ws->second, a.workspace_bytes, stream) ==
           HIPBLAS_STATUS_SUCCESS;
}

}  // namespace casv
#else
namespace casv {
void vendor_gemm_release(hipStream_t) {}
bool vendor_gemm_nt(const float*, long long, long long, int, const float*, int, const float*, float*, long long, int, hipStream_t) { return false; }
}  // namespace casv
#endif
