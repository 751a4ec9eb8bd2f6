// OPTIONAL library GEMMs (DVQ_USE_HIPBLASLT=1): the large bias-only NT / NN products of the stage-2 transformer's Linear layers can
// be handed to hipBLASLt for comparison.  The default is the hand-written path -- the software-pipelined 256 x 256 / 192 x 256
// kernels of igemm.hip (gemm_nt_wide_pipe_kernel: 760 - 1105 TFLOP/s on the StackGPT shapes of tools/gemm_probe.py against the
// library's 840 - 1210; the whole stage-2 step is 5 % slower without the library, profiles/r02_*) -- and the own TN kernel for the
// weight gradients, which beats the library's on every probe shape.
// The library is bound at run time with dlopen / dlsym (the copy already mapped into the process if there is one), so
// libdvq_hip.so has no link-time dependency on it.
#include <dlfcn.h>
#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

#include "dvq_common.h"

namespace {

struct Api {
    decltype(&hipblasLtCreate) create = nullptr;
    decltype(&hipblasLtMatmulDescCreate) desc_create = nullptr;
    decltype(&hipblasLtMatmulDescSetAttribute) desc_set = nullptr;
    decltype(&hipblasLtMatrixLayoutCreate) layout_create = nullptr;
    decltype(&hipblasLtMatmulPreferenceCreate) pref_create = nullptr;
    decltype(&hipblasLtMatmulPreferenceSetAttribute) pref_set = nullptr;
    decltype(&hipblasLtMatmulAlgoGetHeuristic) heuristic = nullptr;
    decltype(&hipblasLtMatmul) matmul = nullptr;
    hipblasLtHandle_t handle = nullptr;
    void* workspace = nullptr;
    size_t workspace_bytes = 0;
    bool ok = false;
};

struct Plan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
    hipblasLtMatmulAlgo_t algo;
    size_t ws = 0;
    bool ok = false;
};

using Key = std::tuple<int64_t, int64_t, int64_t, int64_t, int64_t, int64_t, int, int>;

std::mutex g_mu;
Api g_api;
bool g_tried = false;
std::map<Key, Plan> g_plans;

Api& api() {
    if (g_tried) return g_api;
    g_tried = true;
    const char* on = getenv("DVQ_USE_HIPBLASLT");
    if (on == nullptr || on[0] != '1') return g_api;
    void* h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_NOLOAD);      // the copy the process already uses (PyTorch's), if any
    if (h == nullptr) h = dlopen("libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) h = dlopen("/opt/rocm/lib/libhipblaslt.so.1", RTLD_NOW | RTLD_LOCAL);
    if (h == nullptr) return g_api;
    Api a;
#define DVQ_SYM(field, name) a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name))
    DVQ_SYM(create, "hipblasLtCreate");
    DVQ_SYM(desc_create, "hipblasLtMatmulDescCreate");
    DVQ_SYM(desc_set, "hipblasLtMatmulDescSetAttribute");
    DVQ_SYM(layout_create, "hipblasLtMatrixLayoutCreate");
    DVQ_SYM(pref_create, "hipblasLtMatmulPreferenceCreate");
    DVQ_SYM(pref_set, "hipblasLtMatmulPreferenceSetAttribute");
    DVQ_SYM(heuristic, "hipblasLtMatmulAlgoGetHeuristic");
    DVQ_SYM(matmul, "hipblasLtMatmul");
#undef DVQ_SYM
    if (!a.create || !a.desc_create || !a.desc_set || !a.layout_create || !a.pref_create || !a.pref_set || !a.heuristic || !a.matmul)
        return g_api;
    if (a.create(&a.handle) != HIPBLAS_STATUS_SUCCESS) return g_api;
    a.workspace_bytes = 32u << 20;
    if (hipMalloc(&a.workspace, a.workspace_bytes) != hipSuccess) {
        a.workspace = nullptr;
        a.workspace_bytes = 0;
    }
    a.ok = true;
    g_api = a;
    return g_api;
}

Plan& plan_for(Api& a, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb, int64_t ldc, bool bias, bool b_kn) {
    const Key key{M, N, K, lda, ldb, ldc, bias ? 1 : 0, b_kn ? 1 : 0};
    auto it = g_plans.find(key);
    if (it != g_plans.end()) return it->second;
    Plan p;
    // row-major C[M][N] = A[M][K] B[N][K]^T  ==  column-major D (N x M) = op_T(B as K x N) * (A as K x M);
    // with B stored [K][N] (b_kn, the weight matrix of an input gradient as it lies in memory) the first operand is the
    // column-major N x K matrix itself: no transposed copy of the weights is ever made
    const int32_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
    bool ok = a.desc_create(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && a.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, b_kn ? &opN : &opT, sizeof(opT)) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && a.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)) == HIPBLAS_STATUS_SUCCESS;
    if (ok && bias) {
        const uint32_t epi = HIPBLASLT_EPILOGUE_BIAS;
        const int32_t btype = HIP_R_32F;
        ok = a.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)) == HIPBLAS_STATUS_SUCCESS &&
             a.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &btype, sizeof(btype)) == HIPBLAS_STATUS_SUCCESS;
    }
    ok = ok && (b_kn ? a.layout_create(&p.la, HIP_R_16BF, (uint64_t)N, (uint64_t)K, ldb)
                     : a.layout_create(&p.la, HIP_R_16BF, (uint64_t)K, (uint64_t)N, ldb)) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && a.layout_create(&p.lb, HIP_R_16BF, (uint64_t)K, (uint64_t)M, lda) == HIPBLAS_STATUS_SUCCESS;
    ok = ok && a.layout_create(&p.lc, HIP_R_16BF, (uint64_t)N, (uint64_t)M, ldc) == HIPBLAS_STATUS_SUCCESS;
    if (ok) {
        hipblasLtMatmulPreference_t pref = nullptr;
        ok = a.pref_create(&pref) == HIPBLAS_STATUS_SUCCESS;
        const uint64_t wsb = a.workspace_bytes;
        ok = ok && a.pref_set(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb)) == HIPBLAS_STATUS_SUCCESS;
        hipblasLtMatmulHeuristicResult_t res[1];
        int found = 0;
        ok = ok && a.heuristic(a.handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, res, &found) == HIPBLAS_STATUS_SUCCESS && found > 0 &&
             res[0].state == HIPBLAS_STATUS_SUCCESS && res[0].workspaceSize <= a.workspace_bytes;
        if (ok) {
            p.algo = res[0].algo;
            p.ws = res[0].workspaceSize;
        }
    }
    p.ok = ok;
    return g_plans.emplace(key, p).first->second;
}

}  // namespace

// 1 = done by hipBLASLt, 0 = not taken (library absent / shape declined): the caller runs its own kernel.
// C[M][N] (bf16, row stride ldc) = alpha * A[M][K] op(B) (+ bias[N] fp32);  b_kn == 0: B is [N][K] (NT), 1: B is [K][N] (NN)
int dvq_blaslt_gemm_try(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                        int64_t ldc, float alpha, const float* bias, int b_kn, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_mu);
    Api& a = api();
    if (!a.ok) return 0;
    Plan& p = plan_for(a, M, N, K, lda, ldb, ldc, bias != nullptr, b_kn != 0);
    if (!p.ok) return 0;
    if (bias != nullptr &&
        a.desc_set(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias)) != HIPBLAS_STATUS_SUCCESS)
        return 0;
    const float beta = 0.f;
    const hipblasStatus_t st = a.matmul(a.handle, p.desc, &alpha, B, p.la, A, p.lb, &beta, C, p.lc, C, p.lc, &p.algo, a.workspace, p.ws,
                                        stream);
    return st == HIPBLAS_STATUS_SUCCESS ? 1 : 0;
}

int dvq_blaslt_gemm_nt_try(const void* A, const void* B, void* C, int64_t M, int64_t N, int64_t K, int64_t lda, int64_t ldb,
                           int64_t ldc, float alpha, const float* bias, hipStream_t stream) {
    return dvq_blaslt_gemm_try(A, B, C, M, N, K, lda, ldb, ldc, alpha, bias, 0, stream);
}

// C[M][N] = A[M][K] B[K][N], bf16, library path only (the caller transposes B and uses dvq_gemm_nt when this declines)
extern "C" int dvq_gemm_nn_lib(const void* A, const void* B, void* C, int dtype, int64_t M, int64_t N, int64_t K, int64_t lda,
                               int64_t ldb, int64_t ldc, dvq_stream_t stream) {
    DVQ_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, DVQ_EINVAL, "dvq_gemm_nn_lib: bad arguments");
    DVQ_REQUIRE(dtype == DVQ_BF16 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && N % 8 == 0 && K % 8 == 0, DVQ_ESHAPE,
                "dvq_gemm_nn_lib: bf16 with 16-byte aligned rows only");
    if (dvq_blaslt_gemm_try(A, B, C, M, N, K, lda, ldb, ldc, 1.f, nullptr, 1, (hipStream_t)stream) == 1) return DVQ_OK;
    dvq_set_error("dvq_gemm_nn_lib: hipBLASLt is not available or declined the shape (transpose B and call dvq_gemm_nt)");
    return DVQ_ESHAPE;
}

extern "C" int dvq_blaslt_available(void) {
    std::lock_guard<std::mutex> lock(g_mu);
    return api().ok ? 1 : 0;
}
