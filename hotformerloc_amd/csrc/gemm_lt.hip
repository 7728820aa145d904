// Split-precision Linear through hipBLASLt with the epilogue the block needs, in ONE GEMM launch:
//
//     out (M,N) f32 = a (M,K') bf16 . w (N,K')^T bf16   [+ bias (N) f32]   [+ residual (M,N) f32]
//
// a / w are the K-concatenated operands [x_hi|x_hi|x_lo] / [w_hi|w_lo|w_hi] (include/hotformerloc_hip.h,
// section 9), accumulation and output are fp32.  Replaces `torch.nn.Linear` + the residual add of
// models/octformer_backbone.py:275-278 and models/hotformerloc_backbone.py:213-216 (x = x + proj(attn),
// x = x + mlp(x)): torch.addmm(residual, a, w.T) copies the residual into the output first (an extra
// M x N read + write) and cannot take bias and residual together; here C != D and the bias is the
// library's BIAS epilogue, so the separate add-bias pass over the residual stream disappears.
//
// Row-major operands map onto the column-major API as D^T (N x M) = op(A) . B with A = w (K' x N,
// ld K', transposed) and B = a (K' x M, ld K').  Algorithms come from the library heuristic for the exact
// problem and are cached; one 128 MiB workspace per stream (concurrent GEMMs on the pyramid streams must
// not share one).  hipBLASLt state is per device, workspaces per (device, stream); they live for the process.
#include "hfl_common.h"

#include <hipblaslt/hipblaslt.h>

#include <map>
#include <mutex>
#include <tuple>

namespace {

constexpr size_t kLtWorkspace = 128u << 20;   // stream-K schedules keep partial tiles here (32 MiB faulted)

struct LtPlan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t la = nullptr, lb = nullptr, lc = nullptr;
  hipblasLtMatmulAlgo_t algo;
  bool ok = false;
};

typedef std::tuple<int, int64_t, int, int, int, int> LtKey;     // device, M, N, K, bias (2 = weight-gradient form), residual

// All state is per device (the handle binds to the device current at creation; the null stream is a different
// queue on every device), workspaces per (device, stream).  Callers pass tensors of the CURRENT device.
std::mutex g_lt_mutex;
std::map<int, hipblasLtHandle_t> g_lt_handles;
thread_local hipblasLtHandle_t g_lt_handle = nullptr;           // handle of the call in progress (set under the mutex)
std::map<LtKey, LtPlan> g_lt_plans;
std::map<std::pair<int, hipStream_t>, void*> g_lt_ws;

// handle + workspace of the current device for `s`; HFL_OK or an error code
int lt_context(hipStream_t s, int* device, void** ws_out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  hipblasLtHandle_t& h = g_lt_handles[dev];
  if (h == nullptr) {
    const hipblasStatus_t st = hipblasLtCreate(&h);
    if (st != HIPBLAS_STATUS_SUCCESS) {
      h = nullptr;
      return HFL_EBACKEND - (int)st;
    }
  }
  g_lt_handle = h;
  void*& ws = g_lt_ws[std::make_pair(dev, s)];
  if (ws == nullptr) {
    e = hipMalloc(&ws, kLtWorkspace);
    if (e != hipSuccess) {
      ws = nullptr;
      return (int)e;
    }
  }
  *device = dev;
  *ws_out = ws;
  return HFL_OK;
}

void lt_destroy(LtPlan& p) {
  if (p.desc) hipblasLtMatmulDescDestroy(p.desc);
  if (p.la) hipblasLtMatrixLayoutDestroy(p.la);
  if (p.lb) hipblasLtMatrixLayoutDestroy(p.lb);
  if (p.lc) hipblasLtMatrixLayoutDestroy(p.lc);
  p = LtPlan();
}

int lt_make_plan(LtPlan& p, int64_t M, int N, int K, bool bias) {
#define HFL_LT(call)                                   \
  {                                                    \
    hipblasStatus_t st_ = (call);                      \
    if (st_ != HIPBLAS_STATUS_SUCCESS) {               \
      lt_destroy(p);                                   \
      return HFL_EBACKEND - (int)st_;                  \
    }                                                  \
  }
  HFL_LT(hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  const hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
  HFL_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
  HFL_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
  if (bias) {
    const hipblasLtEpilogue_t ep = HIPBLASLT_EPILOGUE_BIAS;
    const hipDataType bt = HIP_R_32F;
    const void* dummy = &ep;                           // heuristics want a non-null pointer; set per call
    HFL_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep)));
    HFL_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt)));
    HFL_LT(hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &dummy, sizeof(dummy)));
  }
  HFL_LT(hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, (uint64_t)K, (uint64_t)N, (int64_t)K));
  HFL_LT(hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, (uint64_t)K, (uint64_t)M, (int64_t)K));
  HFL_LT(hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, (uint64_t)N, (uint64_t)M, (int64_t)N));
  hipblasLtMatmulPreference_t pref = nullptr;
  HFL_LT(hipblasLtMatmulPreferenceCreate(&pref));
  const uint64_t ws = kLtWorkspace;
  hipblasStatus_t st = hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES,
                                                             &ws, sizeof(ws));
  hipblasLtMatmulHeuristicResult_t heur[1];
  int found = 0;
  if (st == HIPBLAS_STATUS_SUCCESS)
    st = hipblasLtMatmulAlgoGetHeuristic(g_lt_handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, heur, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || found < 1) {
    lt_destroy(p);
    return st != HIPBLAS_STATUS_SUCCESS ? HFL_EBACKEND - (int)st : HFL_EBACKEND;
  }
  p.algo = heur[0].algo;
  p.ok = true;
  return HFL_OK;
#undef HFL_LT
}

}  // namespace

extern "C" int hfl_gemm_bf16(float* out, const uint16_t* a, const uint16_t* w, const float* bias,
                              const float* residual, int64_t n_rows, int out_features, int k_concat,
                              hfl_stream_t stream) {
  if (n_rows < 0 || out_features <= 0 || k_concat <= 0) return HFL_EINVAL;
  if (out == nullptr || a == nullptr || w == nullptr || (const float*)out == residual) return HFL_EINVAL;
  if (n_rows == 0) return HFL_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(g_lt_mutex);
  int dev = 0;
  void* ws = nullptr;
  {
    const int rc = lt_context(s, &dev, &ws);
    if (rc != HFL_OK) return rc;
  }
  const LtKey key(dev, n_rows, out_features, k_concat, bias != nullptr, residual != nullptr);
  if (g_lt_plans.size() > 4096) {                       // token counts change with every batch
    for (auto& kv : g_lt_plans) lt_destroy(kv.second);
    g_lt_plans.clear();
  }
  LtPlan& p = g_lt_plans[key];
  if (!p.ok) {
    const int rc = lt_make_plan(p, n_rows, out_features, k_concat, bias != nullptr);
    if (rc != HFL_OK) {
      g_lt_plans.erase(key);
      return rc;
    }
  }
  if (bias != nullptr) {
    const void* bp = bias;
    if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bp, sizeof(bp)) !=
        HIPBLAS_STATUS_SUCCESS)
      return HFL_EBACKEND;
  }
  const float alpha = 1.0f, beta = residual != nullptr ? 1.0f : 0.0f;
  const hipblasStatus_t st =
      hipblasLtMatmul(g_lt_handle, p.desc, &alpha, w, p.la, a, p.lb, &beta,
                      residual != nullptr ? (const void*)residual : (const void*)out, p.lc, out, p.lc, &p.algo,
                      ws, kLtWorkspace, s);
  return st == HIPBLAS_STATUS_SUCCESS ? HFL_OK : HFL_EBACKEND - (int)st;
}

// Weight gradient of the split Linear: out (N,K) f32 = a (R,N)^T bf16 . b (R,K) bf16, the contraction
// running over the R = 3M stacked rows  a = [dy_hi; dy_hi; dy_lo],  b = [x_hi; x_lo; x_hi]
// (dW = dy^T x as dy_hi^T x_hi + dy_hi^T x_lo + dy_lo^T x_hi).  Column-major view: D^T (K x N) =
// B_c (K x R, ld K) . A_c^T (A_c: N x R, ld N).
extern "C" int hfl_gemm_bf16_tn(float* out, const uint16_t* a, const uint16_t* b, int64_t n_rows_stacked,
                                 int n_out, int k_out, hfl_stream_t stream) {
  if (n_rows_stacked <= 0 || n_out <= 0 || k_out <= 0 || out == nullptr || a == nullptr || b == nullptr)
    return HFL_EINVAL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  std::lock_guard<std::mutex> lock(g_lt_mutex);
  int dev = 0;
  void* ws = nullptr;
  {
    const int rc = lt_context(s, &dev, &ws);
    if (rc != HFL_OK) return rc;
  }
  const LtKey key(dev, n_rows_stacked, n_out, k_out, 2, 0);
  if (g_lt_plans.size() > 4096) {
    for (auto& kv : g_lt_plans) lt_destroy(kv.second);
    g_lt_plans.clear();
  }
  LtPlan& p = g_lt_plans[key];
  if (!p.ok) {
    hipblasStatus_t st = hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F);
    const hipblasOperation_t ta = HIPBLAS_OP_N, tb = HIPBLAS_OP_T;
    if (st == HIPBLAS_STATUS_SUCCESS)
      st = hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta));
    if (st == HIPBLAS_STATUS_SUCCESS)
      st = hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb));
    if (st == HIPBLAS_STATUS_SUCCESS)
      st = hipblasLtMatrixLayoutCreate(&p.la, HIP_R_16BF, (uint64_t)k_out, (uint64_t)n_rows_stacked, (int64_t)k_out);
    if (st == HIPBLAS_STATUS_SUCCESS)
      st = hipblasLtMatrixLayoutCreate(&p.lb, HIP_R_16BF, (uint64_t)n_out, (uint64_t)n_rows_stacked, (int64_t)n_out);
    if (st == HIPBLAS_STATUS_SUCCESS)
      st = hipblasLtMatrixLayoutCreate(&p.lc, HIP_R_32F, (uint64_t)k_out, (uint64_t)n_out, (int64_t)k_out);
    hipblasLtMatmulPreference_t pref = nullptr;
    if (st == HIPBLAS_STATUS_SUCCESS) st = hipblasLtMatmulPreferenceCreate(&pref);
    const uint64_t wsb = kLtWorkspace;
    if (st == HIPBLAS_STATUS_SUCCESS)
      st = hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsb, sizeof(wsb));
    hipblasLtMatmulHeuristicResult_t heur[1];
    int found = 0;
    if (st == HIPBLAS_STATUS_SUCCESS)
      st = hipblasLtMatmulAlgoGetHeuristic(g_lt_handle, p.desc, p.la, p.lb, p.lc, p.lc, pref, 1, heur, &found);
    if (pref) hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || found < 1) {
      lt_destroy(p);
      g_lt_plans.erase(key);
      return st != HIPBLAS_STATUS_SUCCESS ? HFL_EBACKEND - (int)st : HFL_EBACKEND;
    }
    p.algo = heur[0].algo;
    p.ok = true;
  }
  const float alpha = 1.0f, beta = 0.0f;
  const hipblasStatus_t st = hipblasLtMatmul(g_lt_handle, p.desc, &alpha, b, p.la, a, p.lb, &beta, out, p.lc, out,
                                             p.lc, &p.algo, ws, kLtWorkspace, s);
  return st == HIPBLAS_STATUS_SUCCESS ? HFL_OK : HFL_EBACKEND - (int)st;
}
