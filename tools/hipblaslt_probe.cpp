// Probe (not part of the library): hipBLASLt fp32 matmul with the bias + ReLU epilogue vs rocBLAS sgemm + a separate pass, on
// SetRank's skinny shapes.  Y[T, M] = relu(X[T, K] W[M, K]^T + b)  == column-major  D[M, T] = W^T(op T)[M, K] X[K, T].
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <rocblas/rocblas.h>
#include <cstdio>
#include <vector>
#define CK(x) do { auto e_ = (x); if ((int)e_ != 0) { printf("fail %s = %d line %d\n", #x, (int)e_, __LINE__); return 1; } } while (0)
int main() {
  const int64_t T = 102400;
  const int shapes[3][2] = {{256, 64}, {64, 256}, {256, 256}};  // K, M
  hipblasLtHandle_t lt; CK(hipblasLtCreate(&lt));
  rocblas_handle rb; CK(rocblas_create_handle(&rb));
  float *X, *W, *Y, *b; void* wsp;
  CK(hipMalloc(&X, T * 256 * 4)); CK(hipMalloc(&W, 256 * 256 * 4)); CK(hipMalloc(&Y, T * 256 * 4)); CK(hipMalloc(&b, 1024));
  CK(hipMalloc(&wsp, 64 << 20));
  CK(hipMemset(X, 0, T * 256 * 4)); CK(hipMemset(W, 0, 256 * 256 * 4)); CK(hipMemset(b, 0, 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& sh : shapes) {
    const int K = sh[0], M = sh[1];
    const float alpha = 1.f, beta = 0.f;
    // rocBLAS
    for (int it = 0; it < 3; ++it) rocblas_sgemm(rb, rocblas_operation_transpose, rocblas_operation_none, M, (int)T, K, &alpha, W, K, X, K, &beta, Y, M);
    CK(hipEventRecord(e0));
    for (int it = 0; it < 20; ++it) rocblas_sgemm(rb, rocblas_operation_transpose, rocblas_operation_none, M, (int)T, K, &alpha, W, K, X, K, &beta, Y, M);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("K %d M %d rocblas %.1f us\n", K, M, ms * 50.f);
    // hipBLASLt
    hipblasLtMatmulDesc_t md; CK(hipblasLtMatmulDescCreate(&md, HIPBLAS_COMPUTE_32F, HIP_R_32F));
    hipblasOperation_t ta = HIPBLAS_OP_T, tb = HIPBLAS_OP_N;
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSA, &ta, sizeof(ta)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_TRANSB, &tb, sizeof(tb)));
    hipblasLtEpilogue_t epi = HIPBLASLT_EPILOGUE_RELU_BIAS;
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_EPILOGUE, &epi, sizeof(epi)));
    CK(hipblasLtMatmulDescSetAttribute(md, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &b, sizeof(b)));
    hipblasLtMatrixLayout_t la, lb, lc;
    CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_32F, K, M, K));   // A stored [K x M] col-major (= W row-major [M, K]), op T -> [M, K]
    CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_32F, K, T, K));   // B [K x T]
    CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_32F, M, T, M));   // D [M x T]
    hipblasLtMatmulPreference_t pref; CK(hipblasLtMatmulPreferenceCreate(&pref));
    size_t wsz = 64 << 20; CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
    hipblasLtMatmulHeuristicResult_t res[8]; int nres = 0;
    CK(hipblasLtMatmulAlgoGetHeuristic(lt, md, la, lb, lc, lc, pref, 8, res, &nres));
    printf("  hipblaslt algos %d\n", nres);
    for (int a = 0; a < nres && a < 4; ++a) {
      for (int it = 0; it < 3; ++it) hipblasLtMatmul(lt, md, &alpha, W, la, X, lb, &beta, Y, lc, Y, lc, &res[a].algo, wsp, wsz, 0);
      CK(hipEventRecord(e0));
      for (int it = 0; it < 20; ++it) hipblasLtMatmul(lt, md, &alpha, W, la, X, lb, &beta, Y, lc, Y, lc, &res[a].algo, wsp, wsz, 0);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("  algo %d: %.1f us (bias + relu fused)\n", a, ms * 50.f);
    }
  }
  return 0;
}
