// gemm.h — descriptors of the batched small-GEMM launcher (k_gemm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace mppo {

enum GemmAct { ACT_NONE = 0, ACT_TANH = 1, ACT_RELU = 2 };
enum GemmEpi {
  EPI_BIAS_ACT = 0,  // C = act(A.B + bias[n])                          (MLP forward, train.py:62-68)
  EPI_DACT = 1,      // C = (A.B) * act'(aux[m][n])   aux = stored activations (MLP backward)
  EPI_STORE = 2,     // C = A.B                                           (weight-gradient slabs)
};

// One problem C[M,N] = op(A)[M,K] . op(B)[K,N].
//   a_t = 0 : A(m,k) = A[row(m)*lda + k]        row(m) = gather ? gather[m] : m
//   a_t = 1 : A(m,k) = A[row(k)*lda + m]        (A^T stored [K,M])
//   bias_out (EPI_STORE): also writes the column sums of B (sum over k of B(k,n)) to bias_out[n] (per split-K slab)
//   b_t = 0 : B(k,n) = B[k*ldb + n]
//   b_t = 1 : B(k,n) = B[n*ldb + k]
struct GemmProb {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* aux;
  const int* gather;
  int M, N, K;
  int lda, ldb, ldc, ldaux;
  int act;
  float* bias_out;
  float* a_copy;  // forward only: contiguous copy of the (gathered) A rows, [M, lda]
};

constexpr int kGemmMaxProb = 6;
struct GemmBatch {
  GemmProb p[kGemmMaxProb];
  int count;
  int ksplit;          // split-K factor (EPI_STORE only); slab s is written at C + s*slab_stride
  size_t slab_stride;  // floats
};

// variant = a_t*2 + b_t ; epi as above; bf16 = 1 uses bf16-in/f32-acc MFMA (inputs rounded to bf16 when staged)
int32_t gemm_launch(const GemmBatch& batch, int a_t, int b_t, int epi, int bf16, hipStream_t stream);
int32_t gemm_launch_lds(const GemmBatch& batch, int a_t, int b_t, int epi, hipStream_t stream);  // k_gemm_lds.hip

}  // namespace mppo
