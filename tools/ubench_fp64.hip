// FP64 division / square-root / normalize cost on gfx950 (MI355X): what one wave64 pays, in SIMD cycles, for the
// operations the shading kernels are made of, as the compiler expands them under the product's flags
// (-O3 -ffp-contract=off, IEEE division and sqrt).  Design input only; not part of the product library.
//
// Build: hipcc -O3 -ffp-contract=off --offload-arch=gfx950 tools/ubench_fp64.hip -o tools/ubench_fp64
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int ITERS = 2048;

struct d3 { double x, y, z; };

// the reference's normalize: three true divisions by the square root
__device__ inline d3 normalize_ref(d3 a) {
  const double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  return {a.x / len, a.y / len, a.z / len};
}

// one division for the three quotients: r = 1/len correctly rounded, then per component q = x*r,
// e = fma(-q, len, x), result = fma(e, r, q)
__device__ inline d3 normalize_shared(d3 a) {
  const double len = sqrt(a.x * a.x + a.y * a.y + a.z * a.z);
  const double r = 1.0 / len;
  auto quot = [&](double x) {
    const double q = x * r;
    const double e = __builtin_fma(-q, len, x);
    return __builtin_fma(e, r, q);
  };
  return {quot(a.x), quot(a.y), quot(a.z)};
}

template <int CHAINS>
__global__ void k_div(double* out, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + threadIdx.x * 1e-3 + c;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = b / x[c];
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CHAINS>
__global__ void k_sqrt(double* out, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + threadIdx.x * 1e-3 + c;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = sqrt(x[c]) + b;
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool SHARED>
__global__ void k_normalize(double* out, double a, double b) {
  d3 v = {a + threadIdx.x * 1e-3, b, a * b};
  for (int it = 0; it < ITERS; ++it) {
    d3 n = SHARED ? normalize_shared(v) : normalize_ref(v);
    v = {n.x + b, n.y - a, n.z + a};  // keeps the chain dependent and the operands away from 0
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = v.x + v.y + v.z;
}

template <int CHAINS>
__global__ void k_fma(double* out, double a, double b) {
  double x[CHAINS];
  for (int c = 0; c < CHAINS; ++c) x[c] = a + threadIdx.x * 1e-3 + c;
  for (int it = 0; it < ITERS; ++it)
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) x[c] = __builtin_fma(x[c], a, b);
  double s = 0;
  for (int c = 0; c < CHAINS; ++c) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define ASM_CHAIN(NAME, TEXT)                                                              \
  template <int CHAINS>                                                                    \
  __global__ void NAME(double* out, double a, double b) {                                  \
    double x[CHAINS];                                                                      \
    for (int c = 0; c < CHAINS; ++c) x[c] = a + threadIdx.x * 1e-3 + c;                    \
    for (int it = 0; it < ITERS; ++it)                                                     \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) asm volatile(TEXT : "+v"(x[c]));  \
    double s = 0;                                                                          \
    for (int c = 0; c < CHAINS; ++c) s += x[c];                                            \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                        \
  }
ASM_CHAIN(k_rcp, "v_rcp_f64 %0, %0")
ASM_CHAIN(k_rsq, "v_rsq_f64 %0, %0")
ASM_CHAIN(k_fixup, "v_div_fixup_f64 %0, %0, %0, %0")
ASM_CHAIN(k_scale, "v_div_scale_f64 %0, vcc, %0, %0, %0")
ASM_CHAIN(k_fmas, "v_div_fmas_f64 %0, %0, %0, %0")

template <class K>
static void run(const char* name, K kernel, int ops_per_iter, int cus, double mhz, double* out) {
  for (int waves = 1; waves <= 8; waves *= 2) {
    const int blocks = cus * waves;  // 256 threads = 4 waves = one per SIMD of a CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.5, 0.75);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 1.5, 0.75);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double cyc = ms * 1e-3 * mhz * 1e6 / ((double)ITERS * ops_per_iter * waves);
    printf("%-34s waves/SIMD %d: %8.3f ms  %8.2f SIMD-cycles per wave-level operation\n", name, waves, ms, cyc);
  }
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double mhz = prop.clockRate / 1000.0;
  printf("device %s  CUs %d  clock %.0f MHz (cycles below assume this clock)\n", prop.gcnArchName, cus, mhz);
  double* out;
  CK(hipMalloc(&out, sizeof(double) * cus * 8 * 256));
  run("v_fma_f64, 1 chain", k_fma<1>, 1, cus, mhz, out);
  run("v_fma_f64, 4 chains", k_fma<4>, 4, cus, mhz, out);
  run("v_rcp_f64, 4 chains", k_rcp<4>, 4, cus, mhz, out);
  run("v_rsq_f64, 4 chains", k_rsq<4>, 4, cus, mhz, out);
  run("v_div_scale_f64, 4 chains", k_scale<4>, 4, cus, mhz, out);
  run("v_div_fmas_f64, 4 chains", k_fmas<4>, 4, cus, mhz, out);
  run("v_div_fixup_f64, 4 chains", k_fixup<4>, 4, cus, mhz, out);
  run("IEEE division b/x, 1 chain", k_div<1>, 1, cus, mhz, out);
  run("IEEE division b/x, 3 chains", k_div<3>, 3, cus, mhz, out);
  run("IEEE sqrt, 1 chain", k_sqrt<1>, 1, cus, mhz, out);
  run("IEEE sqrt, 3 chains", k_sqrt<3>, 3, cus, mhz, out);
  run("normalize: sqrt + 3 divisions", k_normalize<false>, 1, cus, mhz, out);
  run("normalize: sqrt + 1 div + 3x(mul,2 fma)", k_normalize<true>, 1, cus, mhz, out);
  CK(hipFree(out));
  return 0;
}
