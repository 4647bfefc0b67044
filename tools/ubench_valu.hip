// Instruction-rate microbenchmark for gfx950 (MI355X): measures how many wave64
// VALU instructions per cycle per SIMD the chip sustains for the instruction
// forms the ray/triangle filter loop can be built from. Design input only; not
// part of the product library.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;

// 8 independent accumulators, 8 instructions per inner step, 4 steps unrolled.
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

__global__ void k_fma_vvv(float* out, float a, float b) {
  float acc[8];
  float va = a + threadIdx.x * 1e-9f, vb = b;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = i;
  for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_fma_svv(float* out, float a, float b) {
  float acc[8];
  float vb = b + threadIdx.x * 1e-9f;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = i;
  for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(a), "v"(vb));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pkfma_vvv(float* out, float a, float b) {
  f2 acc[8];
  f2 va = {a + threadIdx.x * 1e-9f, a}, vb = {b, b * 0.5f};
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f2{(float)i, (float)-i};
  for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// SGPR pair as one source, broadcast of the low dword of the VGPR source to both halves.
__global__ void k_pkfma_svv_bcast(float* out, float a, float b) {
  f2 acc[8];
  f2 sa = {a, a * 0.5f};
  f2 vb = {b + threadIdx.x * 1e-9f, b};
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f2{(float)i, (float)-i};
  for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[i]) : "s"(sa), "v"(vb));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_pkmul_vvv(float* out, float a, float b) {
  f2 acc[8];
  f2 va = {a + threadIdx.x * 1e-9f, a};
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = f2{(float)i, (float)-i};
  for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0" : "+v"(acc[i]) : "v"(va));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_max3(float* out, float a, float b) {
  float acc[8];
  float va = a + threadIdx.x * 1e-9f, vb = b;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = i;
  for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_max3_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_cmp_sgpr(float* out, float a, float b) {
  float va = a + threadIdx.x * 1e-3f, vb = b;
  unsigned long long m = 0;
  for (int it = 0; it < ITERS; ++it) {
#define X(i) { unsigned long long t; asm volatile("v_cmp_le_f32 %0, %1, %2" : "=s"(t) : "v"(va), "v"(vb)); m |= t; }
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(m & 0xffff);
}

__global__ void k_fma64(float* out, float a, float b) {
  double acc[8];
  double va = a + threadIdx.x * 1e-9, vb = b;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = i;
  for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(va), "v"(vb));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

__global__ void k_mul64(float* out, float a, float b) {
  double acc[8];
  double va = a + threadIdx.x * 1e-9;
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = i + 1;
  for (int it = 0; it < ITERS; ++it) {
#define X(i) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(va));
    REP8(X) REP8(X) REP8(X) REP8(X)
#undef X
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
}

// LDS broadcast read (all lanes one address) rate: ds_read_b128, 8 in flight.
__global__ void k_lds_bcast(float* out, float a, float b) {
  __shared__ float4 sm[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) sm[i] = make_float4(a, b, a, b);
  __syncthreads();
  float4 acc = make_float4(0, 0, 0, 0);
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int j = 0; j < 32; ++j) {
      float4 v = sm[(it * 32 + j) & 1023];
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// ---- prototype filter loops: 9 FMA + max3 + cmp per (ray, triangle) -------------
// Triangle data: 12 floats per triangle (3 scaled edge-plane vectors, padded).

// (A) lane = ray, triangle operands wave-uniform from global memory (scalar loads).
template <int R>
__global__ void __launch_bounds__(256) k_proto_sgpr(const float* __restrict__ tri, int ntri,
                                                     const float* __restrict__ rays, unsigned* hits) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float dx[R], dy[R], dz[R];
  unsigned cnt = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    dx[r] = rays[(gid * R + r) * 3 + 0];
    dy[r] = rays[(gid * R + r) * 3 + 1];
    dz[r] = rays[(gid * R + r) * 3 + 2];
  }
  for (int t = 0; t < ntri; t += 4) {
    const float* p = tri + t * 12;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float* q = p + u * 12;
      float m[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) m[k] = q[k];
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float s0 = __builtin_fmaf(m[2], dz[r], __builtin_fmaf(m[1], dy[r], m[0] * dx[r]));
        float s1 = __builtin_fmaf(m[5], dz[r], __builtin_fmaf(m[4], dy[r], m[3] * dx[r]));
        float s2 = __builtin_fmaf(m[8], dz[r], __builtin_fmaf(m[7], dy[r], m[6] * dx[r]));
        float mx = __builtin_fmaxf(__builtin_fmaxf(s0, s1), s2);
        any |= (mx <= 1.0f);
      }
      if (__builtin_expect(__any(any), 0)) cnt += any ? 1u : 0u;
    }
  }
  hits[gid] = cnt;
}

// (B) lane = ray, triangle chunk staged in LDS, broadcast reads.
template <int R>
__global__ void __launch_bounds__(256) k_proto_lds(const float* __restrict__ tri, int ntri,
                                                    const float* __restrict__ rays, unsigned* hits) {
  constexpr int CH = 256;  // triangles per chunk
  __shared__ float4 sm[CH * 3];
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float dx[R], dy[R], dz[R];
  unsigned cnt = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    dx[r] = rays[(gid * R + r) * 3 + 0];
    dy[r] = rays[(gid * R + r) * 3 + 1];
    dz[r] = rays[(gid * R + r) * 3 + 2];
  }
  for (int c = 0; c < ntri; c += CH) {
    __syncthreads();
    const float4* src = reinterpret_cast<const float4*>(tri + c * 12);
    for (int i = threadIdx.x; i < CH * 3; i += blockDim.x) sm[i] = src[i];
    __syncthreads();
#pragma unroll 4
    for (int t = 0; t < CH; ++t) {
      float4 a = sm[t * 3 + 0], b = sm[t * 3 + 1], cc = sm[t * 3 + 2];
      bool any = false;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        float s0 = __builtin_fmaf(a.z, dz[r], __builtin_fmaf(a.y, dy[r], a.x * dx[r]));
        float s1 = __builtin_fmaf(b.y, dz[r], __builtin_fmaf(b.x, dy[r], a.w * dx[r]));
        float s2 = __builtin_fmaf(cc.x, dz[r], __builtin_fmaf(b.w, dy[r], b.z * dx[r]));
        float mx = __builtin_fmaxf(__builtin_fmaxf(s0, s1), s2);
        any |= (mx <= 1.0f);
      }
      if (__builtin_expect(__any(any), 0)) cnt += any ? 1u : 0u;
    }
  }
  hits[gid] = cnt;
}

// (C) lane = triangle (T per lane in registers), rays wave-uniform via scalar loads.
template <int T>
__global__ void __launch_bounds__(256) k_proto_trilane(const float* __restrict__ triSoA, int ntri,
                                                        const float* __restrict__ rays, int nrays,
                                                        unsigned* hits) {
  // block handles 256*T triangles; grid.y splits rays
  int tbase = blockIdx.x * 256 * T + threadIdx.x;
  float m[T][9];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int k = 0; k < 9; ++k) m[t][k] = triSoA[(size_t)k * ntri + tbase + t * 256];
  int r0 = blockIdx.y * (nrays / gridDim.y), r1 = r0 + nrays / gridDim.y;
  unsigned cnt = 0;
  for (int r = r0; r < r1; r += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float dx = rays[(r + u) * 3 + 0], dy = rays[(r + u) * 3 + 1], dz = rays[(r + u) * 3 + 2];
      bool any = false;
#pragma unroll
      for (int t = 0; t < T; ++t) {
        float s0 = __builtin_fmaf(m[t][2], dz, __builtin_fmaf(m[t][1], dy, m[t][0] * dx));
        float s1 = __builtin_fmaf(m[t][5], dz, __builtin_fmaf(m[t][4], dy, m[t][3] * dx));
        float s2 = __builtin_fmaf(m[t][8], dz, __builtin_fmaf(m[t][7], dy, m[t][6] * dx));
        float mx = __builtin_fmaxf(__builtin_fmaxf(s0, s1), s2);
        any |= (mx <= 1.0f);
      }
      if (__builtin_expect(__any(any), 0)) cnt += any ? 1u : 0u;
    }
  }
  hits[blockIdx.y * gridDim.x * 256 + blockIdx.x * 256 + threadIdx.x] = cnt;
}

// ---- the production cone loop: 3 packed FMA-class instructions per PAIR of triangles and ray --------------
// A record holds 8 triangles as 4 pairs x 3 components x (triangle a, triangle b) = 12 f2 values (+4 padding);
// a ray passes a triangle iff d . w >= 1.  R rays per lane, as in k_search_primary<R> / k_search_tree<R>.
// (D) record operands wave-uniform from global memory: scalar loads, SGPR operands of v_pk_mul/v_pk_fma.
template <int R>
__global__ void __launch_bounds__(256) k_cone_sgpr(const f2* __restrict__ rec, int nrec, const float* __restrict__ rays,
                                                    unsigned* hits) {
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float dx[R], dy[R], dz[R];
  unsigned cnt = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    dx[r] = rays[(gid * R + r) * 3 + 0];
    dy[r] = rays[(gid * R + r) * 3 + 1];
    dz[r] = rays[(gid * R + r) * 3 + 2];
  }
  for (int b = 0; b < nrec; ++b) {
    const f2* rp = rec + (size_t)b * 16;
    f2 cur[12];
#pragma unroll
    for (int c = 0; c < 12; ++c) cur[c] = rp[c];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      f2 q[R];
#pragma unroll
      for (int r = 0; r < R; ++r) { f2 X = {dx[r], dx[r]}; q[r] = cur[pr * 3 + 0] * X; }
#pragma unroll
      for (int r = 0; r < R; ++r) { f2 Y = {dy[r], dy[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 1], Y, q[r]); }
#pragma unroll
      for (int r = 0; r < R; ++r) { f2 Z = {dz[r], dz[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 2], Z, q[r]); }
      float ma = q[0].x, mb = q[0].y;
#pragma unroll
      for (int r = 1; r < R; ++r) { ma = __builtin_fmaxf(ma, q[r].x); mb = __builtin_fmaxf(mb, q[r].y); }
      if (__builtin_expect(__any(__builtin_fmaxf(ma, mb) >= 1.0f), 0)) cnt += 1u;
    }
  }
  hits[gid] = cnt;
}

// (E) the same loop with the records staged through LDS in chunks and read back with broadcast reads.
template <int R>
__global__ void __launch_bounds__(256) k_cone_lds(const f2* __restrict__ rec, int nrec, const float* __restrict__ rays,
                                                   unsigned* hits) {
  constexpr int CH = 64;  // records per chunk: 64 x 128 B = 8 KB
  __shared__ f2 sm[CH * 16];
  int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float dx[R], dy[R], dz[R];
  unsigned cnt = 0;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    dx[r] = rays[(gid * R + r) * 3 + 0];
    dy[r] = rays[(gid * R + r) * 3 + 1];
    dz[r] = rays[(gid * R + r) * 3 + 2];
  }
  for (int c0 = 0; c0 < nrec; c0 += CH) {
    __syncthreads();
    for (int i = threadIdx.x; i < CH * 16; i += blockDim.x) sm[i] = rec[(size_t)c0 * 16 + i];
    __syncthreads();
    for (int b = 0; b < CH; ++b) {
      f2 cur[12];
#pragma unroll
      for (int c = 0; c < 12; ++c) cur[c] = sm[b * 16 + c];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        f2 q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { f2 X = {dx[r], dx[r]}; q[r] = cur[pr * 3 + 0] * X; }
#pragma unroll
        for (int r = 0; r < R; ++r) { f2 Y = {dy[r], dy[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 1], Y, q[r]); }
#pragma unroll
        for (int r = 0; r < R; ++r) { f2 Z = {dz[r], dz[r]}; q[r] = __builtin_elementwise_fma(cur[pr * 3 + 2], Z, q[r]); }
        float ma = q[0].x, mb = q[0].y;
#pragma unroll
        for (int r = 1; r < R; ++r) { ma = __builtin_fmaxf(ma, q[r].x); mb = __builtin_fmaxf(mb, q[r].y); }
        if (__builtin_expect(__any(__builtin_fmaxf(ma, mb) >= 1.0f), 0)) cnt += 1u;
      }
    }
  }
  hits[gid] = cnt;
}

template <typename F>
static float time_ms(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGetLastError());
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount;
  double clk = prop.clockRate * 1e3;  // Hz
  printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
  float* out;
  CK(hipMalloc(&out, sizeof(float) * 256 * cus * 16));

  struct T { const char* name; void (*k)(float*, float, float); double flop_per_inst; };
  T tests[] = {
      {"v_fma_f32 vvv", k_fma_vvv, 2}, {"v_fma_f32 svv", k_fma_svv, 2},
      {"v_pk_fma_f32 vvv", k_pkfma_vvv, 4}, {"v_pk_fma_f32 svv bcast", k_pkfma_svv_bcast, 4},
      {"v_pk_mul_f32", k_pkmul_vvv, 2}, {"v_max3_f32", k_max3, 2}, {"v_cmp->sgpr", k_cmp_sgpr, 1},
      {"v_fma_f64", k_fma64, 2}, {"v_mul_f64", k_mul64, 1}, {"ds_read_b128 bcast", k_lds_bcast, 0},
  };
  for (auto& t : tests) {
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD
      int blocks = cus * wps;       // 256-thread blocks: 4 waves -> one per SIMD
      float ms = time_ms([&] { hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f); });
      double inst_per_wave = (double)ITERS * 32;
      double waves_per_simd = wps;
      double cyc = ms * 1e-3 * clk;
      double cyc_per_inst = cyc / (inst_per_wave * waves_per_simd);
      double tflops = t.flop_per_inst * 64 * inst_per_wave * 4.0 * blocks / (ms * 1e-3) / 1e12;
      printf("%-26s waves/SIMD %d: %8.3f ms  %.2f cyc/inst/SIMD  %.1f TFLOP/s\n", t.name, wps, ms,
             cyc_per_inst, tflops);
    }
  }

  // prototypes
  const int ntri = 5120;
  const int nrays = 256 * cus * 8 * 2;  // enough for R=2 at 8 blocks/CU
  std::vector<float> htri((size_t)ntri * 12), hsoa((size_t)ntri * 9), hray((size_t)nrays * 3);
  srand(1);
  for (auto& v : htri) v = (rand() / (float)RAND_MAX - 0.5f) * 1e3f;
  for (int t = 0; t < ntri; ++t) for (int k = 0; k < 9; ++k) hsoa[(size_t)k * ntri + t] = htri[t * 12 + k];
  for (auto& v : hray) v = rand() / (float)RAND_MAX - 0.5f;
  float *dtri, *dsoa, *dray; unsigned* dh;
  CK(hipMalloc(&dtri, htri.size() * 4)); CK(hipMalloc(&dsoa, hsoa.size() * 4));
  CK(hipMalloc(&dray, hray.size() * 4)); CK(hipMalloc(&dh, sizeof(unsigned) * nrays));
  CK(hipMemcpy(dtri, htri.data(), htri.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsoa, hsoa.data(), hsoa.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dray, hray.data(), hray.size() * 4, hipMemcpyHostToDevice));

  auto report = [&](const char* name, double pairs, float ms) {
    double cyc_per_pair_simd = ms * 1e-3 * clk * cus * 4 / (pairs / 64.0);
    printf("%-28s %8.3f ms  %.3e pairs/s  %.2f SIMD-cyc per wave-pair (11 VALU => ideal 22)\n", name, ms,
           pairs / (ms * 1e-3), cyc_per_pair_simd);
  };
  for (int bpc : {2, 4, 8}) {
    int blocks = cus * bpc;
    printf("-- blocks/CU %d\n", bpc);
    float ms;
    ms = time_ms([&] { hipLaunchKernelGGL(k_proto_sgpr<1>, dim3(blocks), dim3(256), 0, 0, dtri, ntri, dray, dh); });
    report("sgpr R=1", (double)blocks * 256 * 1 * ntri, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_proto_sgpr<2>, dim3(blocks), dim3(256), 0, 0, dtri, ntri, dray, dh); });
    report("sgpr R=2", (double)blocks * 256 * 2 * ntri, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_proto_lds<1>, dim3(blocks), dim3(256), 0, 0, dtri, ntri, dray, dh); });
    report("lds R=1", (double)blocks * 256 * 1 * ntri, ms);
    ms = time_ms([&] { hipLaunchKernelGGL(k_proto_lds<2>, dim3(blocks), dim3(256), 0, 0, dtri, ntri, dray, dh); });
    report("lds R=2", (double)blocks * 256 * 2 * ntri, ms);
  }
  {
    // the production cone loop (3 packed FMA-class instructions per pair of triangles and ray): operands through
    // scalar loads (what k_search_primary / k_search_tree do) against operands staged in LDS
    const int nrec = ntri / 8;
    std::vector<float> hrec((size_t)nrec * 32);
    for (auto& v : hrec) v = (rand() / (float)RAND_MAX - 0.5f) * 0.5f;  // |d . w| stays below 1: nothing passes, as in the steady state
    float* drec;
    CK(hipMalloc(&drec, hrec.size() * 4));
    CK(hipMemcpy(drec, hrec.data(), hrec.size() * 4, hipMemcpyHostToDevice));
    auto report_cone = [&](const char* name, double pairs, float ms) {
      double cyc = ms * 1e-3 * clk * cus * 4 / (pairs / 64.0);
      printf("%-28s %8.3f ms  %.3e pairs/s  %.2f SIMD-cyc per wave-pair (1.5 packed FMA + 0.5 max => ideal 8.5)\n", name, ms,
             pairs / (ms * 1e-3), cyc);
    };
    for (int bpc : {4, 8}) {
      int blocks = cus * bpc;
      printf("-- cone loop, blocks/CU %d\n", bpc);
      float ms;
      ms = time_ms([&] { hipLaunchKernelGGL(k_cone_sgpr<2>, dim3(blocks), dim3(256), 0, 0, (const f2*)drec, nrec, dray, dh); });
      report_cone("cone sgpr R=2", (double)blocks * 256 * 2 * ntri, ms);
      ms = time_ms([&] { hipLaunchKernelGGL(k_cone_lds<2>, dim3(blocks), dim3(256), 0, 0, (const f2*)drec, nrec, dray, dh); });
      report_cone("cone lds R=2", (double)blocks * 256 * 2 * ntri, ms);
    }
    // R=4 needs twice the rays: a second ray buffer
    float* dray4;
    std::vector<float> hray4((size_t)256 * cus * 8 * 4 * 3);
    for (auto& v : hray4) v = rand() / (float)RAND_MAX - 0.5f;
    CK(hipMalloc(&dray4, hray4.size() * 4));
    CK(hipMemcpy(dray4, hray4.data(), hray4.size() * 4, hipMemcpyHostToDevice));
    for (int bpc : {4, 8}) {
      int blocks = cus * bpc;
      printf("-- cone loop, blocks/CU %d\n", bpc);
      float ms;
      ms = time_ms([&] { hipLaunchKernelGGL(k_cone_sgpr<4>, dim3(blocks), dim3(256), 0, 0, (const f2*)drec, nrec, dray4, dh); });
      report_cone("cone sgpr R=4", (double)blocks * 256 * 4 * ntri, ms);
      ms = time_ms([&] { hipLaunchKernelGGL(k_cone_lds<4>, dim3(blocks), dim3(256), 0, 0, (const f2*)drec, nrec, dray4, dh); });
      report_cone("cone lds R=4", (double)blocks * 256 * 4 * ntri, ms);
    }
  }
  {
    // lane = triangle: ntri/(256*T) blocks in x, rays split over y to fill the chip
    const int nr = 1 << 16;
    for (int T : {1, 2, 4}) {
      int bx = ntri / (256 * T);
      int by = (cus * 8) / bx;
      while (nr % by) --by;
      float ms;
      if (T == 1) ms = time_ms([&] { hipLaunchKernelGGL(k_proto_trilane<1>, dim3(bx, by), dim3(256), 0, 0, dsoa, ntri, dray, nr, dh); });
      else if (T == 2) ms = time_ms([&] { hipLaunchKernelGGL(k_proto_trilane<2>, dim3(bx, by), dim3(256), 0, 0, dsoa, ntri, dray, nr, dh); });
      else ms = time_ms([&] { hipLaunchKernelGGL(k_proto_trilane<4>, dim3(bx, by), dim3(256), 0, 0, dsoa, ntri, dray, nr, dh); });
      char nm[64]; snprintf(nm, sizeof nm, "trilane T=%d (%dx%d blocks)", T, bx, by);
      report(nm, (double)nr * ntri, ms);
    }
  }
  return 0;
}
