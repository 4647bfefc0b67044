// What it costs on gfx950 (MI355X) to add the samples of a pixel up with floating-point atomics instead of storing
// every sample and summing in a second launch.  Design input only; not part of the product library.
//
// The pattern is the one the lighting kernel of the pixels-only path would produce: a workgroup handles 256 consecutive
// candidates, which are (mostly) the 64 pixels of one 8x8 block at four sub-samples; per candidate either
//   store:  one 16-byte store to rgbf[sample] and one 4-byte store to hit_id[sample]      (what the kernel does now)
//   atomic: four global_atomic_add_f32 (no return) to acc[pixel][0..3]
// followed, per pixel, by either a pass over 16 x (4 + 16) bytes or one over 16 bytes.
//
// Build: hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics tools/ubench_accum.hip -o tools/ubench_accum
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int SPP = 16;

// candidate i of the frame: block = i / 1024, sub-sample group and pixel as the walk emits them (64 pixels x 4 sub-samples
// per wave of the walk, 4 such waves per block)
__device__ inline void where(int i, int* block, int* q, int* pix) {
  *block = i >> 10;
  const int w = (i >> 8) & 3, r = (i >> 6) & 3;
  *q = w * 4 + r;
  *pix = i & 63;
}

__global__ void __launch_bounds__(256) k_store(float4* rgbf, int* hit_id, int n, int stride) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    int b, q, pix;
    where(i, &b, &q, &pix);
    const int s = ((b * SPP + q) << 6) + pix;
    rgbf[s] = make_float4(0.25f, 0.5f, 0.125f, 1.0e-6f);
    hit_id[s] = i;
  }
}

__global__ void __launch_bounds__(256) k_atomic(float* acc, int n, int stride) {
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    int b, q, pix;
    where(i, &b, &q, &pix);
    float* a = acc + ((size_t)b * 64 + pix) * 4;
    unsafeAtomicAdd(a + 0, 0.25f);
    unsafeAtomicAdd(a + 1, 0.5f);
    unsafeAtomicAdd(a + 2, 0.125f);
    unsafeAtomicAdd(a + 3, 1.0e-6f);
  }
}

__global__ void __launch_bounds__(256) k_sum_samples(const float4* rgbf, const int* hit_id, int npix, unsigned* out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const int b = p >> 6, pix = p & 63;
  float r = 0.f, g = 0.f, bl = 0.f, e = 0.f;
  for (int q = 0; q < SPP; ++q) {
    const int s = ((b * SPP + q) << 6) + pix;
    if (hit_id[s] >= 0) {
      const float4 v = rgbf[s];
      r += v.x; g += v.y; bl += v.z; e += v.w;
    }
  }
  out[p] = (unsigned)(r * 15.9f) | ((unsigned)(g * 15.9f) << 8) | ((unsigned)(bl * 15.9f) << 16) | ((unsigned)(e * 1e5f) << 24);
}

__global__ void __launch_bounds__(256) k_sum_acc(const float4* acc, int npix, unsigned* out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= npix) return;
  const float4 v = acc[p];
  out[p] = (unsigned)(v.x * 15.9f) | ((unsigned)(v.y * 15.9f) << 8) | ((unsigned)(v.z * 15.9f) << 16) | ((unsigned)(v.w * 1e5f) << 24);
}

template <class F>
static float timed(F f, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main() {
  const int blocks = 32768;                 // live 8x8 blocks (the 4K frame has about 43,000)
  const int nsamples = blocks * 64 * SPP;   // 33.5 M candidates: every sample a hit
  const int npix = blocks * 64;
  float4* rgbf; int* hit_id; float* acc; unsigned* out;
  CK(hipMalloc(&rgbf, sizeof(float4) * nsamples));
  CK(hipMalloc(&hit_id, sizeof(int) * nsamples));
  CK(hipMalloc(&acc, sizeof(float) * 4 * npix));
  CK(hipMalloc(&out, sizeof(unsigned) * npix));
  CK(hipMemset(acc, 0, sizeof(float) * 4 * npix));
  const int grid = 256 * 5;  // resident workgroups striding over the candidates, as the lighting kernel runs
  const int stride = grid * 256;
  const float t_store = timed([&] { hipLaunchKernelGGL(k_store, dim3(grid), dim3(256), 0, 0, rgbf, hit_id, nsamples, stride); }, 10);
  const float t_atomic = timed([&] { hipLaunchKernelGGL(k_atomic, dim3(grid), dim3(256), 0, 0, acc, nsamples, stride); }, 10);
  const float t_sum_s = timed([&] { hipLaunchKernelGGL(k_sum_samples, dim3((npix + 255) / 256), dim3(256), 0, 0, rgbf, hit_id, npix, out); }, 10);
  const float t_sum_a = timed([&] { hipLaunchKernelGGL(k_sum_acc, dim3((npix + 255) / 256), dim3(256), 0, 0, (const float4*)acc, npix, out); }, 10);
  const float t_zero = timed([&] { CK(hipMemsetAsync(acc, 0, sizeof(float) * 4 * npix, 0)); }, 10);
  printf("%d candidates, %d pixels\n", nsamples, npix);
  printf("store 20 B per candidate         %8.3f ms  (%.0f GB/s)\n", t_store, nsamples * 20.0 / t_store / 1e6);
  printf("4 float atomics per candidate    %8.3f ms  (%.1f G atomics/s)\n", t_atomic, nsamples * 4.0 / t_atomic / 1e6);
  printf("sum 16 samples per pixel         %8.3f ms\n", t_sum_s);
  printf("read the accumulators            %8.3f ms\n", t_sum_a);
  printf("clear the accumulators           %8.3f ms\n", t_zero);
  printf("store + sum %.3f ms, atomics + read + clear %.3f ms\n", t_store + t_sum_s, t_atomic + t_sum_a + t_zero);
  return 0;
}
