// Checks of the CPU stand-in for the HIP runtime itself (tests/emu/hip/hip_runtime.h): TEST INFRASTRUCTURE.
//   emu_selftest ok        wave-level operations, divergent callers, barriers, atomics: prints "selftest ok"
//   emu_selftest oob       a kernel writes one element past a hipMalloc'ed buffer: AddressSanitizer must end the run
//   emu_selftest lds       a kernel writes one element past a __shared__ array: likewise
//   emu_selftest race      (race-detection build only) LDS written and read across waves without a barrier:
//                          ThreadSanitizer must report a data race
//   emu_selftest barrier   two waves wait at different __syncthreads(): the emulator must end the run
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <vector>

__global__ void k_votes(unsigned long long* ballots, int* firsts, int* shf, int* sums) {
  const int t = threadIdx.x, lane = t & 63;
  ballots[t] = __ballot(lane % 3 == 0);
  firsts[t] = __builtin_amdgcn_readfirstlane(t * 7 + 1);
  shf[t] = __shfl(t * 2, 5);
  int v = lane;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  sums[t] = v;
}

// Odd lanes vote among themselves inside a branch; then the whole wave votes.
__global__ void k_divergent(unsigned long long* inner, unsigned long long* outer, int* loop_votes) {
  const int t = threadIdx.x, lane = t & 63;
  inner[t] = 0;
  if (lane & 1) inner[t] = __ballot(lane < 32);
  outer[t] = __ballot(lane >= 60);
  // lanes leave the loop at different trips; each trip's vote sees the lanes still in it
  int n = 0;
  for (int i = 0; i < (lane & 3) + 1; ++i) n += __popcll(__ballot(1));
  loop_votes[t] = n;
}

__global__ void k_lds(int* out) {
  __shared__ int buf[256];
  __shared__ int total;
  const int t = threadIdx.x;
  if (t == 0) total = 0;
  buf[t] = t;
  __syncthreads();
  atomicAdd(&total, buf[255 - t]);
  __syncthreads();
  out[blockIdx.x * 256 + t] = total + blockIdx.x;
}

__global__ void k_lds_oob(int* out) {
  __shared__ int buf[64];
  buf[threadIdx.x + 1] = (int)threadIdx.x;  // the last lane writes buf[64]
  __syncthreads();
  out[threadIdx.x] = buf[threadIdx.x];
}

// Work-items of one workgroup exchange a value through LDS without a barrier in between.
__global__ void k_race(int* out) {
  __shared__ int fill;
  if (threadIdx.x == 0) fill = 7;
  // (a __syncthreads() belongs here)
  out[threadIdx.x] = fill + (int)threadIdx.x;
}

__global__ void k_oob(int* p, int n) { p[threadIdx.x + n - 63] = 1; }  // the last lane writes p[n]

__global__ void k_two_barriers(volatile int* out) {
  // two barriers an optimiser cannot merge: different work before and after each
  if (threadIdx.x < 64) {
    out[threadIdx.x] = 1;
    __syncthreads();
  } else {
    __syncthreads();
    out[threadIdx.x] = 2;
    __syncthreads();
  }
}

__global__ void k_early_exit(int* out) {
  if (threadIdx.x >= 64) return;  // the second wave is gone before the barrier
  __syncthreads();
  out[threadIdx.x] = 1;
}

#define CHECK(c) do { if (!(c)) { printf("selftest FAILED: %s (line %d)\n", #c, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const char* what = argc > 1 ? argv[1] : "ok";
  if (!strcmp(what, "oob")) {
    int* p;
    (void)hipMalloc(&p, 100 * sizeof(int));
    hipLaunchKernelGGL(k_oob, dim3(1), dim3(64), 0, nullptr, p, 100);
    printf("no fault reported\n");
    return 0;
  }
  if (!strcmp(what, "lds")) {
    int* p;
    (void)hipMalloc(&p, 64 * sizeof(int));
    hipLaunchKernelGGL(k_lds_oob, dim3(1), dim3(64), 0, nullptr, p);
    printf("no fault reported\n");
    return 0;
  }
  if (!strcmp(what, "race")) {
    int* p;
    (void)hipMalloc(&p, 128 * sizeof(int));
    hipLaunchKernelGGL(k_race, dim3(1), dim3(128), 0, nullptr, p);
    (void)hipFree(p);
    printf("race kernel ran\n");
    return 0;
  }
  if (!strcmp(what, "barrier")) {
    int* p;
    (void)hipMalloc(&p, 128 * sizeof(int));
    hipLaunchKernelGGL(k_two_barriers, dim3(1), dim3(128), 0, nullptr, p);
    printf("no mismatch reported\n");
    return 0;
  }
  unsigned long long *b, *inner, *outer;
  int *f, *s, *sums, *lv, *lds, *ee;
  (void)hipMalloc(&b, 128 * 8); (void)hipMalloc(&inner, 128 * 8); (void)hipMalloc(&outer, 128 * 8);
  (void)hipMalloc(&f, 128 * 4); (void)hipMalloc(&s, 128 * 4); (void)hipMalloc(&sums, 128 * 4); (void)hipMalloc(&lv, 128 * 4);
  (void)hipMalloc(&lds, 3 * 256 * 4); (void)hipMalloc(&ee, 64 * 4);
  CHECK(((unsigned char*)b)[0] != 0);  // fresh memory is not zero

  hipLaunchKernelGGL(k_votes, dim3(1), dim3(128), 0, nullptr, b, f, s, sums);
  unsigned long long expect = 0;
  for (int l = 0; l < 64; l += 3) expect |= 1ull << l;
  for (int t = 0; t < 128; ++t) {
    CHECK(b[t] == expect);
    CHECK(f[t] == (t & ~63) * 7 + 1);
    CHECK(s[t] == ((t & ~63) + 5) * 2);
  }
  CHECK(sums[0] == 63 * 64 / 2 && sums[64] == 63 * 64 / 2);

  hipLaunchKernelGGL(k_divergent, dim3(1), dim3(128), 0, nullptr, inner, outer, lv);
  unsigned long long odd_low = 0;
  for (int l = 1; l < 32; l += 2) odd_low |= 1ull << l;
  for (int t = 0; t < 128; ++t) {
    CHECK(inner[t] == ((t & 1) ? odd_low : 0ull));
    CHECK(outer[t] == 0xf000000000000000ull);
    // trip i is run by the lanes with (lane & 3) >= i: 64, 48, 32, 16 lanes
    const int trips = (t & 3) + 1, seen[4] = {64, 48, 32, 16};
    int n = 0;
    for (int i = 0; i < trips; ++i) n += seen[i];
    CHECK(lv[t] == n);
  }

  hipLaunchKernelGGL(k_lds, dim3(3), dim3(256), 0, nullptr, lds);
  for (int i = 0; i < 3 * 256; ++i) CHECK(lds[i] == 255 * 256 / 2 + i / 256);

  hipLaunchKernelGGL(k_early_exit, dim3(1), dim3(128), 0, nullptr, ee);
  for (int i = 0; i < 64; ++i) CHECK(ee[i] == 1);

  for (void* p : {(void*)b, (void*)inner, (void*)outer, (void*)f, (void*)s, (void*)sums, (void*)lv, (void*)lds, (void*)ee}) (void)hipFree(p);
  printf("selftest ok\n");
  emu::report(stdout);
  return 0;
}
