#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef v4i v4i_u __attribute__((aligned(4)));
// each WG: loop R rounds: load NL vectors from rows written last round, store 5 vectors, barrier.
template <int UNAL>
__global__ void k(int* buf, int width, int rounds, long long* out) {
  int* base = buf + (size_t)blockIdx.x * width * 64;
  int tid = threadIdx.x;
  long long tl = 0, ts = 0, tb = 0;
  int acc = 0;
  for (int r = 1; r <= rounds; ++r) {
    long long t0 = clock64();
    v4i s = {0,0,0,0};
    for (int j = 1; j <= 9; ++j) {
      const int* p = base + ((r - j) & 31) * width + 8 + tid * 4 + (UNAL ? ((j & 1) ? 1 : -1) : 0);
      v4i v = *(const v4i_u*)p;
      s += v;
    }
    acc += s[0] + s[1] + s[2] + s[3];
    asm volatile("" :: "v"(acc) : "memory");
    long long t1 = clock64();
    for (int j = 0; j < 5; ++j) {
      int* q = base + ((r + j * 7) & 31) * width + 8 + tid * 4;
      v4i v = {acc, r, j, tid};
      *(v4i*)q = v;
    }
    long long t2 = clock64();
    __syncthreads();
    long long t3 = clock64();
    tl += t1 - t0; ts += t2 - t1; tb += t3 - t2;
  }
  if (tid == 0 && blockIdx.x == 0) { out[0] = tl; out[1] = ts; out[2] = tb; out[3] = acc; }
}
int main() {
  int nwg = 64, width = 1 << 15;
  int* buf; long long* out;
  hipMalloc(&buf, (size_t)nwg * 16 * width * 64 * 4);
  hipMemset(buf, 0, (size_t)nwg * 16 * width * 64 * 4);
  hipMalloc(&out, 64);
  long long h[4];
  for (int threads : {256, 1024}) for (int wgs : {1, 64, 256, 1024}) for (int un = 0; un < 2; ++un) {
    int rounds = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    if (un) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(threads), 0, 0, buf, width, rounds, out);
    else hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(threads), 0, 0, buf, width, rounds, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, out, 32, hipMemcpyDeviceToHost);
    printf("threads %4d wgs %4d unaligned %d: %.3f us/round | cycles/round load %lld store-issue %lld barrier %lld\n", threads, wgs, un, ms * 1e3 / rounds, h[0] / rounds, h[1] / rounds, h[2] / rounds);
  }
  return 0;
}
