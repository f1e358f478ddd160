// Latency micro-benchmarks for gfx950 (diagnostic; not part of the product).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lat.hip -o tools/ubench/lat && tools/ubench/lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

__device__ inline long long Now() { return (long long)__builtin_amdgcn_s_memtime(); }

// 1. pointer chase: one lane, dependent loads through a random cycle
__global__ void Chase(const unsigned *next, int steps, long long *out, unsigned *sink) {
  unsigned i = threadIdx.x + blockIdx.x * 977;
  long long t0 = Now();
  for (int k = 0; k < steps; ++k) i = next[i];
  long long t1 = Now();
  if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / steps;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = i;
}

// 2. gather: every lane of W waves loads 16 bytes at a random place, then waits (one round trip under load)
__global__ void Gather(const int4 *data, const unsigned *idx, int rounds, long long *out, int *sink) {
  int acc = 0;
  unsigned j = idx[threadIdx.x];
  long long t0 = Now();
  for (int k = 0; k < rounds; ++k) {
    int4 v = data[j];
    acc += v.x;
    j = (unsigned)(v.y);  // dependent
  }
  long long t1 = Now();
  if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0) / rounds;
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + j;
}

// 3. LDS atomic chain, 4. barriers
__global__ void LdsStuff(long long *out, int iters) {
  __shared__ int ctr[64];
  __shared__ unsigned long long c64[64];
  __shared__ int bar;
  const int tid = threadIdx.x;
  if (tid < 64) { ctr[tid] = 0; c64[tid] = ~0ull; }
  if (tid == 0) bar = 0;
  __syncthreads();
  long long t0 = Now();
  int v = 0;
  for (int k = 0; k < iters; ++k) v = atomicAdd(&ctr[(v + tid) & 63], 1);   // dependent rtn atomics, spread addresses
  long long t1 = Now();
  unsigned long long w = tid;
  for (int k = 0; k < iters; ++k) w = atomicMin(&c64[(w + tid) & 63], w + k);
  long long t2 = Now();
  for (int k = 0; k < iters; ++k) __syncthreads();
  long long t3 = Now();
  for (int k = 0; k < iters; ++k) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  long long t4 = Now();
  // same-address atomic add of 1 from all lanes (compiler may wave-aggregate)
  int z = 0;
  for (int k = 0; k < iters; ++k) z += atomicAdd(&ctr[0], 1);
  long long t5 = Now();
  // soft barrier among first 4 waves
  long long t6 = t5;
  if (tid < 256) {
    int gen = 0;
    for (int k = 0; k < iters; ++k) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      gen += 4;
      if ((tid & 63) == 0) atomicAdd(&bar, 1);
      while (__hip_atomic_load(&bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) - gen < 0) __builtin_amdgcn_s_sleep(1);
    }
    t6 = Now();
  }
  __syncthreads();
  if (tid == 0) {
    out[0] = (t1 - t0) / iters; out[1] = (t2 - t1) / iters; out[2] = (t3 - t2) / iters;
    out[3] = (t4 - t3) / iters; out[4] = (t5 - t4) / iters; out[5] = (t6 - t5) / iters;
    out[6] = v + (int)w + z;
  }
}

int main() {
  const size_t N = 64u << 20;  // 64M unsigned = 256 MB
  std::vector<unsigned> h(N);
  // random single cycle (Sattolo)
  std::iota(h.begin(), h.end(), 0u);
  std::mt19937_64 rng(1);
  for (size_t i = N - 1; i > 0; --i) { size_t j = rng() % i; std::swap(h[i], h[j]); }
  unsigned *d_next; long long *d_out; unsigned *d_sink;
  CK(hipMalloc(&d_next, N * 4)); CK(hipMalloc(&d_out, 4096 * 8)); CK(hipMalloc(&d_sink, 1 << 24));
  CK(hipMemcpy(d_next, h.data(), N * 4, hipMemcpyHostToDevice));
  long long out[4096];
  for (int blocks : {1, 128}) for (int threads : {1, 64, 1024}) {
    Chase<<<blocks, threads>>>(d_next, 200, d_out, d_sink);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, d_out, blocks * 8, hipMemcpyDeviceToHost));
    long long mx = 0, sm = 0; for (int b = 0; b < blocks; ++b) { mx = std::max(mx, out[b]); sm += out[b]; }
    printf("chase 256MB: blocks %3d threads %4d: %lld cycles/step (max %lld)\n", blocks, threads, sm / blocks, mx);
  }
  // small footprint chase (4 MB -> L2 resident)
  {
    const size_t M = 1u << 20;
    std::vector<unsigned> g(M); std::iota(g.begin(), g.end(), 0u);
    for (size_t i = M - 1; i > 0; --i) { size_t j = rng() % i; std::swap(g[i], g[j]); }
    CK(hipMemcpy(d_next, g.data(), M * 4, hipMemcpyHostToDevice));
    for (int threads : {1, 1024}) {
      Chase<<<1, threads>>>(d_next, 2000, d_out, d_sink); CK(hipDeviceSynchronize());
      Chase<<<1, threads>>>(d_next, 2000, d_out, d_sink); CK(hipDeviceSynchronize());
      CK(hipMemcpy(out, d_out, 8, hipMemcpyDeviceToHost));
      printf("chase 4MB (L2): threads %4d: %lld cycles/step\n", threads, out[0]);
    }
    const size_t S = 1u << 12;  // 16 KB -> L1
    std::vector<unsigned> q(S); std::iota(q.begin(), q.end(), 0u);
    for (size_t i = S - 1; i > 0; --i) { size_t j = rng() % i; std::swap(q[i], q[j]); }
    CK(hipMemcpy(d_next, q.data(), S * 4, hipMemcpyHostToDevice));
    Chase<<<1, 1>>>(d_next, 20000, d_out, d_sink); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out, d_out, 8, hipMemcpyDeviceToHost));
    printf("chase 16KB (L1): threads    1: %lld cycles/step\n", out[0]);
  }
  LdsStuff<<<1, 1024>>>(d_out, 1000); CK(hipDeviceSynchronize());
  CK(hipMemcpy(out, d_out, 7 * 8, hipMemcpyDeviceToHost));
  printf("1024 threads: ds_add_rtn chain %lld, ds_min_rtn_u64 chain %lld, __syncthreads %lld, lds barrier %lld, same-addr atomicAdd %lld, soft barrier(4 waves) %lld cycles\n",
         out[0], out[1], out[2], out[3], out[4], out[5]);
  LdsStuff<<<1, 256>>>(d_out, 1000); CK(hipDeviceSynchronize());
  CK(hipMemcpy(out, d_out, 7 * 8, hipMemcpyDeviceToHost));
  printf(" 256 threads: ds_add_rtn chain %lld, ds_min_rtn_u64 chain %lld, __syncthreads %lld, lds barrier %lld, same-addr atomicAdd %lld, soft barrier(4 waves) %lld cycles\n",
         out[0], out[1], out[2], out[3], out[4], out[5]);
  return 0;
}
