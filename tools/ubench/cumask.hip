// tools/ubench/cumask.hip -- where do the workgroups of two concurrent kernels of different
// shapes land (XCD, SE, CU), with and without CU masks on their streams?  (Round 6: the
// placement question behind "two kinds of persistent workgroups".)
//   hipcc --offload-arch=gfx950 -O2 cumask.hip -o cumask && ./cumask
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <map>
#include <set>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec { uint32_t xcc, hw; unsigned long long t0, t1; };

__global__ void Spin(Rec *out, unsigned long long ticks) {
  extern __shared__ unsigned char lds[];
  if (threadIdx.x == 0) {
    lds[0] = 1;
    Rec r;
    r.xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 15u;
    r.hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    r.t0 = wall_clock64();
    while (wall_clock64() - r.t0 < ticks) __builtin_amdgcn_s_sleep(32);
    r.t1 = wall_clock64();
    out[blockIdx.x] = r;
  }
  __syncthreads();
}

static uint32_t CuKey(const Rec &r) { return r.xcc << 16 | ((r.hw >> 13) & 7) << 8 | ((r.hw >> 12) & 1) << 4 | ((r.hw >> 8) & 15); }

static void Report(const char *name, const std::vector<Rec> &a, const std::vector<Rec> &b) {
  std::map<uint32_t, int> ca, cb;
  for (auto &r : a) ca[CuKey(r)]++;
  for (auto &r : b) cb[CuKey(r)]++;
  int shared = 0;
  for (auto &kv : ca) if (cb.count(kv.first)) shared++;
  int perxcc_a[8] = {0}, perxcc_b[8] = {0};
  for (auto &kv : ca) perxcc_a[kv.first >> 16]++;
  for (auto &kv : cb) perxcc_b[kv.first >> 16]++;
  std::map<int, int> hist;
  for (auto &kv : cb) hist[kv.second]++;
  unsigned long long a0 = ~0ull, a1 = 0, b0 = ~0ull, b1 = 0, amin_end = ~0ull, bmax_start = 0, amax_start = 0;
  for (auto &r : a) { a0 = std::min(a0, r.t0); a1 = std::max(a1, r.t1); amin_end = std::min(amin_end, r.t1); amax_start = std::max(amax_start, r.t0); }
  for (auto &r : b) { b0 = std::min(b0, r.t0); b1 = std::max(b1, r.t1); bmax_start = std::max(bmax_start, r.t0); }
  printf("%s: large on %zu CUs, small on %zu CUs, shared CUs %d; small WGs per CU:", name, ca.size(), cb.size(), shared);
  for (auto &kv : hist) printf(" %dx%d", kv.second, kv.first);
  printf("\n   CUs per XCD large:"); for (int i = 0; i < 8; ++i) printf(" %d", perxcc_a[i]);
  printf("  small:"); for (int i = 0; i < 8; ++i) printf(" %d", perxcc_b[i]);
  printf("\n   large starts span %.3f ms, last small start - first large start %.3f ms, first large end - first large start %.3f ms (all resident together: %s)\n",
         (amax_start - a0) / 1e5, ((double)bmax_start - (double)a0) / 1e5, (amin_end - a0) / 1e5,
         (bmax_start < amin_end && amax_start < amin_end) ? "yes" : "NO");
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("CUs %d, wall clock rate %d kHz\n", cus, prop.clockInstructionRate);
  int wc = 0; (void)hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0); printf("wall clock %d kHz\n", wc);
  const unsigned long long ticks = 3ull * wc;  // 3 ms
  CK(hipFuncSetAttribute((const void *)Spin, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  Rec *da, *db;
  const int X = 192, NB = 3 * (cus - X);
  CK(hipMalloc(&da, sizeof(Rec) * 1024)); CK(hipMalloc(&db, sizeof(Rec) * 1024));
  std::vector<Rec> a(X), b(NB);
  auto run = [&](const char *name, hipStream_t sa, hipStream_t sb, bool small_first) -> int {
    if (!small_first) hipLaunchKernelGGL(Spin, dim3(X), dim3(768), 159 * 1024, sa, da, ticks);
    hipLaunchKernelGGL(Spin, dim3(NB), dim3(256), 53 * 1024, sb, db, ticks);
    if (small_first) hipLaunchKernelGGL(Spin, dim3(X), dim3(768), 159 * 1024, sa, da, ticks);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(a.data(), da, sizeof(Rec) * X, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), db, sizeof(Rec) * NB, hipMemcpyDeviceToHost));
    Report(name, a, b);
    return 0;
  };
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  for (int rep = 0; rep < 2; ++rep) {
    if (run("no mask, large first", s0, s1, false)) return 1;
    if (run("no mask, small first", s0, s1, true)) return 1;
  }
  // masks: the first X bits for the large kernel's stream, the rest for the small one's
  std::vector<uint32_t> ma((cus + 31) / 32, 0), mb((cus + 31) / 32, 0);
  for (int i = 0; i < cus; ++i) (i < X ? ma : mb)[i / 32] |= 1u << (i % 32);
  hipStream_t ta, tb;
  hipError_t e = hipExtStreamCreateWithCUMask(&ta, ma.size(), ma.data());
  printf("hipExtStreamCreateWithCUMask: %s\n", hipGetErrorString(e));
  if (e != hipSuccess) return 0;
  CK(hipExtStreamCreateWithCUMask(&tb, mb.size(), mb.data()));
  for (int rep = 0; rep < 2; ++rep) {
    if (run("masks [0,X) / [X,256), large first", ta, tb, false)) return 1;
    if (run("masks [0,X) / [X,256), small first", ta, tb, true)) return 1;
  }
  // which CUs does a mask of the first 8 / 32 / 64 bits give?
  for (int nb : {8, 32, 64}) {
    std::vector<uint32_t> m((cus + 31) / 32, 0);
    for (int i = 0; i < nb; ++i) m[i / 32] |= 1u << (i % 32);
    hipStream_t t;
    CK(hipExtStreamCreateWithCUMask(&t, m.size(), m.data()));
    hipLaunchKernelGGL(Spin, dim3(256), dim3(768), 159 * 1024, t, da, ticks / 30);
    CK(hipStreamSynchronize(t));
    std::vector<Rec> r(256);
    CK(hipMemcpy(r.data(), da, sizeof(Rec) * 256, hipMemcpyDeviceToHost));
    std::set<uint32_t> cusu; int perx[8] = {0};
    for (auto &x : r) if (cusu.insert(CuKey(x)).second) perx[x.xcc]++;
    printf("mask of the first %d bits: %zu CUs; per XCD:", nb, cusu.size());
    for (int i = 0; i < 8; ++i) printf(" %d", perx[i]);
    printf("\n");
    CK(hipStreamDestroy(t));
  }
  return 0;
}
