// micro-benchmark / semantics check: global_load_lds_dwordx4 (gfx950): does lane l of a wave land at M0 base + 16 l ?
// build + run: hipcc --offload-arch=gfx950 -O3 scripts/micro/global_load_lds.hip -o /tmp/gll && /tmp/gll
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ void copy_kernel(const double* g, double* out, int n_chunks) {
  extern __shared__ __align__(16) double lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int c = wave; c < n_chunks; c += nw) {
    const char* src = reinterpret_cast<const char*>(g) + (size_t)c * 1024 + lane * 16;
    char* dst = reinterpret_cast<char*>(lds) + c * 1024;  // wave-uniform
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < n_chunks * 128; i += blockDim.x) out[i] = lds[i];
}

int main() {
  const int n_chunks = 93;
  std::vector<double> h(n_chunks * 128);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)i + 0.25;
  double *d, *o;
  hipMalloc(&d, h.size() * 8);
  hipMalloc(&o, h.size() * 8);
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)copy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(copy_kernel, dim3(1), dim3(1024), n_chunks * 1024, 0, d, o, n_chunks);
  std::vector<double> r(h.size());
  hipMemcpy(r.data(), o, r.size() * 8, hipMemcpyDeviceToHost);
  size_t bad = 0;
  for (size_t i = 0; i < h.size(); ++i) bad += r[i] != h[i];
  printf("global_load_lds_dwordx4: %zu of %zu doubles differ (%s)\n", bad, h.size(), bad ? "lane l does NOT land at base + 16 l" : "lane l lands at base + 16 l");
  return bad != 0;
}
