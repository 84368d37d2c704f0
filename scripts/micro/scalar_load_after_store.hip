// Micro-test (round 6): does a wave-uniform (SCALAR, s_load through the constant address space) read in kernel B see what kernel A --
// launched immediately in front of it on the same stream -- just stored with vector stores?  YES, always (MI355X, 3 x 2 000 pairs:
// 0 stale reads, scalar or vector, fast or slow writer).  Written while chasing a root de-duplication failure in sp_order_kernel
// that looked like a stale read of the flags sp_scan had just written; the real cause was a hipcc miscompile of the sort key
// (tests/test_kernel_resources.py::test_parity_kernels_have_no_flat_loads).
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/sl scripts/micro/scalar_load_after_store.hip ; run: /tmp/sl
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

// `spin`: dependent work in front of the stores (a writer whose stores come LATE in its run, like sp_scan's -- behind a prefix scan
// with a dozen barriers): would show a reader that starts before the writer has finished
__global__ void writer(int* buf, int n, int epoch, int spin) {
  float x = (float)epoch;
  for (int k = 0; k < spin; ++k) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = epoch + (x < 0.f ? 1 : 0);
}
// reads buf[k] for wave-uniform k two ways: through a constant-address-space pointer (s_load_dword) and as a plain global load
__global__ void reader(const int* buf, int n, int* out) {
  typedef const int __attribute__((address_space(4)))* cint_p;
  int s_sum = 0, v_sum = 0;
  for (int k = 0; k < n; ++k) {
    s_sum += ((cint_p)buf)[k];                  // scalar path
    v_sum += buf[(k + threadIdx.x * 0) % n];     // vector path (same value for every lane)
  }
  if (threadIdx.x == 0) {
    out[0] = s_sum;
    out[1] = v_sum;
  }
}
// something else for the other CUs to do between epochs (as the CFR / net kernels do)
__global__ void filler(float* x, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = x[i] * 1.0001f + 1.0f;
}

int main() {
  const int n = 64, epochs = 2000;
  int *buf, *out;
  float* x;
  hipMalloc(&buf, n * sizeof(int));
  hipMalloc(&out, 2 * sizeof(int) * epochs);
  hipMalloc(&x, (1 << 22) * sizeof(float));
  hipMemset(buf, 0, n * sizeof(int));
  hipMemset(x, 0, (1 << 22) * sizeof(float));
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  for (int mode = 0; mode < 3; ++mode) {  // 0: reader right behind writer; 1: a filler kernel between them; 2: a slow writer (~40 us)
    for (int e = 1; e <= epochs; ++e) {
      hipLaunchKernelGGL(writer, dim3(1), dim3(1024), 0, st, buf, n, e, mode == 2 ? 20000 : 0);
      if (mode == 1) hipLaunchKernelGGL(filler, dim3(1 << 12), dim3(256), 0, st, x, 1 << 20);
      hipLaunchKernelGGL(reader, dim3(1), dim3(128), 0, st, buf, n, out + 2 * (e - 1));
      for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(filler, dim3(1 << 14), dim3(256), 0, st, x, 1 << 22);
    }
    hipStreamSynchronize(st);
    std::vector<int> h(2 * epochs);
    hipMemcpy(h.data(), out, h.size() * sizeof(int), hipMemcpyDeviceToHost);
    int stale_s = 0, stale_v = 0, first = -1;
    for (int e = 1; e <= epochs; ++e) {
      if (h[2 * (e - 1)] != n * e) {
        ++stale_s;
        if (first < 0) first = e;
      }
      if (h[2 * (e - 1) + 1] != n * e) ++stale_v;
    }
    std::printf("%s: %d epochs; scalar reads stale in %d (first at epoch %d, read %d expected %d); vector reads stale in %d\n",
                mode == 0 ? "reader directly behind writer" : (mode == 1 ? "a kernel between writer and reader" : "slow writer, reader directly behind"), epochs, stale_s, first,
                first > 0 ? h[2 * (first - 1)] : 0, first > 0 ? n * first : 0, stale_v);
  }
  return 0;
}
