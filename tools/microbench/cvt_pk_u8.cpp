// Checks v_cvt_pk_u8_f32 against rintf + clamp on a dense set of floats (rounding mode / saturation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* x, unsigned* a, unsigned* b, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
  float v = x[i];
  a[i] = __builtin_amdgcn_cvt_pk_u8_f32(v, 0u, 0u) & 255u;
  float r = rintf(v); r = r < 0.f ? 0.f : (r > 255.f ? 255.f : r); b[i] = (unsigned)r;
}
int main() {
  std::vector<float> h;
  for (int k = -4; k <= 260; k++) for (float d : {-0.5f, -0.49999997f, -0.25f, 0.f, 0.25f, 0.49999997f, 0.5f, 0.50000006f, 0.75f}) h.push_back((float)k + d);
  for (int i = 0; i < 100000; i++) h.push_back((float)(i * 0.0031f - 20.f));
  h.push_back(1e30f); h.push_back(-1e30f); h.push_back(300.5f);
  int n = h.size(); float* dx; unsigned *da, *db; hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
  hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, dx, da, db, n);
  std::vector<unsigned> a(n), b(n); hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < n; i++) if (a[i] != b[i]) { if (bad < 10) printf("x=%.9g pk=%u ref=%u\n", h[i], a[i], b[i]); bad++; }
  printf("mismatches %d of %d\n", bad, n); return 0;
}
