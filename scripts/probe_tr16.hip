// Probe of ds_read_b64_tr_b16 (gfx950): which 16-bit element does lane c, element j receive for arbitrary per-lane addresses?
// Hypothesis checked: within a 16-lane group, result[c][j] = element (c & 3) of the 8-byte word fetched by lane 4j + (c >> 2).
//   hipcc --offload-arch=gfx950 -O3 scripts/probe_tr16.hip -o scripts/probe_tr16.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(const int* word_of_lane, short* out) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3)))*)(lds + 4 * word_of_lane[l]));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  int h_word[64]; short h_out[256];
  int* d_word; short* d_out;
  hipMalloc(&d_word, sizeof(h_word)); hipMalloc(&d_out, sizeof(h_out));
  int bad = 0;
  for (int trial = 0; trial < 20; ++trial) {
    for (int l = 0; l < 64; ++l) h_word[l] = trial == 0 ? l : rand() % 2048;
    hipMemcpy(d_word, h_word, sizeof(h_word), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_word, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      const int grp = l & ~15, c = l & 15;
      const int expect = 4 * h_word[grp + 4 * j + (c >> 2)] + (c & 3);
      if (h_out[l * 4 + j] != (short)expect) { if (bad < 8) printf("trial %d lane %d elem %d: got %d expected %d\n", trial, l, j, h_out[l*4+j], expect); ++bad; }
    }
    if (trial == 0) { printf("consecutive words: lane 0: %d %d %d %d  lane 1: %d %d %d %d  lane 17: %d %d %d %d\n", h_out[0],h_out[1],h_out[2],h_out[3],h_out[4],h_out[5],h_out[6],h_out[7],h_out[68],h_out[69],h_out[70],h_out[71]); }
  }
  printf(bad ? "MISMATCHES: %d\n" : "hypothesis holds (%d mismatches)\n", bad);
  return bad != 0;
}
