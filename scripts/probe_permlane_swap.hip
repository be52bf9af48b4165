// Probe of gfx950's v_permlane32_swap / v_permlane16_swap and of wave_rows_transpose4 (csrc/tcnn_device.h), which is built on them.
// Prints what each lane receives from the two raw swaps (rows = groups of 16 lanes) and checks the transpose against its
// definition: lane (g, c) receives t[r] = element g of what lane (r, c) passed.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude scripts/probe_permlane_swap.hip -o scripts/probe_permlane_swap.bin
#include "../tiny-cuda-nn_amd/csrc/tcnn_device.h"
using namespace tcnn_hip;
__global__ void k_raw(unsigned* out) {
  const unsigned l = threadIdx.x, a = 0x1000u + l, b = 0x2000u + l;
  const auto s32 = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const auto s16 = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l * 4 + 0] = s32[0]; out[l * 4 + 1] = s32[1]; out[l * 4 + 2] = s16[0]; out[l * 4 + 3] = s16[1];
}
__global__ void k_transpose(const unsigned short* in, unsigned short* out) {
  const unsigned l = threadIdx.x;
  typedef unsigned short us4 __attribute__((ext_vector_type(4)));
  const us4 v = *(const us4*)(in + l * 4);
  const h4 t = wave_rows_transpose4(__builtin_bit_cast(h4, v));
  *(us4*)(out + l * 4) = __builtin_bit_cast(us4, t);
}
int main() {
  unsigned h_raw[256]; unsigned* d_raw;
  hipMalloc(&d_raw, sizeof(h_raw));
  hipLaunchKernelGGL(k_raw, dim3(1), dim3(64), 0, 0, d_raw);
  hipMemcpy(h_raw, d_raw, sizeof(h_raw), hipMemcpyDeviceToHost);
  for (int row = 0; row < 4; ++row) {
    const int l = row * 16 + 3;
    printf("lane %2d (row %d): permlane32_swap -> {%#x, %#x}   permlane16_swap -> {%#x, %#x}   (a = 0x1000 + lane, b = 0x2000 + lane)\n", l, row, h_raw[l * 4],
           h_raw[l * 4 + 1], h_raw[l * 4 + 2], h_raw[l * 4 + 3]);
  }
  unsigned short h_in[256], h_out[256]; unsigned short *d_in, *d_out;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
  int bad = 0;
  for (int trial = 0; trial < 10; ++trial) {
    for (int i = 0; i < 256; ++i) h_in[i] = trial == 0 ? (unsigned short)i : (unsigned short)(rand() & 0xFFFF);  // arbitrary bit patterns incl. NaN / inf / -0
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_transpose, dim3(1), dim3(64), 0, 0, d_in, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    for (int g = 0; g < 4; ++g) for (int c = 0; c < 16; ++c) for (int r = 0; r < 4; ++r) {
      const unsigned short expect = h_in[(r * 16 + c) * 4 + g], got = h_out[(g * 16 + c) * 4 + r];
      if (expect != got) { if (bad < 8) printf("trial %d lane (%d, %d) element %d: got %#x expected %#x\n", trial, g, c, r, got, expect); ++bad; }
    }
  }
  printf(bad ? "wave_rows_transpose4: %d MISMATCHES\n" : "wave_rows_transpose4: ok (%d mismatches)\n", bad);
  return bad != 0;
}
