// LDS atomic throughput on MI355X, VALU-light (LCG indices, power-of-two tables), dense and sparse lanes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
// kind: 0 ds_add_f32, 1 ds_pk_add_f16, 2 ds_add_u32, 3 ds_add_u64, 4 ds_write_b32 (non atomic), 5 ds_add_f32 x2 (two features)
template<int KIND>
__global__ void __launch_bounds__(1024) k(float* out, uint32_t iters, uint32_t mask, uint32_t active_mod){
  extern __shared__ float lds[];
  for(uint32_t e=threadIdx.x;e<=mask+1;e+=blockDim.x) lds[e]=0;
  __syncthreads();
  uint32_t x = (blockIdx.x*blockDim.x+threadIdx.x)*2654435761u + 12345u;
  const bool active = (threadIdx.x % active_mod)==0;
  for(uint32_t j=0;j<iters;++j){
    x = x*1664525u + 1013904223u;
    uint32_t idx = (x>>8) & mask;
    if (active){
      if (KIND==0) atomicAdd(&lds[idx], 1.0f);
      else if (KIND==1){ h2 v={(_Float16)1.0f,(_Float16)0.5f}; __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)&lds[idx], v); }
      else if (KIND==2) atomicAdd((uint32_t*)&lds[idx], 1u);
      else if (KIND==3) atomicAdd((unsigned long long*)&lds[idx & ~1u], 0x100000001ull);
      else if (KIND==4) lds[idx]=1.0f;
      else if (KIND==5){ atomicAdd(&lds[idx & ~1u], 1.0f); atomicAdd(&lds[(idx & ~1u)+1], 1.0f); }
    }
  }
  __syncthreads();
  if (threadIdx.x==0) out[blockIdx.x]=lds[0];
}
template<int KIND> void run(const char* name, float* out, uint32_t words, uint32_t active_mod, uint32_t threads){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  const uint32_t blocks=512, iters=256; float best=1e9;
  CK(hipFuncSetAttribute((const void*)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 140*1024));
  for(int rep=0;rep<4;++rep){ CK(hipEventRecord(a)); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), words*4+16, 0, out, iters, words-1, active_mod); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best)best=ms; }
  double lane_ops=(double)blocks*threads*iters/active_mod, instr=(double)blocks*(threads/64)*iters;
  printf("%-22s words=%6u threads=%4u active 1/%-2u : %7.3f ms  %7.1f G lane-ops/s  %6.2f G wave-instr/s  (%.2f clk/instr/CU @2.1GHz, 1 WG per CU x2 rounds)\n", name, words, threads, active_mod, best, lane_ops/best/1e6, instr/best/1e6, best*1e-3*2.1e9/(instr/256.0));
}
int main(){ float* out; CK(hipMalloc(&out,4096*4));
  for(uint32_t threads: {1024u, 256u}) for(uint32_t words: {8192u, 32768u}) for(uint32_t am: {1u, 8u, 32u}){
    run<0>("ds_add_f32",out,words,am,threads); run<1>("ds_pk_add_f16",out,words,am,threads); run<2>("ds_add_u32",out,words,am,threads);
    run<3>("ds_add_u64",out,words,am,threads); run<4>("ds_write_b32",out,words,am,threads); run<5>("2x ds_add_f32",out,words,am,threads); }
  return 0; }
