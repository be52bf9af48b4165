// Microbenchmark: can the scalar data cache's path to L2 carry random table gathers NEXT TO the vector memory path?
// (The forward gather is bound by the vector L1's line rate, profiles/r02_microbench_l1.txt.)  Per wave and iteration:
// V vector gathers of 64 random dwords (64 lines) and/or S scalar loads of one random line each.
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_scalar.hip -o scripts/microbench_scalar.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d: %s\n",#x,__LINE__,hipGetErrorString(e)); exit(1);} }while(0)

__device__ inline uint32_t hash32(uint32_t x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }
typedef uint32_t u2 __attribute__((ext_vector_type(2)));

template<int V, int S>
__global__ void __launch_bounds__(256) k(const uint32_t* __restrict__ table, uint32_t mask, uint32_t iters, uint32_t* sink){
  const uint32_t tid = blockIdx.x*256+threadIdx.x;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  uint32_t acc=0, sacc=0;
  for(uint32_t j=0;j<iters;++j){
    uint32_t v[V > 0 ? V : 1];
    u2 s[S > 0 ? S : 1];
#pragma unroll
    for(int u=0;u<S;++u){
      const uint32_t off = (hash32(wave*7919u + (j*S+u)*0x9E3779B9u) & mask & ~1u) * 4u;  // uniform: SALU
      asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(s[u]) : "s"(table), "s"(off) : "memory");
    }
#pragma unroll
    for(int u=0;u<V;++u){
      const uint32_t idx = hash32(tid*977u + (j*V+u)*0x9E3779B9u) & mask;
      v[u]=table[idx];
    }
    if (S > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for(int u=0;u<S;++u) sacc ^= s[u][0] ^ s[u][1];
#pragma unroll
    for(int u=0;u<V;++u) acc^=v[u];
  }
  if ((acc ^ sacc)==0x12345678u) sink[0]=acc;
}

template<int V, int S> void run(const uint32_t* table, uint32_t entries, uint32_t* sink){
  const uint32_t blocks=4096, iters=64;
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best=1e9;
  for(int rep=0;rep<5;++rep){
    CK(hipEventRecord(a));
    hipLaunchKernelGGL((k<V,S>), dim3(blocks), dim3(256), 0, 0, table, entries-1, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best) best=ms;
  }
  const double waves = (double)blocks*4, per_cu = waves*iters/256;  // wave-iterations per CU
  const double clk = best*1e-3*2.4e9/per_cu;
  const double vlines = (double)V*64, slines = S;
  printf("  V=%2d vector gathers + S=%2d scalar loads per wave-iteration: %8.3f ms  %7.1f clk per wave-iteration and CU  -> %6.2f clk per line (%5.1f %% of the lines scalar)\n",
         V, S, best, clk, clk/(vlines+slines), 100.0*slines/(vlines+slines));
}

int main(){
  uint32_t* table; uint32_t* sink; CK(hipMalloc(&table,(size_t)(1u<<22)*4)); CK(hipMalloc(&sink,4096)); CK(hipMemset(table,1,(size_t)(1u<<22)*4));
  for (uint32_t entries : {2048u, 524288u}) {
    printf("-- table %u KB\n", entries*4/1024);
    run<8,0>(table, entries, sink);
    run<4,0>(table, entries, sink);
    run<2,0>(table, entries, sink);
    run<0,8>(table, entries, sink);
    run<0,32>(table, entries, sink);
    run<8,8>(table, entries, sink);
    run<8,32>(table, entries, sink);
    run<4,32>(table, entries, sink);
    run<2,32>(table, entries, sink);
    run<1,32>(table, entries, sink);
  }
  return 0;
}
