// Microbenchmark: what does one vector-memory instruction cost a CU when its data is L1/L2 resident?
// Informs the forward gather: is the unit of cost the instruction, the active lane, the quad or the line?
//   hipcc --offload-arch=gfx950 -O3 scripts/microbench_l1.hip -o scripts/microbench_l1.bin
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d: %s\n",#x,__LINE__,hipGetErrorString(e)); exit(1);} }while(0)

__device__ inline uint32_t hash32(uint32_t x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }
typedef uint32_t u2 __attribute__((ext_vector_type(2)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

enum Mode { DWORD=0, DWORD_HALF_LANES=1, DWORD_QUARTER_LANES=2, DWORDX2=3, DWORDX4=4, STRIDE12=5, PAIR_SAME_LINE=6, LDS_B32=7, LDS_B64=8, COALESCED=9, DWORD_HALF_QUADS=10, DWORD_NT=11, DWORD_SC1=12, DWORD_SC0SC1=13, DWORDX2_NT=14, DWORD_SC0=15 };

template<int MODE>
__global__ void __launch_bounds__(256) k(const uint32_t* __restrict__ table, uint32_t mask, uint32_t iters, uint32_t* sink){
  __shared__ uint32_t lds[8192];
  if (MODE==LDS_B32 || MODE==LDS_B64){ for(uint32_t e=threadIdx.x;e<8192;e+=256) lds[e]=table[e&mask]; __syncthreads(); }
  const uint32_t tid = blockIdx.x*256+threadIdx.x, lane=threadIdx.x&63;
  uint32_t acc=0;
  for(uint32_t j=0;j<iters;j+=8){
    uint32_t v[8];
#pragma unroll
    for(uint32_t u=0;u<8;++u){
      const uint32_t h = hash32(tid*977u + (j+u)*0x9E3779B9u);
      const uint32_t idx = h & mask;
      v[u]=0;
      if (MODE==DWORD) v[u]=table[idx];
      else if (MODE==DWORD_HALF_LANES){ if (h>>31) v[u]=table[idx]; }
      else if (MODE==DWORD_QUARTER_LANES){ if ((h>>30)==3) v[u]=table[idx]; }
      else if (MODE==DWORD_HALF_QUADS){ if ((lane>>2)&1) v[u]=table[idx]; }
      else if (MODE==DWORDX2){ u2 t=*(const u2*)(table+(idx&~1u)); v[u]=t[0]^t[1]; }
      else if (MODE==DWORDX4){ u4 t=*(const u4*)(table+(idx&~3u)); v[u]=t[0]^t[1]^t[2]^t[3]; }
      else if (MODE==STRIDE12){ v[u]=table[((tid*3u + (j+u)*49152u + (u%3u)) & mask)]; }
      else if (MODE==PAIR_SAME_LINE){ v[u]=table[(u&1)? (hash32(tid*977u + (j+u-1)*0x9E3779B9u)&mask)^1u : idx]; }
      else if (MODE==LDS_B32){ v[u]=lds[idx&8191u]; }
      else if (MODE==LDS_B64){ u2 t=*(const u2*)(lds+(idx&8190u)); v[u]=t[0]^t[1]; }
      else if (MODE==DWORD_NT){ v[u]=__builtin_nontemporal_load(table+idx); }
      else if (MODE==DWORDX2_NT){ u2 t=__builtin_nontemporal_load((const u2*)(table+(idx&~1u))); v[u]=t[0]^t[1]; }
      else if (MODE==DWORD_SC1){ asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[u]) : "v"(table+idx) : "memory"); }
      else if (MODE==DWORD_SC0){ asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v[u]) : "v"(table+idx) : "memory"); }
      else if (MODE==DWORD_SC0SC1){ asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v[u]) : "v"(table+idx) : "memory"); }
      else if (MODE==COALESCED){ v[u]=table[(tid + (j+u)*65536u) & mask]; }
    }
    if (MODE==DWORD_SC1 || MODE==DWORD_SC0 || MODE==DWORD_SC0SC1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for(uint32_t u=0;u<8;++u) acc^=v[u];
  }
  if (acc==0x12345678u) sink[0]=acc;
}

template<int MODE> void run(const char* name, const uint32_t* table, uint32_t entries, uint32_t* sink){
  const uint32_t blocks=4096, iters=64; const double ops=(double)blocks*256*iters;
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best=1e9;
  for(int rep=0;rep<5;++rep){
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, table, entries-1, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best) best=ms;
  }
  // wave-instructions per CU per second -> clocks per wave-instruction at 2.4 GHz
  const double winstr_per_cu = ops/64/256;
  printf("  %-34s %8.3f ms  %8.1f G lane-ops/s  %6.1f clk/wave-instr/CU (2.4 GHz)\n", name, best, ops/best/1e6, best*1e-3*2.4e9/winstr_per_cu);
}

int main(){
  uint32_t* table; uint32_t* sink; CK(hipMalloc(&table,(size_t)(1u<<22)*4)); CK(hipMalloc(&sink,4096)); CK(hipMemset(table,1,(size_t)(1u<<22)*4));
  for (uint32_t entries : {2048u, 32768u, 524288u}) {
    printf("-- table %u KB\n", entries*4/1024);
    run<DWORD>("dword, 64 random lanes", table, entries, sink);
    run<DWORD_HALF_LANES>("dword, ~32 random lanes active", table, entries, sink);
    run<DWORD_QUARTER_LANES>("dword, ~16 random lanes active", table, entries, sink);
    run<DWORD_HALF_QUADS>("dword, every other quad active", table, entries, sink);
    run<DWORDX2>("dwordx2 (aligned pair)", table, entries, sink);
    run<DWORDX4>("dwordx4 (aligned quad)", table, entries, sink);
    run<PAIR_SAME_LINE>("dword + dword same line (idx, idx^1)", table, entries, sink);
    run<STRIDE12>("dword, 12-byte lane stride", table, entries, sink);
    run<COALESCED>("dword, coalesced", table, entries, sink);
    run<DWORD_NT>("dword nt, 64 random lanes", table, entries, sink);
    run<DWORDX2_NT>("dwordx2 nt (aligned pair)", table, entries, sink);
    run<DWORD_SC0>("dword sc0", table, entries, sink);
    run<DWORD_SC1>("dword sc1", table, entries, sink);
    run<DWORD_SC0SC1>("dword sc0 sc1", table, entries, sink);
  }
  printf("-- LDS 32 KB\n");
  run<LDS_B32>("ds_read_b32 random", table, 8192, sink);
  run<LDS_B64>("ds_read_b64 random (aligned pair)", table, 8192, sink);
  return 0;
}
