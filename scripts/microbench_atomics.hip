// Microbenchmark: how fast (and how coherent) are scattered fp16x2 / fp32 atomics on MI355X?
// Informs the grid-backward design (DESIGN.md "grid backward").  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
#include <cstring>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s line %d: %s\n",#x,__LINE__,hipGetErrorString(e)); exit(1);} }while(0)

__device__ inline uint32_t hash32(uint32_t x){ x^=x>>16; x*=0x7feb352dU; x^=x>>15; x*=0x846ca68bU; x^=x>>16; return x; }

enum Mode { PK_PLAIN=0, PK_SC1=1, F32_PLAIN=2, U32_ATOMIC=3, STORE=4, GATHER=5, PK_SC0SC1=6, F32_AGENT=7 };

template<int MODE>
__global__ void k_scatter(uint32_t* table, uint32_t mask, uint32_t ops_per_thread, int by_xcd, uint32_t n_tables, uint32_t table_stride, float* sink){
  uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x;
  uint32_t t_sel = by_xcd ? (blockIdx.x & 7u) % n_tables : (blockIdx.x/8u) % n_tables;   // which table this block hits
  uint32_t* tab = table + (size_t)t_sel*table_stride;
  float acc=0;
  for(uint32_t j=0;j<ops_per_thread;++j){
    uint32_t idx = hash32(tid*977u + j*0x9E3779B9u) & mask;
    if (MODE==PK_PLAIN){ h2 v={(_Float16)1.0f,(_Float16)0.5f}; __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) h2*)(tab+idx), v); }
    else if (MODE==PK_SC1){ h2 v={(_Float16)1.0f,(_Float16)0.5f}; uint32_t* p=tab+idx; asm volatile("global_atomic_pk_add_f16 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory"); }
    else if (MODE==PK_SC0SC1){ h2 v={(_Float16)1.0f,(_Float16)0.5f}; uint32_t* p=tab+idx; h2 r; asm volatile("global_atomic_pk_add_f16 %0, %1, %2, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory"); acc+=(float)r[0]; }
    else if (MODE==F32_PLAIN){ unsafeAtomicAdd((float*)(tab+idx), 1.0f); }
    else if (MODE==F32_AGENT){ __hip_atomic_fetch_add((float*)(tab+idx), 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    else if (MODE==U32_ATOMIC){ atomicAdd(tab+idx, 1u); }
    else if (MODE==STORE){ tab[idx]=j; }
    else if (MODE==GATHER){ acc += (float)tab[idx]; }
  }
  if (acc==123.456f) sink[0]=acc;
}

__global__ void k_lds(float* out, uint32_t ops_per_thread, uint32_t lds_words, int pk){
  extern __shared__ float lds[];
  for(uint32_t e=threadIdx.x;e<lds_words;e+=blockDim.x) lds[e]=0;
  __syncthreads();
  uint32_t tid = blockIdx.x*blockDim.x+threadIdx.x;
  for(uint32_t j=0;j<ops_per_thread;++j){
    uint32_t idx = hash32(tid*977u + j*0x9E3779B9u) % lds_words;
    if (pk){ h2 v={(_Float16)1.0f,(_Float16)0.5f}; __builtin_amdgcn_ds_atomic_fadd_v2f16((__attribute__((address_space(3))) h2*)&lds[idx], v); }
    else atomicAdd(&lds[idx], 1.0f);
  }
  __syncthreads();
  if (threadIdx.x==0) out[blockIdx.x]=lds[0];
}

template<int MODE> float run(uint32_t* table, uint32_t entries, uint32_t blocks, uint32_t opt, int by_xcd, uint32_t n_tables, uint32_t stride, float* sink){
  hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best=1e9;
  for(int rep=0;rep<4;++rep){
    CK(hipMemset(table,0,(size_t)n_tables*stride*4));
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_scatter<MODE>, dim3(blocks), dim3(256), 0, 0, table, entries-1, opt, by_xcd, n_tables, stride, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best) best=ms;
  }
  return best;
}

int main(){
  const uint32_t blocks=4096, opt=32; const double ops=(double)blocks*256*opt;   // 33.5M ops
  uint32_t* table; float* sink; const uint32_t n_tables=16, stride=1u<<19;
  CK(hipMalloc(&table,(size_t)n_tables*stride*4)); CK(hipMalloc(&sink,4096*4));
  printf("ops per launch: %.1fM\n", ops/1e6);
  struct C{const char* name; uint32_t entries; int by_xcd; uint32_t nt;};
  C cases[]={{"1 table 2MB, all XCDs",1u<<19,0,1},{"16 tables 2MB, table by block/8 (mixed XCDs)",1u<<19,0,16},{"8 tables 2MB, table = XCD (b%8)",1u<<19,1,8},
             {"1 table 16KB (4096 entries), all XCDs",4096,0,1},{"8 tables 16KB, table = XCD",4096,1,8},{"1 table 128KB",32768,0,1}};
  for(auto&c:cases){
    printf("-- %s\n", c.name);
    float t;
    t=run<PK_PLAIN>(table,c.entries,blocks,opt,c.by_xcd,c.nt,stride,sink);  printf("  pk_add_f16 (no scope bits) %8.3f ms  %7.1f Gop/s\n",t,ops/t/1e6);
    t=run<PK_SC1>(table,c.entries,blocks,opt,c.by_xcd,c.nt,stride,sink);    printf("  pk_add_f16 sc1             %8.3f ms  %7.1f Gop/s\n",t,ops/t/1e6);
    t=run<F32_PLAIN>(table,c.entries,blocks,opt,c.by_xcd,c.nt,stride,sink); printf("  add_f32 unsafeAtomicAdd    %8.3f ms  %7.1f Gop/s\n",t,ops/t/1e6);
    t=run<F32_AGENT>(table,c.entries,blocks,opt,c.by_xcd,c.nt,stride,sink); printf("  add_f32 hip_atomic agent   %8.3f ms  %7.1f Gop/s\n",t,ops/t/1e6);
    t=run<U32_ATOMIC>(table,c.entries,blocks,opt,c.by_xcd,c.nt,stride,sink);printf("  atomicAdd u32              %8.3f ms  %7.1f Gop/s\n",t,ops/t/1e6);
    t=run<STORE>(table,c.entries,blocks,opt,c.by_xcd,c.nt,stride,sink);     printf("  plain 4B store             %8.3f ms  %7.1f Gop/s\n",t,ops/t/1e6);
    t=run<GATHER>(table,c.entries,blocks,opt,c.by_xcd,c.nt,stride,sink);    printf("  plain 4B gather            %8.3f ms  %7.1f Gop/s\n",t,ops/t/1e6);
  }
  // coherence check: every thread adds 1.0f (no scope bits) into a 1024-entry table from all XCDs; sum must equal ops exactly
  {
    CK(hipMemset(table,0,(size_t)n_tables*stride*4));
    hipLaunchKernelGGL(k_scatter<F32_PLAIN>, dim3(blocks), dim3(256), 0, 0, table, 1023u, 4u, 0, 1u, stride, sink);
    CK(hipDeviceSynchronize());
    std::vector<float> h(1024); CK(hipMemcpy(h.data(),table,4096,hipMemcpyDeviceToHost));
    double s=0; for(float v:h) s+=v; printf("coherence f32 no-scope: sum=%.0f expected=%.0f %s\n", s,(double)blocks*256*4, s==(double)blocks*256*4?"OK":"LOST UPDATES");
    CK(hipMemset(table,0,(size_t)n_tables*stride*4));
    hipLaunchKernelGGL(k_scatter<U32_ATOMIC>, dim3(blocks), dim3(256), 0, 0, table, 1023u, 4u, 0, 1u, stride, sink);
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> hu(1024); CK(hipMemcpy(hu.data(),table,4096,hipMemcpyDeviceToHost));
    uint64_t su=0; for(auto v:hu) su+=v; printf("coherence u32 atomicAdd: sum=%llu expected=%llu\n",(unsigned long long)su,(unsigned long long)blocks*256*4);
    CK(hipMemset(table,0,(size_t)n_tables*stride*4));
    hipLaunchKernelGGL(k_scatter<PK_PLAIN>, dim3(64), dim3(256), 0, 0, table, 1023u, 4u, 0, 1u, stride, sink);   // 65536 adds of (1, .5): 64 per entry -> exact in fp16
    CK(hipDeviceSynchronize());
    std::vector<uint32_t> hp(1024); CK(hipMemcpy(hp.data(),table,4096,hipMemcpyDeviceToHost));
    double sp=0; for(auto v:hp){ uint16_t lo=v&0xffff; _Float16 f; memcpy(&f,&lo,2); sp+=(float)f; } printf("coherence pk_f16 no-scope: sum(lo)=%.0f expected=%.0f\n", sp,(double)64*256*4);
  }
  // LDS atomics
  { float* out; CK(hipMalloc(&out,4096*4)); hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for(int pk=0;pk<2;++pk) for(uint32_t words: {8192u, 32768u}){
      float best=1e9; for(int rep=0;rep<4;++rep){ CK(hipEventRecord(a)); hipLaunchKernelGGL(k_lds, dim3(1024), dim3(1024), words*4, 0, out, 32u, words, pk); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms,a,b)); if(ms<best)best=ms; }
      double o=1024.0*1024*32; printf("LDS %s random over %u words: %.3f ms %.1f Gop/s\n", pk?"ds_pk_add_f16":"ds_add_f32", words, best, o/best/1e6);
    } }
  return 0;
}
