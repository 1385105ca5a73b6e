// Standalone micro-benchmark of K/V row compaction variants (gfx950).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// variant 0: current kernel shape (thread handles 4 rows, 16 lanes per row)
template<int UNROLL, bool IDX64>
__global__ void __launch_bounds__(256) k_v0(const uint16_t* __restrict__ k, const uint16_t* __restrict__ v, int64_t ss, int64_t sh,
        const void* __restrict__ idxv, int Hkv, int S, int W, int cap, uint16_t* __restrict__ ko, uint16_t* __restrict__ vo){
  const int LPR=16, RPI=16;
  int bg=blockIdx.y; int g=bg%Hkv; int bb=bg/Hkv; bool isv=blockIdx.z;
  const uint16_t* src=(isv?v:k)+(int64_t)bb*S*ss+(int64_t)g*sh; uint16_t* dst=(isv?vo:ko)+(size_t)bg*cap*128;
  int kk=cap-W, n=S-W;
  int sub=threadIdx.x%LPR, rloc=threadIdx.x/LPR;
  int r0=blockIdx.x*(RPI*UNROLL)+rloc;
  int64_t srow[UNROLL];
  #pragma unroll
  for(int u=0;u<UNROLL;u++){ int r=r0+u*RPI; int64_t id; if(r<kk){ if(IDX64) id=((const int64_t*)idxv)[(size_t)bg*kk+r]; else id=((const int32_t*)idxv)[(size_t)bg*kk+r]; } else id=n+(r-kk); srow[u]=id; }
  uint4 val[UNROLL];
  #pragma unroll
  for(int u=0;u<UNROLL;u++){ int r=r0+u*RPI; if(r<cap) val[u]=*(const uint4*)(src+srow[u]*ss+sub*8); }
  #pragma unroll
  for(int u=0;u<UNROLL;u++){ int r=r0+u*RPI; if(r<cap) *(uint4*)(dst+(size_t)r*128+sub*8)=val[u]; }
}
// plain copy of same byte volume for reference
__global__ void __launch_bounds__(256) k_copy(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n){
  size_t i=blockIdx.x*(size_t)blockDim.x+threadIdx.x; size_t st=(size_t)gridDim.x*blockDim.x;
  for(;i<n;i+=st) b[i]=a[i];
}
int main(int argc,char**argv){
  int S= argc>1?atoi(argv[1]):32768; int B= argc>2?atoi(argv[2]):1; int cap=2048, W=8, Hkv=8, D=128; int kk=cap-W, n=S-W;
  int NL= B>=8?3:8; int HB=Hkv*B; // rotate distinct buffers
  std::vector<uint16_t*> K(NL),V(NL),KO(NL),VO(NL);
  size_t kb=(size_t)B*S*Hkv*D*2, ob=(size_t)B*Hkv*cap*D*2;
  for(int i=0;i<NL;i++){ CK(hipMalloc(&K[i],kb)); CK(hipMalloc(&V[i],kb)); CK(hipMalloc(&KO[i],ob)); CK(hipMalloc(&VO[i],ob)); CK(hipMemset(K[i],1,kb)); CK(hipMemset(V[i],2,kb)); }
  std::mt19937_64 rng(1);
  std::vector<int64_t> idx64((size_t)HB*kk); std::vector<int32_t> idx32((size_t)HB*kk); std::vector<int64_t> idxs((size_t)HB*kk);
  for(int g=0;g<HB;g++){ std::vector<int> p(n); for(int i=0;i<n;i++)p[i]=i; std::shuffle(p.begin(),p.end(),rng);
    for(int i=0;i<kk;i++){ idx64[(size_t)g*kk+i]=p[i]; idx32[(size_t)g*kk+i]=p[i]; }
    std::vector<int> q(p.begin(),p.begin()+kk); std::sort(q.begin(),q.end()); for(int i=0;i<kk;i++) idxs[(size_t)g*kk+i]=q[i]; }
  int64_t *d64,*ds; int32_t* d32; CK(hipMalloc(&d64,idx64.size()*8)); CK(hipMalloc(&ds,idx64.size()*8)); CK(hipMalloc(&d32,idx32.size()*4));
  CK(hipMemcpy(d64,idx64.data(),idx64.size()*8,hipMemcpyHostToDevice)); CK(hipMemcpy(ds,idxs.data(),idxs.size()*8,hipMemcpyHostToDevice)); CK(hipMemcpy(d32,idx32.data(),idx32.size()*4,hipMemcpyHostToDevice));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int64_t ss=Hkv*D, sh=D;
  auto run=[&](const char* name, auto launch){
    for(int i=0;i<10;i++) launch(i%NL);
    CK(hipDeviceSynchronize());
    int N=200; CK(hipEventRecord(e0));
    for(int i=0;i<N;i++) launch(i%NL);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms,e0,e1));
    double us=ms*1e3/N; double bytes=2.0*2*HB*cap*D*2; printf("%-40s %8.2f us  %7.1f GB/s\n",name,us,bytes/us/1e3);
  };
  run("v0 unroll4 idx64 random", [&](int i){ hipLaunchKernelGGL((k_v0<4,true>),dim3((cap+63)/64,HB,2),dim3(256),0,0,K[i],V[i],ss,sh,d64,Hkv,S,W,cap,KO[i],VO[i]); });
  run("v0 unroll4 idx64 sorted", [&](int i){ hipLaunchKernelGGL((k_v0<4,true>),dim3((cap+63)/64,HB,2),dim3(256),0,0,K[i],V[i],ss,sh,ds,Hkv,S,W,cap,KO[i],VO[i]); });
  run("v0 unroll4 idx32 random", [&](int i){ hipLaunchKernelGGL((k_v0<4,false>),dim3((cap+63)/64,HB,2),dim3(256),0,0,K[i],V[i],ss,sh,d32,Hkv,S,W,cap,KO[i],VO[i]); });
  run("v0 unroll1 idx64 random", [&](int i){ hipLaunchKernelGGL((k_v0<1,true>),dim3((cap+15)/16,HB,2),dim3(256),0,0,K[i],V[i],ss,sh,d64,Hkv,S,W,cap,KO[i],VO[i]); });
  run("v0 unroll2 idx64 random", [&](int i){ hipLaunchKernelGGL((k_v0<2,true>),dim3((cap+31)/32,HB,2),dim3(256),0,0,K[i],V[i],ss,sh,d64,Hkv,S,W,cap,KO[i],VO[i]); });
  run("v0 unroll8 idx64 random", [&](int i){ hipLaunchKernelGGL((k_v0<8,true>),dim3((cap+127)/128,HB,2),dim3(256),0,0,K[i],V[i],ss,sh,d64,Hkv,S,W,cap,KO[i],VO[i]); });
  run("plain copy same bytes (8.4MB->8.4MB)", [&](int i){ hipLaunchKernelGGL(k_copy,dim3(2048),dim3(256),0,0,(const uint4*)K[i],(uint4*)KO[i],(size_t)ob/16); });
  run("empty-ish copy 1 elem", [&](int i){ hipLaunchKernelGGL(k_copy,dim3(1),dim3(64),0,0,(const uint4*)K[i],(uint4*)KO[i],(size_t)1); });
  return 0;
}
