// Probe: is the f16-input MFMA's accumulation reproducible on a CPU?  Compares v_mfma_f32_32x32x16_f16 / 32x32x8_f16
// outputs against candidate arithmetic models evaluated exactly (__float128) on the host.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} }while(0)

// A: [32 rows][K] row-major fp16, B: [K][32 cols] stored as Bt [32 cols][K]; C: [32][32] f32
template<int K>
__global__ void k_mfma(const _Float16* A, const _Float16* Bt, const float* C, float* D){
  int l = threadIdx.x; int r = l & 31, h = l >> 5;
  f32x16 acc;
  for (int i=0;i<16;i++){ int m=(i&3)+8*(i>>2)+4*h; acc[i]=C[m*32+r]; }
  if constexpr (K==16){
    f16x8 a,b; for(int j=0;j<8;j++){ a[j]=A[r*16+8*h+j]; b[j]=Bt[r*16+8*h+j]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a,b,acc,0,0,0);
  } else {
    f16x4 a,b; for(int j=0;j<4;j++){ a[j]=A[r*8+4*h+j]; b[j]=Bt[r*8+4*h+j]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a,b,acc,0,0,0);
  }
  for (int i=0;i<16;i++){ int m=(i&3)+8*(i>>2)+4*h; D[m*32+r]=acc[i]; }
}
static float hf(uint16_t h){ _Float16 x; memcpy(&x,&h,2); return (float)x; }
static uint16_t rnd_half(uint64_t &st, int lo, int hi){ st=st*6364136223846793005ULL+1442695040888963407ULL; uint32_t r=st>>33; uint16_t e=lo+(r>>16)%(hi-lo+1); return (uint16_t)((r&0x8000)|(e<<10)|(r&0x3ff)); }
static uint32_t fb(float f){ uint32_t u; memcpy(&u,&f,4); return u; }
template<int K> void run(int trials, int elo, int ehi, float cscale){
  std::vector<uint16_t> A(32*K), Bt(32*K); std::vector<float> C(1024), D(1024);
  _Float16 *dA,*dB; float *dC,*dD; CK(hipMalloc(&dA,32*K*2)); CK(hipMalloc(&dB,32*K*2)); CK(hipMalloc(&dC,4096)); CK(hipMalloc(&dD,4096));
  uint64_t st=777+K; long n=0, mm[6]={0}; int shown=0;
  for(int t=0;t<trials;t++){
    for(auto&x:A) x=rnd_half(st,elo,ehi); for(auto&x:Bt) x=rnd_half(st,elo,ehi);
    for(auto&x:C){ st=st*6364136223846793005ULL+1442695040888963407ULL; x=((float)((int32_t)(st>>40)-(1<<23))/(float)(1<<20))*cscale; if((st&7)==0) x=0; }
    CK(hipMemcpy(dA,A.data(),32*K*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dB,Bt.data(),32*K*2,hipMemcpyHostToDevice)); CK(hipMemcpy(dC,C.data(),4096,hipMemcpyHostToDevice));
    k_mfma<K><<<1,64>>>(dA,dB,dC,dD); CK(hipMemcpy(D.data(),dD,4096,hipMemcpyDeviceToHost));
    for(int m=0;m<32;m++) for(int c=0;c<32;c++){
      float g=D[m*32+c]; float cc=C[m*32+c];
      // models
      __float128 ex=cc; float seq=cc; __float128 exh[4]={0,0,0,0}; 
      for(int k=0;k<K;k++){ float a=hf(A[m*K+k]), b=hf(Bt[c*K+k]); ex+=(__float128)a*b; seq=fmaf(a,b,seq); exh[k/(K/4)]+=(__float128)a*b; }
      float M1=(float)ex; float M2=seq;
      __float128 prods=0; for(int k=0;k<K;k++) prods+=(__float128)hf(A[m*K+k])*hf(Bt[c*K+k]);
      float M4=(float)((float)prods+(__float128)cc);   // products summed exactly, rounded, then + C rounded
      float M5=cc; for(int q=0;q<4;q++) M5=(float)((__float128)M5+exh[q]);   // 4 exact quarter-blocks added sequentially with rounding
      float M6=cc; { __float128 h0=exh[0]+exh[1], h1=exh[2]+exh[3]; M6=(float)((__float128)M6+h0); M6=(float)((__float128)M6+h1);} // two exact halves sequentially
      float cand[5]={M1,M2,M4,M5,M6}; n++;
      for(int q=0;q<5;q++) if(fb(cand[q])!=fb(g)) mm[q]++;
      if(fb(M1)!=fb(g) && shown<6){ shown++; printf("  K=%d ex: gpu=%08x M1=%08x seq=%08x M5=%08x M6=%08x c=%g\n",K,fb(g),fb(M1),fb(M2),fb(M5),fb(M6),cc); }
    }
  }
  printf("K=%d exp[%d..%d] cscale=%g n=%ld mismatches: exact-single-rounding=%ld seq-fma=%ld prods-rounded-then-C=%ld 4-quarter-blocks=%ld 2-half-blocks=%ld\n",K,elo,ehi,cscale,n,mm[0],mm[1],mm[2],mm[3],mm[4]);
}
int main(){
  run<16>(300,13,17,1.0f); run<16>(300,8,20,64.0f); run<16>(300,10,16,0.0f);
  run<8>(300,13,17,1.0f); run<8>(300,8,20,64.0f);
  return 0;
}
